// msx.hip -- gfx950 (CDNA4) kernels + C ABI for mcmc_spec's per-walker log-likelihood path.
//
// What is in here (rows of SURVEY.md §8a; reference = /root/reference/mft6.py):
//   A0  staged grid [nt][ng][nwl] in HBM                      (specs dict, :342-383)
//   A1  isochrone Teff -> logg / luminosity lookup            (get_logg :87-98, get_radius :66-85)
//   A2  nearest-node bracket + bilinear blend                 (get_spec :387-563)
//   A3  Gaussian instrumental broadening, LDS-tiled FIR       (broaden :124-152 -> instrBroadGaussFast)
//   A4  flux scaling + component sum                          (make_composite :687-707,:740-751)
//   A5  contrast magnitudes via per-node band integrals       (:713-741)
//   A6  unresolved photometry via per-node band integrals     (:755-783)
//   A7  CCM89 reddening                                       (extinct :46-64)
//   A8  resample to data pixels, median scale, quadratic fit  (:1169-1174, norm_spec :193-196)
//   A9  chi^2 + combine                                       (chisq :115-122, :1178-1205)
//   f1  prior box + Gaussian terms                            (logprior :1207-1272, logposterior :1459-1470)
//
// Design notes (details in DESIGN.md):
//   * one workgroup per walker; the walker's Npix-long model vector lives in LDS for the exact
//     median (radix select), the 3-term fit and the chi^2 pass; all sums are float64 with a fixed
//     reduction order (wave shuffles, then LDS across waves) -> bit-reproducible run to run.
//   * the resample (A8) only ever touches the two model samples bracketing each data pixel, and
//     those indices are static per dataset, so staging gathers them once into a pixel-major
//     "pair table" pairs[node][pix] = {flux[lo], flux[lo+1]}: every hot-loop load is a 16-byte
//     per-lane, fully coalesced read (1 KiB per wave instruction).
//   * wave = 64 lanes everywhere; no MFMA (nothing here is a GEMM); no CUDA-compat shims.

#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "../../include/msx.h"

namespace {

constexpr int kWave = 64;
constexpr int kMaxWaves = 16;            // 1024 threads
constexpr int kMaxCorners = MSX_MAX_SPEC * 4;
constexpr int kSelectFinish = 256;       // radix select switches to all-pairs ranking at this many candidates
constexpr double kRsunCm = 6.957e10;     // mft6.py:691
constexpr double kPcCm = 3.086e18;       // mft6.py:691
constexpr double kLog2Of10 = 3.321928094887362347870319429489390175864831393024580612054;

// ------------------------------------------------------------------------------------------------
// Device-side view of everything staged.  Passed to kernels by value (well under the 4 KiB limit).
// ------------------------------------------------------------------------------------------------
// NOTE: kernels take this struct BY VALUE (kernarg segment).  Every device helper that receives it by
// reference must be __forceinline__: an out-of-line call needs the struct's address, which makes the
// compiler copy all ~1.2 KB of it into per-lane scratch and route every later access through scratch
// (measured: 21.8 -> 44.8 us per 256-walker launch).  `-Rpass-analysis=kernel-resource-usage` must show
// ScratchSize 0 for logprob_kernel; tests/test_abi.py checks it.  (Reading the struct through a pointer to a
// device copy instead was tried: no scratch hazard, but 61 instead of 16 spilled SGPRs and +10 % kernel time
// for 256-thread workgroups, so by-value + forced inlining stays.)
// Optional compact pair storage (msx_problem.compact_pairs): {flux[lo] as float64, flux[lo+1]-flux[lo] as
// float32}, 12 bytes instead of 16.  The difference of neighbouring 0.2 A samples is ~1e-2..1e-3 of the flux,
// so rounding it to float32 perturbs the upper sample by ~1e-9..1e-10 relative -- NOT bit-faithful to the
// float64 reference arithmetic; off by default, measured in DESIGN.md.
struct __attribute__((packed, aligned(4))) PairC {
    double lo;
    float d;
};
static_assert(sizeof(PairC) == 12, "PairC must be 12 bytes");

struct DevProblem {
    // grid (A0)
    const double *grid;   // [nt*ng][nwl]
    const double *kgrid;  // [nwl]  CCM89 a + b/3.1 per grid sample (A7)
    int64_t nwl;
    int32_t nt, ng;
    const double *teff_nodes;
    const double *logg_nodes;
    const uint8_t *present;
    // pixel tables (A8)
    const double2 *pairs;  // [nt*ng][npix] {flux[lo], flux[lo+1]}
    const PairC *pairs_c;  // [nt*ng][npix] compact form, or nullptr
    const double2 *pix_k;  // [npix] {k[lo], k[lo+1]}
    const double *pix_t, *pix_u, *pix_flux, *pix_ivar;  // pix_ivar = 1/err^2 (chisq squares sigma, mft6.py:120)
    int64_t npix;
    double median_flux;
    double minv[9];
    // bands (A5/A6)
    int32_t nc, np;
    const double *band_tab;  // [nt*ng][nc+np]
    double cmag[MSX_MAX_BANDS], cerr[MSX_MAX_BANDS];
    double pmag[MSX_MAX_BANDS], perr[MSX_MAX_BANDS], pzero[MSX_MAX_BANDS], pk[MSX_MAX_BANDS];
    double civar[MSX_MAX_BANDS], pivar[MSX_MAX_BANDS];  // 1/cerr^2, 1/perr^2
    int64_t win_j0, win_n;
    // isochrone (A1)
    int32_t niso;
    const double *iso_t, *iso_g, *iso_l;
    // prior (f1)
    int32_t nav;
    const double *av_edges, *av_mu, *av_sig;
    double tmin, tmax;
    double pmean[MSX_MAX_DIM], psig[MSX_MAX_DIM];
    int32_t use_av, dist_fit, rad_prior, has_prior;
    int32_t nspec;
    int32_t no_spectrum;  // mft6_nospec.py: contrast + photometry chi^2 only
    // pre-optimiser (f4): per-chain normalised data vectors / their medians, walker -> chain map
    double *opt_flux;          // [nchains][npix]
    double *opt_med;           // [nchains]
    const int32_t *opt_chain;  // [n] (OPT_STEP launches)
    double *model_scratch;     // [n][npix] only when the model vector does not fit LDS (GM kernel variants)
    // device-resident stretch move (f2): when smp_on, walker wk of the launch is the wk-th walker of the
    // active half; the kernel builds its own proposal and applies the accept rule in its last lines
    int32_t smp_on;
    double *smp_coords, *smp_logp;          // [nw][ndim], [nw]   ensemble state (updated in place)
    double *smp_q;                          // [ns][ndim]         proposals of this half-step
    const int32_t *smp_sidx, *smp_cidx, *smp_partner;  // [ns]; smp_partner holds cidx[partner]: the ensemble
                                                       // index of the complementary walker (resolved on the host)
    const double *smp_zz, *smp_zfac, *smp_logu;        // [ns]
    int64_t *smp_naccept;                   // [nw]
    double *smp_chain_row, *smp_lp_row;     // chain[step] [nw][ndim], logp chain[step] [nw]
    int32_t *smp_worst;
#ifdef MSX_STAMPS
    unsigned long long *stamps;  // diagnostic build only: [walker][16] shader-clock stamps
#endif
};

#ifdef MSX_STAMPS
#define MSX_STAMP(P, wk, i) do { if (threadIdx.x == 0) (P).stamps[(wk) * 16 + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define MSX_STAMP(P, wk, i) do { } while (0)
#endif

// Per-walker recipe computed once by one lane and broadcast through LDS.
struct WalkerDesc {
    int32_t node[kMaxCorners];   // flat node index it*ng+ig of each corner
    double w[kMaxCorners];       // bilinear weight * (R/d)^2 scale
    double redc;                 // exp2 coefficient -0.4*log2(10)*A_V; 0 -> no reddening (A_V <= 0)
    double lp;                   // log prior (0 in LOGLIKE mode)
    double chi_extra;            // icontrast + iphot                        mft6.py:1183,1189
    double contrast[MSX_MAX_BANDS];
    double phot[MSX_MAX_BANDS];
    double mag[MSX_MAX_BANDS * MSX_MAX_SPEC + MSX_MAX_BANDS];  // per-lane magnitudes of the wave recipe
    int32_t status;
    int32_t ncorner;
    int32_t stat[MSX_MAX_SPEC];  // fast recipe: per-star status of part 1 (one wave per star)
    // device-resident sampler: everything the accept step needs is fetched in the kernel's first lines (in
    // parallel with the proposal's own loads) so that its last lines touch no memory they have to wait for
    double theta[2 * MSX_MAX_SPEC + 2];   // the proposal q (this walker's coordinates for the launch)
    double smp_sv[2 * MSX_MAX_SPEC + 2];  // the walker's current coordinates
    double smp_old, smp_zfac, smp_logu;   // its current log-probability, (ndim-1) ln z, ln u
    int64_t smp_s;                        // its index in the ensemble
    int64_t smp_nacc;                     // its acceptance count so far
};

// ------------------------------------------------------------------------------------------------
// small device helpers
// ------------------------------------------------------------------------------------------------
// When is the composite (and its photometry) reddened?
//   loglikelihood / logposterior : `av == True and extinct_guess > 0`                 mft6.py:1161
//   fit_spec, proposals          : `var_par[1] > 0`                                    mft6.py:1002
//   fit_spec, initial guess      : never (the extinct() call is commented out and the chi^2 uses the
//                                  un-reddened `phot`)                                 mft6.py:880,901
__device__ __forceinline__ bool redden_rule(int mode, int use_av, double a_v) {
    if (mode == MSX_MODE_OPT_INIT) return false;
    if (mode == MSX_MODE_OPT_STEP) return a_v > 0.0;
    return use_av && a_v > 0.0;
}

// ---- cross-lane reductions on the DPP path (VALU speed) instead of ds_bpermute shuffles (an LDS round
// trip, ~50-100 cycles each, per 32-bit half, per step).  Four DPP steps leave every lane of a 16-lane
// row with its row's result; the four rows are then combined through v_readlane in a fixed order, so
// every lane returns the same, run-to-run reproducible value.
template <int CTRL>
__device__ __forceinline__ int dpp_i32(int v) {
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false);
}
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
    return __hiloint2double(dpp_i32<CTRL>(__double2hiint(v)), dpp_i32<CTRL>(__double2loint(v)));
}
template <int CTRL>
__device__ __forceinline__ unsigned long long dpp_u64(unsigned long long v) {
    const unsigned int lo = (unsigned int)dpp_i32<CTRL>((int)(unsigned int)v);
    const unsigned int hi = (unsigned int)dpp_i32<CTRL>((int)(unsigned int)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}
constexpr int kDppQuadSwap1 = 0xB1;  // quad_perm:[1,0,3,2]
constexpr int kDppQuadSwap2 = 0x4E;  // quad_perm:[2,3,0,1]
constexpr int kDppRowRor4 = 0x124;   // row_ror:4
constexpr int kDppRowRor8 = 0x128;   // row_ror:8

__device__ __forceinline__ double lane_f64(double v, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l),
                            __builtin_amdgcn_readlane(__double2loint(v), l));
}
__device__ __forceinline__ unsigned long long lane_u64(unsigned long long v, int l) {
    const unsigned int lo = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)v, l);
    const unsigned int hi = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)(v >> 32), l);
    return ((unsigned long long)hi << 32) | lo;
}

__device__ __forceinline__ double wave_sum(double v) {
    v += dpp_f64<kDppQuadSwap1>(v);
    v += dpp_f64<kDppQuadSwap2>(v);
    v += dpp_f64<kDppRowRor4>(v);
    v += dpp_f64<kDppRowRor8>(v);
    return ((lane_f64(v, 0) + lane_f64(v, 16)) + lane_f64(v, 32)) + lane_f64(v, 48);
}
__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v) {
    unsigned long long t;
    t = dpp_u64<kDppQuadSwap1>(v); v = t < v ? t : v;
    t = dpp_u64<kDppQuadSwap2>(v); v = t < v ? t : v;
    t = dpp_u64<kDppRowRor4>(v); v = t < v ? t : v;
    t = dpp_u64<kDppRowRor8>(v); v = t < v ? t : v;
    const unsigned long long a = lane_u64(v, 0), b = lane_u64(v, 16), c = lane_u64(v, 32), d = lane_u64(v, 48);
    const unsigned long long ab = a < b ? a : b, cd = c < d ? c : d;
    return ab < cd ? ab : cd;
}
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
    unsigned long long t;
    t = dpp_u64<kDppQuadSwap1>(v); v = t > v ? t : v;
    t = dpp_u64<kDppQuadSwap2>(v); v = t > v ? t : v;
    t = dpp_u64<kDppRowRor4>(v); v = t > v ? t : v;
    t = dpp_u64<kDppRowRor8>(v); v = t > v ? t : v;
    const unsigned long long a = lane_u64(v, 0), b = lane_u64(v, 16), c = lane_u64(v, 32), d = lane_u64(v, 48);
    const unsigned long long ab = a > b ? a : b, cd = c > d ? c : d;
    return ab > cd ? ab : cd;
}
// inclusive prefix sum over the 64 lanes: DPP row_shr steps inside each row of 16, then the three row
// carries through readlane
__device__ __forceinline__ unsigned int wave_scan_u32(unsigned int v) {
    const int lane = threadIdx.x & 63;
    unsigned int x = v;
    x += (unsigned int)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, true);  // row_shr:1
    x += (unsigned int)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, true);  // row_shr:2
    x += (unsigned int)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, true);  // row_shr:4
    x += (unsigned int)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, true);  // row_shr:8
    const unsigned int r0 = (unsigned int)__builtin_amdgcn_readlane((int)x, 15);
    const unsigned int r1 = (unsigned int)__builtin_amdgcn_readlane((int)x, 31);
    const unsigned int r2 = (unsigned int)__builtin_amdgcn_readlane((int)x, 47);
    const int row = lane >> 4;
    return x + (row > 0 ? r0 : 0u) + (row > 1 ? r1 : 0u) + (row > 2 ? r2 : 0u);
}

// order-preserving map double -> uint64 (NaN with sign bit clear sorts above +inf, like np.sort)
__device__ __forceinline__ unsigned long long key_of(double x) {
    unsigned long long b = (unsigned long long)__double_as_longlong(x);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double val_of(unsigned long long k) {
    unsigned long long b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    return __longlong_as_double((long long)b);
}

// mft6.py:439-453 / :467-477.  Nearest node first (first index on ties), then its neighbour on the
// other side; Python index semantics: -1 wraps to the last node, == n is an IndexError.
__device__ int bracket_nodes(const double *nodes, int n, double v, int *i1, int *i2) {
    int best = 0;
    double bd = fabs(nodes[0] - v);
    for (int i = 1; i < n; ++i) {
        double d = fabs(nodes[i] - v);
        if (d < bd) { bd = d; best = i; }
    }
    int other;
    if (nodes[best] == v) other = best;
    else if (nodes[best] > v) other = best - 1;
    else other = best + 1;
    if (other == -1) other = n - 1;
    if (other >= n) return MSX_W_INDEXERROR;
    *i1 = best;
    *i2 = other;
    return MSX_W_OK;
}

// Build the corner list + weights for every star (A2 + A4) and the band terms (A5/A6).
// Executed by ONE lane.  rad[] is the reference's rad_guess = [R1, R2/R1, (R3/R1)].
__device__ void build_desc(const DevProblem &P, const double *teff, const double *logg, const double *rad,
                           bool use_distance, double plx, double a_v, WalkerDesc *D) {
    const int ns = P.nspec;
    D->status = MSX_W_OK;
    D->ncorner = ns * 4;
    double starscale[MSX_MAX_SPEC];
    for (int s = 0; s < ns; ++s) {
        int t1, t2, g1, g2;
        int st = bracket_nodes(P.teff_nodes, P.nt, teff[s], &t1, &t2);
        if (st == MSX_W_OK) st = bracket_nodes(P.logg_nodes, P.ng, logg[s], &g1, &g2);
        if (st != MSX_W_OK) { D->status = st; return; }
        // the reference looks up all four keys unless both axes are on-node (mft6.py:488-500)
        int n11 = t1 * P.ng + g1, n12 = t1 * P.ng + g2, n21 = t2 * P.ng + g1, n22 = t2 * P.ng + g2;
        if (!P.present[n11] || !P.present[n12] || !P.present[n21] || !P.present[n22]) {
            D->status = MSX_W_KEYERROR;
            return;
        }
        double a = (g1 == g2) ? 0.0 : (logg[s] - P.logg_nodes[g1]) / (P.logg_nodes[g2] - P.logg_nodes[g1]);
        double b = (t1 == t2) ? 0.0 : (teff[s] - P.teff_nodes[t1]) / (P.teff_nodes[t2] - P.teff_nodes[t1]);
        double sc;
        if (use_distance) {
            double di = 1.0 / plx;  // mft6.py:690
            double r = (s == 0) ? rad[0] : rad[0] * rad[s];
            double q = r * kRsunCm / (di * kPcCm);  // mft6.py:691,700
            sc = q * q;
        } else {
            sc = (s == 0) ? 1.0 : rad[s - 1] * rad[s - 1];  // mft6.py:703
        }
        starscale[s] = sc;
        D->node[4 * s + 0] = n11; D->w[4 * s + 0] = (1.0 - b) * (1.0 - a) * sc;
        D->node[4 * s + 1] = n12; D->w[4 * s + 1] = (1.0 - b) * a * sc;
        D->node[4 * s + 2] = n21; D->w[4 * s + 2] = b * (1.0 - a) * sc;
        D->node[4 * s + 3] = n22; D->w[4 * s + 3] = b * a * sc;
    }
    (void)starscale;
    const bool redden = P.use_av && a_v > 0.0;  // mft6.py:1161
    D->redc = redden ? -0.4 * kLog2Of10 * a_v : 0.0;
    const int nb = P.nc + P.np;
    double chi = 0.0;
    // contrasts: instrumental magnitude of each star through each filter (A5)
    for (int f = 0; f < P.nc; ++f) {
        double mag[MSX_MAX_SPEC];
        for (int s = 0; s < ns; ++s) {
            double m = 0.0;
            for (int c = 0; c < 4; ++c) m += D->w[4 * s + c] * P.band_tab[(int64_t)D->node[4 * s + c] * nb + f];
            mag[s] = -2.5 * log10(m);  // mft6.py:733
        }
        int sec = 1;
        if (ns == 3 && f >= P.nc / 2) sec = 2;  // mft6.py:747-749
        double con = mag[sec] - mag[0];
        D->contrast[f] = con;
        double z = (con - P.cmag[f]);
        chi += (z * z) / (P.cerr[f] * P.cerr[f]);  // mft6.py:120,1182
    }
    // unresolved photometry of the composite (A6) + reddening of the magnitudes (mft6.py:1163)
    for (int f = 0; f < P.np; ++f) {
        double flux = 0.0;
        for (int c = 0; c < ns * 4; ++c) flux += D->w[c] * P.band_tab[(int64_t)D->node[c] * nb + P.nc + f];
        double mag = -2.5 * log10(flux / P.pzero[f]);  // mft6.py:780-782
        D->phot[f] = mag;
        double mred = redden ? mag + a_v * P.pk[f] : mag;
        double z = mred - P.pmag[f];
        chi += (z * z) / (P.perr[f] * P.perr[f]);  // mft6.py:1188
    }
    D->chi_extra = chi;
}

// The hard gates of logprior (a value of -inf, not an error) for theta = [T.., A_V, R1, ratios.., plx]:
//   dist_fit, binary   : T box, every radius entry >= 0.05, R1 <= 1.5, 1/3000 <= plx <= 1/4   mft6.py:1227
//   dist_fit, triple   : T box, every radius entry >= 0.05, 1/1000 <= plx <= 1/4               mft6.py:1347
//   no dist_fit, binary: T box, both radius entries >= 0.05                                    mft6.py:1286
//   no dist_fit, triple: T box, the two RATIOS >= 0.05 (R1 is not tested), plx >= 0            mft6.py:1411
//   and A_V >= 0 whenever extinction is fitted                                                 mft6.py:1229
template <int NS>
__device__ __forceinline__ bool prior_gates(const DevProblem &P, const double *t) {
    const double a_v = t[NS], plx = t[2 * NS + 1];
    const double *rad = t + NS + 1;
    bool ok = true;
#pragma unroll
    for (int s = 0; s < NS; ++s) ok = ok && !(t[s] > P.tmax) && !(t[s] < P.tmin);
    if (P.dist_fit) {
#pragma unroll
        for (int s = 0; s < NS; ++s) ok = ok && !(rad[s] < 0.05);
        if (NS == 2) ok = ok && !(rad[0] > 1.5) && !(plx < 1.0 / 3000) && !(plx > 1.0 / 4);
        else ok = ok && !(plx < 1.0 / 1000) && !(plx > 1.0 / 4);
    } else if (NS == 2) {
        ok = ok && !(rad[0] < 0.05) && !(rad[1] < 0.05);
    } else {
#pragma unroll
        for (int s = 1; s < NS; ++s) ok = ok && !(rad[s] < 0.05);
        ok = ok && !(plx < 0.0);
    }
    if (P.use_av) ok = ok && !(a_v < 0.0);
    return ok;
}

// ------------------------------------------------------------------------------------------------
// wave-parallel recipe helpers (phase 0 of the hot kernel runs on wave 0, all 64 lanes)
// ------------------------------------------------------------------------------------------------
// number of entries of the sorted table xs[0..n) that are <= x (an upper_bound), 64 entries per step
__device__ __forceinline__ int wave_count_le(const double *__restrict__ xs, int n, double x, int lane) {
    int cnt = 0;
    for (int base = 0; base < n; base += kWave) {
        const int i = base + lane;
        const bool pred = (i < n) && (xs[i] <= x);
        cnt += __popcll(__ballot(pred));
    }
    return cnt;
}

// np.interp on a sorted table given cnt = #{xs <= x}; caller has checked xs[0] <= x <= xs[n-1]
__device__ __forceinline__ double interp_from_count(const double *__restrict__ xs, const double *__restrict__ ys,
                                                    int n, double x, int cnt) {
    const int j = cnt - 1;
    if (j >= n - 1) return ys[n - 1];
    const double x0 = xs[j], y0 = ys[j];
    if (x0 == x) return y0;
    const double slope = (ys[j + 1] - y0) / (xs[j + 1] - x0);
    return slope * (x - x0) + y0;
}

// mft6.py:439-453 / :467-477 for SORTED, unique node values (staging sorts them; so does the
// reference, :436,:457-465): the nearest node is one of the two neighbours of v, ties go to the lower
// index like argmin; then the neighbour on the other side of v.  Python index semantics as in
// bracket_nodes(): -1 wraps to the last node, == n is an IndexError.
__device__ __forceinline__ int wave_bracket(const double *__restrict__ nodes, int n, double v, int lane, int *i1,
                                            int *i2, double *e1, double *e2) {
    const int j = wave_count_le(nodes, n, v, lane) - 1;  // nodes[j] <= v < nodes[j+1]
    int best;
    if (j < 0) best = 0;
    else if (j >= n - 1) best = n - 1;
    else best = (fabs(nodes[j + 1] - v) < fabs(nodes[j] - v)) ? j + 1 : j;
    const double nb = nodes[best];
    int other;
    if (nb == v) other = best;
    else if (nb > v) other = best - 1;
    else other = best + 1;
    if (other == -1) other = n - 1;
    if (other >= n) return MSX_W_INDEXERROR;
    *i1 = best;
    *i2 = other;
    *e1 = nb;
    *e2 = nodes[other];
    return MSX_W_OK;
}

// The small lookup tables of phase 0.  The hot kernel copies them into LDS (into the region that later
// holds the model vector) with all threads at once, so the recipe's dependent lookups cost an LDS
// round trip (~100 cycles) instead of an L2/MALL one (~500+); pointers are generic on purpose.
struct RecipeTabs {
    const double *iso_t, *iso_g, *iso_l, *av_edges, *av_mu, *av_sig, *teff_nodes, *logg_nodes;
};

// Phase 0 on wave 0: prior gate (f1), A1, A2, A4 weights, A5/A6 band terms.  Writes D (LDS).
template <int NS>
__device__ __forceinline__ void build_recipe_wave(const DevProblem &P, const RecipeTabs &T, int mode, const double *__restrict__ th,
                                  int ndim, WalkerDesc &D, int lane, int64_t wk) {
    double t[2 * NS + 2];
    bool alive = true;
#pragma unroll
    for (int k = 0; k < 2 * NS + 2; ++k) {
        t[k] = th[k];
        alive = alive && isfinite(t[k]);  // emcee refuses non-finite coordinates anyway
    }
    const double a_v = t[NS];
    const double plx = t[2 * NS + 1];
    const double *rad = &t[NS + 1];
    int st = MSX_W_OK;
    double lp = 0.0;
    if (alive && (mode == MSX_MODE_LOGPOST || mode == MSX_MODE_LOGPRIOR)) {
        alive = alive && prior_gates<NS>(P, t);
        if (alive && P.use_av) {
            if (P.nav > 0) {
                const double d = 1.0 / plx;  // pc, mft6.py:1233
                int b = wave_count_le(T.av_edges, P.nav + 1, d, lane) - 1;
                b = b < 0 ? 0 : (b > P.nav - 1 ? P.nav - 1 : b);
                double sig = T.av_sig[b];
                if (sig == 0.0) sig = 0.05;  // mft6.py:1237-1238
                const double z = (a_v - T.av_mu[b]) / sig;
                lp += -0.5 * (z * z);
            }
        }
        if (alive && P.has_prior) {
#pragma clang loop unroll(full)
            for (int k = 0; k < 2 * NS + 2; ++k) {
                if (P.pmean[k] != 0.0) {  // mft6.py:1258
                    const double z = (t[k] - P.pmean[k]) / P.psig[k];
                    lp += -0.5 * (z * z);
                }
            }
        }
        if (alive && P.rad_prior) {  // mft6.py:1262-1269
            double mr[NS];
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                if (!(t[s] >= T.iso_t[0]) || !(t[s] <= T.iso_t[P.niso - 1])) { st = MSX_W_VALUEERROR; mr[s] = 1.0; continue; }
                const int cnt = wave_count_le(T.iso_t, P.niso, t[s], lane);
                const double lum = interp_from_count(T.iso_t, T.iso_l, P.niso, t[s], cnt);
                const double sigma_sb = 5.670374e-5, lsun = 3.839e33;
                const double t2 = t[s] * t[s];
                mr[s] = sqrt(lum * lsun / (4 * M_PI * sigma_sb * (t2 * t2))) / kRsunCm;  // mft6.py:83
            }
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const double target = (s == 0) ? mr[0] : mr[s] / mr[0];
                const double z = (rad[s] - target) / (0.02 * target);
                lp += -0.5 * (z * z);
            }
        }
    }
    if (st != MSX_W_OK || !alive) {
        if (lane == 0) D.status = (st != MSX_W_OK) ? st : MSX_W_REJECT;
        return;
    }
    if (mode == MSX_MODE_LOGPRIOR) {
        if (lane == 0) { D.lp = lp; D.status = MSX_W_OK; }
        return;
    }
    // A1 + A2 + A4
    int node[NS * 4];
    double w[NS * 4];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        if (!(t[s] >= T.iso_t[0]) || !(t[s] <= T.iso_t[P.niso - 1])) { st = MSX_W_VALUEERROR; break; }
        const int cnt = wave_count_le(T.iso_t, P.niso, t[s], lane);
        const double lg = interp_from_count(T.iso_t, T.iso_g, P.niso, t[s], cnt);  // mft6.py:1149
        int t1, t2, g1, g2;
        double te1, te2, ge1, ge2;
        st = wave_bracket(T.teff_nodes, P.nt, t[s], lane, &t1, &t2, &te1, &te2);
        if (st == MSX_W_OK) st = wave_bracket(T.logg_nodes, P.ng, lg, lane, &g1, &g2, &ge1, &ge2);
        if (st != MSX_W_OK) break;
        const int n11 = t1 * P.ng + g1, n12 = t1 * P.ng + g2, n21 = t2 * P.ng + g1, n22 = t2 * P.ng + g2;
        if (!P.present[n11] || !P.present[n12] || !P.present[n21] || !P.present[n22]) { st = MSX_W_KEYERROR; break; }
        const double a = (g1 == g2) ? 0.0 : (lg - ge1) / (ge2 - ge1);
        const double b = (t1 == t2) ? 0.0 : (t[s] - te1) / (te2 - te1);
        const double di = 1.0 / plx;  // mft6.py:690
        const double r = (s == 0) ? rad[0] : rad[0] * rad[s];
        const double q = r * kRsunCm / (di * kPcCm);  // mft6.py:691,700
        const double sc = q * q;
        node[4 * s + 0] = n11; w[4 * s + 0] = (1.0 - b) * (1.0 - a) * sc;
        node[4 * s + 1] = n12; w[4 * s + 1] = (1.0 - b) * a * sc;
        node[4 * s + 2] = n21; w[4 * s + 2] = b * (1.0 - a) * sc;
        node[4 * s + 3] = n22; w[4 * s + 3] = b * a * sc;
    }
    if (st != MSX_W_OK) {
        if (lane == 0) D.status = st;
        return;
    }
    if (lane == 0) {
#pragma unroll
        for (int c = 0; c < NS * 4; ++c) { D.node[c] = node[c]; D.w[c] = w[c]; }
    }
    // same-wave LDS hand-off (lane 0 -> all lanes): LDS ops of one wave complete in order; the fence
    // keeps the compiler from moving the reads above the writes
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    // A5/A6: one (filter, star) or one photometric band per lane; magnitudes land in LDS
    const bool redden = redden_rule(mode, P.use_av, a_v);
    const int nb = P.nc + P.np;
    const int njobs = P.nc * NS + P.np;
    if (lane < njobs) {
        double val;
        if (lane < P.nc * NS) {
            const int f = lane / NS, s = lane - f * NS;
            double m = 0.0;
            for (int c = 0; c < 4; ++c) m += D.w[4 * s + c] * P.band_tab[(int64_t)D.node[4 * s + c] * nb + f];
            val = -2.5 * log10(m);  // mft6.py:733
        } else {
            const int f = lane - P.nc * NS;
            double flux = 0.0;
            for (int c = 0; c < NS * 4; ++c) flux += D.w[c] * P.band_tab[(int64_t)D.node[c] * nb + P.nc + f];
            val = -2.5 * log10(flux / P.pzero[f]);  // mft6.py:780-782
        }
        D.mag[lane] = val;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) {
        double chi = 0.0;
        for (int f = 0; f < P.nc; ++f) {
            int sec = 1;
            if (NS == 3 && f >= P.nc / 2) sec = 2;  // mft6.py:747-749
            const double con = D.mag[f * NS + sec] - D.mag[f * NS];  // mft6.py:741
            const double z = con - P.cmag[f];
            chi += (z * z) / (P.cerr[f] * P.cerr[f]);  // mft6.py:120,1182
        }
        for (int f = 0; f < P.np; ++f) {
            const double mag = D.mag[P.nc * NS + f];
            const double mred = redden ? mag + a_v * P.pk[f] : mag;  // mft6.py:1163
            const double z = mred - P.pmag[f];
            chi += (z * z) / (P.perr[f] * P.perr[f]);  // mft6.py:1188
        }
        D.chi_extra = chi;
        D.redc = redden ? -0.4 * kLog2Of10 * a_v : 0.0;
        D.lp = lp;
        D.status = MSX_W_OK;
    }
}

// ------------------------------------------------------------------------------------------------
// Phase 0, fast form: every small table is loaded ONCE into registers of wave 0 (one batch of
// independent loads), searches are ballots on registers and element fetches are v_readlane with a
// uniform index -- no dependent memory round trips.  Same arithmetic as build_recipe_wave.
// Limits (checked by the caller): niso <= 256, nt, ng <= 64, nt*ng <= 128, nav+1 <= 128.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double readlane_f64(double v, int l) {  // l must be wave-uniform
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double pick4(const double (&r)[4], int idx) {  // idx uniform, 0..255
    const int k = idx >> 6;
    const double v = (k == 0) ? r[0] : (k == 1) ? r[1] : (k == 2) ? r[2] : r[3];
    return readlane_f64(v, idx & 63);
}
__device__ __forceinline__ double pick2(const double (&r)[2], int idx) {  // idx uniform, 0..127
    return readlane_f64((idx >> 6) ? r[1] : r[0], idx & 63);
}
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

// np.interp on the register-resident isochrone; caller checked the range
__device__ __forceinline__ double iso_interp_regs(const double (&xs)[4], const double (&ys)[4], int n, double x) {
    int cnt = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) cnt += __popcll(__ballot(xs[k] <= x));  // pads are +inf
    const int j = uni(cnt) - 1;
    if (j >= n - 1) return pick4(ys, n - 1);
    const double x0 = pick4(xs, j), y0 = pick4(ys, j);
    if (x0 == x) return y0;
    const double slope = (pick4(ys, j + 1) - y0) / (pick4(xs, j + 1) - x0);
    return slope * (x - x0) + y0;
}

// sorted-node bracket on a register-resident node list (lane i holds nodes[i], pads +inf)
__device__ __forceinline__ int bracket_regs(double nodes, int n, double v, int *i1, int *i2, double *e1, double *e2) {
    const int j = uni(__popcll(__ballot(nodes <= v))) - 1;
    int best;
    if (j < 0) best = 0;
    else if (j >= n - 1) best = n - 1;
    else best = (fabs(readlane_f64(nodes, j + 1) - v) < fabs(readlane_f64(nodes, j) - v)) ? j + 1 : j;
    best = uni(best);
    const double nb = readlane_f64(nodes, best);
    int other;
    if (nb == v) other = best;
    else if (nb > v) other = best - 1;
    else other = best + 1;
    if (other == -1) other = n - 1;
    if (other >= n) return MSX_W_INDEXERROR;
    other = uni(other);
    *i1 = best;
    *i2 = other;
    *e1 = nb;
    *e2 = readlane_f64(nodes, other);
    return MSX_W_OK;
}

// Part 1 (gates phase A): finite + box check, A1 logg, A2 brackets, A4 weights.  Wave 0, before the
// first barrier.  Writes D.node, D.w, D.redc, D.status.
template <int NS>
__device__ __forceinline__ void recipe_part1_regs(const DevProblem &P, int mode, const double *__restrict__ th, WalkerDesc &D,
                                  int lane, int64_t wk, const int star) {
    // one wave per star (wave `star` of the block): the two or three dependent lookup chains run side by
    // side; every wave evaluates the (cheap) gates itself and reports through D.stat[star]
    // ---- one batch of independent loads -------------------------------------------------------------
    double t[2 * NS + 2];
#pragma unroll
    for (int k = 0; k < 2 * NS + 2; ++k) t[k] = th[k];
    double isot[4], isog[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = lane + kWave * k;
        const bool ok = i < P.niso;
        isot[k] = ok ? P.iso_t[i] : INFINITY;
        isog[k] = ok ? P.iso_g[i] : 0.0;
    }
    const double tn = lane < P.nt ? P.teff_nodes[lane] : INFINITY;
    const double gn = lane < P.ng ? P.logg_nodes[lane] : INFINITY;
    const int nn = P.nt * P.ng;
    const int pres0 = lane < nn ? (int)P.present[lane] : 0;
    const int pres1 = lane + kWave < nn ? (int)P.present[lane + kWave] : 0;
    MSX_STAMP(P, wk, 9);
    // ---- the hard gates of the prior (mft6.py:1227-1230 binary, :1347-1350 triple) -----------------------
    bool alive = true;
#pragma unroll
    for (int k = 0; k < 2 * NS + 2; ++k) alive = alive && isfinite(t[k]);  // emcee refuses non-finite coords
    const double a_v = t[NS];
    const double plx = t[2 * NS + 1];
    const double *rad = &t[NS + 1];
    if (mode == MSX_MODE_LOGPOST || mode == MSX_MODE_LOGPRIOR) {
        alive = alive && prior_gates<NS>(P, t);
    }
    if (!alive) {
        if (lane == 0) D.stat[star] = MSX_W_REJECT;
        return;
    }
    if (mode == MSX_MODE_LOGPRIOR) {  // no spectrum pass: the prior terms finish the job
        if (lane == 0) D.stat[star] = MSX_W_OK;
        return;
    }
    MSX_STAMP(P, wk, 10);
    // ---- A1 + A2 + A4 -----------------------------------------------------------------------------------
    const double iso_lo = pick4(isot, 0), iso_hi = pick4(isot, P.niso - 1);
    int st = MSX_W_OK;
    int node[4];
    double w[4];
    const double di = 1.0 / plx;  // mft6.py:690
    {
        const int s = star;
        do {
            if (!(t[s] >= iso_lo) || !(t[s] <= iso_hi)) { st = MSX_W_VALUEERROR; break; }
            const double lg = iso_interp_regs(isot, isog, P.niso, t[s]);  // mft6.py:1149
            int t1, t2, g1, g2;
            double te1, te2, ge1, ge2;
            st = bracket_regs(tn, P.nt, t[s], &t1, &t2, &te1, &te2);
            if (st == MSX_W_OK) st = bracket_regs(gn, P.ng, lg, &g1, &g2, &ge1, &ge2);
            if (st != MSX_W_OK) break;
            const int n11 = t1 * P.ng + g1, n12 = t1 * P.ng + g2, n21 = t2 * P.ng + g1, n22 = t2 * P.ng + g2;
            bool have = true;
            const int four[4] = {n11, n12, n21, n22};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int nd = uni(four[c]);
                have = have && (__builtin_amdgcn_readlane(nd < kWave ? pres0 : pres1, nd & 63) != 0);
            }
            if (!have) { st = MSX_W_KEYERROR; break; }
            const double a = (g1 == g2) ? 0.0 : (lg - ge1) / (ge2 - ge1);
            const double b = (t1 == t2) ? 0.0 : (t[s] - te1) / (te2 - te1);
            const double r = (s == 0) ? rad[0] : rad[0] * rad[s];
            const double q = r * kRsunCm / (di * kPcCm);  // mft6.py:691,700
            const double sc = q * q;
            node[0] = n11; w[0] = (1.0 - b) * (1.0 - a) * sc;
            node[1] = n12; w[1] = (1.0 - b) * a * sc;
            node[2] = n21; w[2] = b * (1.0 - a) * sc;
            node[3] = n22; w[3] = b * a * sc;
        } while (false);
    }
    if (lane == 0) {
        if (st == MSX_W_OK) {
#pragma unroll
            for (int c = 0; c < 4; ++c) { D.node[4 * star + c] = node[c]; D.w[4 * star + c] = w[c]; }
            if (star == 0) D.redc = redden_rule(mode, P.use_av, a_v) ? -0.4 * kLog2Of10 * a_v : 0.0;
        }
        D.stat[star] = st;
    }
    MSX_STAMP(P, wk, 11);
}

// Part 2 (off the critical path): the Gaussian prior terms (f1) and the contrast / photometry chi^2
// (A5/A6).  They are only read by the last lines of the kernel, so two otherwise idle waves compute
// them during the median's bin-scan stage (which keeps only wave 0 busy).  Both re-read theta and the
// small tables (L2 hits) instead of carrying registers across phase A.
template <int NS>
__device__ __forceinline__ void recipe_prior_terms(const DevProblem &P, int mode, const double *__restrict__ th, WalkerDesc &D,
                                   int lane) {
    double t[2 * NS + 2];
#pragma unroll
    for (int k = 0; k < 2 * NS + 2; ++k) t[k] = th[k];
    const double a_v = t[NS];
    const double plx = t[2 * NS + 1];
    const double *rad = &t[NS + 1];
    double lp = 0.0;
    int st = MSX_W_OK;
    if (mode == MSX_MODE_LOGPOST || mode == MSX_MODE_LOGPRIOR) {
        if (P.use_av && P.nav > 0) {
            double ave[2], avm[2], avs[2];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int i = lane + kWave * k;
                ave[k] = (i < P.nav + 1) ? P.av_edges[i] : INFINITY;
                avm[k] = (i < P.nav) ? P.av_mu[i] : 0.0;
                avs[k] = (i < P.nav) ? P.av_sig[i] : 0.0;
            }
            const double d = 1.0 / plx;  // pc, mft6.py:1233
            int b = uni(__popcll(__ballot(ave[0] <= d)) + __popcll(__ballot(ave[1] <= d))) - 1;
            b = b < 0 ? 0 : (b > P.nav - 1 ? P.nav - 1 : b);
            double sig = pick2(avs, b);
            if (sig == 0.0) sig = 0.05;  // mft6.py:1237-1238
            const double z = (a_v - pick2(avm, b)) / sig;
            lp += -0.5 * (z * z);
        }
        if (P.has_prior) {
#pragma clang loop unroll(full)
            for (int k = 0; k < 2 * NS + 2; ++k) {
                if (P.pmean[k] != 0.0) {  // mft6.py:1258
                    const double z = (t[k] - P.pmean[k]) / P.psig[k];
                    lp += -0.5 * (z * z);
                }
            }
        }
        if (P.rad_prior) {  // mft6.py:1262-1269
            double isot[4], isol[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int i = lane + kWave * k;
                const bool ok = i < P.niso;
                isot[k] = ok ? P.iso_t[i] : INFINITY;
                isol[k] = ok ? P.iso_l[i] : 0.0;
            }
            const double iso_lo = pick4(isot, 0), iso_hi = pick4(isot, P.niso - 1);
            double mr[NS];
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                if (!(t[s] >= iso_lo) || !(t[s] <= iso_hi)) { st = MSX_W_VALUEERROR; mr[s] = 1.0; continue; }
                const double lum = iso_interp_regs(isot, isol, P.niso, t[s]);
                const double sigma_sb = 5.670374e-5, lsun = 3.839e33;
                const double t2 = t[s] * t[s];
                mr[s] = sqrt(lum * lsun / (4 * M_PI * sigma_sb * (t2 * t2))) / kRsunCm;  // mft6.py:83
            }
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const double target = (s == 0) ? mr[0] : mr[s] / mr[0];
                const double z = (rad[s] - target) / (0.02 * target);
                lp += -0.5 * (z * z);
            }
        }
    }
    if (lane == 0) {
        D.lp = lp;
        if (mode == MSX_MODE_LOGPRIOR) D.status = st;  // only reachable there: part 1 range-checked Teff otherwise
    }
}

template <int NS>
__device__ __forceinline__ void recipe_band_terms(const DevProblem &P, int mode, const double *__restrict__ th, WalkerDesc &D,
                                  int lane) {
    const double a_v = th[NS];
    const bool redden = redden_rule(mode, P.use_av, a_v);
    const int nb = P.nc + P.np;
    const int njobs = P.nc * NS + P.np;
    double val = 0.0;
    if (lane < njobs) {  // one (filter, star) or one photometric band per lane
        if (lane < P.nc * NS) {
            const int f = lane / NS, s = lane - f * NS;
            double m = 0.0;
            for (int c = 0; c < 4; ++c) m += D.w[4 * s + c] * P.band_tab[(int64_t)D.node[4 * s + c] * nb + f];
            val = -2.5 * log10(m);  // mft6.py:733
        } else {
            const int f = lane - P.nc * NS;
            double flux = 0.0;
            for (int c = 0; c < NS * 4; ++c) flux += D.w[c] * P.band_tab[(int64_t)D.node[c] * nb + P.nc + f];
            val = -2.5 * log10(flux / P.pzero[f]);  // mft6.py:780-782
        }
    }
    double chi = 0.0;
    for (int f = 0; f < P.nc; ++f) {
        int sec = 1;
        if (NS == 3 && f >= P.nc / 2) sec = 2;  // mft6.py:747-749
        const double con = readlane_f64(val, f * NS + sec) - readlane_f64(val, f * NS);  // mft6.py:741
        const double z = con - P.cmag[f];
        chi += (z * z) * P.civar[f];  // mft6.py:120,1182
    }
    for (int f = 0; f < P.np; ++f) {
        const double mag = readlane_f64(val, P.nc * NS + f);
        const double mred = redden ? mag + a_v * P.pk[f] : mag;  // mft6.py:1163
        const double z = mred - P.pmag[f];
        chi += (z * z) * P.pivar[f];  // mft6.py:1188
    }
    if (lane == 0) D.chi_extra = chi;
}

// ------------------------------------------------------------------------------------------------
// block scratch.  Every reduction site has its own slots so that one barrier per reduction suffices
// (fixed order everywhere: lanes via shuffles, then waves 0..nw-1 serially -> deterministic).
// ------------------------------------------------------------------------------------------------
constexpr int kBins = 1024;     // linear value bins of the median select
constexpr int kLogBins = 2048;  // logarithmic bins of the early-histogram median (3 exponent + 8 mantissa bits)
struct alignas(16) BlockScratch {
    double q[3][kMaxWaves];
    unsigned long long kmin[kMaxWaves], kmax[kMaxWaves];
    unsigned long long above[kMaxWaves];
    double chi[kMaxWaves];
    unsigned int wave_tot[kMaxWaves];
    unsigned int hist[kLogBins];  // block_median and radix_select use the first kBins / 256
    unsigned long long cand[kSelectFinish];
    unsigned long long sel_result[2];
    unsigned int sel_bin, sel_k, sel_cnt, cand_n, has_second;
    unsigned int cnt_le;
};

// Exact k-th smallest (0-based) of the keys of model[0..npix) by MSB radix passes; the general,
// always-terminating fallback of the median.  Uses hist[0..256).  All threads must call it.
__device__ __forceinline__ unsigned long long radix_select(const double *model, int npix, unsigned int k, unsigned long long kmin,
                                           unsigned long long kmax, BlockScratch &S) {
    const int tid = threadIdx.x, B = blockDim.x, lane = tid & 63, wave = tid >> 6;
    if (kmin == kmax) return kmin;
    const int hb = 63 - __clzll((long long)(kmin ^ kmax));  // highest differing bit
    int shift = hb + 1;
    unsigned long long pmask = (shift >= 64) ? 0ull : ~((1ull << shift) - 1ull);
    unsigned long long pval = kmin & pmask;
    unsigned long long v1 = kmin;
    bool done = false;
    while (shift > 0 && !done) {
        const int bits = shift < 8 ? shift : 8;
        shift -= bits;
        const unsigned int dmask = (1u << bits) - 1u;
        if (tid < 256) S.hist[tid] = 0;
        __syncthreads();
        for (int p = tid; p < npix; p += B) {
            const unsigned long long key = key_of(model[p]);
            if ((key & pmask) == pval) atomicAdd(&S.hist[(unsigned int)(key >> shift) & dmask], 1u);
        }
        __syncthreads();
        if (wave == 0) {  // locate the bin holding rank k: 4 bins per lane + wave inclusive scan
            const unsigned int c0 = S.hist[4 * lane], c1 = S.hist[4 * lane + 1], c2 = S.hist[4 * lane + 2],
                               c3 = S.hist[4 * lane + 3];
            const unsigned int tot = c0 + c1 + c2 + c3;
            const unsigned int inc = wave_scan_u32(tot);
            const unsigned long long ball = __ballot(inc > k);
            const int L = __ffsll((long long)ball) - 1;
            if (lane == L) {
                unsigned int kk = k - (inc - tot);
                unsigned int bin, cnt;
                if (kk < c0) { bin = 0; cnt = c0; }
                else if ((kk -= c0) < c1) { bin = 1; cnt = c1; }
                else if ((kk -= c1) < c2) { bin = 2; cnt = c2; }
                else { kk -= c2; bin = 3; cnt = c3; }
                S.sel_bin = 4 * L + bin;
                S.sel_k = kk;
                S.sel_cnt = cnt;
                S.cand_n = 0;
            }
        }
        __syncthreads();
        k = S.sel_k;
        const unsigned int cnt = S.sel_cnt;
        pval |= ((unsigned long long)S.sel_bin) << shift;
        pmask |= ((unsigned long long)dmask) << shift;
        if (shift == 0) {
            v1 = pval;  // every remaining candidate equals the prefix
            done = true;
        } else if (cnt <= (unsigned int)kSelectFinish) {
            for (int p = tid; p < npix; p += B) {
                const unsigned long long key = key_of(model[p]);
                if ((key & pmask) == pval) S.cand[atomicAdd(&S.cand_n, 1u)] = key;
            }
            __syncthreads();
            if (tid < (int)cnt) {
                const unsigned long long mine = S.cand[tid];
                unsigned int r = 0;
                for (unsigned int j = 0; j < cnt; ++j) {
                    const unsigned long long o = S.cand[j];
                    r += (o < mine) || (o == mine && j < (unsigned int)tid);
                }
                if (r == k) S.sel_result[0] = mine;
            }
            __syncthreads();
            v1 = S.sel_result[0];
            done = true;
        }
        __syncthreads();
    }
    return v1;
}

#ifdef MSX_STAMPS
__device__ unsigned long long g_med_stamps[65536 * 8];
#define MED_STAMP(i) do { if (threadIdx.x == 0) g_med_stamps[blockIdx.x * 8 + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define MED_STAMP(i) do { } while (0)
#endif

// ------------------------------------------------------------------------------------------------
// Exact np.median of v[0..npix) held in LDS, given the min / max keys of the vector.  All threads of
// the block call it; S.hist[0..kBins) must be zero on entry (it is left dirty).
//   1024 linear value bins between min and max (a monotone map, so ranks resolve bin by bin) -> block
//   scan -> the <= 256 candidates of the median's bin are ranked all-pairs; the upper middle value
//   comes from the same ranking or from the minimum of the higher bins.  Distributions that defeat
//   the binning (heavy duplication, infinities) fall back to the bitwise radix select.
// ------------------------------------------------------------------------------------------------
struct NoSide {
    __device__ __forceinline__ void operator()() const {}
};
// Per-element work that can ride along the median's first pass over the vector (it already reads every
// element): process4() gets four (index, value, valid) triples, flush() publishes the wave partials
// right before the pass's barrier.
struct NoElem {
    __device__ __forceinline__ void process4(const int (&)[4], const double (&)[4], const bool (&)[4]) {}
    __device__ __forceinline__ void flush(BlockScratch &) {}
};

__device__ __forceinline__ unsigned long long readlane_u64(unsigned long long v, int l) {  // l wave-uniform
    const unsigned int lo = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)v, l);
    const unsigned int hi = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)(v >> 32), l);
    return ((unsigned long long)hi << 32) | lo;
}

template <class Side, class Elem>
__device__ __forceinline__ double block_median(const double *model, int npix, unsigned long long kmin, unsigned long long kmax,
                               BlockScratch &S, Side side, Elem &elem, bool *elem_done) {
    bool side_done = false;  // `side` runs exactly once, preferably in the stage that keeps only wave 0 busy
    *elem_done = false;
    const int tid = threadIdx.x, B = blockDim.x, lane = tid & 63, wave = tid >> 6, nw = B >> 6;
    // k1 = lower middle rank (0-based); for even npix the median averages ranks k1 and k1+1.
    const unsigned int k1 = (unsigned int)((npix - 1) >> 1);
    const bool need_two = (npix & 1) == 0;
    unsigned long long v1 = kmin, v2 = kmin;
    if (kmin != kmax) {
        const double vmin = val_of(kmin), vmax = val_of(kmax);
        // monotone map value -> bin: (x - vmin) * scale is non-decreasing in x, so every key in a lower
        // bin is <= every key in a higher bin and ranks can be resolved bin by bin.
        const double scale = (double)kBins / (vmax - vmin);
        const bool lin_ok = isfinite(scale) && scale > 0.0;
        bool solved = false;
        MED_STAMP(0);
        if (lin_ok) {
            for (int base = 0; base < npix; base += 4 * B) {  // 4 elements per trip: loads first, then use
                int pp[4];
                double xv[4];
                bool ok[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int p = base + u * B + tid;
                    ok[u] = p < npix;
                    pp[u] = ok[u] ? p : npix - 1;
                    xv[u] = model[pp[u]];
                }
                elem.process4(pp, xv, ok);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    int bin = (int)((xv[u] - vmin) * scale);
                    bin = bin > kBins - 1 ? kBins - 1 : bin;
                    if (ok[u]) atomicAdd(&S.hist[bin], 1u);
                }
            }
            elem.flush(S);
            *elem_done = true;
            __syncthreads();
            MED_STAMP(1);
            // bin scan by wave 0 alone (16 bins per lane + one wave scan); the other waves are idle here,
            // so two of them do the walker's off-critical-path side work meanwhile
            side();
            side_done = true;
            if (wave == 0) {
                constexpr int per = kBins / kWave;
                unsigned int own = 0;
                const uint4 *h4 = reinterpret_cast<const uint4 *>(&S.hist[lane * per]);
#pragma unroll
                for (int i = 0; i < per / 4; ++i) {
                    const uint4 h = h4[i];
                    own += h.x + h.y + h.z + h.w;
                }
                const unsigned int inc = wave_scan_u32(own);
                const unsigned int excl = inc - own;
                if (own > 0 && excl <= k1 && k1 < excl + own) {  // exactly one lane
                    unsigned int kk = k1 - excl;
                    int bin = lane * per;
                    unsigned int cnt = S.hist[bin];
                    while (kk >= cnt) { kk -= cnt; ++bin; cnt = S.hist[bin]; }
                    S.sel_bin = (unsigned int)bin;
                    S.sel_k = kk;
                    S.sel_cnt = cnt;
                    S.cand_n = 0;
                    S.has_second = 0;
                }
            }
            __syncthreads();
            MED_STAMP(2);
            const unsigned int cnt = S.sel_cnt, kk = S.sel_k;
            const int sel = (int)S.sel_bin;
            if (cnt <= (unsigned int)kSelectFinish) {
                // gather the candidates of the selected bin; keep the smallest key of the higher bins
                unsigned long long above = ~0ull;
                for (int base = 0; base < npix; base += 4 * B) {
                    double xv[4];
                    bool ok[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int p = base + u * B + tid;
                        ok[u] = p < npix;
                        xv[u] = model[ok[u] ? p : npix - 1];
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        int bin = (int)((xv[u] - vmin) * scale);
                        bin = bin > kBins - 1 ? kBins - 1 : bin;
                        const unsigned long long key = key_of(xv[u]);
                        if (ok[u] && bin == sel) S.cand[atomicAdd(&S.cand_n, 1u)] = key;
                        else if (ok[u] && bin > sel && key < above) above = key;
                    }
                }
                above = wave_min_u64(above);
                if (lane == 0) S.above[wave] = above;
                __syncthreads();
                MED_STAMP(3);
                if (cnt <= (unsigned int)kWave) {
                    // all-pairs rank inside wave 0: one candidate per lane, the others arrive by readlane
                    if (wave == 0) {
                        const unsigned long long mine = lane < (int)cnt ? S.cand[lane] : ~0ull;
                        unsigned int r = 0;
                        for (int j = 0; j < (int)cnt; ++j) {
                            const unsigned long long o = readlane_u64(mine, j);
                            r += (o < mine) || (o == mine && j < lane);
                        }
                        if (lane < (int)cnt && r == kk) S.sel_result[0] = mine;
                        if (lane < (int)cnt && r == kk + 1) { S.sel_result[1] = mine; S.has_second = 1; }
                    }
                } else if (tid < (int)cnt) {  // all-pairs rank through LDS, ties broken by slot
                    const unsigned long long mine = S.cand[tid];
                    unsigned int r = 0;
                    for (unsigned int j = 0; j < cnt; ++j) {
                        const unsigned long long o = S.cand[j];
                        r += (o < mine) || (o == mine && j < (unsigned int)tid);
                    }
                    if (r == kk) S.sel_result[0] = mine;
                    if (r == kk + 1) { S.sel_result[1] = mine; S.has_second = 1; }
                }
                __syncthreads();
                MED_STAMP(4);
#ifdef MSX_STAMPS
                if (tid == 0) g_med_stamps[blockIdx.x * 8 + 6] = cnt;
#endif
                v1 = S.sel_result[0];
                if (S.has_second) {
                    v2 = S.sel_result[1];
                } else {
                    v2 = S.above[0];
                    for (int x = 1; x < nw; ++x) v2 = S.above[x] < v2 ? S.above[x] : v2;
                }
                solved = true;
            }
        }
        if (!solved) {  // adversarial value distribution: bitwise radix select (always terminates)
            __syncthreads();
            v1 = radix_select(model, npix, k1, kmin, kmax, S);
            v2 = v1;
            if (need_two) {
                // rank k1+1: equals v1 when v1 is duplicated past rank k1, else the smallest key above v1
                unsigned int cle = 0;
                unsigned long long nxt = ~0ull;
                for (int p = tid; p < npix; p += B) {
                    const unsigned long long key = key_of(model[p]);
                    cle += key <= v1;
                    if (key > v1 && key < nxt) nxt = key;
                }
                if (tid == 0) S.cnt_le = 0;
                __syncthreads();
                const unsigned int wc0 = wave_scan_u32(cle);
                const unsigned int wc = (unsigned int)__builtin_amdgcn_readlane((int)wc0, 63);
                nxt = wave_min_u64(nxt);
                if (lane == 0) { atomicAdd(&S.cnt_le, wc); S.above[wave] = nxt; }
                __syncthreads();
                v2 = S.above[0];
                for (int x = 1; x < nw; ++x) v2 = S.above[x] < v2 ? S.above[x] : v2;
                if (S.cnt_le >= k1 + 2) v2 = v1;
            }
        }
    }
    // np.median: mean of the two middle values for even npix
    if (!side_done) side();
    return need_two ? (val_of(v1) + val_of(v2)) / 2.0 : val_of(v1);
}

// ------------------------------------------------------------------------------------------------
// The same exact median when the histogram was filled DURING phase A.  That needs a bin map that does not
// depend on the vector's min / max: for positive finite doubles the low 3 exponent bits and the top 8
// mantissa bits (256 logarithmic sub-bins per binade, cyclic in the exponent).  The map is monotone along the
// cycle starting at min's bin as long as the vector spans < 8 binades, which is checked here from min / max
// (anything else -- zeros, negatives, infinities, huge ranges -- returns false and the caller takes
// block_median).  On entry: S.hist complete (a barrier has passed), S.cand_n == 0, S.has_second == 0.
//   every wave scans the 2048 counters itself (no publish, no barrier) -> ONE pass over the vector does the
//   chi^2 terms (elem) and gathers the median bin's candidates -> barrier -> wave 0 ranks the candidates in
//   registers.  The result is valid in wave 0 only (its lane 0 finishes the walker): no closing barrier.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned int logbin(double x) {
    return ((unsigned int)__double2hiint(x) >> 12) & (unsigned int)(kLogBins - 1);
}

template <class Elem>
__device__ __forceinline__ bool logbin_median(const double *model, int npix, unsigned long long kmin, unsigned long long kmax,
                                              BlockScratch &S, Elem &elem, double *med_out) {
    const int tid = threadIdx.x, B = blockDim.x, lane = tid & 63, wave = tid >> 6, nw = B >> 6;
    const unsigned int k1 = (unsigned int)((npix - 1) >> 1);
    const bool need_two = (npix & 1) == 0;
    if (!(kmin > key_of(0.0)) || kmin == kmax) return false;
    const unsigned int hmin = (unsigned int)__double2hiint(val_of(kmin)) >> 12;
    const unsigned int hmax = (unsigned int)__double2hiint(val_of(kmax)) >> 12;
    constexpr int per = kLogBins / kWave;  // counters per lane in the scan
    // scan origin: min's bin rounded down to a lane's group, so no group straddles the end of the array
    const unsigned int a = (hmin & (unsigned int)(kLogBins - 1)) & ~(unsigned int)(per - 1);
    if ((hmax - hmin) + (hmin & (unsigned int)(per - 1)) >= (unsigned int)kLogBins) return false;  // the cycle would lap itself
    MED_STAMP(0);
    MED_STAMP(1);
    // ---- per-wave scan: lane l owns the `per` counters from physical bin (a + per*l) mod kLogBins ----------
    const unsigned int phys = (a + (unsigned int)(per * lane)) & (unsigned int)(kLogBins - 1);
    unsigned int own = 0;
    {
        const uint4 *h4 = reinterpret_cast<const uint4 *>(&S.hist[phys]);
#pragma unroll
        for (int i = 0; i < per / 4; ++i) {
            const uint4 h = h4[i];
            own += h.x + h.y + h.z + h.w;
        }
    }
    const unsigned int inc = wave_scan_u32(own);
    const unsigned int excl = inc - own;
    const bool mine_it = own > 0 && excl <= k1 && k1 < excl + own;      // exactly one lane (total = npix > k1)
    const int L = uni(__ffsll((long long)__ballot(mine_it)) - 1);
    // second level, again on the whole wave: lane j < per takes counter j of lane L's group
    const unsigned int phys_l = (unsigned int)__builtin_amdgcn_readlane((int)phys, L);
    const unsigned int t = k1 - (unsigned int)__builtin_amdgcn_readlane((int)excl, L);
    const unsigned int c = lane < per ? S.hist[phys_l + lane] : 0u;
    const unsigned int inc2 = wave_scan_u32(c);
    const int J = uni(__ffsll((long long)__ballot(c > 0 && inc2 - c <= t && t < inc2)) - 1);
    const unsigned int kk = t - (unsigned int)__builtin_amdgcn_readlane((int)(inc2 - c), J);
    const unsigned int cnt = (unsigned int)__builtin_amdgcn_readlane((int)c, J);
    const unsigned int lsel = (phys_l + (unsigned int)J - a) & (unsigned int)(kLogBins - 1);
    if (cnt > (unsigned int)kSelectFinish) return false;  // heavy duplication: the general path sorts it out
    MED_STAMP(2);
    // ---- one pass: chi^2 terms + candidates of the median's bin + smallest key of the later bins -----------
    unsigned long long above = ~0ull;
    for (int base = 0; base < npix; base += 4 * B) {
        int pp[4];
        double xv[4];
        bool ok[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int p = base + u * B + tid;
            ok[u] = p < npix;
            pp[u] = ok[u] ? p : npix - 1;
            xv[u] = model[pp[u]];
        }
        elem.process4(pp, xv, ok);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const unsigned int lx = (logbin(xv[u]) - a) & (unsigned int)(kLogBins - 1);
            const unsigned long long key = key_of(xv[u]);
            if (ok[u] && lx == lsel) S.cand[atomicAdd(&S.cand_n, 1u)] = key;
            else if (ok[u] && lx > lsel && key < above) above = key;
        }
    }
    elem.flush(S);
    above = wave_min_u64(above);
    if (lane == 0) S.above[wave] = above;
    __syncthreads();
    MED_STAMP(3);
    // ---- rank.  Up to 64 candidates (the usual case): wave 0 alone, in registers, and only wave 0 (whose lane 0
    // finishes the walker) learns the median -- no further barrier.  More: the first waves through LDS.
    unsigned long long v1 = 0, v2 = 0;
    bool second = false;
    if (cnt <= (unsigned int)kWave) {
        if (wave == 0) {
            // one candidate per lane; the others arrive as LDS broadcast reads, eight per trip.  Counting the keys
            // below and not above a candidate pins its VALUE's rank interval [lt, le), which is all the median needs
            // (duplicates share a value), so no tie-break by slot.  Pad slots hold ~0 and rank last.
            if (lane < 8) S.cand[cnt + lane] = ~0ull;   // same wave: LDS operations execute in order
            const unsigned long long mine = S.cand[lane < (int)cnt ? lane : (int)cnt];
            unsigned int lt = 0, le = 0;
            const ulonglong2 *c2 = reinterpret_cast<const ulonglong2 *>(S.cand);
            for (int j = 0; j < (int)cnt; j += 8) {
                ulonglong2 x[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) x[u] = c2[(j >> 1) + u];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    lt += (x[u].x < mine) + (x[u].y < mine);
                    le += (x[u].x <= mine) + (x[u].y <= mine);
                }
            }
            const unsigned long long b1 = __ballot(lane < (int)cnt && lt <= kk && kk < le);
            const unsigned long long b2 = __ballot(lane < (int)cnt && lt <= kk + 1 && kk + 1 < le);
            v1 = readlane_u64(mine, uni(__ffsll((long long)b1) - 1));
            second = b2 != 0ull;
            if (second) v2 = readlane_u64(mine, uni(__ffsll((long long)b2) - 1));
        }
    } else {
        if (tid < (int)cnt) {
            const unsigned long long mine = S.cand[tid];
            unsigned int r = 0;
            for (unsigned int j = 0; j < cnt; ++j) {
                const unsigned long long o = S.cand[j];
                r += (o < mine) || (o == mine && j < (unsigned int)tid);
            }
            if (r == kk) S.sel_result[0] = mine;
            if (r == kk + 1) { S.sel_result[1] = mine; S.has_second = 1; }
        }
        __syncthreads();
        v1 = S.sel_result[0];
        second = S.has_second != 0;
        if (second) v2 = S.sel_result[1];
    }
    MED_STAMP(4);
#ifdef MSX_STAMPS
    if (tid == 0) g_med_stamps[blockIdx.x * 8 + 6] = cnt;
#endif
    if (!second) {
        v2 = S.above[0];
        for (int x = 1; x < nw; ++x) v2 = S.above[x] < v2 ? S.above[x] : v2;
    }
    *med_out = need_two ? (val_of(v1) + val_of(v2)) / 2.0 : val_of(v1);
    return true;
}

// Last lines of a walker (one lane): publish the value and, for the device-resident sampler, apply the
// stretch move's accept rule  log(u) < (ndim-1) ln z + ln p(q) - ln p(s)  (NaN differences compare false,
// like -inf - -inf on the host) and record the walker's row of the chain: a walker only changes in its
// own half-step, so its row after the step is written here.
__device__ __forceinline__ void walker_done(const DevProblem &P, const WalkerDesc &D, int64_t wk, int ndim, double out, int st,
                            double *__restrict__ logp, int32_t *__restrict__ status) {
    logp[wk] = out;
    status[wk] = st;
    if (!P.smp_on) return;
    if (st > MSX_W_REJECT) atomicMax(P.smp_worst, st);
    const int64_t s = D.smp_s;
    const double lnpdiff = (D.smp_zfac + out) - D.smp_old;
    const bool acc = D.smp_logu < lnpdiff;
    if (acc) {
        P.smp_logp[s] = out;
        P.smp_naccept[s] = D.smp_nacc + 1;
    }
    for (int d = 0; d < ndim; ++d) {
        const double v = acc ? D.theta[d] : D.smp_sv[d];
        if (acc) P.smp_coords[s * ndim + d] = v;
        P.smp_chain_row[s * ndim + d] = v;
    }
    P.smp_lp_row[s] = acc ? out : D.smp_old;
}

// ------------------------------------------------------------------------------------------------
// THE HOT KERNEL: one workgroup per walker.
//   phase 0  wave 0 builds the walker's recipe on 64 lanes (prior gate, A1, A2, A4, A5, A6)
//   phase A  blend + redden + resample into LDS; fit sums; value range           (A2, A4, A7, A8.1)
//   phase B  exact median: 1024 linear value bins -> <=256 candidates -> all-pairs rank
//            (falls back to the bitwise radix select for adversarial distributions)     (A8.2)
//   phase C  continuum fit coefficients, chi^2                                          (A8.3, A9)
// ------------------------------------------------------------------------------------------------
extern __shared__ __attribute__((aligned(16))) unsigned char dyn_lds[];

// MAXT = largest workgroup the variant is launched with.  The 512-thread variants may use up to 256
// VGPRs (the two-pixels-per-trip body wants ~146: no spills, 3 waves per SIMD); the 1024-thread variants
// are capped at 128 VGPRs by the hardware and use one pixel per trip.
// GM = the walker's model vector lives in global memory (spectra longer than ~19k pixels) instead of LDS.
// PF = while the recipe waves work, the idle waves copy three walker-independent pixel vectors (u, data flux,
//      resample weight t) into LDS; phase A and the chi^2 pass then read them from LDS, which takes 160 of the
//      786 KB a walker pulls through its CU's L2 port off the critical path (one workgroup per CU only: 4 npix
//      doubles of LDS).
template <int NS, int U, int MAXT, bool GM = false, bool CP = false, bool PF = false>
__global__ void __launch_bounds__(MAXT, MAXT == 256 ? 3 : (MAXT == 512 && U == 1) ? 2 : 1)
logprob_kernel(DevProblem P, int mode, const double *theta, int64_t n, int ndim,
               double *__restrict__ logp, int32_t *__restrict__ status) {
    __shared__ WalkerDesc D;
    __shared__ BlockScratch S;
    const int64_t wk = blockIdx.x;
    if (wk >= n) return;
    double *model = GM ? P.model_scratch + wk * P.npix : reinterpret_cast<double *>(dyn_lds);  // [npix]
    const int tid = threadIdx.x;
    constexpr int B = MAXT;  // every variant is launched with exactly MAXT threads (msx_logprob_batch_dev)
    const int lane = tid & 63, wave = tid >> 6;
    constexpr int nw = B >> 6;
    const int npix = (int)P.npix;
    double *const lds_u = PF ? reinterpret_cast<double *>(dyn_lds) + npix : nullptr;
    double *const lds_f = PF ? lds_u + npix : nullptr;
    double *const lds_t = PF ? lds_f + npix : nullptr;

    MSX_STAMP(P, wk, 0);
    MSX_STAMP(P, wk, 8);
    const double *th_row = theta + wk * ndim;
    if (P.smp_on) {  // stretch-move proposal q = c - (c - s) z for this walker (mft6.py:1494 drives emcee's move)
        // two dependent levels only: {own index, complement index, z} -> the two coordinate rows.  The proposal
        // goes to LDS (the recipe waves read it there, no round trip through memory); wave 1 meanwhile fetches
        // what the accept step will need at the very end.
        if (tid < ndim) {
#pragma clang fp contract(off)
            // no FMA contraction: the proposal must have the bits NumPy's `c - (c - s) * z` produces so that
            // the device-resident and the host-driven sampler stay in lock-step
            const int64_t si = P.smp_sidx[wk], ci = P.smp_partner[wk];
            const double zz = P.smp_zz[wk];
            const double sv = P.smp_coords[si * ndim + tid];
            const double cv = P.smp_coords[ci * ndim + tid];
            const double diff = cv - sv;
            const double prod = diff * zz;
            const double qv = cv - prod;
            D.theta[tid] = qv;
            D.smp_sv[tid] = sv;
            P.smp_q[wk * ndim + tid] = qv;  // (kept for inspection; nothing reads it back)
        } else if (tid == kWave) {
            const int64_t si = P.smp_sidx[wk];
            D.smp_s = si;
            D.smp_old = P.smp_logp[si];
            D.smp_nacc = P.smp_naccept[si];
            D.smp_zfac = P.smp_zfac[wk];
            D.smp_logu = P.smp_logu[wk];
        }
        __syncthreads();
        th_row = D.theta;
    }
    for (int i = tid; i < kLogBins; i += B) S.hist[i] = 0;
    if (tid == 0) { S.cand_n = 0; S.has_second = 0; }
    // register-resident tables when they fit one wave (the usual case), else the generic walk
    const bool fast = P.niso <= 4 * kWave && P.nt <= kWave && P.ng <= kWave && P.nt * P.ng <= 2 * kWave &&
                      P.nav + 1 <= 2 * kWave;
    // Early-histogram path (logbin_median): the median's histogram is filled while phase A computes the model,
    // and the walker's prior terms move to an idle wave of phase 0.  Likelihood / posterior / chi^2 modes with
    // the register-resident recipe and the model vector in LDS; everything else keeps block_median.
    const bool early = !GM && fast && !P.no_spectrum &&
                       (mode == MSX_MODE_LOGLIKE || mode == MSX_MODE_LOGPOST || mode == MSX_MODE_CHISQ);
    // the prior terms (f1) depend on theta alone: an idle wave computes them beside the recipe waves, for every
    // mode (rejected walkers never read them)
    if (fast && wave == NS) recipe_prior_terms<NS>(P, mode, th_row, D, lane);
    if (PF && wave > NS) {  // the waves with no recipe work stage pixel statics (published by the barrier below)
        const int nthr = B - (NS + 1) * kWave, id = tid - (NS + 1) * kWave;
#pragma unroll 4
        for (int p = id; p < npix; p += nthr) {
            lds_u[p] = P.pix_u[p];
            lds_f[p] = P.pix_flux[p];
            lds_t[p] = P.pix_t[p];
        }
    }
    if (fast) {
        if (wave < NS) recipe_part1_regs<NS>(P, mode, th_row, D, lane, wk, wave);
    } else if (wave == 0) {
        const RecipeTabs T = {P.iso_t, P.iso_g, P.iso_l, P.av_edges, P.av_mu, P.av_sig, P.teff_nodes, P.logg_nodes};
        build_recipe_wave<NS>(P, T, mode, th_row, ndim, D, lane, wk);
    }
    __syncthreads();
    int wst = D.status;
    if (fast) {  // first star that failed decides, like the reference's star-by-star loop
        wst = D.stat[0];
#pragma unroll
        for (int k = 1; k < NS; ++k) wst = (wst == MSX_W_OK) ? D.stat[k] : wst;
    }
    if (wst != MSX_W_OK) {
        if (tid == 0) {
            walker_done(P, D, wk, ndim, (wst == MSX_W_REJECT) ? -INFINITY : NAN, wst, logp, status);
        }
        return;
    }
    if (mode == MSX_MODE_LOGPRIOR) {  // logprior alone (mft6.py:1207-1272): no spectrum pass
        if (wave == 0) {  // (fast recipe: wave NS left D.lp / D.status before the barrier above)
            if (lane == 0) {
                logp[wk] = (D.status == MSX_W_OK) ? D.lp : NAN;
                status[wk] = D.status;
            }
        }
        return;
    }
    if (P.no_spectrum) {
        // the mft6_nospec.py variant (mft6_nospec.py:1163-1196): the spectrum term is commented out there and
        // the total is contrast + photometry chi^2 only -- no spectral phases at all
        if (fast && wave == 2) recipe_band_terms<NS>(P, mode, th_row, D, lane);
        __syncthreads();
        if (tid == 0) {
            const double total = D.chi_extra;
            const bool chi_valued = mode == MSX_MODE_CHISQ || mode == MSX_MODE_OPT_STEP || mode == MSX_MODE_OPT_INIT;
            walker_done(P, D, wk, ndim, chi_valued ? total : (isnan(total) ? -INFINITY : D.lp + (-0.5 * total)), MSX_W_OK, logp,
                        status);
        }
        return;
    }
    MSX_STAMP(P, wk, 1);

    // ---- phase A ------------------------------------------------------------------------------------
    const double2 *rows[NS * 4];
    const PairC *rows_c[NS * 4];
    double w[NS * 4];
#pragma unroll
    for (int c = 0; c < NS * 4; ++c) {
        const int64_t off = (int64_t)__builtin_amdgcn_readfirstlane(D.node[c]) * npix;
        rows[c] = P.pairs + off;
        rows_c[c] = CP ? P.pairs_c + off : nullptr;
        w[c] = D.w[c];
    }
    const double redc = D.redc;
    const bool redden = redc != 0.0;
    // Sums are taken in an order that does not depend on the workgroup size: pixel p belongs to row p / B of the
    // launch, i.e. to 64-pixel chunk c = p / 64, and chunk c is owned by VIRTUAL wave c mod 16.  A real wave of a
    // 4- or 8-wave workgroup plays 4 or 2 virtual waves (its rows alternate between them), each with its own
    // accumulator; every virtual wave sees its chunks in ascending order, lanes are reduced by the same DPP tree
    // and the 16 partials are added serially -- the same association for 256, 512 and 1024 threads, so a
    // walker's log-probability has the same bits whatever launch (batch size, shard, rank) evaluates it.
    // (every variant is launched with exactly MAXT threads, so the count is a compile-time constant)
    constexpr int vk = kMaxWaves / (MAXT / kWave);  // virtual waves per real wave: 4, 2 or 1
    double qa[vk][3];
#pragma unroll
    for (int k = 0; k < vk; ++k) qa[k][0] = qa[k][1] = qa[k][2] = 0.0;
    double q[3];
    unsigned long long kmin = ~0ull, kmax = 0ull;
    // rows are taken vk at a time (SUB sub-trips of U rows) so that every row's virtual-wave slot is static
    constexpr int SUB = (vk > U) ? vk / U : 1;
    for (int base0 = 0; base0 < npix; base0 += B * U * SUB) {
#pragma unroll
      for (int sub = 0; sub < SUB; ++sub) {
        const int base = base0 + sub * B * U;
        if (base >= npix) break;
        double2 v[U][NS * 4];
        double2 kk[U];
        double tt[U], ff[U], uu[U];
        int pp[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int p = base + u * B + tid;
            pp[u] = p < npix ? p : npix - 1;
#pragma unroll
            for (int c = 0; c < NS * 4; ++c) {
                if (CP) {
                    const PairC pc = rows_c[c][pp[u]];
                    v[u][c] = make_double2(pc.lo, (double)pc.d);  // .y holds the DIFFERENCE in compact mode
                } else {
                    v[u][c] = rows[c][pp[u]];
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            kk[u] = redden ? P.pix_k[pp[u]] : make_double2(0.0, 0.0);
            tt[u] = PF ? lds_t[pp[u]] : P.pix_t[pp[u]];
            ff[u] = PF ? lds_f[pp[u]] : P.pix_flux[pp[u]];
            uu[u] = PF ? lds_u[pp[u]] : P.pix_u[pp[u]];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            double ylo = 0.0, yhi = 0.0;
#pragma unroll
            for (int c = 0; c < NS * 4; ++c) {
                ylo = fma(w[c], v[u][c].x, ylo);
                yhi = fma(w[c], v[u][c].y, yhi);
            }
            if (CP) yhi += ylo;  // blended difference -> blended upper sample
            if (redden) {
                const double elo = exp2(redc * kk[u].x);  // 10^(-0.4 A_V k)     mft6.py:62-63
                // neighbouring grid samples: y = ln2 * c * (k_hi - k_lo) is tiny, so e^y from four series
                // terms is exact to < 1e-17 for |y| < 1e-3; anything larger takes the full exp2
                const double y = 0.6931471805599453 * (redc * (kk[u].y - kk[u].x));
                const double ehi = (fabs(y) < 1e-3)
                                       ? elo * fma(y, fma(y, fma(y, fma(y, 1.0 / 24, 1.0 / 6), 0.5), 1.0), 1.0)
                                       : exp2(redc * kk[u].y);
                ylo *= elo;
                yhi *= ehi;
            }
            const double m = fma(yhi - ylo, tt[u], ylo);  // mft6.py:1169-1170
            if (base + u * B + tid < npix) {
                model[pp[u]] = m;
                const double f = ff[u] / m;  // frac before the median scale, mft6.py:194
                const double f1 = f * uu[u], f2 = f * (uu[u] * uu[u]);
                const int slot = (sub * U + u) & (vk - 1);  // static: sub and u are unrolled
#pragma unroll
                for (int k = 0; k < vk; ++k)
                    if (slot == k) { qa[k][0] += f; qa[k][1] += f1; qa[k][2] += f2; }
                const unsigned long long key = key_of(m);
                kmin = key < kmin ? key : kmin;
                kmax = key > kmax ? key : kmax;
                if (early) atomicAdd(&S.hist[logbin(m)], 1u);
            }
        }
      }
    }
    MSX_STAMP(P, wk, 2);
    // The contrast / photometry terms (A5/A6) need the recipe's nodes and weights and nothing else.  Phase A is
    // bound by the CU's L2 port and wave 0's loads are served first, so wave 0 leaves the pixel loop thousands of
    // cycles before the last wave: it computes the terms in that wait.  (Other modes: inside block_median.)
    if (early && wave == 0) recipe_band_terms<NS>(P, mode, th_row, D, lane);
    {
#pragma unroll
        for (int k = 0; k < vk; ++k) {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const double r = wave_sum(qa[k][i]);
                if (lane == 0) S.q[i][k * nw + wave] = r;  // virtual wave = row class * nw + wave
            }
        }
        const unsigned long long a = wave_min_u64(kmin), b = wave_max_u64(kmax);
        if (lane == 0) { S.kmin[wave] = a; S.kmax[wave] = b; }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            double r = 0.0;
            for (int x = 0; x < kMaxWaves; ++x) r += S.q[i][x];
            q[i] = r;
        }
        kmin = S.kmin[0]; kmax = S.kmax[0];
        for (int x = 1; x < nw; ++x) {
            kmin = S.kmin[x] < kmin ? S.kmin[x] : kmin;
            kmax = S.kmax[x] > kmax ? S.kmax[x] : kmax;
        }
    }
    MSX_STAMP(P, wk, 3);
    // np.median of a vector holding a NaN is NaN -> total NaN -> -inf (mft6.py:1202-1203)
    if (kmax > key_of(INFINITY) || kmin < key_of(-INFINITY)) {
        if (tid == 0) {
            const bool chi_valued = mode == MSX_MODE_CHISQ || mode == MSX_MODE_OPT_STEP || mode == MSX_MODE_OPT_INIT;
            if (mode == MSX_MODE_OPT_INIT) P.opt_med[wk] = NAN;
            walker_done(P, D, wk, ndim, chi_valued ? NAN : -INFINITY, MSX_W_OK, logp, status);
        }
        return;
    }

    // ---- phase B: exact median (np.median, mft6.py:1173) -----------------------------------------------
    // wave 2 computes the contrast / photometry terms inside the median's scan stage (fast recipe only)
    const double *th_w = th_row;
    auto side = [&]() __attribute__((always_inline)) {
        if (fast && wave == 2) recipe_band_terms<NS>(P, mode, th_w, D, lane);
    };
    // The spectrum chi^2 factorises: with P(u) = c0 + c1 u + c2 u^2 the raw fit of data/model (from the q
    // sums), the fit of data/(scale*model) is P/scale, data' = scale*data/P and
    //   sum (scale*m - data')^2/err^2 = scale^2 * sum (m - data/P)^2/err^2,
    // so everything but the final scalar multiply is independent of the median and rides along the
    // median's first pass over the model vector (fused modes only; the optimiser modes keep phase C).
    const bool fused = !(mode == MSX_MODE_OPT_STEP || mode == MSX_MODE_OPT_INIT);
    struct ChiElem {  // holds plain pointers, never a reference to the by-value kernel argument (see DevProblem)
        enum { VK = kMaxWaves / (MAXT / kWave) };  // (a local class cannot have static data members)
        const double *pix_u, *pix_flux, *pix_ivar;
        double c0, c1, c2;
        double acc[VK];  // one per virtual wave this wave plays (see phase A): rows k, k + VK, ... of the pass
        bool on;
        // four consecutive rows of the pass (row = p / blockDim.x; a trip starts at a multiple of four rows)
        __device__ __forceinline__ void process4(const int (&pp)[4], const double (&xv)[4], const bool (&ok)[4]) {
            if (!on) return;
            double u[4], f[4], e[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) { u[k] = pix_u[pp[k]]; f[k] = pix_flux[pp[k]]; e[k] = pix_ivar[pp[k]]; }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const double poly = fma(fma(c2, u[k], c1), u[k], c0);
                const double r = xv[k] - f[k] / poly;  // (model - data/P); mft6.py:196,120 up to scale^2
                acc[k & (VK - 1)] += ok[k] ? (r * r) * e[k] : 0.0;
            }
        }
        __device__ __forceinline__ void flush(BlockScratch &S) {
            if (!on) return;
#pragma unroll
            for (int k = 0; k < VK; ++k) {
                const double r = wave_sum(acc[k]);
                if ((threadIdx.x & 63) == 0) S.chi[k * (MAXT / kWave) + (threadIdx.x >> 6)] = r;
            }
        }
    };
    ChiElem chi_elem{PF ? lds_u : P.pix_u, PF ? lds_f : P.pix_flux, P.pix_ivar,
                     P.minv[0] * q[0] + P.minv[1] * q[1] + P.minv[2] * q[2],
                     P.minv[3] * q[0] + P.minv[4] * q[1] + P.minv[5] * q[2],
                     P.minv[6] * q[0] + P.minv[7] * q[1] + P.minv[8] * q[2], {}, fused};
    bool chi_done = false;
    double med_model = 0.0;
    bool solved = false;
    if (early) {
        solved = logbin_median(model, npix, kmin, kmax, S, chi_elem, &med_model);
        chi_done = solved;
        if (!solved) {  // not a positive vector spanning < 8 binades, or > 256 equal-bin candidates: start over
            __syncthreads();  // every wave decided from the counters by itself: none may still be reading them
            for (int i = tid; i < kLogBins; i += B) S.hist[i] = 0;
            __syncthreads();
        }
    }
    if (!solved) med_model = block_median(model, npix, kmin, kmax, S, side, chi_elem, &chi_done);
    if (fused && !chi_done) {  // degenerate vectors (all equal): the median took no pass, do it here
        for (int base = 0; base < npix; base += 4 * B) {
            int pp[4];
            double xv[4];
            bool ok[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int p = base + u * B + tid;
                ok[u] = p < npix;
                pp[u] = ok[u] ? p : npix - 1;
                xv[u] = model[pp[u]];
            }
            chi_elem.process4(pp, xv, ok);
        }
        chi_elem.flush(S);
        __syncthreads();
    }
    MSX_STAMP(P, wk, 4);
    MSX_STAMP(P, wk, 5);

    // ---- phase C: median scale, quadratic continuum fit, chi^2 (A8.2, A8.3, A9) ------------------
    // Pre-optimiser variants (fit_spec, mft6.py:856-1137): OPT_INIT normalises the data against the
    // chain's initial model like the hot path does and KEEPS the normalised vector + its median
    // (:888-889); OPT_STEP compares every proposal with that stored vector, with no per-proposal
    // continuum fit (:1011-1015).  Both weight the spectrum term by 3 (:893,:1015).
    const bool opt_step = mode == MSX_MODE_OPT_STEP, opt_init = mode == MSX_MODE_OPT_INIT;
    const int64_t chain = opt_step ? (int64_t)P.opt_chain[wk] : wk;
    const double *__restrict__ dflux = opt_step ? P.opt_flux + chain * npix : P.pix_flux;
    const double med_data = opt_step ? P.opt_med[chain] : P.median_flux;
    const double scale = med_data / med_model;  // mft6.py:1173 / :1011
    double coef[3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
        coef[i] = (P.minv[3 * i] * q[0] + P.minv[3 * i + 1] * q[1] + P.minv[3 * i + 2] * q[2]) / scale;
    double chia[vk];  // per virtual wave, like phase A
#pragma unroll
    for (int k = 0; k < vk; ++k) chia[k] = 0.0;
    unsigned long long dmin = ~0ull, dmax = 0ull;
    for (int p = tid, row = 0; p < npix && !fused; p += B, ++row) {
        const double ms = model[p] * scale;
        double dn;
        if (opt_step) {
            dn = dflux[p];
        } else {
            const double u = P.pix_u[p];
            const double poly = fma(fma(coef[2], u, coef[1]), u, coef[0]);
            dn = dflux[p] / poly;  // mft6.py:196
        }
        const double r = ms - dn;
        const double t = (r * r) * P.pix_ivar[p];  // mft6.py:120
        const int slot = row & (vk - 1);
#pragma unroll
        for (int k = 0; k < vk; ++k)
            if (slot == k) chia[k] += t;
        if (opt_init) {
            P.opt_flux[wk * npix + p] = dn;
            model[p] = dn;  // the model value is dead now; reuse the LDS vector for median(data')
            const unsigned long long key = key_of(dn);
            dmin = key < dmin ? key : dmin;
            dmax = key > dmax ? key : dmax;
        }
    }
    MSX_STAMP(P, wk, 6);
    if (!fused) {
#pragma unroll
        for (int k = 0; k < vk; ++k) {
            const double r = wave_sum(chia[k]);
            if (lane == 0) S.chi[k * nw + wave] = r;
        }
    }
    if (opt_init) {
        const unsigned long long a = wave_min_u64(dmin), b = wave_max_u64(dmax);
        if (lane == 0) { S.kmin[wave] = a; S.kmax[wave] = b; }
        for (int i = tid; i < kBins; i += B) S.hist[i] = 0;
    }
    if (!fused) __syncthreads();  // (fused: S.chi was published before the median's first barrier)
    MSX_STAMP(P, wk, 7);
    double tot = 0.0;
    for (int x = 0; x < kMaxWaves; ++x) tot += S.chi[x];
    if (fused) tot = tot * (scale * scale);
    if (opt_init) {
        dmin = S.kmin[0]; dmax = S.kmax[0];
        for (int x = 1; x < nw; ++x) {
            dmin = S.kmin[x] < dmin ? S.kmin[x] : dmin;
            dmax = S.kmax[x] > dmax ? S.kmax[x] : dmax;
        }
        const bool bad = dmax > key_of(INFINITY) || dmin < key_of(-INFINITY);
        NoElem no_elem;
        bool unused = false;
        const double md = bad ? NAN : block_median(model, npix, dmin, dmax, S, NoSide(), no_elem, &unused);  // np.median(flux), :1011
        if (tid == 0) P.opt_med[wk] = md;
    }
    if (tid == 0) {
        double iic = tot / (double)npix;  // mft6.py:1179
        if (opt_step || opt_init) iic = iic * 3;  // mft6.py:893,1015
        const double total = iic * (double)(P.nc + P.np) + D.chi_extra;  // mft6.py:1191 / :904 / :1028
        double out;
        if (mode == MSX_MODE_CHISQ || opt_step || opt_init) out = total;  // mft6.py:1198-1199
        else out = isnan(total) ? -INFINITY : D.lp + (-0.5 * total);  // mft6.py:1202-1205, 1470
        walker_done(P, D, wk, ndim, out, MSX_W_OK, logp, status);
    }
}


// ------------------------------------------------------------------------------------------------
// staging kernels
// ------------------------------------------------------------------------------------------------
// CCM89 k(lambda) = a(x) + b(x)/R_V, x = 1e4/lambda[A] (A7; coefficients of Cardelli+ 1989)
__global__ void ccm89_kernel(const double *__restrict__ wl, int64_t n, double rv, double *__restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double x = 1e4 / wl[i];
    double a, b;
    if (x < 1.1) {
        const double y = pow(x, 1.61);
        a = 0.574 * y;
        b = -0.527 * y;
    } else if (x < 3.3) {
        const double y = x - 1.82;
        a = ((((((0.329990 * y - 0.77530) * y + 0.01979) * y + 0.72085) * y - 0.02427) * y - 0.50447) * y + 0.17699) * y + 1.0;
        b = ((((((-2.09002 * y + 5.30260) * y - 0.62251) * y - 5.38434) * y + 1.07233) * y + 2.28305) * y + 1.41338) * y;
    } else if (x < 8.0) {
        a = 1.752 - 0.316 * x - 0.104 / ((x - 4.67) * (x - 4.67) + 0.341);
        b = -3.090 + 1.825 * x + 1.206 / ((x - 4.62) * (x - 4.62) + 0.263);
        if (x >= 5.9) {
            const double y = x - 5.9;
            a += -0.04473 * (y * y) - 0.009779 * (y * y * y);
            b += 0.2130 * (y * y) + 0.1207 * (y * y * y);
        }
    } else {
        const double y = x - 8.0;
        a = -0.070 * (y * y * y) + 0.137 * (y * y) - 0.628 * y - 1.073;
        b = 0.374 * (y * y * y) - 0.420 * (y * y) + 4.257 * y + 13.670;
    }
    out[i] = a + b / rv;
}

// pairs[node][p] = {grid[node][lo_p], grid[node][lo_p+1]};  node = blockIdx.y
__global__ void gather_pairs_kernel(const double *__restrict__ grid, int64_t nwl, const int64_t *__restrict__ lo,
                                    int64_t npix, double2 *__restrict__ pairs) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npix) return;
    const double *row = grid + (int64_t)blockIdx.y * nwl;
    const int64_t j = lo[p];
    pairs[(int64_t)blockIdx.y * npix + p] = make_double2(row[j], row[j + 1]);
}

__global__ void gather_pairs_compact_kernel(const double *__restrict__ grid, int64_t nwl, const int64_t *__restrict__ lo,
                                            int64_t npix, PairC *__restrict__ pairs) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npix) return;
    const double *row = grid + (int64_t)blockIdx.y * nwl;
    const int64_t j = lo[p];
    PairC out;
    out.lo = row[j];
    out.d = (float)(row[j + 1] - row[j]);
    pairs[(int64_t)blockIdx.y * npix + p] = out;
}

// band_tab[node][b] = sum_i w_b[i] * grid[node][i0_b + i];  grid.x = band, grid.y = node
__global__ void band_integral_kernel(const double *__restrict__ grid, int64_t nwl, const double *__restrict__ w,
                                     const int64_t *__restrict__ woff, const int64_t *__restrict__ i0,
                                     const int64_t *__restrict__ len, int nb, double *__restrict__ tab) {
    __shared__ double part[kMaxWaves];
    const int b = blockIdx.x, node = blockIdx.y;
    const double *row = grid + (int64_t)node * nwl + i0[b];
    const double *wb = w + woff[b];
    double acc = 0.0;
    for (int64_t i = threadIdx.x; i < len[b]; i += blockDim.x) acc = fma(wb[i], row[i], acc);
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double r = 0.0;
        for (int i = 0; i < (int)(blockDim.x >> 6); ++i) r += part[i];
        tab[(int64_t)node * nb + b] = r;
    }
}

// A3: out[n] = sum_k e[k] * y[n + c - k], zero outside [0, N)  (np.convolve(y, e, 'same'), c=(lx-1)/2)
// e = normalised Gaussian taps built in LDS by every block; tile of y staged through LDS.
constexpr int kConvTile = 1024;
__global__ void __launch_bounds__(256)
broaden_conv_kernel(const double *__restrict__ in, int64_t in_stride, double *__restrict__ out, int64_t out_stride,
                    int64_t n, int lx, double dx, double sigma) {
    double *taps = reinterpret_cast<double *>(dyn_lds);  // [lx]
    double *tile = taps + lx;                             // [kConvTile + lx - 1]
    __shared__ double part[4];
    const int tid = threadIdx.x;
    const int c = (lx - 1) / 2;
    const int off0 = lx / 2 + lx % 2 - 1;  // nx[k] = (k - off0) * dx   (PyAstronomy broadGaussFast)
    double acc = 0.0;
    for (int k = tid; k < lx; k += 256) {
        const double x = (double)(k - off0) * dx;
        const double e = exp(-(x * x) / (2.0 * (sigma * sigma)));
        taps[k] = e;
        acc += e;
    }
    acc = wave_sum(acc);
    if ((tid & 63) == 0) part[tid >> 6] = acc;
    __syncthreads();
    const double norm = (part[0] + part[1]) + (part[2] + part[3]);
    const double *row = in + (int64_t)blockIdx.y * in_stride;
    const int64_t t0 = (int64_t)blockIdx.x * kConvTile;
    const int64_t g0 = t0 + c - (lx - 1);  // global index of tile[0]
    for (int j = tid; j < kConvTile + lx - 1; j += 256) {
        const int64_t g = g0 + j;
        tile[j] = (g >= 0 && g < n) ? row[g] : 0.0;
    }
    for (int k = tid; k < lx; k += 256) taps[k] = taps[k] / norm;
    __syncthreads();
    double *orow = out + (int64_t)blockIdx.y * out_stride;
#pragma unroll
    for (int r = 0; r < kConvTile / 256; ++r) {
        const int nl = tid + r * 256;
        if (t0 + nl >= n) break;
        double s = 0.0;
        const double *tp = tile + nl + (lx - 1);
        for (int k = 0; k < lx; ++k) s = fma(taps[k], tp[-k], s);
        orow[t0 + nl] = s;
    }
}

// f3: linear resample of one tabulated spectrum (x sorted ascending) onto query wavelengths with
// np.interp / scipy interp1d(kind='linear') arithmetic (mft6.py:369-371): one thread per query.
__global__ void resample_kernel(const double *__restrict__ xs, const double *__restrict__ ys, int64_t n,
                                const double *__restrict__ xq, int64_t m, double *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const double x = xq[i];
    int64_t lo = 0, hi = n;  // first index with xs > x
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (xs[mid] <= x) lo = mid + 1; else hi = mid;
    }
    const int64_t j = lo - 1;
    double r;
    if (j < 0) r = NAN;  // caller range-checks; unreachable
    else if (j >= n - 1) r = ys[n - 1];
    else if (xs[j] == x) r = ys[j];
    else {
        const double slope = (ys[j + 1] - ys[j]) / (xs[j + 1] - xs[j]);
        r = slope * (x - xs[j]) + ys[j];
    }
    out[i] = r;
}

// edge patches broad[0:5] = broad[5]; broad[n-10:n] = broad[n-11] (mft6.py:129-130) while copying
__global__ void broaden_patch_kernel(const double *__restrict__ tmp, int64_t tmp_stride, double *__restrict__ dst,
                                     int64_t dst_stride, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int64_t src = i;
    if (i < 5) src = 5;
    if (i >= n - 10) src = n - 11;
    dst[(int64_t)blockIdx.y * dst_stride + i] = tmp[(int64_t)blockIdx.y * tmp_stride + src];
}

// make_composite (A4-A6): one lane builds the recipe, then an elementwise blend over the window
__global__ void composite_setup_kernel(DevProblem P, const double *__restrict__ args, int use_distance,
                                       WalkerDesc *__restrict__ out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    // args = teff[ns], logg[ns], rad[ns], plx
    const int ns = P.nspec;
    WalkerDesc D;
    build_desc(P, args, args + ns, args + 2 * ns, use_distance != 0, args[3 * ns], 0.0, &D);
    *out = D;
}

__global__ void composite_kernel(DevProblem P, const WalkerDesc *__restrict__ Dp, double *__restrict__ spec) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P.win_n || Dp->status != MSX_W_OK) return;
    const int nc = Dp->ncorner;
    // the reference sums star by star: spec1 = pri + sec (+ ter)     mft6.py:744,751
    double total = 0.0;
    for (int s = 0; s < nc / 4; ++s) {
        double acc = 0.0;
        for (int c = 0; c < 4; ++c)
            acc = fma(Dp->w[4 * s + c], P.grid[(int64_t)Dp->node[4 * s + c] * P.nwl + P.win_j0 + i], acc);
        total += acc;
    }
    spec[i] = total;
}

__global__ void __launch_bounds__(256)
copy_float4_kernel(const float4 *__restrict__ src, float4 *__restrict__ dst, int64_t n4) {
    // 4 independent 16-B loads in flight per lane, then 4 stores; grid-stride over 1024-element tiles
    const int64_t stride = (int64_t)gridDim.x * 1024;
    for (int64_t base = (int64_t)blockIdx.x * 1024 + threadIdx.x; base < n4; base += stride) {
        float4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = (base + 256 * k < n4) ? src[base + 256 * k] : make_float4(0, 0, 0, 0);
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (base + 256 * k < n4) dst[base + 256 * k] = v[k];
    }
}

}  // namespace

// ================================================================================================
// host side: context + C ABI
// ================================================================================================
struct msx_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    hipDeviceProp_t prop;
    // grid
    int64_t nwl = 0;
    int nt = 0, ng = 0;
    std::vector<double> h_wl;
    double *d_grid = nullptr, *d_wl = nullptr, *d_kgrid = nullptr, *d_teff = nullptr, *d_logg = nullptr;
    uint8_t *d_present = nullptr;
    bool grid_staged = false;
    // problem
    bool problem_staged = false;
    DevProblem P;
    std::vector<void *> prob_allocs;
    // scratch for host-pointer entry points
    double *d_theta = nullptr, *d_logp = nullptr;
    int32_t *d_status = nullptr;  // points into d_logp's allocation
    void *h_pin = nullptr;        // pinned host staging for the host-pointer entry points
    int64_t cap_walkers = 0;
    double *d_misc = nullptr;  // composite args / desc / small outputs
    double *d_spec = nullptr;
    int64_t cap_spec = 0;
    double *d_opt_flux = nullptr, *d_opt_med = nullptr;
    int32_t *d_opt_chain = nullptr;
    int64_t opt_chains = 0, cap_chain = 0;
    int max_dyn_lds = 0;
    bool pf_ok = false;   // the LDS-staged-statics variants fit (msx_stage_problem)
    bool use_pf = true;   // MSX_NO_PF=1 in the environment turns them off (A/B measurements)
    bool model_in_global = false;
    // RCCL all-gather of log-probabilities: communicator + its own stream + per-slot events
    void *rccl_comm = nullptr;
    hipStream_t comm_stream = nullptr;
    hipEvent_t ev_ready = nullptr;
    hipEvent_t ev_done[4] = {nullptr, nullptr, nullptr, nullptr};
    int comm_world = 0, comm_rank = 0;
    double *d_model_scratch = nullptr;
    int64_t cap_model_scratch = 0;  // doubles
    struct SamplerRun *smp = nullptr;  // device-resident sampler in flight (msx_sampler_begin .. _end)
};
static void sampler_free(msx_ctx *c);

namespace {

// RCCL is resolved at run time from the copy PyTorch-ROCm already mapped (same SONAME librccl.so.1 as
// /opt/rocm's), so the process never holds two RCCL instances; nothing is linked at build time.
struct RcclUniqueId { char internal[128]; };
struct RcclApi {
    void *handle = nullptr;
    int (*GetUniqueId)(RcclUniqueId *) = nullptr;
    int (*CommInitRank)(void **, int, RcclUniqueId, int) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    bool ok = false;
};
RcclApi &rccl() {
    static RcclApi api;
    if (api.handle) return api;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *nm : names) {
        api.handle = dlopen(nm, RTLD_NOW | RTLD_NOLOAD);
        if (api.handle) break;
    }
    for (int i = 0; !api.handle && i < 3; ++i) api.handle = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
    if (!api.handle) return api;
    api.GetUniqueId = (int (*)(RcclUniqueId *))dlsym(api.handle, "ncclGetUniqueId");
    api.CommInitRank = (int (*)(void **, int, RcclUniqueId, int))dlsym(api.handle, "ncclCommInitRank");
    api.AllGather = (int (*)(const void *, void *, size_t, int, void *, hipStream_t))dlsym(api.handle, "ncclAllGather");
    api.CommDestroy = (int (*)(void *))dlsym(api.handle, "ncclCommDestroy");
    api.GetErrorString = (const char *(*)(int))dlsym(api.handle, "ncclGetErrorString");
    api.ok = api.GetUniqueId && api.CommInitRank && api.AllGather && api.CommDestroy && api.GetErrorString;
    return api;
}
constexpr int kNcclFloat64 = 8;  // ncclFloat64 / ncclDouble (rccl.h)

int fail(msx_ctx *c, int code, const std::string &msg) {
    if (c) c->err = msg;
    return code;
}

#define HIP_TRY(ctx, expr)                                                                         \
    do {                                                                                           \
        hipError_t e__ = (expr);                                                                   \
        if (e__ != hipSuccess)                                                                     \
            return fail(ctx, MSX_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e__));     \
    } while (0)

template <typename T>
int dev_alloc_copy(msx_ctx *c, std::vector<void *> *track, const T *host, int64_t count, T **out) {
    *out = nullptr;
    if (count <= 0) {
        // keep kernels simple: always a valid pointer
        HIP_TRY(c, hipMalloc((void **)out, 16));
        if (track) track->push_back(*out);
        return MSX_OK;
    }
    HIP_TRY(c, hipMalloc((void **)out, sizeof(T) * count));
    if (track) track->push_back(*out);
    HIP_TRY(c, hipMemcpy(*out, host, sizeof(T) * count, hipMemcpyHostToDevice));
    return MSX_OK;
}

void free_problem(msx_ctx *c) {
    sampler_free(c);  // a sampler in flight holds pointers into the problem's tables
    for (void *p : c->prob_allocs) (void)hipFree(p);
    c->prob_allocs.clear();
    c->problem_staged = false;
    if (c->d_opt_flux) (void)hipFree(c->d_opt_flux);
    if (c->d_opt_med) (void)hipFree(c->d_opt_med);
    c->d_opt_flux = c->d_opt_med = nullptr;
    c->opt_chains = 0;
}

void free_grid(msx_ctx *c) {
    void *ptrs[] = {c->d_grid, c->d_wl, c->d_kgrid, c->d_teff, c->d_logg, c->d_present};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    c->d_grid = c->d_wl = c->d_kgrid = c->d_teff = c->d_logg = nullptr;
    c->d_present = nullptr;
    c->grid_staged = false;
}

int conv_taps(double mean_wl, double dx, double resolution, double maxsig, double *sigma_out, int *lx_out) {
    // pyasl.instrBroadGaussFast: fwhm = mean(wl)/R; sigma = fwhm/(2 sqrt(2 ln 2)); broadGaussFast:
    // lx = int(((sigma*maxsig)/dx)*2.0) + 1
    const double fwhm = 1.0 / resolution * mean_wl;
    const double sigma = fwhm / (2.0 * sqrt(2.0 * log(2.0)));
    const int lx = (int)(((sigma * maxsig) / dx) * 2.0) + 1;
    *sigma_out = sigma;
    *lx_out = lx;
    return lx;
}

int check_even_spacing(msx_ctx *c, const double *wl, int64_t n) {
    // broadGaussFast: abs(max(dxs) - min(dxs)) > mean(dxs)*1e-6 -> error
    double mx = -INFINITY, mn = INFINITY, sum = 0.0;
    for (int64_t i = 1; i < n; ++i) {
        const double d = wl[i] - wl[i - 1];
        mx = d > mx ? d : mx;
        mn = d < mn ? d : mn;
        sum += d;
    }
    if (fabs(mx - mn) > (sum / (double)(n - 1)) * 1e-6)
        return fail(c, MSX_ERR_RANGE, "broaden: the wavelength axis is not equidistant");
    return MSX_OK;
}

int launch_conv(msx_ctx *c, const double *d_in, int64_t in_stride, double *d_tmp, int64_t rows, int64_t n, int lx,
                double dx, double sigma) {
    const size_t lds = sizeof(double) * ((size_t)lx + kConvTile + lx - 1);
    if (lds > 150 * 1024) return fail(c, MSX_ERR_RANGE, "broaden: kernel too long for the LDS tile");
    if ((int)lds > 64 * 1024)
        HIP_TRY(c, hipFuncSetAttribute((const void *)broaden_conv_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)lds));
    dim3 g((unsigned)((n + kConvTile - 1) / kConvTile), (unsigned)rows);
    hipLaunchKernelGGL(broaden_conv_kernel, g, dim3(256), lds, c->stream, d_in, in_stride, d_tmp, n, n, lx, dx, sigma);
    HIP_TRY(c, hipGetLastError());
    return MSX_OK;
}

int pick_block(const msx_ctx *c, int64_t n, int64_t npix) {
    // Long spectra: one walker over 16 waves.  Otherwise, measured at 4096 px (us, 256 / 512 / 1024 / 4096 /
    // 16384 walkers):  512 threads, one workgroup per CU, two pixels per trip + LDS-staged statics   20 / -- ...
    //                  512 threads, two workgroups per CU, one pixel per trip (<= 128 VGPRs)         -- / 31 / 55 / 179 / 668
    //                  256 threads, three workgroups per CU                                          25 / 34 / 55 / 165 / 590
    // The choice only affects speed: every variant sums in the same order (see phase A), so a walker's value
    // has the same bits whichever one evaluates it.
    const int64_t cus = c->prop.multiProcessorCount > 0 ? c->prop.multiProcessorCount : 256;
    if (npix >= 8192) return 1024;
    if (n <= 4 * cus) return 512;
    return 256;
}

}  // namespace

extern "C" {

int msx_create(int device, msx_ctx **out) {
    if (!out) return MSX_ERR_INVALID;
    *out = nullptr;
    msx_ctx *c = new msx_ctx();
    c->device = device;
    memset(&c->P, 0, sizeof(c->P));
    if (const char *e = getenv("MSX_NO_PF")) c->use_pf = !(e[0] == '1');
    *out = c;  // returned even on failure so the caller can read msx_last_error
    HIP_TRY(c, hipSetDevice(device));
    HIP_TRY(c, hipGetDeviceProperties(&c->prop, device));
    HIP_TRY(c, hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    HIP_TRY(c, hipMalloc((void **)&c->d_misc, 4096));
    return MSX_OK;
}

void msx_destroy(msx_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    sampler_free(c);
    free_problem(c);
    free_grid(c);
    void *ptrs[] = {c->d_theta, c->d_logp, c->d_misc, c->d_spec, c->d_opt_flux, c->d_opt_med, c->d_opt_chain,
                    c->d_model_scratch};
    if (c->h_pin) (void)hipHostFree(c->h_pin);
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    if (c->rccl_comm && rccl().ok) (void)rccl().CommDestroy(c->rccl_comm);
    if (c->comm_stream) (void)hipStreamDestroy(c->comm_stream);
    if (c->ev_ready) (void)hipEventDestroy(c->ev_ready);
    for (hipEvent_t e : c->ev_done)
        if (e) (void)hipEventDestroy(e);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

const char *msx_last_error(msx_ctx *c) { return c ? c->err.c_str() : "null ctx"; }

int msx_device_info(msx_ctx *c, int64_t *out3, char *name, int name_len) {
    if (!c || !out3) return MSX_ERR_INVALID;
    out3[0] = c->prop.multiProcessorCount;
    out3[1] = (int64_t)c->prop.totalGlobalMem;
    out3[2] = c->prop.clockRate;
    if (name && name_len > 0) {
        strncpy(name, c->prop.name, name_len - 1);
        name[name_len - 1] = 0;
    }
    return MSX_OK;
}

int msx_stage_grid(msx_ctx *c, const double *wl, int64_t nwl, const double *teff_nodes, int32_t nt,
                   const double *logg_nodes, int32_t ng, const double *flux, const uint8_t *present) {
    if (!c || !wl || !teff_nodes || !logg_nodes || !flux || nwl < 2 || nt < 1 || ng < 1)
        return fail(c, MSX_ERR_INVALID, "msx_stage_grid: bad arguments");
    HIP_TRY(c, hipSetDevice(c->device));
    free_problem(c);
    free_grid(c);
    const int64_t nn = (int64_t)nt * ng;
    c->nwl = nwl; c->nt = nt; c->ng = ng;
    c->h_wl.assign(wl, wl + nwl);
    HIP_TRY(c, hipMalloc((void **)&c->d_grid, sizeof(double) * nn * nwl));
    HIP_TRY(c, hipMemcpy(c->d_grid, flux, sizeof(double) * nn * nwl, hipMemcpyHostToDevice));
    int rc;
    if ((rc = dev_alloc_copy(c, nullptr, wl, nwl, &c->d_wl))) return rc;
    if ((rc = dev_alloc_copy(c, nullptr, teff_nodes, (int64_t)nt, &c->d_teff))) return rc;
    if ((rc = dev_alloc_copy(c, nullptr, logg_nodes, (int64_t)ng, &c->d_logg))) return rc;
    std::vector<uint8_t> pres(nn, 1);
    if (present) pres.assign(present, present + nn);
    if ((rc = dev_alloc_copy(c, nullptr, pres.data(), nn, &c->d_present))) return rc;
    HIP_TRY(c, hipMalloc((void **)&c->d_kgrid, sizeof(double) * nwl));
    hipLaunchKernelGGL(ccm89_kernel, dim3((unsigned)((nwl + 255) / 256)), dim3(256), 0, c->stream, c->d_wl, nwl, 3.1,
                       c->d_kgrid);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->grid_staged = true;
    return MSX_OK;
}

int msx_ccm89_k(msx_ctx *c, const double *wl, int64_t n, double rv, double *out) {
    if (!c || !wl || !out || n < 0) return fail(c, MSX_ERR_INVALID, "msx_ccm89_k: bad arguments");
    if (n == 0) return MSX_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    double *d_in = nullptr, *d_out = nullptr;
    HIP_TRY(c, hipMalloc((void **)&d_in, sizeof(double) * n));
    HIP_TRY(c, hipMalloc((void **)&d_out, sizeof(double) * n));
    HIP_TRY(c, hipMemcpy(d_in, wl, sizeof(double) * n, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(ccm89_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, d_in, n, rv, d_out);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipMemcpy(out, d_out, sizeof(double) * n, hipMemcpyDeviceToHost));
    (void)hipFree(d_in);
    (void)hipFree(d_out);
    return MSX_OK;
}

int msx_resample_linear(msx_ctx *c, const double *x, const double *y, int64_t n, const double *xq, int64_t m,
                        double *out) {
    if (!c || !x || !y || !xq || !out || n < 2 || m < 0) return fail(c, MSX_ERR_INVALID, "msx_resample_linear: bad arguments");
    if (m == 0) return MSX_OK;
    for (int64_t i = 1; i < n; ++i)
        if (!(x[i] >= x[i - 1])) return fail(c, MSX_ERR_INVALID, "msx_resample_linear: x must be sorted ascending");
    double qmin = INFINITY, qmax = -INFINITY;
    for (int64_t i = 0; i < m; ++i) { qmin = xq[i] < qmin ? xq[i] : qmin; qmax = xq[i] > qmax ? xq[i] : qmax; }
    if (qmin < x[0]) return fail(c, MSX_ERR_RANGE, "A value in x_new is below the interpolation range's minimum value.");
    if (qmax > x[n - 1]) return fail(c, MSX_ERR_RANGE, "A value in x_new is above the interpolation range's maximum value.");
    HIP_TRY(c, hipSetDevice(c->device));
    double *d_x = nullptr, *d_y = nullptr, *d_q = nullptr, *d_o = nullptr;
    HIP_TRY(c, hipMalloc((void **)&d_x, sizeof(double) * n));
    HIP_TRY(c, hipMalloc((void **)&d_y, sizeof(double) * n));
    HIP_TRY(c, hipMalloc((void **)&d_q, sizeof(double) * m));
    HIP_TRY(c, hipMalloc((void **)&d_o, sizeof(double) * m));
    hipError_t e = hipMemcpy(d_x, x, sizeof(double) * n, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_y, y, sizeof(double) * n, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_q, xq, sizeof(double) * m, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(resample_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, c->stream, d_x, d_y, n, d_q, m, d_o);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) e = hipMemcpy(out, d_o, sizeof(double) * m, hipMemcpyDeviceToHost);
    (void)hipFree(d_x); (void)hipFree(d_y); (void)hipFree(d_q); (void)hipFree(d_o);
    if (e != hipSuccess) return fail(c, MSX_ERR_HIP, hipGetErrorString(e));
    return MSX_OK;
}

int msx_broaden(msx_ctx *c, const double *wl, const double *flux, int64_t n, double resolution, double maxsig,
                double *out) {
    if (!c || !wl || !flux || !out || n < 16 || !(resolution > 0) || !(maxsig > 0))
        return fail(c, MSX_ERR_INVALID, "msx_broaden: bad arguments (need n >= 16)");
    HIP_TRY(c, hipSetDevice(c->device));
    int rc = check_even_spacing(c, wl, n);
    if (rc) return rc;
    double mean = 0.0;
    for (int64_t i = 0; i < n; ++i) mean += wl[i];
    mean /= (double)n;
    double sigma;
    int lx;
    conv_taps(mean, wl[1] - wl[0], resolution, maxsig, &sigma, &lx);
    if (lx < 1) return fail(c, MSX_ERR_RANGE, "msx_broaden: empty kernel");
    double *d_in = nullptr, *d_tmp = nullptr, *d_out = nullptr;
    HIP_TRY(c, hipMalloc((void **)&d_in, sizeof(double) * n));
    HIP_TRY(c, hipMalloc((void **)&d_tmp, sizeof(double) * n));
    HIP_TRY(c, hipMalloc((void **)&d_out, sizeof(double) * n));
    HIP_TRY(c, hipMemcpy(d_in, flux, sizeof(double) * n, hipMemcpyHostToDevice));
    rc = launch_conv(c, d_in, n, d_tmp, 1, n, lx, wl[1] - wl[0], sigma);
    if (rc == MSX_OK) {
        hipLaunchKernelGGL(broaden_patch_kernel, dim3((unsigned)((n + 255) / 256), 1), dim3(256), 0, c->stream, d_tmp, n,
                           d_out, n, n);
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e == hipSuccess) e = hipMemcpy(out, d_out, sizeof(double) * n, hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = fail(c, MSX_ERR_HIP, hipGetErrorString(e));
    }
    (void)hipFree(d_in);
    (void)hipFree(d_tmp);
    (void)hipFree(d_out);
    return rc;
}

int msx_broaden_grid(msx_ctx *c, int64_t i0, int64_t n, double resolution, double maxsig) {
    if (!c) return MSX_ERR_INVALID;
    if (!c->grid_staged) return fail(c, MSX_ERR_STATE, "msx_broaden_grid: no grid staged");
    if (i0 < 0 || n < 16 || i0 + n > c->nwl || !(resolution > 0) || !(maxsig > 0))
        return fail(c, MSX_ERR_INVALID, "msx_broaden_grid: bad window");
    HIP_TRY(c, hipSetDevice(c->device));
    const double *wl = c->h_wl.data() + i0;
    int rc = check_even_spacing(c, wl, n);
    if (rc) return rc;
    double mean = 0.0;
    for (int64_t i = 0; i < n; ++i) mean += wl[i];
    mean /= (double)n;
    double sigma;
    int lx;
    conv_taps(mean, wl[1] - wl[0], resolution, maxsig, &sigma, &lx);
    const int64_t rows = (int64_t)c->nt * c->ng;
    double *d_tmp = nullptr;
    HIP_TRY(c, hipMalloc((void **)&d_tmp, sizeof(double) * rows * n));
    rc = launch_conv(c, c->d_grid + i0, c->nwl, d_tmp, rows, n, lx, wl[1] - wl[0], sigma);
    if (rc == MSX_OK) {
        hipLaunchKernelGGL(broaden_patch_kernel, dim3((unsigned)((n + 255) / 256), (unsigned)rows), dim3(256), 0,
                           c->stream, d_tmp, n, c->d_grid + i0, c->nwl, n);
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) rc = fail(c, MSX_ERR_HIP, hipGetErrorString(e));
    }
    (void)hipFree(d_tmp);
    // any staged problem was derived from the pre-broadening grid
    free_problem(c);
    return rc;
}

int msx_read_node(msx_ctx *c, int32_t it, int32_t ig, double *out) {
    if (!c || !out) return MSX_ERR_INVALID;
    if (!c->grid_staged) return fail(c, MSX_ERR_STATE, "msx_read_node: no grid staged");
    if (it < 0 || it >= c->nt || ig < 0 || ig >= c->ng) return fail(c, MSX_ERR_INVALID, "msx_read_node: bad node");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipMemcpy(out, c->d_grid + ((int64_t)it * c->ng + ig) * c->nwl, sizeof(double) * c->nwl,
                         hipMemcpyDeviceToHost));
    return MSX_OK;
}

int msx_stage_problem(msx_ctx *c, const msx_problem *p) {
    if (!c || !p) return MSX_ERR_INVALID;
    if (p->struct_size != (int32_t)sizeof(msx_problem))
        return fail(c, MSX_ERR_INVALID, "msx_stage_problem: struct_size mismatch (header/library skew)");
    if (!c->grid_staged) return fail(c, MSX_ERR_STATE, "msx_stage_problem: stage the grid first");
    if (p->nspec < 2 || p->nspec > MSX_MAX_SPEC) return fail(c, MSX_ERR_INVALID, "nspec must be 2 or 3");
    if (p->npix < 4) return fail(c, MSX_ERR_INVALID, "npix too small");
    if (p->n_contrast < 0 || p->n_contrast > MSX_MAX_BANDS || p->n_phot < 0 || p->n_phot > MSX_MAX_BANDS)
        return fail(c, MSX_ERR_INVALID, "too many bands");
    if (p->niso < 2) return fail(c, MSX_ERR_INVALID, "isochrone table needs >= 2 rows");
    if (p->win_j0 < 0 || p->win_n < 1 || p->win_j0 + p->win_n > c->nwl)
        return fail(c, MSX_ERR_RANGE, "composite window is outside the staged grid");
    const int64_t need_lds = (int64_t)sizeof(double) * p->npix;
    const bool model_in_global = need_lds > 150 * 1024;  // > 19,200 pixels: the GM kernel variants
    for (int64_t i = 0; i < p->npix; ++i)
        if (p->pix_lo[i] < 0 || p->pix_lo[i] + 1 >= c->nwl)
            return fail(c, MSX_ERR_RANGE, "A value in x_new is outside the interpolation range (data pixel vs model grid)");
    const int nb = p->n_contrast + p->n_phot;
    for (int b = 0; b < nb; ++b)
        if (p->band_i0[b] < 0 || p->band_len[b] < 0 || p->band_i0[b] + p->band_len[b] > c->nwl)
            return fail(c, MSX_ERR_RANGE, "band weights run outside the staged grid");
    HIP_TRY(c, hipSetDevice(c->device));
    free_problem(c);
    std::vector<void *> &tr = c->prob_allocs;
    DevProblem &P = c->P;
    memset(&P, 0, sizeof(P));
    P.grid = c->d_grid; P.kgrid = c->d_kgrid; P.nwl = c->nwl; P.nt = c->nt; P.ng = c->ng;
    P.teff_nodes = c->d_teff; P.logg_nodes = c->d_logg; P.present = c->d_present;
    P.npix = p->npix; P.median_flux = p->median_flux; P.nspec = p->nspec;
    memcpy(P.minv, p->fit_minv, sizeof(P.minv));
    P.nc = p->n_contrast; P.np = p->n_phot;
    for (int i = 0; i < P.nc; ++i) { P.cmag[i] = p->cmag[i]; P.cerr[i] = p->cerr[i]; P.civar[i] = 1.0 / (p->cerr[i] * p->cerr[i]); }
    for (int i = 0; i < P.np; ++i) {
        P.pmag[i] = p->pmag[i]; P.perr[i] = p->perr[i]; P.pzero[i] = p->phot_zero[i]; P.pk[i] = p->phot_k[i];
        P.pivar[i] = 1.0 / (p->perr[i] * p->perr[i]);
    }
    P.win_j0 = p->win_j0; P.win_n = p->win_n;
    P.niso = p->niso; P.nav = p->nav; P.tmin = p->tmin; P.tmax = p->tmax;
    memcpy(P.pmean, p->prior_mean, sizeof(P.pmean));
    memcpy(P.psig, p->prior_sig, sizeof(P.psig));
    P.use_av = p->use_av; P.dist_fit = p->dist_fit; P.rad_prior = p->rad_prior; P.has_prior = p->has_prior_list;
    P.no_spectrum = p->no_spectrum;
    int rc;
    double *d;
    if ((rc = dev_alloc_copy(c, &tr, p->pix_t, p->npix, &d))) return rc; P.pix_t = d;
    if ((rc = dev_alloc_copy(c, &tr, p->pix_u, p->npix, &d))) return rc; P.pix_u = d;
    if ((rc = dev_alloc_copy(c, &tr, p->pix_flux, p->npix, &d))) return rc; P.pix_flux = d;
    {
        std::vector<double> ivar(p->npix);
        for (int64_t i = 0; i < p->npix; ++i) ivar[i] = 1.0 / (p->pix_err[i] * p->pix_err[i]);
        if ((rc = dev_alloc_copy(c, &tr, ivar.data(), p->npix, &d))) return rc;
        P.pix_ivar = d;
    }
    if ((rc = dev_alloc_copy(c, &tr, p->iso_teff, (int64_t)p->niso, &d))) return rc; P.iso_t = d;
    if ((rc = dev_alloc_copy(c, &tr, p->iso_logg, (int64_t)p->niso, &d))) return rc; P.iso_g = d;
    if ((rc = dev_alloc_copy(c, &tr, p->iso_lum, (int64_t)p->niso, &d))) return rc; P.iso_l = d;
    if ((rc = dev_alloc_copy(c, &tr, p->av_edges_pc, (int64_t)(p->nav > 0 ? p->nav + 1 : 0), &d))) return rc; P.av_edges = d;
    if ((rc = dev_alloc_copy(c, &tr, p->av_mu, (int64_t)p->nav, &d))) return rc; P.av_mu = d;
    if ((rc = dev_alloc_copy(c, &tr, p->av_sig, (int64_t)p->nav, &d))) return rc; P.av_sig = d;

    const int64_t nn = (int64_t)c->nt * c->ng;
    // pair table + k pairs
    int64_t *d_lo = nullptr;
    if ((rc = dev_alloc_copy(c, &tr, p->pix_lo, p->npix, &d_lo))) return rc;
    double2 *d_pairs = nullptr, *d_pk = nullptr;
    HIP_TRY(c, hipMalloc((void **)&d_pairs, sizeof(double2) * nn * p->npix)); tr.push_back(d_pairs);
    HIP_TRY(c, hipMalloc((void **)&d_pk, sizeof(double2) * p->npix)); tr.push_back(d_pk);
    dim3 gg((unsigned)((p->npix + 255) / 256), (unsigned)nn);
    hipLaunchKernelGGL(gather_pairs_kernel, gg, dim3(256), 0, c->stream, c->d_grid, c->nwl, d_lo, p->npix, d_pairs);
    HIP_TRY(c, hipGetLastError());
    hipLaunchKernelGGL(gather_pairs_kernel, dim3(gg.x, 1), dim3(256), 0, c->stream, c->d_kgrid, c->nwl, d_lo, p->npix, d_pk);
    HIP_TRY(c, hipGetLastError());
    P.pairs = d_pairs; P.pix_k = d_pk;
    P.pairs_c = nullptr;
    if (p->compact_pairs) {
        PairC *d_pc = nullptr;
        HIP_TRY(c, hipMalloc((void **)&d_pc, sizeof(PairC) * nn * p->npix)); tr.push_back(d_pc);
        hipLaunchKernelGGL(gather_pairs_compact_kernel, gg, dim3(256), 0, c->stream, c->d_grid, c->nwl, d_lo, p->npix, d_pc);
        HIP_TRY(c, hipGetLastError());
        P.pairs_c = d_pc;
    }
    // band integrals
    double *d_tab = nullptr;
    HIP_TRY(c, hipMalloc((void **)&d_tab, sizeof(double) * std::max<int64_t>(1, nn * nb))); tr.push_back(d_tab);
    if (nb > 0) {
        std::vector<int64_t> woff(nb);
        int64_t tot = 0;
        for (int b = 0; b < nb; ++b) { woff[b] = tot; tot += p->band_len[b]; }
        double *d_w = nullptr;
        int64_t *d_woff = nullptr, *d_i0 = nullptr, *d_len = nullptr;
        if ((rc = dev_alloc_copy(c, &tr, p->band_w, tot, &d_w))) return rc;
        if ((rc = dev_alloc_copy(c, &tr, woff.data(), (int64_t)nb, &d_woff))) return rc;
        if ((rc = dev_alloc_copy(c, &tr, p->band_i0, (int64_t)nb, &d_i0))) return rc;
        if ((rc = dev_alloc_copy(c, &tr, p->band_len, (int64_t)nb, &d_len))) return rc;
        hipLaunchKernelGGL(band_integral_kernel, dim3((unsigned)nb, (unsigned)nn), dim3(256), 0, c->stream, c->d_grid,
                           c->nwl, d_w, d_woff, d_i0, d_len, nb, d_tab);
        HIP_TRY(c, hipGetLastError());
    }
    P.band_tab = d_tab;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    // the hot kernel may need more than the default 64 KiB of dynamic LDS
    c->max_dyn_lds = (int)need_lds;
    c->model_in_global = model_in_global;
    if (need_lds > 48 * 1024 && !model_in_global) {
        const void *variants[] = {(const void *)logprob_kernel<2, 2, 256>, (const void *)logprob_kernel<2, 2, 512>,
                                  (const void *)logprob_kernel<2, 1, 512>,
                                  (const void *)logprob_kernel<2, 1, 1024>,
                                  (const void *)logprob_kernel<2, 2, 256, false, true>,
                                  (const void *)logprob_kernel<2, 2, 512, false, true>,
                                  (const void *)logprob_kernel<3, 1, 256>, (const void *)logprob_kernel<3, 1, 512>,
                                  (const void *)logprob_kernel<3, 1, 1024>};
        for (const void *k : variants)
            HIP_TRY(c, hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)need_lds));
    }
    c->pf_ok = !model_in_global && 4 * need_lds <= 148 * 1024;  // + ~9 KB static scratch <= 160 KB
    if (c->pf_ok) {
        const void *variants[] = {(const void *)logprob_kernel<2, 2, 512, false, false, true>,
                                  (const void *)logprob_kernel<3, 1, 512, false, false, true>};
        for (const void *k : variants)
            HIP_TRY(c, hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(4 * need_lds)));
    }
#ifdef MSX_STAMPS
    {   // diagnostic build only: per-walker shader-clock stamps
        unsigned long long *st = nullptr;
        HIP_TRY(c, hipMalloc((void **)&st, sizeof(unsigned long long) * 16 * 65536)); tr.push_back(st);
        HIP_TRY(c, hipMemset(st, 0, sizeof(unsigned long long) * 16 * 65536));
        P.stamps = st;
    }
#endif
    c->problem_staged = true;
    return MSX_OK;
}

#ifdef MSX_STAMPS
int msx_diag_read_med_stamps(msx_ctx *c, int64_t n, unsigned long long *out) {
    if (!c || n > 65536) return MSX_ERR_INVALID;
    HIP_TRY(c, hipDeviceSynchronize());
    HIP_TRY(c, hipMemcpyFromSymbol(out, HIP_SYMBOL(g_med_stamps), sizeof(unsigned long long) * 8 * n));
    return MSX_OK;
}
int msx_diag_read_stamps(msx_ctx *c, int64_t n, unsigned long long *out) {
    if (!c || !c->problem_staged || n > 65536) return MSX_ERR_INVALID;
    HIP_TRY(c, hipDeviceSynchronize());
    HIP_TRY(c, hipMemcpy(out, c->P.stamps, sizeof(unsigned long long) * 16 * n, hipMemcpyDeviceToHost));
    return MSX_OK;
}
#endif

int msx_logprob_batch_dev(msx_ctx *c, int32_t mode, const double *d_theta, int64_t n, int32_t ndim, double *d_logp,
                          int32_t *d_status, void *hip_stream, int32_t block_threads) {
    if (!c) return MSX_ERR_INVALID;
    if (!c->problem_staged) return fail(c, MSX_ERR_STATE, "msx_logprob_batch: no problem staged");
    if (n < 0 || !d_theta || !d_logp || !d_status) return fail(c, MSX_ERR_INVALID, "msx_logprob_batch: bad arguments");
    if (ndim != 2 * c->P.nspec + 2)
        return fail(c, MSX_ERR_INVALID, "P0 doesn't match what I was expecting (ndim must be 2*nspec+2)");
    if (mode < 0 || mode > 5) return fail(c, MSX_ERR_INVALID, "msx_logprob_batch: bad mode");
    if ((mode == MSX_MODE_OPT_STEP || mode == MSX_MODE_OPT_INIT) && !c->P.opt_flux)
        return fail(c, MSX_ERR_STATE, "optimiser modes go through msx_opt_init / msx_opt_step");
    if (n == 0) return MSX_OK;
    int B = block_threads > 0 ? block_threads : pick_block(c, n, c->P.npix);
    // the median's bin scan assigns kBins/B bins to each thread and the radix fallback clears its
    // 256-bin histogram with tid < 256
    if (B != 256 && B != 512 && B != 1024) return fail(c, MSX_ERR_INVALID, "block_threads must be 256, 512 or 1024");
    hipStream_t s = (hipStream_t)hip_stream;
    const size_t lds = sizeof(double) * (size_t)c->P.npix;
    if (c->model_in_global) {
        // long spectra: model vector in global memory, no dynamic LDS, 1024 threads.  The scratch grows on
        // demand (a synchronous hipMalloc: not capturable into a graph the first time a size is seen)
        const int64_t need = n * c->P.npix;
        if (need > c->cap_model_scratch) {
            HIP_TRY(c, hipSetDevice(c->device));
            HIP_TRY(c, hipStreamSynchronize(s));
            if (c->d_model_scratch) (void)hipFree(c->d_model_scratch);
            c->d_model_scratch = nullptr; c->cap_model_scratch = 0;
            HIP_TRY(c, hipMalloc((void **)&c->d_model_scratch, sizeof(double) * need));
            c->cap_model_scratch = need;
        }
        c->P.model_scratch = c->d_model_scratch;
        const dim3 gg((unsigned)n), bb(1024);
        if (c->P.nspec == 2)
            hipLaunchKernelGGL((logprob_kernel<2, 1, 1024, true>), gg, bb, 0, s, c->P, mode, d_theta, n, ndim, d_logp, d_status);
        else
            hipLaunchKernelGGL((logprob_kernel<3, 1, 1024, true>), gg, bb, 0, s, c->P, mode, d_theta, n, ndim, d_logp, d_status);
        HIP_TRY(c, hipGetLastError());
        return MSX_OK;
    }
    const dim3 g((unsigned)n), b((unsigned)B);
    // pixel statics staged in LDS by the idle waves: 512-thread workgroups that own their CU (one per CU anyway)
    // and whose 4 npix doubles fit beside the scratch
    const bool pf = B == 512 && n <= c->prop.multiProcessorCount && !c->P.pairs_c && c->pf_ok && c->use_pf;
    // every variant is compiled for, and launched with, exactly its thread count (the canonical summation order
    // relies on it)
#define MSX_GO(NS_, U_, T_, CP_, PF_, LDS_)                                                                           \
    hipLaunchKernelGGL((logprob_kernel<NS_, U_, T_, false, CP_, PF_>), g, b, (LDS_), s, c->P, mode, d_theta, n, ndim, \
                       d_logp, d_status)
    const bool cp = c->P.pairs_c != nullptr;
    if (c->P.nspec == 2) {
        if (B == 256) { if (cp) MSX_GO(2, 2, 256, true, false, lds); else MSX_GO(2, 2, 256, false, false, lds); }
        else if (B == 512) {
            if (pf) MSX_GO(2, 2, 512, false, true, 4 * lds);
            else if (cp) MSX_GO(2, 2, 512, true, false, lds);
            else if (n > c->prop.multiProcessorCount) MSX_GO(2, 1, 512, false, false, lds);  // two workgroups per CU
            else MSX_GO(2, 2, 512, false, false, lds);
        } else MSX_GO(2, 1, 1024, false, false, lds);
    } else {
        if (B == 256) MSX_GO(3, 1, 256, false, false, lds);
        else if (B == 512) { if (pf) MSX_GO(3, 1, 512, false, true, 4 * lds); else MSX_GO(3, 1, 512, false, false, lds); }
        else MSX_GO(3, 1, 1024, false, false, lds);
    }
#undef MSX_GO
    HIP_TRY(c, hipGetLastError());
    return MSX_OK;
}

int msx_logprob_batch(msx_ctx *c, int32_t mode, const double *theta, int64_t n, int32_t ndim, double *logp_out,
                      int32_t *status_out) {
    if (!c) return MSX_ERR_INVALID;
    if (!theta || !logp_out || !status_out || n < 0) return fail(c, MSX_ERR_INVALID, "msx_logprob_batch: bad arguments");
    if (n == 0) return MSX_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    if (n > c->cap_walkers) {
        if (c->d_theta) (void)hipFree(c->d_theta);
        if (c->d_logp) (void)hipFree(c->d_logp);
        if (c->h_pin) (void)hipHostFree(c->h_pin);
        c->d_theta = c->d_logp = nullptr; c->d_status = nullptr; c->h_pin = nullptr; c->cap_walkers = 0;
        const int64_t cap = std::max<int64_t>(n, 1024);
        HIP_TRY(c, hipMalloc((void **)&c->d_theta, sizeof(double) * cap * MSX_MAX_DIM));
        // log-probs and statuses share one device allocation
        HIP_TRY(c, hipMalloc((void **)&c->d_logp, (sizeof(double) + sizeof(int32_t)) * cap));
        c->d_status = reinterpret_cast<int32_t *>(c->d_logp + cap);
        // pinned staging: async copies from / to pageable memory are staged synchronously by the runtime and
        // cost ~15 us each; through pinned memory the whole call is launch + ~12 us
        HIP_TRY(c, hipHostMalloc((void **)&c->h_pin, (sizeof(double) * (MSX_MAX_DIM + 1) + sizeof(int32_t)) * cap, hipHostMallocDefault));
        c->cap_walkers = cap;
    }
    const int64_t cap = c->cap_walkers;
    double *h_theta = reinterpret_cast<double *>(c->h_pin);
    double *h_out = h_theta + cap * MSX_MAX_DIM;  // [cap] log-probs followed by [cap] int32 statuses
    memcpy(h_theta, theta, sizeof(double) * n * ndim);
    HIP_TRY(c, hipMemcpyAsync(c->d_theta, h_theta, sizeof(double) * n * ndim, hipMemcpyHostToDevice, c->stream));
    int rc = msx_logprob_batch_dev(c, mode, c->d_theta, n, ndim, c->d_logp, c->d_status, c->stream, 0);
    if (rc) return rc;
    HIP_TRY(c, hipMemcpyAsync(h_out, c->d_logp, sizeof(double) * n, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(h_out + cap, c->d_status, sizeof(int32_t) * n, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    memcpy(logp_out, h_out, sizeof(double) * n);
    memcpy(status_out, reinterpret_cast<int32_t *>(h_out + cap), sizeof(int32_t) * n);
    return MSX_OK;
}

int msx_opt_init(msx_ctx *c, const double *theta0, int64_t nchains, int32_t ndim, double *chi2_out,
                 int32_t *status_out) {
    if (!c) return MSX_ERR_INVALID;
    if (!c->problem_staged) return fail(c, MSX_ERR_STATE, "msx_opt_init: no problem staged");
    if (!theta0 || !chi2_out || !status_out || nchains < 1) return fail(c, MSX_ERR_INVALID, "msx_opt_init: bad arguments");
    HIP_TRY(c, hipSetDevice(c->device));
    if (c->d_opt_flux) (void)hipFree(c->d_opt_flux);
    if (c->d_opt_med) (void)hipFree(c->d_opt_med);
    c->d_opt_flux = c->d_opt_med = nullptr; c->opt_chains = 0;
    c->P.opt_flux = c->P.opt_med = nullptr;
    HIP_TRY(c, hipMalloc((void **)&c->d_opt_flux, sizeof(double) * nchains * c->P.npix));
    HIP_TRY(c, hipMalloc((void **)&c->d_opt_med, sizeof(double) * nchains));
    c->opt_chains = nchains;
    c->P.opt_flux = c->d_opt_flux; c->P.opt_med = c->d_opt_med; c->P.opt_chain = nullptr;
    return msx_logprob_batch(c, MSX_MODE_OPT_INIT, theta0, nchains, ndim, chi2_out, status_out);
}

int msx_opt_step(msx_ctx *c, const double *theta, const int32_t *chain, int64_t n, int32_t ndim, double *chi2_out,
                 int32_t *status_out) {
    if (!c) return MSX_ERR_INVALID;
    if (!c->problem_staged || !c->P.opt_flux) return fail(c, MSX_ERR_STATE, "msx_opt_step: call msx_opt_init first");
    if (!theta || !chain || !chi2_out || !status_out || n < 0) return fail(c, MSX_ERR_INVALID, "msx_opt_step: bad arguments");
    if (n == 0) return MSX_OK;
    for (int64_t i = 0; i < n; ++i)
        if (chain[i] < 0 || chain[i] >= c->opt_chains) return fail(c, MSX_ERR_INVALID, "msx_opt_step: chain index out of range");
    HIP_TRY(c, hipSetDevice(c->device));
    if (n > c->cap_chain) {
        if (c->d_opt_chain) (void)hipFree(c->d_opt_chain);
        c->d_opt_chain = nullptr; c->cap_chain = 0;
        HIP_TRY(c, hipMalloc((void **)&c->d_opt_chain, sizeof(int32_t) * std::max<int64_t>(n, 1024)));
        c->cap_chain = std::max<int64_t>(n, 1024);
    }
    HIP_TRY(c, hipMemcpyAsync(c->d_opt_chain, chain, sizeof(int32_t) * n, hipMemcpyHostToDevice, c->stream));
    c->P.opt_chain = c->d_opt_chain;
    return msx_logprob_batch(c, MSX_MODE_OPT_STEP, theta, n, ndim, chi2_out, status_out);
}

// ---- device-resident sampler, pipelined -------------------------------------------------------------------
// begin: ensemble state + two slots (device chunk buffers, pinned host staging, events).  enqueue(slot): the
// chunk's randomness goes host -> pinned -> device on the copy stream while the previous chunk's kernels run,
// its 2*nsteps fused launches go on the compute stream, its chain comes back on the copy stream.  collect(slot)
// waits for that slot only.  With two slots the compute stream never drains between chunks.
struct SamplerRun {
    int32_t mode = 0, ndim = 0;
    int64_t nw = 0, ns = 0, cap_steps = 0;
    char *d_state = nullptr;
    double *d_coords = nullptr, *d_logp = nullptr, *d_q = nullptr, *d_newlp = nullptr;
    int64_t *d_nacc = nullptr;
    int32_t *d_wst = nullptr;
    hipStream_t copy = nullptr, up = nullptr;  // downloads / uploads: separate queues, or chunk i+1's upload
                                               // would wait behind chunk i's download (which waits for its kernels)
    struct Slot {
        char *d_in = nullptr, *d_out = nullptr, *h_in = nullptr, *h_out = nullptr;
        hipEvent_t in_ready = nullptr, kernels_done = nullptr, out_ready = nullptr;
        int64_t nsteps = 0;
        bool busy = false;
    } slot[2];
    size_t in_bytes(int64_t st) const { return (size_t)(st * 2 * ns) * (3 * sizeof(double) + 3 * sizeof(int32_t)); }
    size_t out_bytes(int64_t st) const {
        return sizeof(double) * (size_t)(st * nw * ndim + st * nw) + sizeof(int64_t) * (size_t)nw + 16;
    }
};

static void sampler_free(msx_ctx *c) {
    SamplerRun *r = c->smp;
    if (!r) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    if (r->copy) (void)hipStreamSynchronize(r->copy);
    if (r->up) (void)hipStreamSynchronize(r->up);
    for (auto &sl : r->slot) {
        if (sl.d_in) (void)hipFree(sl.d_in);
        if (sl.d_out) (void)hipFree(sl.d_out);
        if (sl.h_in) (void)hipHostFree(sl.h_in);
        if (sl.h_out) (void)hipHostFree(sl.h_out);
        if (sl.in_ready) (void)hipEventDestroy(sl.in_ready);
        if (sl.kernels_done) (void)hipEventDestroy(sl.kernels_done);
        if (sl.out_ready) (void)hipEventDestroy(sl.out_ready);
    }
    if (r->d_state) (void)hipFree(r->d_state);
    if (r->copy) (void)hipStreamDestroy(r->copy);
    if (r->up) (void)hipStreamDestroy(r->up);
    delete r;
    c->smp = nullptr;
    c->P.smp_on = 0;
}

int msx_sampler_begin(msx_ctx *c, int32_t mode, int64_t nw, int32_t ndim, int64_t max_chunk_steps, const double *coords,
                      const double *logp, const int64_t *naccept) {
    if (!c) return MSX_ERR_INVALID;
    if (!c->problem_staged) return fail(c, MSX_ERR_STATE, "msx_sampler_begin: no problem staged");
    if (!coords || !logp || nw < 2 || (nw & 1) || max_chunk_steps < 1)
        return fail(c, MSX_ERR_INVALID, "msx_sampler_begin: bad arguments (need an even number of walkers)");
    if (ndim != 2 * c->P.nspec + 2) return fail(c, MSX_ERR_INVALID, "P0 doesn't match what I was expecting");
    if (mode != MSX_MODE_LOGPOST && mode != MSX_MODE_LOGLIKE) return fail(c, MSX_ERR_INVALID, "msx_sampler_begin: bad mode");
    HIP_TRY(c, hipSetDevice(c->device));
    sampler_free(c);
    SamplerRun *r = new SamplerRun;
    c->smp = r;
    r->mode = mode; r->ndim = ndim; r->nw = nw; r->ns = nw / 2; r->cap_steps = max_chunk_steps;
    const int64_t ns = r->ns;
    const size_t state_bytes = sizeof(double) * (size_t)(nw * ndim + nw + ns * ndim + ns) + sizeof(int64_t) * (size_t)nw +
                               sizeof(int32_t) * (size_t)ns + 64;
    hipError_t e = hipMalloc((void **)&r->d_state, state_bytes);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&r->copy, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&r->up, hipStreamNonBlocking);
    for (auto &sl : r->slot) {
        if (e == hipSuccess) e = hipMalloc((void **)&sl.d_in, r->in_bytes(max_chunk_steps));
        if (e == hipSuccess) e = hipMalloc((void **)&sl.d_out, r->out_bytes(max_chunk_steps));
        if (e == hipSuccess) e = hipHostMalloc((void **)&sl.h_in, r->in_bytes(max_chunk_steps), hipHostMallocDefault);
        if (e == hipSuccess) e = hipHostMalloc((void **)&sl.h_out, r->out_bytes(max_chunk_steps), hipHostMallocDefault);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&sl.in_ready, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&sl.kernels_done, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&sl.out_ready, hipEventDisableTiming);
    }
    if (e == hipSuccess) {
        r->d_coords = (double *)r->d_state; r->d_logp = r->d_coords + nw * ndim; r->d_q = r->d_logp + nw;
        r->d_newlp = r->d_q + ns * ndim; r->d_nacc = (int64_t *)(r->d_newlp + ns); r->d_wst = (int32_t *)(r->d_nacc + nw);
        e = hipMemcpyAsync(r->d_coords, coords, sizeof(double) * nw * ndim, hipMemcpyHostToDevice, c->stream);
    }
    if (e == hipSuccess) e = hipMemcpyAsync(r->d_logp, logp, sizeof(double) * nw, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess)
        e = naccept ? hipMemcpyAsync(r->d_nacc, naccept, sizeof(int64_t) * nw, hipMemcpyHostToDevice, c->stream)
                    : hipMemsetAsync(r->d_nacc, 0, sizeof(int64_t) * nw, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);  // the caller's arrays are consumed on return
    if (e != hipSuccess) {
        sampler_free(c);
        return fail(c, MSX_ERR_HIP, std::string("msx_sampler_begin: ") + hipGetErrorString(e));
    }
    return MSX_OK;
}

int msx_sampler_enqueue(msx_ctx *c, int32_t slot, int64_t nsteps, const int32_t *sidx, const int32_t *cidx,
                        const int32_t *partner, const double *zz, const double *zfac, const double *logu) {
    if (!c) return MSX_ERR_INVALID;
    SamplerRun *r = c->smp;
    if (!r) return fail(c, MSX_ERR_STATE, "msx_sampler_enqueue: call msx_sampler_begin first");
    if (slot < 0 || slot > 1 || nsteps < 1 || nsteps > r->cap_steps || !sidx || !cidx || !partner || !zz || !zfac || !logu)
        return fail(c, MSX_ERR_INVALID, "msx_sampler_enqueue: bad arguments");
    SamplerRun::Slot &sl = r->slot[slot];
    if (sl.busy) return fail(c, MSX_ERR_STATE, "msx_sampler_enqueue: slot not collected yet");
    HIP_TRY(c, hipSetDevice(c->device));
    const int64_t ns = r->ns, nw = r->nw, nh = nsteps * 2 * ns;
    const int ndim = r->ndim;
    // pinned staging, doubles first: [zz | zfac | logu | sidx | cidx | partner]
    double *hz = (double *)sl.h_in;
    int32_t *hi = (int32_t *)(hz + 3 * nh);
    memcpy(hz, zz, sizeof(double) * nh); memcpy(hz + nh, zfac, sizeof(double) * nh); memcpy(hz + 2 * nh, logu, sizeof(double) * nh);
    memcpy(hi, sidx, sizeof(int32_t) * nh); memcpy(hi + nh, cidx, sizeof(int32_t) * nh);
    memcpy(hi + 2 * nh, partner, sizeof(int32_t) * nh);
    // every index is dereferenced on the device: check them here
    for (int64_t i = 0; i < nh; ++i)
        if ((uint32_t)hi[i] >= (uint32_t)nw || (uint32_t)hi[nh + i] >= (uint32_t)nw || (uint32_t)hi[2 * nh + i] >= (uint32_t)ns)
            return fail(c, MSX_ERR_INVALID, "msx_sampler_enqueue: walker / partner index out of range");
    // resolve partner -> ensemble index of the complementary walker here, so that the kernel's proposal needs
    // two dependent loads (index, coordinates) instead of three
    for (int64_t i = 0; i < nh; ++i) hi[2 * nh + i] = hi[nh + (i / ns) * ns + hi[2 * nh + i]];
    HIP_TRY(c, hipMemcpyAsync(sl.d_in, sl.h_in, r->in_bytes(nsteps), hipMemcpyHostToDevice, r->up));
    HIP_TRY(c, hipEventRecord(sl.in_ready, r->up));
    HIP_TRY(c, hipStreamWaitEvent(c->stream, sl.in_ready, 0));
    double *d_zz = (double *)sl.d_in, *d_zfac = d_zz + nh, *d_logu = d_zfac + nh;
    int32_t *d_sidx = (int32_t *)(d_logu + nh), *d_cidx = d_sidx + nh, *d_partner = d_cidx + nh;
    double *d_chain = (double *)sl.d_out, *d_lpchain = d_chain + nsteps * nw * ndim;
    int64_t *d_nacc_snap = (int64_t *)(d_lpchain + nsteps * nw);
    int32_t *d_worst = (int32_t *)(d_nacc_snap + nw);
    HIP_TRY(c, hipMemsetAsync(d_worst, 0, sizeof(int32_t), c->stream));
    DevProblem &P = c->P;
    P.smp_on = 1;
    P.smp_coords = r->d_coords; P.smp_logp = r->d_logp; P.smp_q = r->d_q; P.smp_naccept = r->d_nacc; P.smp_worst = d_worst;
    int rc = MSX_OK;
    for (int64_t st = 0; st < nsteps && rc == MSX_OK; ++st) {
        for (int half = 0; half < 2 && rc == MSX_OK; ++half) {
            const int64_t off = (st * 2 + half) * ns;
            P.smp_sidx = d_sidx + off; P.smp_cidx = d_cidx + off; P.smp_partner = d_partner + off;
            P.smp_zz = d_zz + off; P.smp_zfac = d_zfac + off; P.smp_logu = d_logu + off;
            P.smp_chain_row = d_chain + st * nw * ndim; P.smp_lp_row = d_lpchain + st * nw;
            rc = msx_logprob_batch_dev(c, r->mode, r->d_q, ns, ndim, r->d_newlp, r->d_wst, c->stream, 0);
        }
    }
    P.smp_on = 0;
    if (rc != MSX_OK) return rc;
    // acceptance counters keep running while this chunk's results travel: snapshot them in stream order
    HIP_TRY(c, hipMemcpyAsync(d_nacc_snap, r->d_nacc, sizeof(int64_t) * nw, hipMemcpyDeviceToDevice, c->stream));
    HIP_TRY(c, hipEventRecord(sl.kernels_done, c->stream));
    HIP_TRY(c, hipStreamWaitEvent(r->copy, sl.kernels_done, 0));
    HIP_TRY(c, hipMemcpyAsync(sl.h_out, sl.d_out, r->out_bytes(nsteps), hipMemcpyDeviceToHost, r->copy));
    HIP_TRY(c, hipEventRecord(sl.out_ready, r->copy));
    sl.nsteps = nsteps;
    sl.busy = true;
    return MSX_OK;
}

int msx_sampler_collect(msx_ctx *c, int32_t slot, double *chain_out, double *logp_out, int64_t *naccept,
                        int32_t *worst_status) {
    if (!c) return MSX_ERR_INVALID;
    SamplerRun *r = c->smp;
    if (!r) return fail(c, MSX_ERR_STATE, "msx_sampler_collect: call msx_sampler_begin first");
    if (slot < 0 || slot > 1 || !chain_out || !logp_out || !naccept || !worst_status)
        return fail(c, MSX_ERR_INVALID, "msx_sampler_collect: bad arguments");
    SamplerRun::Slot &sl = r->slot[slot];
    if (!sl.busy) return fail(c, MSX_ERR_STATE, "msx_sampler_collect: nothing enqueued in this slot");
    HIP_TRY(c, hipEventSynchronize(sl.out_ready));
    const int64_t st = sl.nsteps, nw = r->nw;
    const double *h_chain = (const double *)sl.h_out, *h_lp = h_chain + st * nw * r->ndim;
    const int64_t *h_nacc = (const int64_t *)(h_lp + st * nw);
    memcpy(chain_out, h_chain, sizeof(double) * st * nw * r->ndim);
    memcpy(logp_out, h_lp, sizeof(double) * st * nw);
    memcpy(naccept, h_nacc, sizeof(int64_t) * nw);
    *worst_status = *(const int32_t *)(h_nacc + nw);
    sl.busy = false;
    return MSX_OK;
}

int msx_sampler_end(msx_ctx *c, double *coords, double *logp) {
    if (!c) return MSX_ERR_INVALID;
    SamplerRun *r = c->smp;
    if (!r) return MSX_OK;
    hipError_t e = hipSetDevice(c->device);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess && coords) e = hipMemcpy(coords, r->d_coords, sizeof(double) * r->nw * r->ndim, hipMemcpyDeviceToHost);
    if (e == hipSuccess && logp) e = hipMemcpy(logp, r->d_logp, sizeof(double) * r->nw, hipMemcpyDeviceToHost);
    sampler_free(c);
    if (e != hipSuccess) return fail(c, MSX_ERR_HIP, std::string("msx_sampler_end: ") + hipGetErrorString(e));
    return MSX_OK;
}

// one synchronous chunk (the pipelined entry points above, used back to back)
int msx_sampler_run(msx_ctx *c, int32_t mode, int64_t nw, int32_t ndim, int64_t nsteps, double *coords, double *logp,
                    const int32_t *sidx, const int32_t *cidx, const int32_t *partner, const double *zz,
                    const double *zfac, const double *logu, double *chain_out, double *logp_out, int64_t *naccept,
                    int32_t *worst_status) {
    if (!c) return MSX_ERR_INVALID;
    if (!coords || !logp || !chain_out || !logp_out || !naccept || !worst_status || nsteps < 1)
        return fail(c, MSX_ERR_INVALID, "msx_sampler_run: bad arguments (need an even number of walkers)");
    int rc = msx_sampler_begin(c, mode, nw, ndim, nsteps, coords, logp, naccept);
    if (rc == MSX_OK) rc = msx_sampler_enqueue(c, 0, nsteps, sidx, cidx, partner, zz, zfac, logu);
    if (rc == MSX_OK) rc = msx_sampler_collect(c, 0, chain_out, logp_out, naccept, worst_status);
    if (rc == MSX_OK) return msx_sampler_end(c, coords, logp);
    const std::string keep = c->err;
    sampler_free(c);
    c->err = keep;
    return rc;
}

int msx_make_composite(msx_ctx *c, const double *teff, const double *logg, const double *rad, int32_t use_distance,
                       double plx, double *spec_out, double *contrast_out, double *phot_out, int32_t *status_out) {
    if (!c) return MSX_ERR_INVALID;
    if (!c->problem_staged) return fail(c, MSX_ERR_STATE, "msx_make_composite: no problem staged");
    if (!teff || !logg || !rad || !spec_out || !status_out) return fail(c, MSX_ERR_INVALID, "msx_make_composite: bad arguments");
    HIP_TRY(c, hipSetDevice(c->device));
    const int ns = c->P.nspec;
    double args[3 * MSX_MAX_SPEC + 1];
    for (int i = 0; i < ns; ++i) { args[i] = teff[i]; args[ns + i] = logg[i]; args[2 * ns + i] = rad[i]; }
    args[3 * ns] = plx;
    if (c->P.win_n > c->cap_spec) {
        if (c->d_spec) (void)hipFree(c->d_spec);
        c->d_spec = nullptr; c->cap_spec = 0;
        HIP_TRY(c, hipMalloc((void **)&c->d_spec, sizeof(double) * c->P.win_n));
        c->cap_spec = c->P.win_n;
    }
    double *d_args = c->d_misc;
    WalkerDesc *d_desc = reinterpret_cast<WalkerDesc *>(c->d_misc + 64);
    static_assert(sizeof(WalkerDesc) + 64 * sizeof(double) <= 4096, "misc buffer too small");
    HIP_TRY(c, hipMemcpyAsync(d_args, args, sizeof(double) * (3 * ns + 1), hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(composite_setup_kernel, dim3(1), dim3(64), 0, c->stream, c->P, d_args, (int)use_distance, d_desc);
    HIP_TRY(c, hipGetLastError());
    hipLaunchKernelGGL(composite_kernel, dim3((unsigned)((c->P.win_n + 255) / 256)), dim3(256), 0, c->stream, c->P, d_desc,
                       c->d_spec);
    HIP_TRY(c, hipGetLastError());
    WalkerDesc h;
    HIP_TRY(c, hipMemcpyAsync(&h, d_desc, sizeof(WalkerDesc), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    *status_out = h.status;
    if (h.status != MSX_W_OK) return MSX_OK;
    HIP_TRY(c, hipMemcpy(spec_out, c->d_spec, sizeof(double) * c->P.win_n, hipMemcpyDeviceToHost));
    for (int i = 0; i < c->P.nc && contrast_out; ++i) contrast_out[i] = h.contrast[i];
    for (int i = 0; i < c->P.np && phot_out; ++i) phot_out[i] = h.phot[i];
    return MSX_OK;
}

int msx_comm_unique_id(msx_ctx *c, uint8_t *out128) {
    if (!c || !out128) return MSX_ERR_INVALID;
    if (!rccl().ok) return fail(c, MSX_ERR_STATE, "RCCL (librccl.so.1) could not be resolved in this process");
    RcclUniqueId id;
    const int rc = rccl().GetUniqueId(&id);
    if (rc != 0) return fail(c, MSX_ERR_HIP, std::string("ncclGetUniqueId: ") + rccl().GetErrorString(rc));
    memcpy(out128, id.internal, 128);
    return MSX_OK;
}

int msx_comm_init(msx_ctx *c, const uint8_t *id128, int32_t rank, int32_t world) {
    if (!c || !id128 || world < 1 || rank < 0 || rank >= world) return fail(c, MSX_ERR_INVALID, "msx_comm_init: bad arguments");
    if (!rccl().ok) return fail(c, MSX_ERR_STATE, "RCCL (librccl.so.1) could not be resolved in this process");
    if (c->rccl_comm) return fail(c, MSX_ERR_STATE, "msx_comm_init: communicator already initialised");
    HIP_TRY(c, hipSetDevice(c->device));
    RcclUniqueId id;
    memcpy(id.internal, id128, 128);
    const int rc = rccl().CommInitRank(&c->rccl_comm, world, id, rank);
    if (rc != 0) {
        c->rccl_comm = nullptr;
        return fail(c, MSX_ERR_HIP, std::string("ncclCommInitRank: ") + rccl().GetErrorString(rc));
    }
    HIP_TRY(c, hipStreamCreateWithFlags(&c->comm_stream, hipStreamNonBlocking));
    HIP_TRY(c, hipEventCreateWithFlags(&c->ev_ready, hipEventDisableTiming));
    for (hipEvent_t &e : c->ev_done) HIP_TRY(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    c->comm_world = world;
    c->comm_rank = rank;
    return MSX_OK;
}

int msx_comm_allgather_dev(msx_ctx *c, const double *d_send, double *d_recv, int64_t count, void *compute_stream,
                           int32_t slot) {
    if (!c || !d_send || !d_recv || count < 1 || slot < 0 || slot > 3) return fail(c, MSX_ERR_INVALID, "msx_comm_allgather_dev: bad arguments");
    if (!c->rccl_comm) return fail(c, MSX_ERR_STATE, "msx_comm_allgather_dev: call msx_comm_init first");
    // the collective starts once everything queued so far on the compute stream is done, runs on the
    // communicator's own stream (so the next launch overlaps it) and signals the slot's event
    HIP_TRY(c, hipEventRecord(c->ev_ready, (hipStream_t)compute_stream));
    HIP_TRY(c, hipStreamWaitEvent(c->comm_stream, c->ev_ready, 0));
    const int rc = rccl().AllGather(d_send, d_recv, (size_t)count, kNcclFloat64, c->rccl_comm, c->comm_stream);
    if (rc != 0) return fail(c, MSX_ERR_HIP, std::string("ncclAllGather: ") + rccl().GetErrorString(rc));
    HIP_TRY(c, hipEventRecord(c->ev_done[slot], c->comm_stream));
    return MSX_OK;
}

int msx_comm_wait_slot(msx_ctx *c, int32_t slot, void *compute_stream) {
    if (!c || slot < 0 || slot > 3) return MSX_ERR_INVALID;
    if (!c->rccl_comm) return fail(c, MSX_ERR_STATE, "msx_comm_wait_slot: call msx_comm_init first");
    HIP_TRY(c, hipStreamWaitEvent((hipStream_t)compute_stream, c->ev_done[slot], 0));
    return MSX_OK;
}

int msx_stream_copy_gbps(msx_ctx *c, int64_t bytes, int32_t iters, double *gbps_out) {
    if (!c || !gbps_out || bytes < 4096 || iters < 1) return fail(c, MSX_ERR_INVALID, "msx_stream_copy_gbps: bad arguments");
    HIP_TRY(c, hipSetDevice(c->device));
    const int64_t n4 = bytes / 16;
    float4 *a = nullptr, *b = nullptr;
    HIP_TRY(c, hipMalloc((void **)&a, n4 * 16));
    HIP_TRY(c, hipMalloc((void **)&b, n4 * 16));
    HIP_TRY(c, hipMemsetAsync(a, 1, n4 * 16, c->stream));
    hipEvent_t e0, e1;
    HIP_TRY(c, hipEventCreate(&e0));
    HIP_TRY(c, hipEventCreate(&e1));
    const int blocks = c->prop.multiProcessorCount * 8;  // 2048 workgroups of 256 threads
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(copy_float4_kernel, dim3(blocks), dim3(256), 0, c->stream, a, b, n4);
    HIP_TRY(c, hipEventRecord(e0, c->stream));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(copy_float4_kernel, dim3(blocks), dim3(256), 0, c->stream, a, b, n4);
    HIP_TRY(c, hipEventRecord(e1, c->stream));
    HIP_TRY(c, hipEventSynchronize(e1));
    float ms = 0.f;
    HIP_TRY(c, hipEventElapsedTime(&ms, e0, e1));
    *gbps_out = (2.0 * (double)(n4 * 16) * iters) / ((double)ms * 1e-3) / 1e9;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipFree(a);
    (void)hipFree(b);
    return MSX_OK;
}

int msx_bytes_per_eval(msx_ctx *c, int64_t *requested_bytes) {
    if (!c || !requested_bytes) return MSX_ERR_INVALID;
    if (!c->problem_staged) return fail(c, MSX_ERR_STATE, "msx_bytes_per_eval: no problem staged");
    const int64_t npix = c->P.npix;
    // pair rows (16 B x corners) + per-pixel statics read in phases A and C
    const int64_t pair_bytes = (c->P.pairs_c && c->P.nspec == 2) ? 12 : 16;
    *requested_bytes = npix * (pair_bytes * (int64_t)c->P.nspec * 4 + 16 + 8 * 3 + 8 * 3) + 8 * (2 * c->P.nspec + 2) + 12;
    return MSX_OK;
}

}  // extern "C"
