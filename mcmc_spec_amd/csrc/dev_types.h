// dev_types.h -- part of the single translation unit msx.hip (included there, in this order).
// constants and the device-side structs: DevProblem (the staged problem, a by-value kernel argument), WalkerDesc (a walker's recipe, in LDS).
#ifndef MSX_DEV_TYPES_H
#define MSX_DEV_TYPES_H

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
struct SmpRec {  // the stretch move of one walker of a half-step
    int32_t si, ci;  // ensemble index of the walker and of its partner in the complementary half
    double zz;       // the stretch factor
    // overlapped half-steps (smp_overlap): the VERSIONS (= updates so far) of the two walkers this move reads -- the
    // workgroup waits until the ensemble holds them, and finds version v of a walker in coordinate buffer v & 1
    uint32_t ver_own, ver_partner;
};
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
    // Tables of the blend (A2, A4, A7, A8.1), in ELEMENTS of two pixels {pa, pa + 256}, pa = (e >> 8) * 512 +
    // (e & 255), e < npair (= the pixel count padded to a multiple of 512, halved; pad pixels repeat the last one):
    const double2 *r2;     // [nt*ng][npair]  R = flux[lo] + (flux[lo+1] - flux[lo]) t   (blend_pixel_rh, blend.h)
    const float2 *h2;      // [nt*ng][npair]  H = flux[lo+1] t
    const float2 *r2f;     // [nt*ng][npair]  R rounded to float32: only with msx_set_grid_storage(MSX_STORE_F32), else null
    const float4 *r4f, *r4fb;  // ... and by quad ({eA, eB = eA + 512} / {eA, eA + 256}), like h4 / h4b: one load per corner and trip
    const double2 *kl2;    // [npair]         CCM89 k[lo]
    const float2 *dk2;     // [npair]         k[lo+1] - k[lo]
    const double2 *f2, *u2, *iv2;  // [npair] data flux, mapped wavelength u, 1/err^2 (element copies of pix_flux, pix_u, pix_ivar)
    int64_t npair;
    // ... and the float32 ones in quads (staging_kernels.h::gather_quads_kernel): quad q = elements {1024 (q >> 9) + (q & 511), + 512}
    const float4 *h4;      // [nt*ng][nquad]   512-thread workgroups: eB = eA + 512
    const float4 *dk4;     // [nquad]
    const float4 *h4b;     // [nt*ng][nquad]   256-thread workgroups: eB = eA + 256
    const float4 *dk4b;    // [nquad]
    int64_t nquad;         // ceil(npair / 1024) * 512
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
    double *model_scratch;     // [rows][npix]: the model vectors of the GM variants (spectra beyond the LDS) and of the
                               // linked form's rare walkers whose median needs the whole vector in one place
    struct SegPart *segparts;  // [rows][segments] linked form: the segments' partials
    unsigned long long *seg_flag;  // [rows] linked form: arrivals at the walker's two meeting points, counted up for ever
                                   // (2 x segments per launch; 64 bits: never wraps)
    int32_t *linked_poison;    // linked form: != 0 once a hand-over has timed out on this context -- every later linked
                               // launch fails all its walkers with MSX_W_HANDOVER until msx_stage_problem clears it
    const struct PairItem *pair_items;   // pair form: [rows / 2] the planner's pairs, recipes included
    const struct PairRec *pair_singles;  // ... and [rows] its singles
    int32_t *pair_lease;                 // pair form: [kPairSpillRows] leases of the spill path's scratch rows
    int32_t linked_fault;      // test hook (msx_test_hook / MSX_LINKED_FAULT=1): nobody signals its arrival, every wait must time out
    // device-resident stretch move (f2): when smp_on, walker wk of the launch is the wk-th walker of the
    // active half; the kernel builds its own proposal and applies the accept rule in its last lines
    int32_t smp_on;
    int32_t smp_defer;  // sharded sampler: the kernel only publishes log p(q); accept + state update run in
                        // sampler_apply_kernel after the ranks' all-gather (every rank applies every walker)
    double *smp_coords, *smp_logp;          // [nw][ndim], [nw]   ensemble state (updated in place)
    double *smp_q;                          // [ns][ndim]         proposals of this half-step
    const int32_t *smp_sidx, *smp_cidx, *smp_partner;  // [ns]; smp_partner holds cidx[partner]: the ensemble
                                                       // index of the complementary walker (resolved on the host)
    const double *smp_zz, *smp_zfac, *smp_logu;        // [ns]
    const SmpRec *smp_rec;                  // [ns] {sidx, cidx[partner], zz} again, one record per walker: what the
                                            // proposal needs, behind ONE pointer that reaches the kernel preloaded
    int64_t *smp_naccept;                   // [nw]
    double *smp_chain_row, *smp_lp_row;     // chain[step] [nw][ndim], logp chain[step] [nw]
    int32_t *smp_worst;
    // overlapped half-steps (msx.hip, chunk_half_eval): consecutive half-steps are launched on two streams and run
    // concurrently; a walker's workgroup waits for the versions of the two walkers it reads (SmpRec), the state is
    // double-buffered by version parity (smp_coords = [2][nw][ndim], smp_stride = nw * ndim) and published per walker
    int32_t smp_overlap;
    int64_t smp_stride;
    uint32_t *smp_ver;                      // [nw] (rounds 2-3: the version word behind the data; unused since the granules)
    // ... handed over as TAGGED GRANULES: what a move reads of a walker -- its coordinates, its log-probability, its
    // acceptance count -- lives once more in smp_gran[version & 1][walker][kGranPerWalker], every 8-byte word {32 bits of
    // payload | the walker's version}, written by ONE agent-scope store each.  A reader polls the words it wants until they
    // carry the version the move is defined on: the load that sees the tag has the data -- no flag behind the data, no wait
    // for store acknowledgements on the writer's side, no acquire + second round trip on the reader's.
    unsigned long long *smp_gran;
    int64_t smp_gwalkers;                   // nw (words per parity buffer = nw * kGranPerWalker)
    // clock probe (msx_probe_launch; bit 21 of the packed launch word): thread 0 of a walker's workgroup leaves the 100 MHz
    // wall clock and the shader-cycle counter at its first and last line -- [walker][4] for the first kProbeWalkers
    // walkers: the clock the CUs ran at under THIS kernel's load, and the walker's own time inside the launch
    unsigned long long *clk_probe;
    // in-path broadening (inpath_kernels.h; MSX_PATH_INPATH): the walkers' model vectors, computed before the launch by the
    // in-path kernels -- [walker][given_stride] in PIXEL order -- and read by logprob_kernel's GIVEN variant in place of the blend
    const double *given;
    int64_t given_stride;
#ifdef MSX_STAMPS
    unsigned long long *stamps;  // diagnostic build only: [walker][16] shader-clock stamps
#endif
};
constexpr int kProbeWalkers = 4096;
// what the in-path kernels need of a walker's recipe (inpath_recipe_kernel -> inpath_conv_kernel, inpath_resample_kernel)
struct InpathRec {
    double w[8];
    int32_t node[8];
    double redc;
    int32_t ok, pad;
};
// granules of one walker (smp_gran): coordinate d = words 2 d (high half) and 2 d + 1 (low half); then the log-probability's
// two halves and the low 32 bits of the acceptance count
constexpr int kGranPerWalker = 2 * MSX_MAX_DIM + 4, kGranLogp = 2 * MSX_MAX_DIM, kGranNacc = 2 * MSX_MAX_DIM + 2;
__host__ __device__ inline unsigned long long granule(unsigned int payload, unsigned int version) {
    return ((unsigned long long)payload << 32) | (unsigned long long)version;
}

#ifdef MSX_STAMPS
__shared__ int msx_stamp_off;  // linked form: one of a walker's workgroups writes the stamps
#define MSX_STAMP(P, wk, i) do { if (threadIdx.x == 0 && !msx_stamp_off) (P).stamps[(wk) * 16 + (i)] = __builtin_readcyclecounter(); } while (0)
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
    uint32_t smp_ver;                     // overlapped half-steps: its version before this move
};

// The recipe's small tables, gathered into ONE block so that a single preloaded pointer reaches them all (fixed
// offsets: the register-resident recipe takes at most 256 isochrone points, 64 Teff x 32 logg nodes; the presence
// mask is one uint32 per Teff node, bit g = node (t, g) is in the grid).
constexpr int kRbIsoT = 0, kRbIsoG = 2048, kRbTeff = 4096, kRbLogg = 4608, kRbPresent = 5120;
// ... and once more PACKED for the wave form's lanes (load_recipe_regs): what lane l wants of an entry sits side by side,
// pads included, so that the kernel's prologue is a handful of unconditional 16-byte loads and no arithmetic --
//   kRbIsoPack   [256] {x_i, x_i+1 (+inf past the end), y_i, slope_i = (y_i+1 - y_i) / (x_i+1 - x_i) (0 for the last entry)}
//                      (entries >= niso: {+inf, +inf, 0, 0}; the slope is formed on the host, IEEE division like the
//                      device's `/`: the same bits as the planner's one-thread form computes for itself)
//   kRbTeffPack  [64] {node_l, node_l+1} (+inf pads)      kRbLoggPack [64] likewise
//   kRbMaskPack  [64] {presence bits of Teff node l, of node l + 1} (0 pads)
constexpr int kRbIsoPack = 5376, kRbTeffPack = kRbIsoPack + 256 * 32, kRbLoggPack = kRbTeffPack + 64 * 16,
              kRbMaskPack = kRbLoggPack + 64 * 16, kRecipeBlockBytes = kRbMaskPack + 64 * 8;
static_assert(kRbIsoPack % 16 == 0 && kRecipeBlockBytes == 16128, "recipe block layout");
constexpr int kSegElems = 4096;     // table elements (= 8192 pixels) per segment of the canonical sum / of the linked form
constexpr int kSegBins = 2048;      // (= kLogBins, median.h)
constexpr unsigned long long kHandoverTicks = 2000000ull;  // in-kernel waits (the linked form's meetings, the overlapped sampler's versions) give up after 20 ms of the 100 MHz wall clock

// pair form: a walker's recipe as the planner (one thread per walker) leaves it for the pair kernel
struct alignas(16) PairRec {
    double w[kMaxCorners > 8 ? 8 : kMaxCorners];  // bilinear weight x (R/d)^2 per corner, canonical (sorted-node) order
    double redc;                                  // exp2 coefficient of the reddening; 0: none
    double lp, chi_extra;                         // the Gaussian prior terms; icontrast + iphot
    int32_t node[8];
    int32_t walker, pad;                          // index in the (sub-)batch
};
static_assert(sizeof(PairRec) == 128, "PairRec layout");
struct alignas(16) PairItem { PairRec r[2]; };    // two walkers of one grid cell

// linked form: what the workgroup of one (walker, segment) leaves for the walker's other workgroups
struct alignas(16) SegPart {
    // first exchange (before the median can be located)
    double q[3];                    // the segment's three fit sums
    unsigned long long kmin, kmax;  // its range of F(m) = hi32(m) >> 12 (median.h; ~0 / 0 = empty)
    // second exchange (for whichever workgroup finishes the walker)
    double chi;                     // the segment's chi^2 sum (canonical, before scale^2)
    unsigned int ncand, pad[3];     // candidates of the median's bin(s) among the segment's values
    unsigned int hist[kSegBins];    // first exchange: its share of the median's logarithmic histogram
    unsigned long long cand[kSelectFinish];
};
static_assert(sizeof(SegPart) == 64 + 4 * kSegBins + 8 * kSelectFinish, "SegPart layout");

}  // namespace

#endif  // MSX_DEV_TYPES_H
