// recipe.h -- part of the single translation unit msx.hip (included there, in this order).
// phase 0 of the hot kernel: prior gates, isochrone lookup, node brackets, bilinear weights, prior and band terms (generic memory-walking form and the register-resident wave form).
#ifndef MSX_RECIPE_H
#define MSX_RECIPE_H

namespace {

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
        int nd4[4] = {n11, n12, n21, n22};
        double w4[4] = {(1.0 - b) * (1.0 - a) * sc, (1.0 - b) * a * sc, b * (1.0 - a) * sc, b * a * sc};
        sort4_by_node(nd4, w4);  // canonical corner order (blend.h)
        for (int k = 0; k < 4; ++k) { D->node[4 * s + k] = nd4[k]; D->w[4 * s + k] = w4[k]; }
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
// what the gates read besides theta.  The fast recipe gets these from PRELOADED kernel arguments (logprob_kernel):
// fetched from the DevProblem they would each cost a scalar-cache round trip in the walker's critical chain.
struct GateArgs {
    double tmin, tmax;
    bool dist_fit, use_av;
};
__device__ __forceinline__ GateArgs gates_of(const DevProblem &P) { return GateArgs{P.tmin, P.tmax, P.dist_fit != 0, P.use_av != 0}; }
template <int NS>
__device__ __forceinline__ bool prior_gates(const GateArgs &P, const double *t) {
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
        alive = alive && prior_gates<NS>(gates_of(P), t);
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
    // A1 + A2 + A4.  The reference interpolates every star's logg before it builds the first star's spectrum
    // (mft6.py:1149): a Teff outside the isochrone, on ANY star, raises before any bracket can
    int node[NS * 4];
    double w[NS * 4];
#pragma unroll
    for (int s = 0; s < NS; ++s)
        if (!(t[s] >= T.iso_t[0]) || !(t[s] <= T.iso_t[P.niso - 1])) st = MSX_W_VALUEERROR;
    if (st != MSX_W_OK) {
        if (lane == 0) D.status = st;
        return;
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) {
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
        int nd4[4] = {n11, n12, n21, n22};
        double w4[4] = {(1.0 - b) * (1.0 - a) * sc, (1.0 - b) * a * sc, b * (1.0 - a) * sc, b * a * sc};
        sort4_by_node(nd4, w4);  // canonical corner order (blend.h)
#pragma unroll
        for (int k = 0; k < 4; ++k) { node[4 * s + k] = nd4[k]; w[4 * s + k] = w4[k]; }
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

// ---- the arithmetic of the prior and band terms, in ONE place: the wave forms below (tables in registers, lanes) and
// the one-thread forms of the pair form's planner (tables in LDS, loops) fetch their operands differently and then call
// these, so that a walker's terms have the same bits whichever form computed them -------------------------------------
__device__ __forceinline__ double interp_segment(double x0, double y0, double x1, double y1, double x) {  // np.interp, x0 <= x < x1
    if (x0 == x) return y0;
    const double slope = (y1 - y0) / (x1 - x0);
    return fma(slope, x - x0, y0);
}
__device__ __forceinline__ double gauss_term_z(double lp, double z) { return lp + -0.5 * (z * z); }
__device__ __forceinline__ double gauss_term(double lp, double x, double mean, double sig) {  // lp - 0.5 ((x - mean) / sig)^2
    const double z = (x - mean) / sig;
    return gauss_term_z(lp, z);
}
__device__ __forceinline__ double model_radius(double lum, double teff) {  // get_radius, mft6.py:66-85 (Stefan-Boltzmann)
    const double sigma_sb = 5.670374e-5, lsun = 3.839e33;
    const double t2 = teff * teff;
    return sqrt(lum * lsun / (4 * M_PI * sigma_sb * (t2 * t2))) / kRsunCm;  // mft6.py:83
}
__device__ __forceinline__ double radius_term(double lp, double rad, double target) {  // mft6.py:1262-1269
    const double z = (rad - target) / (0.02 * target);
    return lp + -0.5 * (z * z);
}
// magnitude of one "job" of the band terms: job < nc NS = (contrast filter f, star s), else photometric band f
template <int NS>
__device__ __forceinline__ double band_job_value(const DevProblem &P, const int *node, const double *w, int job) {
    const int nb = P.nc + P.np;
    if (job < P.nc * NS) {
        const int f = job / NS, s = job - f * NS;
        double m = 0.0;
        for (int c = 0; c < 4; ++c) m += w[4 * s + c] * P.band_tab[(int64_t)node[4 * s + c] * nb + f];
        return -2.5 * log10(m);  // mft6.py:733
    }
    const int f = job - P.nc * NS;
    double flux = 0.0;
    for (int c = 0; c < NS * 4; ++c) flux += w[c] * P.band_tab[(int64_t)node[c] * nb + P.nc + f];
    return -2.5 * log10(flux / P.pzero[f]);  // mft6.py:780-782
}
// icontrast + iphot from the jobs' magnitudes (val(k) = magnitude of job k).  The sum's fused multiply-adds are WRITTEN
// (the contraction the compiler chose here anyway): the one-thread form below has the same sum spread over branches, where
// a contraction left to the compiler came out as multiply + add -- one ulp of the total apart.
template <int NS, class V>
__device__ __forceinline__ double band_chi(const DevProblem &P, bool redden, double a_v, V val) {
    double chi = 0.0;
    for (int f = 0; f < P.nc; ++f) {
        int sec = 1;
        if (NS == 3 && f >= P.nc / 2) sec = 2;  // mft6.py:747-749
        const double con = val(f * NS + sec) - val(f * NS);  // mft6.py:741
        const double z = con - P.cmag[f];
        chi = fma(z * z, P.civar[f], chi);  // mft6.py:120,1182
    }
    for (int f = 0; f < P.np; ++f) {
        const double mag = val(P.nc * NS + f);
        const double mred = redden ? fma(a_v, P.pk[f], mag) : mag;  // mft6.py:1163
        const double z = mred - P.pmag[f];
        chi = fma(z * z, P.pivar[f], chi);  // mft6.py:1188
    }
    return chi;
}

// np.interp on a register-resident table (search by count; the radius prior's luminosity lookup); caller checked the range
__device__ __forceinline__ double iso_interp_regs(const double (&xs)[4], const double (&ys)[4], int n, double x) {
    int cnt = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) cnt += __popcll(__ballot(xs[k] <= x));  // pads are +inf
    const int j = uni(cnt) - 1;
    if (j >= n - 1) return pick4(ys, n - 1);
    return interp_segment(pick4(xs, j), pick4(ys, j), pick4(xs, j + 1), pick4(ys, j + 1), x);
}

// ------------------------------------------------------------------------------------------------
// The register-resident recipe (the usual case: <= 256 isochrone points, <= 64 Teff and <= 32 logg nodes).
// One wave per star.  The tables are loaded in the kernel's very first instructions from a pointer that arrives
// PRELOADED in SGPRs (see logprob_kernel), one table entry per lane TOGETHER WITH ITS RIGHT NEIGHBOUR, and
// everything that does not depend on theta (the isochrone's slopes) is computed while theta is still in flight.
// When theta arrives every lane evaluates "its" interval speculatively -- lane l: is x in [x_l, x_l+1), and if so
// what is the interpolated value / the nearest node / the bracket -- and a ballot picks the one lane that is right:
// a handful of dependent instructions where a search by count + readlane took dozens (the chain from theta to the
// walker's weights is the kernel's start-up latency: nothing else can begin before it ends).
// ------------------------------------------------------------------------------------------------
struct RecipeRegs {
    double isot[4], isot_n[4], isog[4], slope[4];  // isochrone element i = lane + 64 k: x_i, x_i+1 (+inf past the end),
                                                   // y_i, (y_i+1 - y_i) / (x_i+1 - x_i) (0 for the last element)
    double tn, tn_n, gn, gn_n;                     // Teff / logg node l and l + 1 (+inf pads)
    unsigned int m0, m1;                           // presence bits (bit g: node (l, g) is in the grid) of Teff node l, l + 1
    double iso_lo, iso_hi;                         // (uniform) the isochrone's Teff range
    double t_first, t_last, g_first, g_last;       // (uniform) first / last node
    unsigned int m_first, m_last;                  // (uniform) presence bits of the first / last Teff node
};
__device__ __forceinline__ void load_recipe_regs(RecipeRegs &R, const unsigned char *__restrict__ rblk, int niso, int nt, int ng,
                                                 int lane) {
    // the PACKED copies of the tables (dev_types.h, kRb*Pack; built by msx_stage_problem): every lane's entry with its right
    // neighbour, the isochrone's slopes and all pads ready-made -- eleven unconditional loads, nothing to compute, so the
    // recipe's chain starts the moment theta (requested BEFORE these, logprob_kernel) arrives
    const double4 *__restrict__ iso = reinterpret_cast<const double4 *>(rblk + kRbIsoPack);
    const double2 *__restrict__ tp = reinterpret_cast<const double2 *>(rblk + kRbTeffPack);
    const double2 *__restrict__ gp = reinterpret_cast<const double2 *>(rblk + kRbLoggPack);
    const uint2 *__restrict__ mp = reinterpret_cast<const uint2 *>(rblk + kRbMaskPack);
    double4 e[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) e[k] = iso[lane + kWave * k];
    const double2 t2 = tp[lane], g2 = gp[lane];
    const uint2 m2 = mp[lane];
#pragma unroll
    for (int k = 0; k < 4; ++k) { R.isot[k] = e[k].x; R.isot_n[k] = e[k].y; R.isog[k] = e[k].z; R.slope[k] = e[k].w; }
    R.tn = t2.x; R.tn_n = t2.y; R.gn = g2.x; R.gn_n = g2.y;
    R.m0 = m2.x; R.m1 = m2.y;
    R.iso_lo = pick4(R.isot, 0);
    R.iso_hi = pick4(R.isot, niso - 1);
    R.t_first = readlane_f64(R.tn, 0);
    R.t_last = readlane_f64(R.tn, nt - 1);
    R.g_first = readlane_f64(R.gn, 0);
    R.g_last = readlane_f64(R.gn, ng - 1);
    R.m_first = (unsigned int)__builtin_amdgcn_readlane((int)R.m0, 0);
    R.m_last = (unsigned int)__builtin_amdgcn_readlane((int)R.m0, nt - 1);
}

// np.interp on the register-resident isochrone for lo <= x <= hi (the caller checked the range).  Exactly one
// (register, lane) holds the interval [x_i, x_i+1) that contains x -- the LAST i with x_i <= x, like the
// search-by-count (duplicated abscissae: their zero-length intervals are never hit).  x on a node gives y_i exactly.
__device__ __forceinline__ double iso_interp_lanes(const RecipeRegs &R, double x) {
    double v = 0.0;
    bool hit = false;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const bool h = R.isot[k] <= x && x < R.isot_n[k];
        const double val = fma(R.slope[k], x - R.isot[k], R.isog[k]);  // (explicit: recipe_scalar must form the same bits)
        v = h ? val : v;
        hit = hit || h;
    }
    const int L = uni(__ffsll((long long)__ballot(hit)) - 1);
    return readlane_f64(v, L & 63);  // (no lane: x is NaN -- the range check sent it away already)
}

// The reference's bracket of v in a sorted node list (mft6.py:439-477: nearest node first, ties to the lower, the
// other side next; on a node both are that node; below the first node the "other" index -1 wraps to the LAST node;
// beyond the last there is no other node -> IndexError), evaluated by the lane whose interval [n_l, n_l+1) holds v.
struct LaneBracket {
    int i1, i2, st;
    double e1, e2;
    bool own;
};
__device__ __forceinline__ LaneBracket lane_bracket(double nl, double nn, int l, int n, double v) {
    LaneBracket B;
    B.own = nl <= v && v < nn;                      // (pads are +inf: never for l >= n)
    const bool up = fabs(nn - v) < fabs(nl - v);   // the upper node is strictly nearer
    const bool eq = nl == v;
    B.i1 = up ? l + 1 : l;
    B.i2 = (up || eq) ? l : l + 1;
    B.e1 = up ? nn : nl;
    B.e2 = (up || eq) ? nl : nn;
    B.st = (!up && !eq && l + 1 >= n) ? MSX_W_INDEXERROR : MSX_W_OK;
    return B;
}
// the same bracket by count + readlane (uniform result): v is NaN, or some other case no lane owns
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
    *i1 = best;
    *i2 = best;
    *e1 = nb;
    *e2 = nb;
    if (other >= n) return MSX_W_INDEXERROR;
    other = uni(other);
    *i2 = other;
    *e2 = readlane_f64(nodes, other);
    return MSX_W_OK;
}

// Part 1 (gates phase A): the gates, A1 logg, A2 brackets + presence, A4 weights.  Writes D.node, D.w, D.redc (star
// 0's wave), D.stat.  `theta_lane`: lane k < 2 NS + 2 holds coordinate k (the kernel's first load); tv: the same
// values, uniform.  The gates of the prior (mft6.py:1227-1230 binary, :1347-1350 triple; emcee refuses non-finite
// coordinates) are ONE compare per lane against that lane's bounds -- off the chain altogether; they still take
// precedence in the reported status, like logprior before loglikelihood in the reference.  The chain itself is safe
// for any theta: no memory access depends on it.
template <int NS>
__device__ __forceinline__ void recipe_part1_regs(const DevProblem &P, const GateArgs &G, const RecipeRegs &R, int niso, int nt,
                                                  int ng, int mode, const double theta_lane, const double (&tv)[2 * NS + 2],
                                                  WalkerDesc &D, int lane, int64_t wk, const int star) {
    constexpr int ND = 2 * NS + 2;
    MSX_STAMP(P, wk, 9);
    // ---- the gates: lane k tests coordinate k ------------------------------------------------------------------
    bool alive;
    {
        double lo = -INFINITY, hi = INFINITY;
        if (mode == MSX_MODE_LOGPOST || mode == MSX_MODE_LOGPRIOR) {
            if (lane < NS) { lo = G.tmin; hi = G.tmax; }
            if (lane == NS && G.use_av) lo = 0.0;
            if (G.dist_fit) {
                if (lane > NS && lane <= 2 * NS) lo = 0.05;
                if (NS == 2 && lane == NS + 1) hi = 1.5;
                if (lane == 2 * NS + 1) { lo = NS == 2 ? 1.0 / 3000 : 1.0 / 1000; hi = 1.0 / 4; }
            } else if (NS == 2) {
                if (lane > NS && lane <= 2 * NS) lo = 0.05;
            } else {
                if (lane > NS + 1 && lane <= 2 * NS) lo = 0.05;
                if (lane == 2 * NS + 1) lo = 0.0;
            }
        }
        const bool ok = lane >= ND || (isfinite(theta_lane) && !(theta_lane < lo) && !(theta_lane > hi));
        alive = __ballot(ok) == ~0ull;
    }
    // this wave's star: Teff, and its radius (R1, or R1 times its ratio)
    double ts = tv[0], rs = tv[NS + 1];
#pragma unroll
    for (int k = 1; k < NS; ++k) {
        ts = (star == k) ? tv[k] : ts;
        rs = (star == k) ? tv[NS + 1] * tv[NS + 1 + k] : rs;
    }
    const double plx = tv[2 * NS + 1];
    const double q = fast_div(rs * kRsunCm, fast_div(1.0, plx) * kPcCm);  // mft6.py:690-691,700
    const double sc = q * q;
    // ---- A2, Teff: needs theta alone -----------------------------------------------------------------------------
    LaneBracket T = lane_bracket(R.tn, R.tn_n, lane, nt, ts);
    unsigned int mA = T.i1 == lane ? R.m0 : R.m1, mB = T.i2 == lane ? R.m0 : R.m1;  // presence bits of nodes i1, i2
    int LT = uni(__ffsll((long long)__ballot(T.own)) - 1);
    int st_t = MSX_W_OK;
    if (LT < 0) {  // (uniform) no lane owns Teff: below the first node, or not a number
        LT = 0;
        if (ts < R.t_first) {
            T.i1 = 0; T.i2 = nt - 1; T.e1 = R.t_first; T.e2 = R.t_last; mA = R.m_first; mB = R.m_last;
        } else {
            st_t = bracket_regs(R.tn, nt, ts, &T.i1, &T.i2, &T.e1, &T.e2);
            mA = (unsigned int)__builtin_amdgcn_readlane((int)R.m0, T.i1 & 63);
            mB = (unsigned int)__builtin_amdgcn_readlane((int)R.m0, T.i2 & 63);
        }
    } else {
        st_t = __builtin_amdgcn_readlane(T.st, LT);
    }
    const double bw = (T.i1 == T.i2) ? 0.0 : fast_div(ts - T.e1, T.e2 - T.e1);
    // ---- A1 logg, then its bracket -----------------------------------------------------------------------------
    int st = MSX_W_OK;
    const bool in_iso = (ts >= R.iso_lo) && (ts <= R.iso_hi);
    if (!in_iso) st = MSX_W_VALUEERROR;
    const double lg = iso_interp_lanes(R, ts);  // mft6.py:1149
    MSX_STAMP(P, wk, 12);
    int g1, g2, st_g = MSX_W_OK;
    double a;
    {
        const LaneBracket Gb = lane_bracket(R.gn, R.gn_n, lane, ng, lg);
        const double al = (Gb.i1 == Gb.i2) ? 0.0 : fast_div(lg - Gb.e1, Gb.e2 - Gb.e1);
        const int LG = uni(__ffsll((long long)__ballot(Gb.own)) - 1);
        if (LG >= 0) {
            g1 = __builtin_amdgcn_readlane(Gb.i1, LG);
            g2 = __builtin_amdgcn_readlane(Gb.i2, LG);
            st_g = __builtin_amdgcn_readlane(Gb.st, LG);
            a = readlane_f64(al, LG);
        } else if (lg < R.g_first) {
            g1 = 0; g2 = ng - 1;
            a = (g1 == g2) ? 0.0 : fast_div(lg - R.g_first, R.g_last - R.g_first);
        } else {
            double ge1, ge2;
            st_g = bracket_regs(R.gn, ng, lg, &g1, &g2, &ge1, &ge2);
            a = (g1 == g2) ? 0.0 : fast_div(lg - ge1, ge2 - ge1);
        }
    }
    MSX_STAMP(P, wk, 13);
    if (st == MSX_W_OK) st = st_t;
    if (st == MSX_W_OK) st = st_g;
    // ---- presence, A4: in every lane; lane LT holds the walker's --------------------------------------------------
    const bool have = (((mA >> g1) & (mA >> g2) & (mB >> g1) & (mB >> g2)) & 1u) != 0u;
    const bool have_u = __builtin_amdgcn_readlane((int)have, LT) != 0;
    if (st == MSX_W_OK && !have_u) st = MSX_W_KEYERROR;
    MSX_STAMP(P, wk, 14);
    int node[4] = {T.i1 * ng + g1, T.i1 * ng + g2, T.i2 * ng + g1, T.i2 * ng + g2};
    double w[4] = {(1.0 - bw) * (1.0 - a) * sc, (1.0 - bw) * a * sc, bw * (1.0 - a) * sc, bw * a * sc};
    sort4_by_node(node, w);  // canonical corner order (blend.h): a function of the grid cell alone
    MSX_STAMP(P, wk, 10);
    if (lane == LT) {
        if (!alive) {
            D.stat[star] = MSX_W_REJECT;
        } else if (mode == MSX_MODE_LOGPRIOR) {  // no spectrum pass: the prior terms finish the job
            D.stat[star] = MSX_W_OK;
        } else {
            if (st == MSX_W_OK) {
#pragma unroll
                for (int c = 0; c < 4; ++c) { D.node[4 * star + c] = node[c]; D.w[4 * star + c] = w[c]; }
                if (star == 0) D.redc = redden_rule(mode, G.use_av, tv[NS]) ? -0.4 * kLog2Of10 * tv[NS] : 0.0;
            }
            D.stat[star] = st;
        }
    }
    MSX_STAMP(P, wk, 11);
}

// ------------------------------------------------------------------------------------------------
// The same recipe by ONE THREAD (pair_kernel.h's planner: one thread per walker, where throughput counts and the
// walker's start-up latency does not): the tables in LDS, binary searches where the wave form ballots.  Operation for
// operation the arithmetic of recipe_part1_regs -- the same expressions in the same order, so the same bits: a
// walker's weights do not depend on which form computed them (tests/test_gpu_pair.py compares whole batches).
// ------------------------------------------------------------------------------------------------
struct ScalarTabs {
    // LDS copies of the recipe block, every table padded with +inf to the size its search walks (kPlanIsoPad / kPlanNodePad)
    const double *isot, *teff, *logg;
    const double4 *isopack;  // kRbIsoPack: {x_i, x_i+1, y_i, slope_i}
    const unsigned int *pmask;
    const double *av_edges;  // (the prior's A_V(distance) table rides along: its search runs beside the recipe's)
    int niso, nt, ng, nav;
};
constexpr int kPlanIsoPad = 256, kPlanNodePad = 64;
// One level of a 4-ary search for c = #{i : xs[i] <= v} in a sorted table padded with +inf: three probes, read together.
// After the levels with step = N / 4, N / 16, ..., 1 the base is c for every c < N - 1 (xs[N - 1] is never probed: callers
// whose table may be full, or whose v may reach the last entry, settle that case themselves).
__device__ __forceinline__ void probe3(const double *xs, int base, int step, double (&p)[3]) {
#pragma unroll
    for (int k = 0; k < 3; ++k) p[k] = xs[base + (k + 1) * step - 1];
}
__device__ __forceinline__ int advance3(int base, int step, const double (&p)[3], double v) {
    return base + step * ((p[0] <= v ? 1 : 0) + (p[1] <= v ? 1 : 0) + (p[2] <= v ? 1 : 0));
}
// lane_bracket for the interval [nodes[l], nodes[l + 1]) that holds v, given c = #{nodes <= v} (nodes padded with +inf:
// nodes[n] = +inf); below the first node the wrap of recipe_part1_regs.  (v is finite: non-finite coordinates are rejected
// before, and a logg is only used when its Teff lies inside the isochrone.)
__device__ __forceinline__ int bracket_from_count(const double *nodes, int n, int c, double first, double last, double v, int *i1, int *i2,
                                                  double *e1, double *e2) {
    const int lo = c > 0 ? c - 1 : 0;
    const double nl = nodes[lo], nn = nodes[lo + 1];  // (lo + 1 <= n <= kPlanNodePad - 1 ... or the pad: +inf)
    const bool up = fabs(nn - v) < fabs(nl - v);
    const bool eq = nl == v;
    const bool below = c == 0;  // !(v >= nodes[0])
    *i1 = below ? 0 : (up ? lo + 1 : lo);
    *i2 = below ? n - 1 : ((up || eq) ? lo : lo + 1);
    *e1 = below ? first : (up ? nn : nl);
    *e2 = below ? last : ((up || eq) ? nl : nn);
    return (!below && !up && !eq && lo + 1 >= n) ? MSX_W_INDEXERROR : MSX_W_OK;
}
// One binary's recipe: node[8] (canonical order per star), w[8], redc; returns the walker's status (MSX_W_*), combined
// over the two stars like logprob_kernel does.  A thread alone with its walker waits out every LDS round trip, so the
// searches are arranged for few of them: 4-ary instead of binary, and the five that depend on theta alone -- two stars x
// {Teff bracket, isochrone interval}, and the prior's A_V bin -- walk their levels TOGETHER (fifteen probes in flight per
// level, four levels); the isochrone's interval comes back as one packed entry (the slope was divided at staging, like the
// wave form's), and the two logg brackets walk together again.  Ten dependent round trips where the binary searches one
// after the other took forty-six: planner 14.6 -> see DESIGN 5.1.  Indices, hence bits, are those of the searches they
// replace (a sorted table has one last entry <= v).
// iso_lo[s] (out): the isochrone interval of star s (the radius prior's luminosity lookup is the same search);
// av_bin (out): #{av_edges <= 1 / plx} - 1, unclamped.
__device__ __forceinline__ int recipe_scalar2(const GateArgs &G, const ScalarTabs &T, int mode, const double (&t)[6],
                                              int (&node)[8], double (&w)[8], double *redc, int (&iso_lo)[2], int *av_bin) {
    constexpr int NS = 2;
    bool alive = true;
#pragma unroll
    for (int k = 0; k < 6; ++k) alive = alive && isfinite(t[k]);
    if (alive && (mode == MSX_MODE_LOGPOST || mode == MSX_MODE_LOGPRIOR)) alive = prior_gates<NS>(G, t);
    const double plx = t[2 * NS + 1];
    const double dist = 1.0 / plx;  // pc, mft6.py:1233 (the prior's; the recipe's own is fast_div below, as before)
    const double t_first = T.teff[0], t_last = T.teff[T.nt - 1], g_first = T.logg[0], g_last = T.logg[T.ng - 1];
    const double iso_first = T.isot[0], iso_last = T.isot[T.niso - 1];
    // ---- the searches that need theta alone, level by level ----
    int ct[NS] = {0, 0}, ci[NS] = {0, 0}, ca = 0;
#pragma unroll
    for (int lev = 0; lev < 4; ++lev) {
        const int step = 64 >> (2 * lev);  // 64, 16, 4, 1 (tables of 256); the Teff table (64) joins at the second level
        double pi[NS][3], pt[NS][3], pa[3];
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            probe3(T.isot, ci[s], step, pi[s]);
            if (lev > 0) probe3(T.teff, ct[s], step, pt[s]);
        }
        probe3(T.av_edges, ca, step, pa);
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            ci[s] = advance3(ci[s], step, pi[s], t[s]);
            if (lev > 0) ct[s] = advance3(ct[s], step, pt[s], t[s]);
        }
        ca = advance3(ca, step, pa, dist);
    }
    *av_bin = ca - 1;  // (nav + 1 <= 128 edges in a table of 256: never near the unprobed last entry)
    // ---- what the intervals hold: one round trip ----
    int lo[NS];
    double4 ent[NS];
    bool in_iso[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        in_iso[s] = (t[s] >= iso_first) && (t[s] <= iso_last);
        lo[s] = (t[s] >= iso_last) ? T.niso - 1 : (ci[s] > 0 ? ci[s] - 1 : 0);  // the last i with isot[i] <= ts
        ent[s] = T.isopack[lo[s]];
        iso_lo[s] = lo[s];
        if (t[s] >= t_last) ct[s] = T.nt;  // (a full table's last entry is never probed)
    }
    int i1[NS], i2[NS], st_t[NS];
    double te1[NS], te2[NS], lg[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        st_t[s] = bracket_from_count(T.teff, T.nt, ct[s], t_first, t_last, t[s], &i1[s], &i2[s], &te1[s], &te2[s]);
        lg[s] = in_iso[s] ? fma(ent[s].w, t[s] - ent[s].x, ent[s].z) : T.isopack[0].z;  // mft6.py:1149
    }
    // ---- the two logg brackets, together ----
    int cg[NS] = {0, 0};
#pragma unroll
    for (int lev = 0; lev < 3; ++lev) {
        const int step = 16 >> (2 * lev);  // 16, 4, 1 (table of 64)
        double pg[NS][3];
#pragma unroll
        for (int s = 0; s < NS; ++s) probe3(T.logg, cg[s], step, pg[s]);
#pragma unroll
        for (int s = 0; s < NS; ++s) cg[s] = advance3(cg[s], step, pg[s], lg[s]);
    }
    int stat[NS];
#pragma unroll
    for (int star = 0; star < NS; ++star) {
        const double ts = t[star];
        const double rs = star == 0 ? t[NS + 1] : t[NS + 1] * t[NS + 1 + star];
        const double q = fast_div(rs * kRsunCm, fast_div(1.0, plx) * kPcCm);  // mft6.py:690-691,700
        const double sc = q * q;
        const double bw = (i1[star] == i2[star]) ? 0.0 : fast_div(ts - te1[star], te2[star] - te1[star]);
        int st = in_iso[star] ? MSX_W_OK : MSX_W_VALUEERROR;
        int g1, g2;
        double ge1, ge2;
        if (lg[star] >= g_last) cg[star] = T.ng;
        const int st_g = bracket_from_count(T.logg, T.ng, cg[star], g_first, g_last, lg[star], &g1, &g2, &ge1, &ge2);
        const double a = (g1 == g2) ? 0.0 : fast_div(lg[star] - ge1, ge2 - ge1);
        if (st == MSX_W_OK) st = st_t[star];
        if (st == MSX_W_OK) st = st_g;
        const unsigned int mA = T.pmask[i1[star]], mB = T.pmask[i2[star]];
        const bool have = (((mA >> g1) & (mA >> g2) & (mB >> g1) & (mB >> g2)) & 1u) != 0u;
        if (st == MSX_W_OK && !have) st = MSX_W_KEYERROR;
        int nd[4] = {i1[star] * T.ng + g1, i1[star] * T.ng + g2, i2[star] * T.ng + g1, i2[star] * T.ng + g2};
        double ww[4] = {(1.0 - bw) * (1.0 - a) * sc, (1.0 - bw) * a * sc, bw * (1.0 - a) * sc, bw * a * sc};
        sort4_by_node(nd, ww);
#pragma unroll
        for (int c = 0; c < 4; ++c) { node[4 * star + c] = nd[c]; w[4 * star + c] = ww[c]; }
        stat[star] = !alive ? MSX_W_REJECT : st;
    }
    *redc = redden_rule(mode, G.use_av, t[NS]) ? -0.4 * kLog2Of10 * t[NS] : 0.0;
    int wst = stat[0];
    wst = (wst == MSX_W_OK) ? stat[1] : wst;
    wst = (wst != MSX_W_REJECT && stat[1] == MSX_W_VALUEERROR) ? MSX_W_VALUEERROR : wst;
    return wst;
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
            lp = gauss_term(lp, a_v, pick2(avm, b), sig);
        }
        if (P.has_prior) {
#pragma clang loop unroll(full)
            for (int k = 0; k < 2 * NS + 2; ++k) {
                if (P.pmean[k] != 0.0) lp = gauss_term(lp, t[k], P.pmean[k], P.psig[k]);  // mft6.py:1258
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
                mr[s] = model_radius(iso_interp_regs(isot, isol, P.niso, t[s]), t[s]);
            }
#pragma unroll
            for (int s = 0; s < NS; ++s) lp = radius_term(lp, rad[s], (s == 0) ? mr[0] : mr[s] / mr[0]);
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
    const int njobs = P.nc * NS + P.np;
    double val = 0.0;
    if (lane < njobs) val = band_job_value<NS>(P, D.node, D.w, lane);  // one (filter, star) or one photometric band per lane
    const double chi = band_chi<NS>(P, redden, a_v, [&](int k) __attribute__((always_inline)) { return readlane_f64(val, k); });
    if (lane == 0) D.chi_extra = chi;
}

// ... in two parts: the jobs' magnitudes (the round trip to the band table and the logarithms; needs the recipe's nodes
// and weights) and, from them, the chi^2 terms
template <int NS>
__device__ __forceinline__ double recipe_band_values(const DevProblem &P, const WalkerDesc &D, int lane) {
    const int njobs = P.nc * NS + P.np;
    return lane < njobs ? band_job_value<NS>(P, D.node, D.w, lane) : 0.0;
}
template <int NS>
__device__ __forceinline__ void recipe_band_finish(const DevProblem &P, int mode, const double *__restrict__ th, WalkerDesc &D,
                                                   int lane, double val) {
    const double a_v = th[NS];
    const bool redden = redden_rule(mode, P.use_av, a_v);
    const double chi = band_chi<NS>(P, redden, a_v, [&](int k) __attribute__((always_inline)) { return readlane_f64(val, k); });
    if (lane == 0) D.chi_extra = chi;
}

// ---- the one-thread forms (pair_kernel.h's planner) ---------------------------------------------------------------------
struct ScalarPriorTabs {
    const double *isot, *isol;     // isochrone Teff, luminosity (LDS)
    const double *av_mu, *av_sig;  // the A_V(distance) table (LDS)
};
// the Gaussian prior terms (f1) of a binary in LOGPOST mode: recipe_prior_terms by one thread.  The two table searches
// are recipe_scalar2's (av_bin = #{edges <= 1 / plx} - 1; iso_lo[s] = the isochrone interval of star s).
__device__ __forceinline__ double prior_terms_scalar2(const DevProblem &P, const ScalarPriorTabs &T, int mode, const double (&t)[6],
                                                      int av_bin, const int (&iso_lo)[2]) {
    constexpr int NS = 2;
    const double a_v = t[NS];
    const double *rad = &t[NS + 1];
    double lp = 0.0;
    if (!(mode == MSX_MODE_LOGPOST || mode == MSX_MODE_LOGPRIOR)) return lp;
    if (P.use_av && P.nav > 0) {
        int b = av_bin;  // b = #{edges <= d} - 1, d = 1 / plx pc (mft6.py:1233)
        b = b < 0 ? 0 : (b > P.nav - 1 ? P.nav - 1 : b);
        double sig = T.av_sig[b];
        if (sig == 0.0) sig = 0.05;  // mft6.py:1237-1238
        lp = gauss_term(lp, a_v, T.av_mu[b], sig);
    }
    if (P.has_prior) {
        // (the six quotients first, side by side -- a thread alone with its walker has nothing else to fill a division's
        // latency with -- then the sum in the reference's order; an unused term's quotient is computed and dropped)
        double pm[2 * NS + 2], z[2 * NS + 2];
#pragma unroll
        for (int k = 0; k < 2 * NS + 2; ++k) { pm[k] = P.pmean[k]; z[k] = (t[k] - pm[k]) / P.psig[k]; }
#pragma unroll
        for (int k = 0; k < 2 * NS + 2; ++k) lp = (pm[k] != 0.0) ? gauss_term_z(lp, z[k]) : lp;  // mft6.py:1258
    }
    if (P.rad_prior) {  // mft6.py:1262-1269 (a Teff outside the isochrone has failed the walker already)
        double mr[NS];
        const double iso_first = T.isot[0], iso_last = T.isot[P.niso - 1], lum_last = T.isol[P.niso - 1];
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const bool inside = (t[s] >= iso_first) && (t[s] <= iso_last);
            const int j = iso_lo[s];  // #{isot <= x} - 1
            const int j1 = j + 1 < P.niso ? j + 1 : P.niso - 1;
            const double seg = interp_segment(T.isot[j], T.isol[j], T.isot[j1], T.isol[j1], t[s]);
            const double lum = j >= P.niso - 1 ? lum_last : seg;
            mr[s] = inside ? model_radius(lum, t[s]) : 1.0;
        }
#pragma unroll
        for (int s = 0; s < NS; ++s) lp = radius_term(lp, rad[s], (s == 0) ? mr[0] : mr[s] / mr[0]);
    }
    return lp;
}
// a value the optimiser must take as given: a magnitude is ROUNDED before anything is added to it, as in the wave form,
// where it crosses lanes (without this the product inside it may be contracted into the caller's sum)
__device__ __forceinline__ double rounded_here(double x) {
    asm volatile("" : "+v"(x));
    return x;
}
// icontrast + iphot of a binary: recipe_band_terms by one thread -- band_job_value's sums and band_chi's terms in their
// order, but the band table's rows are requested up to FOUR BANDS x EIGHT NODES at a time, two bands to a 16-byte load
// when the rows are even (one round trip of eight loads for the two contrast filters of BASELINE's configs, where job
// after job took a round trip each)
// (pre: the rows' first four bands, requested by band_rows_first long before -- the planner walks its pairing in between)
struct BandRows { double2 a[8], b[8]; };
__device__ __forceinline__ void band_rows_first(const DevProblem &P, const int (&node)[8], BandRows &pre) {
    const int nb = P.nc + P.np;
    const bool even = (nb & 1) == 0, more = 2 < nb;  // (uniform)
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const double *row = P.band_tab + (int64_t)node[c] * nb;
        if (nb <= 0) { pre.a[c] = pre.b[c] = make_double2(1.0, 1.0); continue; }
        pre.a[c] = even ? *reinterpret_cast<const double2 *>(row) : make_double2(row[0], row[1 < nb ? 1 : nb - 1]);
        pre.b[c] = !more ? make_double2(1.0, 1.0)
                         : (even ? *reinterpret_cast<const double2 *>(row + 2) : make_double2(row[2], row[3 < nb ? 3 : nb - 1]));
    }
}
__device__ __forceinline__ double band_terms_scalar2(const DevProblem &P, int mode, const double (&t)[6], const int (&node)[8],
                                                     const double (&w)[8], const BandRows &pre) {
    constexpr int NS = 2;
    const double a_v = t[NS];
    const bool redden = redden_rule(mode, P.use_av, a_v);
    const int nc = P.nc, nb = P.nc + P.np;
    double chi = 0.0;
    const double *row[NS * 4];
#pragma unroll
    for (int c = 0; c < NS * 4; ++c) row[c] = P.band_tab + (int64_t)node[c] * nb;
    auto one_band = [&](int b, const double (&v)[NS * 4]) __attribute__((always_inline)) {
        if (b < nc) {  // contrast filter b: the stars' instrumental magnitudes (A5)
            double mag[NS];
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                double m = 0.0;
#pragma unroll
                for (int c = 0; c < 4; ++c) m += w[4 * s + c] * v[4 * s + c];
                mag[s] = rounded_here(-2.5 * log10(m));  // mft6.py:733
            }
            const double con = mag[1] - mag[0];  // mft6.py:741
            const double z = con - P.cmag[b];
            chi = fma(z * z, P.civar[b], chi);  // mft6.py:120,1182
        } else {  // photometric band b - nc (A6)
            const int f = b - nc;
            double flux = 0.0;
#pragma unroll
            for (int c = 0; c < NS * 4; ++c) flux += w[c] * v[c];
            const double mag = rounded_here(-2.5 * log10(flux / P.pzero[f]));  // mft6.py:780-782
            const double mred = redden ? fma(a_v, P.pk[f], mag) : mag;  // mft6.py:1163
            const double z = mred - P.pmag[f];
            chi = fma(z * z, P.pivar[f], chi);  // mft6.py:1188
        }
    };
    auto bands = [&](auto even_c) __attribute__((always_inline)) {
        constexpr bool EVEN = decltype(even_c)::value;
        for (int b0 = 0; b0 < nb; b0 += 4) {  // (uniform)
            const bool more = b0 + 2 < nb;  // (uniform)
            double2 va[NS * 4], vb[NS * 4];
            if (b0 == 0) {  // (uniform) the first group is in registers already
#pragma unroll
                for (int c = 0; c < NS * 4; ++c) { va[c] = pre.a[c]; vb[c] = pre.b[c]; }
            } else {
#pragma unroll
                for (int c = 0; c < NS * 4; ++c) {
                    if constexpr (EVEN) {
                        va[c] = *reinterpret_cast<const double2 *>(row[c] + b0);
                    } else {
                        va[c] = make_double2(row[c][b0], row[c][b0 + 1 < nb ? b0 + 1 : nb - 1]);
                    }
                }
                if (more) {
#pragma unroll
                    for (int c = 0; c < NS * 4; ++c) {
                        if constexpr (EVEN) {
                            vb[c] = *reinterpret_cast<const double2 *>(row[c] + b0 + 2);
                        } else {
                            vb[c] = make_double2(row[c][b0 + 2], row[c][b0 + 3 < nb ? b0 + 3 : nb - 1]);
                        }
                    }
                } else {
#pragma unroll
                    for (int c = 0; c < NS * 4; ++c) vb[c] = make_double2(1.0, 1.0);
                }
            }
            double v[NS * 4];
#pragma unroll
            for (int c = 0; c < NS * 4; ++c) v[c] = va[c].x;
            one_band(b0, v);
            if (b0 + 1 < nb) {
#pragma unroll
                for (int c = 0; c < NS * 4; ++c) v[c] = va[c].y;
                one_band(b0 + 1, v);
            }
            if (more) {
#pragma unroll
                for (int c = 0; c < NS * 4; ++c) v[c] = vb[c].x;
                one_band(b0 + 2, v);
                if (b0 + 3 < nb) {
#pragma unroll
                    for (int c = 0; c < NS * 4; ++c) v[c] = vb[c].y;
                    one_band(b0 + 3, v);
                }
            }
        }
    };
    // (rows of an even number of bands are 16-byte aligned: the table is, and a row is nb doubles)
    if ((nb & 1) == 0) bands(std::true_type{}); else bands(std::false_type{});
    return chi;
}

}  // namespace

#endif  // MSX_RECIPE_H
