// pair_kernel.h -- part of the single translation unit msx.hip (included there, after logprob_kernel.h).
// The PAIR form of the hot kernel: one workgroup evaluates TWO walkers of the same grid cell from ONE set of loads.
//
// Why.  With more than one walker per CU the fused kernel is bound by what a CU can pull through its L2 port (the
// launch requests ~600 KB per walker) and every workgroup pulls the same few grid rows again: an MCMC
// ensemble sits in a handful of grid cells.  Walkers of one cell read the SAME eight rows (the canonical corner
// order, blend.h, makes the row list a function of the cell alone) and the same per-pixel statics; only their
// weights differ.  Here a lane loads a table element once and runs two weight chains on it: row, static and chi^2-pass
// requests per walker are halved.  Rows are not shared implicitly -- same-cell walkers run in lock-step in separate
// workgroups get no L1 hits (measured) -- so the sharing has to be through registers.
//
// How it keeps a CU busy.  Two model vectors of 32 KB do not leave room in LDS for the two or three workgroups a CU
// needs to overlap one workgroup's memory phase with another's latency chain.  But a lane only ever re-reads, in the
// median's pass, the pixels it computed itself in phase A (phase A's element walk and pass_pixel() assign the same
// pixels to a lane, for 256 and for 512 threads) -- so the model values stay in REGISTERS (npix / MAXT doubles per
// lane and walker, spectra of up to 4096 pixels) and LDS holds only the walkers' small state.  The paths that need the
// vector in memory (block_median: vectors the early histogram cannot handle) spill it to the walker's row of the
// global scratch first; they are rare.
//
// Same bits.  Every per-walker operation is the fused kernel's own: the recipe (recipe.h), blend_accumulate /
// blend_finish (blend.h), fit_accumulate, the canonical sum (wave_ops.h), logbin_locate / logbin_rank / block_median
// (median.h), chi_term / fit_coefs / fused_total (logprob_kernel.h).  A walker's value does not depend on its partner,
// nor on whether it was paired at all (tests/test_gpu_pair.py).
//
// Who is paired: pair_plan_kernel (below) groups the batch by grid cell and emits pairs {walker, partner} and singles;
// walkers whose recipes turn out to differ after all (or whose partner fails) are evaluated one after the other by
// the same workgroup.
#ifndef MSX_PAIR_KERNEL_H
#define MSX_PAIR_KERNEL_H

namespace {

constexpr int kPairMaxPix = 4096;   // model values in registers: kPairMaxPix / MAXT doubles per lane and walker
constexpr int kPairSpillRows = 1024; // scratch rows of the spill path (block_median on a vector in memory), leased
// The plan (pair_plan_kernel): int32 plan[kPairHdrInts]
//   [0] pairs, [1] singles: the pair kernel's items (final counts); [2], [3] the same while the planner runs, [4] its
//   finished workgroups (all three back at zero when it ends).  The items themselves -- the walkers' recipes, so that a
//   workgroup of the pair kernel is two dependent loads from its weights -- are P.pair_items[] and P.pair_singles[].
constexpr int kPairHdrInts = 8;

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F &&f) {  // f(integral_constant<int, I>) ... f(integral_constant<int, N - 1>)
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// NT = element trips per lane the variant is compiled for: the launcher takes the smallest one that covers the
// spectrum (ceil(elements / MAXT) <= NT); lanes beyond the tables are predicated off.  A compile-time trip count
// keeps the unrolled sweeps free of branches -- and of the register copies their merge points cost.
// RED = the problem fits extinction (use_av): the sweep loads the H rows and the extinction curve; a walker at
// A_V = 0 exactly still takes the unreddened value (blend_finish), it only pays for the loads.
// FULL = the spectrum fills the variant exactly (npix == 2 NT MAXT: BASELINE's 4096 pixels with NT = 4): every lane of every
// trip holds two live pixels, so the clamps of the element index, the per-pixel validity compares and their selects are
// compiled out -- this kernel runs at the vector ALUs' issue rate with many workgroups in flight, and those were ~5 % of
// what it issued.  Same arithmetic on the same pixels: same bits.
template <int MAXT, int NT, bool RED, bool FULL = false>
__global__ void __launch_bounds__(MAXT, MAXT == 256 ? 2 : 4)
logprob_pair_kernel(const double *theta, const unsigned char *__restrict__ rblk, int niso_nt, int ng_mode_fast, int64_t n,
                    double gate_tmin, double gate_tmax, const int32_t *__restrict__ plan, DevProblem P,
                    double *__restrict__ logp, int32_t *__restrict__ status) {
    // (the leading 14 dwords arrive preloaded in SGPRs, as in logprob_kernel)
    constexpr int NS = 2, ndim = 6, NC = 8, B = MAXT, nw = B >> 6;
    (void)rblk; (void)niso_nt; (void)n; (void)gate_tmin; (void)gate_tmax;
    static_assert(NT * MAXT * 2 <= kPairMaxPix, "model values per lane");
    constexpr int vk = kMaxWaves / nw;          // canonical-sum slots per lane: 4 or 2
    constexpr int G = MAXT == 256 ? NC : 4;     // corners per group of loads (512 threads: <= 128 VGPRs, a star at a time)
    static_assert(MAXT == 256 || MAXT == 512, "256 or 512 threads");
    __shared__ WalkerDesc D[2];
    __shared__ BlockScratch S[2];
    __shared__ double red[2][3][nw][kWave];
    __shared__ double e2tab[kExp2Tab];
    struct WalkerLoc { double pc[3]; LogbinSel Q; int appl; unsigned int fsel, fspan; };
    __shared__ WalkerLoc Loc[2];
    const int mode = (ng_mode_fast >> 8) & 0xff;
    const int tid0 = threadIdx.x, lane0 = tid0 & 63, wave0 = tid0 >> 6;
    // ---- this workgroup's item: the planner's pairs first (the longer workgroups), then its singles ----------------
    const PairRec *rec[2];
    bool is_pair;
    {
        const int2 cnt = *reinterpret_cast<const int2 *>(plan);  // {pairs, singles}
        const int b = (int)blockIdx.x, sb = b - cnt.x;
        if (sb >= cnt.y) return;
        is_pair = sb < 0;
        rec[0] = is_pair ? &P.pair_items[b].r[0] : P.pair_singles + sb;
        rec[1] = is_pair ? &P.pair_items[b].r[1] : rec[0];
    }
    const int npix = (int)P.npix;
    const int ne = (int)P.npair;  // table elements, a multiple of 256; <= kPairMaxPix / 2 (checked by the host)

    // ---- phase 0: the walkers' recipes are the planner's (recipe_scalar2: every walker on the list is live) --------
    if (wave0 < 2 && lane0 < NC) {
        const PairRec *R = rec[wave0];
        D[wave0].node[lane0] = R->node[lane0];
        D[wave0].w[lane0] = R->w[lane0];
        if (lane0 == 0) { D[wave0].redc = R->redc; D[wave0].lp = R->lp; D[wave0].chi_extra = R->chi_extra; D[wave0].ncorner = R->walker; }
    }
    fill_exp2_table(e2tab, tid0 - (B - kWave));
    for (int i = tid0; i < kLogBins; i += B) { S[0].hist[i] = 0; S[1].hist[i] = 0; }
    if (tid0 < 2) { S[tid0].cand_n = 0; S[tid0].has_second = 0; }
    __syncthreads();
    const int64_t wkv[2] = {D[0].ncorner, is_pair ? D[1].ncorner : -1};  // (the walkers' indices travel in an unused field)
    const bool act0 = true, act1 = is_pair;
    bool same = act1;
    if (same) {
#pragma unroll
        for (int c = 0; c < NC; ++c) same = same && D[0].node[c] == D[1].node[c];
    }
    // {0, 1} together when both are live and read the same rows; else the live ones one after the other (below)
    const bool need_two = (npix & 1) == 0;

    // The evaluation proper, compiled twice: for two walkers (machinery slot 0 = walker sa, slot 1 = walker sb) and
    // for one (the second slot's arithmetic, state and barriers' work compiled out).
    auto body = [&](auto two_c, const int sa, const int sb, const bool again) __attribute__((always_inline)) {
        constexpr bool two = decltype(two_c)::value;
        constexpr int NSLOT = two ? 2 : 1;
        const int tid = tid0, lane = lane0, wave = wave0;
        const int64_t wk0 = wkv[sa], wk1 = wkv[sb];
        const WalkerDesc &Da = D[sa], &Db = D[sb];
        if (again) {  // (a second evaluation by this workgroup: the LDS state of the first is stale)
            for (int i = tid; i < kLogBins; i += B) S[0].hist[i] = 0;
            if (tid == 0) { S[0].cand_n = 0; S[0].has_second = 0; }
            __syncthreads();
        }

        // ---- phase A: blend + redden + resample, two weight chains per loaded element ---------------------------
        const double2 *rows_r[NC];
        const float2 *rows_h[NC];
        double wA[NC], wB[NC];
        float wfA[NC], wfB[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int64_t off = (int64_t)__builtin_amdgcn_readfirstlane(Da.node[c]) * ne;
            rows_r[c] = P.r2 + off;
            rows_h[c] = P.h2 + off;
            wA[c] = uniform_f64(Da.w[c]);
            wfA[c] = uniform_f32((float)wA[c]);
            wB[c] = uniform_f64(Db.w[c]);
            wfB[c] = uniform_f32((float)wB[c]);
        }
        const double redcA = uniform_f64(Da.redc), redcB = uniform_f64(Db.redc);
        const bool reddenA = redcA != 0.0, reddenB = two && redcB != 0.0;
        double2 m[2][NT];  // THE MODEL VECTORS: lane tid holds the pixels of elements tid + j B, j = 0 .. NT - 1
        double qa[2][vk][3];
        // The value range as the range of the UNMASKED bin number F(x) = hi32(x) >> 12 -- a by-product of the histogram's
        // bin, two integer instructions per pixel, one DPP instruction per reduction step where float64 min / max /
        // NaN flags take five times that.  It decides everything the early-histogram median needs (all values
        // positive normal numbers: F in [1, 0x7ff00); span < 8 binades: F_max - F_min < 2048; the bin of the minimum);
        // the exact float64 range is worked out only on the path that needs it (below).
        unsigned int fmin[2] = {~0u, ~0u}, fmax[2] = {0u, 0u};
        // what follows a pixel pair's model values (logprob_kernel's finish_elem, per walker)
        auto finish_elem = [&](auto s_c, auto j_c, const double2 m2, const double2 f2, const double2 u2, const int ec,
                               const bool live) __attribute__((always_inline)) {
            constexpr int s = decltype(s_c)::value, j = decltype(j_c)::value;
            constexpr int kbase = MAXT == 256 ? 2 * (j & 1) : 0;
            const int pa = ((ec >> 8) << 9) | (ec & 255), pb = pa + 256;
            const bool ok[2] = {FULL || (live && pa < npix), FULL || (live && pb < npix)};
            const double mm[2] = {m2.x, m2.y}, ff[2] = {f2.x, f2.y}, uu[2] = {u2.x, u2.y};
            unsigned int fxs[2] = {0u, 0u};
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (ok[u]) {
                    fit_accumulate(mm[u], ff[u], uu[u], qa[s][kbase + u][0], qa[s][kbase + u][1], qa[s][kbase + u][2]);
                    const unsigned int fx = (unsigned int)__double2hiint(mm[u]) >> 12;
                    fmin[s] = fx < fmin[s] ? fx : fmin[s];
                    fmax[s] = fx > fmax[s] ? fx : fmax[s];
                    fxs[u] = fx;
                }
            }
            hist_add_pair(S[s].hist, fxs[0], ok[0], fxs[1], ok[1]);  // (odd lanes in the other order: median.h)
        };
        {
            auto trip = [&](auto j_c) __attribute__((always_inline)) {
                constexpr int j = decltype(j_c)::value;
                // (an opaque copy of the thread index per trip: the trip's offsets are formed HERE, not hoisted to the
                // top of the unrolled sweep and parked in registers -- or spilled -- until the trip comes)
                int tj = tid;
                asm volatile("" : "+v"(tj));
                const int e = j * B + tj;
                const int ec = (FULL || e < ne) ? e : ne - 1;
                const unsigned int o16 = (unsigned int)ec << 4, o8 = (unsigned int)ec << 3;
                double2 kl2 = make_double2(0.0, 0.0);
                float2 dk2 = make_float2(0.f, 0.f);
                double sra[2] = {0.0, 0.0}, srb[2] = {0.0, 0.0};
                float sha[2] = {0.f, 0.f}, shb[2] = {0.f, 0.f};
#pragma unroll
                for (int c0 = 0; c0 < NC; c0 += G) {
                    double2 rr[G];
                    float2 hh[G];
#pragma unroll
                    for (int c = 0; c < G; ++c) {
                        rr[c] = ld_off(rows_r[c0 + c], o16);
                        hh[c] = RED ? ld_off(rows_h[c0 + c], o8) : make_float2(0.f, 0.f);
                    }
                    if (c0 == 0 && RED) { kl2 = ld_off(P.kl2, o16); dk2 = ld_off(P.dk2, o8); }
                    double ra[G], rb[G];
                    float ha[G], hb[G];
#pragma unroll
                    for (int c = 0; c < G; ++c) { ra[c] = rr[c].x; rb[c] = rr[c].y; ha[c] = hh[c].x; hb[c] = hh[c].y; }
                    blend_accumulate<G>(ra, ha, wA + c0, wfA + c0, RED, sra[0], sha[0]);
                    blend_accumulate<G>(rb, hb, wA + c0, wfA + c0, RED, srb[0], shb[0]);
                    if (two) {
                        blend_accumulate<G>(ra, ha, wB + c0, wfB + c0, RED, sra[1], sha[1]);
                        blend_accumulate<G>(rb, hb, wB + c0, wfB + c0, RED, srb[1], shb[1]);
                    }
                }
                double2 m0, m1 = make_double2(0.0, 0.0);
                constexpr bool kFin = RED;
                m0.x = blend_finish(sra[0], sha[0], kl2.x, (double)dk2.x, redcA, kFin && reddenA, e2tab);
                m0.y = blend_finish(srb[0], shb[0], kl2.y, (double)dk2.y, redcA, kFin && reddenA, e2tab);
                if (two) {
                    m1.x = blend_finish(sra[1], sha[1], kl2.x, (double)dk2.x, redcB, kFin && reddenB, e2tab);
                    m1.y = blend_finish(srb[1], shb[1], kl2.y, (double)dk2.y, redcB, kFin && reddenB, e2tab);
                }
                m[0][j] = m0;
                m[1][j] = m1;
                // (the trips are unrolled for the static register indices of m[][]; without a fence the scheduler
                // hoists the loads of ALL trips to the top and spills what it cannot hold)
                __builtin_amdgcn_sched_barrier(0);
            };
            static_for<0, NT>(trip);
        }
        // ... then ONE sweep over the registers for what follows the model values (the fit sums, the value range, the
        // median's histogram): kept out of the blend loop, whose row loads want the registers -- 24 accumulators per
        // walker pair and the data flux / u of the trip would push it into spilling
#pragma unroll
        for (int s = 0; s < NSLOT; ++s)
#pragma unroll
            for (int k = 0; k < vk; ++k) qa[s][k][0] = qa[s][k][1] = qa[s][k][2] = 0.0;
        auto fit_trip = [&](auto j_c) __attribute__((always_inline)) {
            constexpr int j = decltype(j_c)::value;
            int tj = tid;
            asm volatile("" : "+v"(tj));
            const int e = j * B + tj;
            const bool live = FULL || e < ne;
            const int ec = live ? e : ne - 1;
            const unsigned int o16 = (unsigned int)ec << 4;
            const double2 f2v = ld_off(P.f2, o16), u2v = ld_off(P.u2, o16);
            finish_elem(std::integral_constant<int, 0>{}, j_c, m[0][j], f2v, u2v, ec, live);
            if (two) finish_elem(std::integral_constant<int, 1>{}, j_c, m[1][j], f2v, u2v, ec, live);
            __builtin_amdgcn_sched_barrier(0);
        };
        static_for<0, NT>(fit_trip);

        // ---- the fit sums (canonical sum), running totals of the histograms, value ranges --------------------------
        static_for<0, NSLOT>([&](auto s_c) __attribute__((always_inline)) {
            constexpr int s = decltype(s_c)::value;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                double a[vk];
#pragma unroll
                for (int k = 0; k < vk; ++k) a[k] = qa[s][k][i];
                red[s][i][wave][lane] = lane_partial<vk>(a);
            }
        });
        __syncthreads();
        for (int idx = wave; idx < (two ? 6 : 3); idx += nw) {  // one wave per (walker, quantity)
            const int s = idx >= 3 ? 1 : 0, i = idx - 3 * s;
            const double v = reduce_published<MAXT>(&red[s][i][0][0], lane);
            if (lane == 0) S[s].q[0][i] = v;
        }
        hist_prefix_inplace<MAXT>(S[0]);
        if (two) hist_prefix_inplace<MAXT>(S[1]);
        static_for<0, NSLOT>([&](auto s_c) __attribute__((always_inline)) {
            constexpr int s = decltype(s_c)::value;
            const unsigned int lo = wave_min_u32(fmin[s]), hi = wave_max_u32(fmax[s]);
            if (lane == 0) { S[s].kmin[wave] = lo; S[s].kmax[wave] = hi; }  // (a wave with no live pixel: ~0 / 0, the neutral pair)
        });
        __syncthreads();
        // wave s works out walker s's fit coefficients, value range and the median's bin(s) and leaves them in LDS for
        // the others (the fused kernel lets every wave locate the rank for itself to save a barrier: with many workgroups
        // in flight the ~120 instructions per wave cost more than the barrier)
        static_for<0, NSLOT>([&](auto s_c) __attribute__((always_inline)) {
            constexpr int s = decltype(s_c)::value;
            if (wave != s) return;
            double q[3], c0, c1, c2;
#pragma unroll
            for (int i = 0; i < 3; ++i) q[i] = S[s].q[0][i];
            unsigned int f0 = (unsigned int)S[s].kmin[0], f1 = (unsigned int)S[s].kmax[0];
            for (int x = 1; x < nw; ++x) {
                f0 = (unsigned int)S[s].kmin[x] < f0 ? (unsigned int)S[s].kmin[x] : f0;
                f1 = (unsigned int)S[s].kmax[x] > f1 ? (unsigned int)S[s].kmax[x] : f1;
            }
            fit_coefs(P, q, c0, c1, c2);
            LogbinSel Qs = {0u, 0u, 0u, 0u};
            // positive normal numbers only (zeros and subnormals: F = 0; infinities, NaNs, negatives: F >= 0x7ff00),
            // spanning less than the histogram's cycle; then the rank from the running totals
            bool ap = frange_applicable(f0, f1);
            if (ap) ap = logbin_locate_h<MAXT>(npix, f0, S[s], &Qs);
            if (lane == 0) {
                Loc[s].pc[0] = c0; Loc[s].pc[1] = c1; Loc[s].pc[2] = c2;
                Loc[s].Q = Qs;
                Loc[s].appl = ap;
                // The candidates' test as ONE range check.  The vector is positive and spans < 8 binades, so the
                // UNMASKED bin number F(x) = hi32(x) >> 12 is monotone in x and unique per physical bin:
                // F(bin p) = F(min) + ((p - F(min)) mod 2048).  nxt_p is the next NON-EMPTY bin after sel_p: a value
                // lies in one of the two iff F(sel) <= F(x) <= F(nxt) -- there is nothing in between.
                const unsigned int fsel = f0 + ((Qs.sel_p - f0) & (unsigned int)(kLogBins - 1));
                const unsigned int fnxt = f0 + ((Qs.nxt_p - f0) & (unsigned int)(kLogBins - 1));
                Loc[s].fsel = ap ? fsel : 1u;
                Loc[s].fspan = ap ? fnxt - fsel : 0u;   // (not applicable: F(x) - 1 <= 0 never holds for a positive x ... and
                                                         //  `appl` gates the test anyway)
            }
        });
        __syncthreads();
        double pc[2][3];
        bool bad[2] = {false, false}, appl[2];
        LogbinSel Q[2];
        unsigned int fsel[2], fspan[2];
        static_for<0, NSLOT>([&](auto s_c) __attribute__((always_inline)) {
            constexpr int s = decltype(s_c)::value;
            pc[s][0] = Loc[s].pc[0]; pc[s][1] = Loc[s].pc[1]; pc[s][2] = Loc[s].pc[2];
            Q[s] = Loc[s].Q;
            appl[s] = Loc[s].appl != 0;
            fsel[s] = Loc[s].fsel; fspan[s] = Loc[s].fspan;
        });

        // ---- ONE pass over the registers: chi^2 terms of both walkers from one load of u, data flux and 1/err^2,
        //      and the candidates of each median's bin(s) --------------------------------------------------------
        double acc[2][vk];
#pragma unroll
        for (int s = 0; s < NSLOT; ++s)
#pragma unroll
            for (int k = 0; k < vk; ++k) acc[s][k] = 0.0;
        auto pass_trip = [&](auto j_c) __attribute__((always_inline)) {
            constexpr int j = decltype(j_c)::value;
            constexpr int kbase = MAXT == 256 ? 2 * (j & 1) : 0;
            int tj = tid;
            asm volatile("" : "+v"(tj));
            const int e = j * B + tj;
            const int ec = (FULL || e < ne) ? e : ne - 1;
            const unsigned int o16 = (unsigned int)ec << 4;
            const double2 nv = ld_off(P.iv2, o16), cu = ld_off(P.u2, o16), cf = ld_off(P.f2, o16);
            const int pa = ((e >> 8) << 9) | (e & 255);  // (beyond the tables: >= npix)
            const bool livep[2] = {FULL || (e < ne && pa < npix), FULL || (e < ne && pa + 256 < npix)};
            const double uu[2] = {cu.x, cu.y}, ff[2] = {cf.x, cf.y}, ee[2] = {nv.x, nv.y};
            static_for<0, NSLOT>([&](auto s_c) __attribute__((always_inline)) {
                constexpr int s = decltype(s_c)::value;
                const double xv[2] = {m[s][j].x, m[s][j].y};
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    acc[s][kbase + u] += chi_term(pc[s][0], pc[s][1], pc[s][2], uu[u], ff[u], ee[u], xv[u], livep[u]);
                    const unsigned int fx = (unsigned int)__double2hiint(xv[u]) >> 12;
                    if (appl[s] && livep[u] && fx - fsel[s] <= fspan[s])
                        S[s].cand[atomicAdd(&S[s].cand_n, 1u)] = key_of(xv[u]);
                }
            });
            __builtin_amdgcn_sched_barrier(0);
        };
        static_for<0, NT>(pass_trip);
        red[0][0][wave][lane] = lane_partial<vk>(acc[0]);
        if (two) red[1][0][wave][lane] = lane_partial<vk>(acc[1]);
        __syncthreads();

        // ---- rank: wave s ranks walker s's candidates ------------------------------------------------------------------
        double med[2] = {0.0, 0.0};
        bool lost[2] = {false, false};  // the spill path's lease was not granted in time
        if (appl[0]) med[0] = logbin_rank<MAXT>(S[0], Q[0], need_two, 0);
        if (two && appl[1]) med[1] = logbin_rank<MAXT>(S[1], Q[1], need_two, 1);
        // vectors the early histogram cannot handle (not positive, >= 8 binades, > 256 equal-bin candidates): the
        // model values go to the walker's scratch row and block_median (linear bins, radix fallback) reads them there
        static_for<0, NSLOT>([&](auto s_c) __attribute__((always_inline)) {
            constexpr int s = decltype(s_c)::value;
            if (appl[s]) return;  // (uniform)
            // the exact value range first (order-preserving keys; a NaN anywhere counts as above +inf), like the fused kernel
            double vlo = INFINITY, vhi = -INFINITY;
            bool nan_here = false;
            static_for<0, NT>([&](auto j_c) __attribute__((always_inline)) {
                constexpr int j = decltype(j_c)::value;
                const int e = j * B + tid;
                const int pa = ((e >> 8) << 9) | (e & 255);
                const double xv[2] = {m[s][j].x, m[s][j].y};
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    if (e < ne && pa + 256 * u < npix) {
                        vlo = min_nc(vlo, xv[u]);
                        vhi = max_nc(vhi, xv[u]);
                        nan_here = nan_here || (xv[u] != xv[u]);
                    }
                }
            });
            {
                const double lo = wave_min_f64(vlo), hi = wave_max_f64(vhi);
                const bool wave_nan = __ballot(nan_here) != 0ull;
                if (lane == 0) {
                    S[s].kmin[wave] = lo == INFINITY && hi == -INFINITY ? ~0ull : key_of(lo == 0.0 ? -0.0 : lo);
                    S[s].kmax[wave] = wave_nan ? ~0ull : (lo == INFINITY && hi == -INFINITY ? 0ull : key_of(hi == 0.0 ? 0.0 : hi));
                }
            }
            __syncthreads();
            unsigned long long kmin = S[s].kmin[0], kmax = S[s].kmax[0];
            for (int x = 1; x < nw; ++x) {
                kmin = S[s].kmin[x] < kmin ? S[s].kmin[x] : kmin;
                kmax = S[s].kmax[x] > kmax ? S[s].kmax[x] : kmax;
            }
            // np.median of a vector holding a NaN is NaN -> total NaN -> -inf (mft6.py:1202-1203)
            bad[s] = kmax > key_of(INFINITY) || kmin < key_of(-INFINITY);
            if (bad[s]) return;  // (uniform)
            // A scratch row on LEASE: kPairSpillRows rows serve every launch (a row per walker of a 16,384-walker batch
            // would be 512 MB for a path that real spectra never take).  The workgroup takes row (block mod rows) and
            // waits while an earlier workgroup still holds it -- that one's progress does not depend on this one.
            // The wait is BOUNDED like every other in-kernel wait (kHandoverTicks of the wall clock): a lease word left at 1
            // by a launch torn down mid-spill must end as a status, not as a hung GPU.  On expiry the walker fails with
            // MSX_W_HANDOVER; the synchronous entry points clear the leases when they see it (msx.hip, note_handover).
            const int lease = (int)(blockIdx.x % (unsigned int)kPairSpillRows);
            if (tid == 0) {
                const unsigned long long t0 = wall_clock64();
                bool got = false;
                for (;;) {
                    if (atomicCAS(P.pair_lease + lease, 0, 1) == 0) { got = true; break; }
                    if (wall_clock64() - t0 > kHandoverTicks) break;
                    __builtin_amdgcn_s_sleep(8);
                }
                S[s].meet_state = got ? 1u : 0u;
            }
            __syncthreads();
            if (S[s].meet_state == 0u) { lost[s] = true; return; }  // (uniform)
            double *row = P.model_scratch + (int64_t)lease * npix;
            auto spill = [&](auto j_c) __attribute__((always_inline)) {
                constexpr int j = decltype(j_c)::value;
                const int e = j * B + tid;
                const int pa = ((e >> 8) << 9) | (e & 255);
                if (e < ne && pa < npix) row[pa] = m[s][j].x;
                if (e < ne && pa + 256 < npix) row[pa + 256] = m[s][j].y;
            };
            static_for<0, NT>(spill);
            for (int i = tid; i < kBins; i += B) S[s].hist[i] = 0;  // (block_median's entry condition)
            __threadfence_block();
            __syncthreads();
            NoElem no_elem;
            bool unused = false;
            med[s] = block_median<MAXT>(row, npix, kmin, kmax, S[s], NoSide(), no_elem, &unused);
            __syncthreads();  // (every thread has its last value of the row)
            if (tid == 0) atomicExch(P.pair_lease + lease, 0);
        });
        // (no barrier: wave s has its walker's median in registers and the chi^2 partials were published before the pass's
        // barrier; the other waves are done)

        // ---- combine (wave s finishes walker s) ---------------------------------------------------------------------
        static_for<0, NSLOT>([&](auto s_c) __attribute__((always_inline)) {
            constexpr int s = decltype(s_c)::value;
            if (wave != s) return;
            const double tot = reduce_published<MAXT>(&red[s][0][0][0], lane);
            if (lane == 0) {
                const WalkerDesc &Dw = s == 0 ? D[sa] : D[sb];
                const int64_t wk = s == 0 ? wk0 : wk1;
                double out;
                if (bad[s]) {
                    out = mode == MSX_MODE_CHISQ ? NAN : -INFINITY;
                } else {
                    const double total = fused_total(P, tot, P.median_flux, med[s], npix, Dw.chi_extra);
                    out = value_of_total(mode, total, Dw.lp);
                }
                logp[wk] = lost[s] ? nan_with_status(MSX_W_HANDOVER) : out;
                status[wk] = lost[s] ? MSX_W_HANDOVER : MSX_W_OK;
            }
        });
    };
    if (same) {
        body(std::true_type{}, 0, 1, false);
    } else {
        // the live ones one after the other -- two more inlined copies of the one-walker body, not a loop: as the body of
        // a loop the compiler keeps the sweeps' unrolled state alive around the back edge and spills all of it
        if (act0) body(std::false_type{}, 0, 0, false);
        __syncthreads();  // a second evaluation re-uses the LDS state
        if (act1) body(std::false_type{}, 1, 1, true);
    }
}

// ------------------------------------------------------------------------------------------------
// The planner: who shares a workgroup.  One thread per walker computes the walker's GRID CELL -- the eight sorted node
// indices its recipe will read (isochrone logg, the two brackets per star: scalar restatements of recipe.h) -- hashed
// to a 40-bit tag.  Partners are found without sorting the batch and without contended atomics, level by level:
//   1. inside each WAVE, by ballots: the lanes of one tag pair up in lane order (no memory traffic at all);
//   2. what a wave cannot place -- at most one walker per cell -- meets the other waves' leftovers inside the WORKGROUP:
//      a short list of cards in LDS, 64 to a wave, the same ballots again, for a few rounds;
//   3. what a workgroup cannot place -- at most one walker per cell -- is a single (64 workgroups of 256 walkers x a
//      handful of cells in a batch of 16,384: 232 singles, 1.4 %; a global third level cost more than it placed, and
//      workgroups of 512 / 1024 walkers -- 124 / 72 singles -- spend longer planning than their pairs save: the planner
//      is a chain of dependent steps per thread, 341.1 / 346.0 us per batch against 340.2).
// Per workgroup: one global atomic add for its pairs, one for its singles.  Walkers the prior box rejects or whose
// recipe will fail are singles from the start.  Pairs and singles are ONE list of items for
// logprob_pair_kernel (pairs first: the longer workgroups start first, the lone walkers fill the tail).  Who meets whom
// depends on the batch order only; values depend on neither: a walker's bits are its own (see the header of this
// file).  And a cell mis-computed here costs time, not correctness: the pair kernel compares the real recipes and
// evaluates walkers that differ one after the other.
// ------------------------------------------------------------------------------------------------
constexpr int kPlanThreads = 256;
constexpr int kPlanCells = 12;
constexpr int kPlanSortMax = 2048;  // capacity of the workgroup's card list (1024 walkers: at most 1024 cards)

// The recipe's tables for the one-thread form (recipe.h, recipe_scalar2), copied into LDS by a workgroup of kPlanThreads
// threads and padded with +inf to the sizes the 4-ary searches walk: every load first (clamped indices, no branches between
// them: one round trip), then the stores.  The caller's barrier publishes them.
__device__ __forceinline__ void stage_scalar_tables(const unsigned char *__restrict__ rblk, const DevProblem &P, int niso, int nt, int ng,
                                                    ScalarTabs &T, ScalarPriorTabs &TP) {
    __shared__ double s_isot[kPlanIsoPad], s_isol[kPlanIsoPad], s_teff[kPlanNodePad + 1], s_logg[kPlanNodePad + 1];
    __shared__ double4 s_isopack[kPlanIsoPad];
    __shared__ double s_ave[kPlanIsoPad], s_avm[2 * kWave], s_avs[2 * kWave];
    __shared__ unsigned int s_pmask[kWave];
    static_assert(kPlanThreads == kPlanIsoPad, "one table entry per thread");
    const int tid = threadIdx.x;
    const double *g_isot = reinterpret_cast<const double *>(rblk + kRbIsoT);
    const double4 *g_pack = reinterpret_cast<const double4 *>(rblk + kRbIsoPack);
    const double *g_teff = reinterpret_cast<const double *>(rblk + kRbTeff), *g_logg = reinterpret_cast<const double *>(rblk + kRbLogg);
    const int nav = P.nav;
    const int ii = tid < niso ? tid : niso - 1, it = tid < nt ? tid : nt - 1, ig = tid < ng ? tid : ng - 1;
    const int ie = tid < nav + 1 ? tid : 0, ia = tid < nav ? tid : 0;
    const double v_isot = g_isot[ii], v_isol = P.iso_l[ii];
    const double4 v_pack = g_pack[tid];  // (the block holds all 256 entries)
    const double v_teff = g_teff[it], v_logg = g_logg[ig];
    const unsigned int v_mask = reinterpret_cast<const unsigned int *>(rblk + kRbPresent)[it];
    const double v_ave = nav > 0 ? P.av_edges[ie] : 0.0, v_avm = nav > 0 ? P.av_mu[ia] : 0.0, v_avs = nav > 0 ? P.av_sig[ia] : 0.0;
    s_isot[tid] = tid < niso ? v_isot : INFINITY;
    s_isol[tid] = v_isol;
    s_isopack[tid] = v_pack;
    s_ave[tid] = (nav > 0 && tid < nav + 1) ? v_ave : INFINITY;
    if (tid < 2 * kWave) { s_avm[tid] = v_avm; s_avs[tid] = v_avs; }
    if (tid <= kPlanNodePad) { s_teff[tid] = tid < nt ? v_teff : INFINITY; s_logg[tid] = tid < ng ? v_logg : INFINITY; }
    if (tid < kWave) s_pmask[tid] = tid < nt ? v_mask : 0u;
    T = ScalarTabs{s_isot, s_teff, s_logg, s_isopack, s_pmask, s_ave, niso, nt, ng, nav};
    TP = ScalarPriorTabs{s_isot, s_isol, s_avm, s_avs};
}

// The lanes of one tag, in lane order: ranks 0 and 1 are a pair, 2 and 3, ...; the odd one out of a tag is `leftover`.
// Ballots only.  partner_lane = the lane this (even-rank) lane pairs with, or -1.
__device__ __forceinline__ void plan_wave_pairs(unsigned long long tag, int lane, int *partner_lane, bool *leftover) {
    *partner_lane = -1;
    *leftover = false;
    unsigned long long rem = __ballot(tag != 0ull);
    unsigned long long mygrp = 0ull;  // the lanes of this lane's tag, once its group's trip has come
    // (uniform: one trip per distinct cell in the wave -- up to kPlanCells of them: a wave with more is part of an
    // ensemble spread over the grid, where there is little to pair and no point in 64 trips; the rest stay unplaced.
    // A trip only finds the group; what a lane does with its group is worked out once, after the loop.)
    for (int trip = 0; rem != 0ull && trip < kPlanCells; ++trip) {
        const int leader = __ffsll((long long)rem) - 1;
        const unsigned long long ltag = readlane_u64(tag, leader);
        const unsigned long long grp = __ballot(tag == ltag);  // (tags are never 0 here, and a lane of this tag is still in rem)
        mygrp = (tag == ltag) ? grp : mygrp;
        rem &= ~grp;
    }
    if (mygrp != 0ull) {
        const int r = __popcll(mygrp & ((1ull << lane) - 1ull));
        if ((r & 1) == 0) {
            const unsigned long long nxt = lane < 63 ? (mygrp & ~((2ull << lane) - 1ull)) : 0ull;
            if (nxt != 0ull) *partner_lane = __ffsll((long long)nxt) - 1; else *leftover = true;
        }
    }
    if ((rem >> lane) & 1ull) *leftover = true;
}

__global__ void __launch_bounds__(kPlanThreads)
pair_plan_kernel(const double *__restrict__ theta, const unsigned char *__restrict__ rblk, int niso_nt, int ng_mode_fast, int64_t n,
                 double gate_tmin, double gate_tmax, int32_t *__restrict__ plan, PairItem *__restrict__ pair_items,
                 PairRec *__restrict__ single_items, double *__restrict__ logp, int32_t *__restrict__ status, int32_t *__restrict__ host_stats, DevProblem P) {
    constexpr int NS = 2, ndim = 6;
    __shared__ unsigned long long s_cards[kPlanSortMax];
    __shared__ int2 s_pairs[kPlanSortMax / 2];
    __shared__ unsigned long long s_left[kPlanSortMax];
    __shared__ int s_nc, s_np, s_nl, s_ns, s_basep, s_basel;
    static_assert(kPlanThreads == kPlanIsoPad, "one table entry per thread");
    const int niso = niso_nt & 0xffff, nt = niso_nt >> 16;
    const int ng = ng_mode_fast & 0xff, mode = (ng_mode_fast >> 8) & 0xff;
    const GateArgs gates = {gate_tmin, gate_tmax, ((ng_mode_fast >> 18) & 1) != 0, ((ng_mode_fast >> 19) & 1) != 0};
    const int tid = threadIdx.x, lane = tid & 63;
    __shared__ int s_where[kPlanThreads];  // where this thread's walker goes: 2 (pair index) + slot, or -2 - (single index)
    s_where[tid] = -1;
#ifdef MSX_STAMPS
    if (tid == 0) msx_stamp_off = 0;
#endif
    MSX_STAMP(P, blockIdx.x, 0);
    // (the walker's coordinates are requested first, with the tables: one memory round trip, not two)
    const int64_t i = (int64_t)blockIdx.x * kPlanThreads + tid;
    const bool mine = i < n;
    double t[ndim];
#pragma unroll
    for (int k = 0; k < ndim; ++k) t[k] = mine ? theta[i * ndim + k] : 0.0;
    ScalarTabs T;
    ScalarPriorTabs TP;
    stage_scalar_tables(rblk, P, niso, nt, ng, T, TP);
    if (tid == 0) { s_nc = 0; s_np = 0; s_nl = 0; s_ns = 0; }
    __syncthreads();
    MSX_STAMP(P, blockIdx.x, 1);
    unsigned long long tag = 0ull;  // 0: nothing left to evaluate (rejected by the prior box, or an error status)
    PairRec mrec;
    int node[NS * 4], iso_lo[NS] = {0, 0}, av_bin = 0;
    double w[NS * 4];
    BandRows band_pre;
    if (mine) {
        double redc;
        const int st = recipe_scalar2(gates, T, mode, t, node, w, &redc, iso_lo, &av_bin);
        MSX_STAMP(P, blockIdx.x, 2);
        if (st != MSX_W_OK) {  // final here, like the fused kernel's first lines
            logp[i] = (st > MSX_W_REJECT) ? nan_with_status(st) : -INFINITY;
            status[i] = st;
        } else {
            PairRec *R = &mrec;
#pragma unroll
            for (int c = 0; c < NS * 4; ++c) { R->w[c] = w[c]; R->node[c] = node[c]; }
            R->redc = redc;
            R->walker = (int32_t)i;
            R->pad = 0;
            // (what only the walker's last line reads -- the prior terms, the contrast / photometry chi^2 -- is worked out
            // further down, while the workgroup's global adds are in flight; the band table's rows are requested here)
            band_rows_first(P, node, band_pre);
            unsigned long long h = 0x9E3779B97F4A7C15ull;
#pragma unroll
            for (int c = 0; c < NS * 4; ++c) {
                h ^= (unsigned long long)(unsigned int)node[c] + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2);
                h *= 0xBF58476D1CE4E5B9ull;
                h ^= h >> 31;
            }
            tag = (h >> 24) | 1ull;  // 40 bits, never zero
        }
    }
    // ---- 1. partners inside the wave ----------------------------------------------------------------------------------
    int partner_lane;
    bool leftover;
    plan_wave_pairs(tag, lane, &partner_lane, &leftover);
    const int partner_walker = __shfl((int)i, partner_lane >= 0 ? partner_lane : lane);
    const unsigned long long below = (1ull << lane) - 1ull;

    {   // the wave's pairs, leftover cards and singles into the workgroup's lists (one LDS atomic each per wave)
        const unsigned long long bp = __ballot(partner_lane >= 0), bl = __ballot(leftover), bs = 0ull;  // (no singles at this level)
        int basep = 0, basel = 0, bases = 0;
        if (lane == 0) {
            if (bp) basep = atomicAdd(&s_np, __popcll(bp));
            if (bl) basel = atomicAdd(&s_nc, __popcll(bl));
            if (bs) bases = atomicAdd(&s_ns, __popcll(bs));
        }
        basep = __builtin_amdgcn_readfirstlane(basep);
        basel = __builtin_amdgcn_readfirstlane(basel);
        bases = __builtin_amdgcn_readfirstlane(bases);
        // (a workgroup's 1024 walkers: at most 512 wave-level pairs, 1024 cards, 1024 singles)
        if (partner_lane >= 0) s_pairs[basep + __popcll(bp & below)] = make_int2((int)i, partner_walker);
        if (leftover) s_cards[basel + __popcll(bl & below)] = (tag << 24) | (unsigned long long)(i + 1);
        (void)bases;
    }
    __syncthreads();
    MSX_STAMP(P, blockIdx.x, 3);
    // ---- 2. the waves' leftovers meet inside the workgroup: the same ballots over the card list, 64 cards per wave, in
    //      rounds (128 cards -> at most one per cell and wave -> one wave -> at most one per cell) -------------------------
    int nc = s_nc;
    const int nsingle = s_ns;
    unsigned long long *src = s_cards, *dst = s_cards + kPlanSortMax / 2;
    for (int round = 0; round < 3 && nc > 1; ++round) {  // (uniform)
        const bool one_wave = nc <= kWave;  // every card meets every other in this round: nothing left to find after it
        if (tid == 0) s_nl = 0;
        __syncthreads();
        const int k = tid;  // (nc <= kPlanThreads)
        const unsigned long long cd = k < nc ? src[k] : 0ull;
        int pl;
        bool lo;
        plan_wave_pairs(cd >> 24, lane, &pl, &lo);
        const unsigned long long other = __shfl(cd, pl >= 0 ? pl : lane);
        const unsigned long long bp = __ballot(pl >= 0), bl = __ballot(lo);
        int basep = 0, basel = 0;
        if (lane == 0) {
            if (bp) basep = atomicAdd(&s_np, __popcll(bp));
            if (bl) basel = atomicAdd(&s_nl, __popcll(bl));
        }
        basep = __builtin_amdgcn_readfirstlane(basep);
        basel = __builtin_amdgcn_readfirstlane(basel);
        if (pl >= 0) s_pairs[basep + __popcll(bp & below)] = make_int2((int)(cd & 0xffffffull) - 1, (int)(other & 0xffffffull) - 1);
        if (lo) dst[basel + __popcll(bl & below)] = cd;
        __syncthreads();
        const int left = s_nl;
        if (left == nc) break;  // nothing met (every card its own cell, or one card per wave): the rest are singles
        nc = left;
        unsigned long long *t_ = src; src = dst; dst = t_;
        if (one_wave) break;
    }
    __syncthreads();
    for (int k = tid; k < nc; k += kPlanThreads) s_left[nsingle + k] = src[k];
    if (tid == 0) s_nl = nc;
    __syncthreads();
    MSX_STAMP(P, blockIdx.x, 4);
    // ---- the workgroup's lists to global memory: one atomic add per list.  What the workgroup could not place -- at
    //      most one walker per cell -- is a single. --------------------------------------------------------------------
    const int np = s_np, nl = s_nl;
    // (the adds' results are not looked at before the terms below are done: their round trip is the terms'.  ONE
    // instruction for the two of them, its address a function of the lane: a uniform address would have the compiler's
    // atomic optimizer rewrite the add with a readfirstlane -- and a wait -- right behind it)
    int gbase = 0;
    int tid_v = tid;  // (the thread index as a value the optimiser cannot see through: keeps the addresses per-lane)
    asm volatile("" : "+v"(tid_v));
    if (tid < 2) gbase = atomicAdd(&plan[2 + tid_v], tid == 0 ? np : nsingle + nl);
    MSX_STAMP(P, blockIdx.x, 5);
    // what only the walker's last line reads: the Gaussian prior terms (f1) and the contrast / photometry chi^2 (A5/A6) --
    // here and not by two waves of the pair kernel, whose workgroup would wait for their table round trips after its
    // median is long done
    if (tag != 0ull) {
        mrec.lp = prior_terms_scalar2(P, TP, mode, t, av_bin, iso_lo);
        MSX_STAMP(P, blockIdx.x, 6);
        mrec.chi_extra = band_terms_scalar2(P, mode, t, node, w, band_pre);
        MSX_STAMP(P, blockIdx.x, 7);
    }
    if (tid == 0) s_basep = gbase;
    if (tid == 1) s_basel = gbase;
    // The ticket of "the last workgroup publishes the counts" is taken HERE -- this workgroup's two adds have returned (both
    // are lanes of this wave: their results are in hand) and that is all the ticket stands for; the records' stores below are
    // the kernel boundary's to publish -- and looked at in the kernel's last lines, a round trip later.  (Address by lane, as
    // above.)
    int ticket = -1;
    if (tid < 1) ticket = atomicAdd(&plan[4 + tid_v], 1);
    __syncthreads();
    // every walker's recipe goes where its workgroup will look for it
    const int base = blockIdx.x * kPlanThreads;
    for (int k = tid; k < np; k += kPlanThreads) {
        s_where[s_pairs[k].x - base] = (s_basep + k) << 1;
        s_where[s_pairs[k].y - base] = ((s_basep + k) << 1) | 1;
    }
    for (int k = tid; k < nsingle + nl; k += kPlanThreads) s_where[(int)(s_left[k] & 0xffffffull) - 1 - base] = -2 - (s_basel + k);
    __syncthreads();
    if (tag != 0ull) {
        const int wh = s_where[tid];
        PairRec *dst = wh >= 0 ? &pair_items[wh >> 1].r[wh & 1] : single_items + (-2 - wh);
        *dst = mrec;
    }
    MSX_STAMP(P, blockIdx.x, 8);
    // ---- the last workgroup to take its ticket publishes the counts and leaves the working counters at zero for the next
    // launch (every other workgroup's adds had returned before its ticket was taken)
    if (tid == 0 && ticket == (int)gridDim.x - 1) {
        const int np_all = atomicExch(&plan[2], 0), ns_all = atomicExch(&plan[3], 0);
        plan[0] = np_all;
        plan[1] = ns_all;
        // ... and tells the host, which decides from it what the NEXT launches take (msx.hip, pair_worth_it): a word
        // of host memory, written and forgotten
        if (host_stats) {
            __hip_atomic_store(host_stats, np_all, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(host_stats + 1, ns_all, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        atomicExch(&plan[4], 0);
    }
    MSX_STAMP(P, blockIdx.x, 9);
}

}  // namespace

#endif  // MSX_PAIR_KERNEL_H
