// median.h -- part of the single translation unit msx.hip (included there, in this order).
// exact np.median of the model vector in LDS: block scratch, radix select, block_median (min/max bins) and logbin_median (histogram filled during phase A).
#ifndef MSX_MEDIAN_H
#define MSX_MEDIAN_H

namespace {

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
    alignas(16) unsigned int hist[kLogBins];  // block_median and radix_select use the first kBins / 256
    unsigned long long cand[kSelectFinish];
    unsigned long long sel_result[2];
    unsigned int sel_bin, sel_k, sel_cnt, cand_n, has_second;
    unsigned int cnt_le;
    unsigned int rk[2][kWave];  // split ranking (logbin_rank_part): per candidate, how many are below / not above it; zero on entry
    // linked form (logprob_kernel.h): what thread 0 learnt at the walker's meeting points
    unsigned int meet_state;
    unsigned long long meet_base;
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
// (slots 5 and 7: the 100 MHz wall clock at the walker's first and last stamp -- shader cycles / wall ticks = the clock the CU ran at)
#define MED_WALL(i) do { if (threadIdx.x == 0) g_med_stamps[blockIdx.x * 8 + (i)] = wall_clock64(); } while (0)
#else
#define MED_STAMP(i) do { } while (0)
#define MED_WALL(i) do { } while (0)
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
// element): prime() is called before the pass (the element may start loading what its first trip needs),
// begin_trip<PAR>() at the top of every trip (before the trip's LDS reads),
// process4<PAR>() gets the trip's base, its four pixel indices (a pixel p is valid iff p < npix; invalid ones carry
// the value of pixel npix - 1) and values -- PAR = the trip's parity: an element that loads one trip ahead keeps TWO
// register sets and alternates, instead of copying "next" to "current" every trip -- and flush() publishes the
// lanes' partials right before the pass's barrier.
struct NoElem {
    __device__ __forceinline__ void prime() {}
    template <int PAR> __device__ __forceinline__ void begin_trip(int) {}
    template <int PAR> __device__ __forceinline__ void process4(int, const int (&)[4], const double (&)[4]) {}
    __device__ __forceinline__ void flush(BlockScratch &) {}
};
// One pass over the pixels [p_lo, p_hi) of the model vector in trips of 4 BT pixels (pass_pixel order; p_lo a multiple
// of 8 BT): elem.process4, then per(p, xv).  A pixel p of a trip is valid iff p < p_hi.
// (Validity travels as the pixel index, not as a flag: flags handed through arrays get packed into bytes and
// unpacked again, a dozen instructions per trip; a compare against the end is one.)
// (FULL: every pixel of every trip is valid -- the caller has checked that the range is whole trips -- so no clamp)
template <int BT, bool FULL = false, class Elem, class Per>
__device__ __forceinline__ void pass_trips_range(const double *model, int p_lo, int p_hi, Elem &elem, Per per) {
    const int tid = threadIdx.x;
    auto one = [&](int base, auto par) __attribute__((always_inline)) {
        elem.template begin_trip<decltype(par)::value>(base);
        int p[4];
        double xv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {  // loads first, then use
            p[u] = pass_pixel<BT>(base, u, tid);
            xv[u] = model[(FULL || p[u] < p_hi) ? p[u] : p_hi - 1];
        }
        elem.template process4<decltype(par)::value>(base, p, xv);
        per(p, xv);
    };
    for (int base = p_lo; base < p_hi; base += 8 * BT) {
        one(base, std::integral_constant<int, 0>{});
        if (base + 4 * BT < p_hi) one(base + 4 * BT, std::integral_constant<int, 1>{});  // (uniform)
    }
}
// ... over the whole vector
template <int BT, bool FULL = false, class Elem, class Per>
__device__ __forceinline__ void pass_trips(const double *model, int npix, Elem &elem, Per per) {
    pass_trips_range<BT, FULL>(model, 0, npix, elem, per);
}

__device__ __forceinline__ unsigned long long readlane_u64(unsigned long long v, int l) {  // l wave-uniform
    const unsigned int lo = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)v, l);
    const unsigned int hi = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)(v >> 32), l);
    return ((unsigned long long)hi << 32) | lo;
}

template <int BT, class Side, class Elem>
__device__ __forceinline__ double block_median(const double *model, int npix, unsigned long long kmin, unsigned long long kmax,
                               BlockScratch &S, Side side, Elem &elem, bool *elem_done) {
    bool side_done = false;  // `side` runs exactly once, preferably in the stage that keeps only wave 0 busy
    *elem_done = false;
    constexpr int B = BT, nw = BT >> 6;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
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
            elem.prime();
            pass_trips<BT>(model, npix, elem, [&](const int (&p)[4], const double (&xv)[4]) __attribute__((always_inline)) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    int bin = (int)((xv[u] - vmin) * scale);
                    bin = bin > kBins - 1 ? kBins - 1 : bin;
                    if (p[u] < npix) atomicAdd(&S.hist[bin], 1u);
                }
            });
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

// The histogram's two increments of one table ELEMENT (pixels pa and pa + 256; f0, f1 their unmasked bin numbers).  A wave's
// atomic instruction would hold 64 NEIGHBOURING pixels -- near-equal values, one bin, one LDS address: the adds serialise
// (rounds 1-3: 17.5 % of the LDS cycles of the 256-walker launch, 24.8 % of the pair kernel's, were bank-conflict cycles).
// Odd lanes therefore send their two increments in the other order: neighbouring lanes of one instruction now hold pixels
// 256 apart.  The histogram is the same multiset of increments, so nothing downstream changes.
__device__ __forceinline__ void hist_add_pair(unsigned int *hist, unsigned int f0, bool ok0, unsigned int f1, bool ok1) {
    const bool odd = (threadIdx.x & 1) != 0;
    const unsigned int b0 = odd ? f1 : f0, b1 = odd ? f0 : f1;
    const bool k0 = odd ? ok1 : ok0, k1 = odd ? ok0 : ok1;
    if (k0) atomicAdd(&hist[b0 & (unsigned int)(kLogBins - 1)], 1u);
    if (k1) atomicAdd(&hist[b1 & (unsigned int)(kLogBins - 1)], 1u);
}

// The exact value range of model[0 .. npix) as order-preserving keys (a NaN anywhere: kmax = ~0, above +inf), for the
// paths that need it (block_median; the NaN -> -inf rule).  All threads call it; uses S.kmin / S.kmax and one barrier.
template <int BT>
__device__ __forceinline__ void exact_range(const double *model, int npix, BlockScratch &S, unsigned long long *kmin_out,
                                            unsigned long long *kmax_out) {
    constexpr int nw = BT >> 6;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double vmin = INFINITY, vmax = -INFINITY;
    bool seen_nan = false;
    for (int p = tid; p < npix; p += BT) {
        const double m = model[p];
        vmin = min_nc(vmin, m);
        vmax = max_nc(vmax, m);
        seen_nan = seen_nan || (m != m);
    }
    const double lo = wave_min_f64(vmin), hi = wave_max_f64(vmax);
    const bool wave_nan = __ballot(seen_nan) != 0ull;
    __syncthreads();  // (whoever still reads the slots' previous contents is done)
    if (lane == 0) {
        // (a wave with no pixel reports the empty range; fmin / fmax do not order -0 and +0: a zero bound stands for both)
        S.kmin[wave] = lo == INFINITY && hi == -INFINITY ? ~0ull : key_of(lo == 0.0 ? -0.0 : lo);
        S.kmax[wave] = wave_nan ? ~0ull : (lo == INFINITY && hi == -INFINITY ? 0ull : key_of(hi == 0.0 ? 0.0 : hi));
    }
    __syncthreads();
    unsigned long long kmin = S.kmin[0], kmax = S.kmax[0];
    for (int x = 1; x < nw; ++x) {
        kmin = S.kmin[x] < kmin ? S.kmin[x] : kmin;
        kmax = S.kmax[x] > kmax ? S.kmax[x] : kmax;
    }
    *kmin_out = kmin;
    *kmax_out = kmax;
}

// The early histogram's counters, turned IN PLACE into running totals: thread t owns counters [t PT, (t + 1) PT) and
// leaves in each the number of values in its wave's block of bins up to and including that bin; S.wave_tot[w] gets
// the block's total.  Every thread of the workgroup calls it once the counters are complete (a barrier has passed);
// one more barrier publishes the result.  logbin_median then finds a rank with three dependent LDS reads (the wave
// totals, one running total per lane, one thread's counters) instead of summing all 2048 counters in every wave.
template <int BT>
__device__ __forceinline__ void hist_prefix_inplace(BlockScratch &S) {
    constexpr int PT = kLogBins / BT;  // counters per thread: 4 (512 threads) or 8 (256)
    static_assert(PT == 4 || PT == 8, "256 or 512 threads");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint4 *h = reinterpret_cast<uint4 *>(&S.hist[tid * PT]);
    uint4 a = h[0], b = make_uint4(0u, 0u, 0u, 0u);
    a.y += a.x; a.z += a.y; a.w += a.z;
    unsigned int tot = a.w;
    if (PT == 8) {
        b = h[1];
        b.x += a.w; b.y += b.x; b.z += b.y; b.w += b.z;
        tot = b.w;
    }
    const unsigned int inc = wave_scan_u32(tot), ex = inc - tot;
    a.x += ex; a.y += ex; a.z += ex; a.w += ex;
    h[0] = a;
    if (PT == 8) {
        b.x += ex; b.y += ex; b.z += ex; b.w += ex;
        h[1] = b;
    }
    if (lane == kWave - 1) S.wave_tot[wave] = inc;
}

// The three pieces of logbin_median, separate so that the pair kernel (pair_kernel.h), whose model values sit in
// registers, runs the same rank location and the same ranking around a pass of its own.
struct LogbinSel {
    unsigned int sel_p, nxt_p;  // PHYSICAL bins whose values are the candidates (nxt_p == sel_p: one bin)
    unsigned int kk, cnt;       // ranks kk (and kk + 1 for even npix) among the cnt candidates are the middle values
};
// Where is the median?  Every wave computes the answer for itself from the running totals of hist_prefix_inplace
// (uniform result; no barrier).  false: more than kSelectFinish candidates -- the caller takes block_median.
// (hmin = the unmasked bin number of the vector's minimum, hi32(min) >> 12)
template <int BT>
__device__ __forceinline__ bool logbin_locate_h(int npix, unsigned int hmin, const BlockScratch &S, LogbinSel *out) {
    constexpr int nw = BT >> 6;
    const int lane = threadIdx.x & 63;
    const unsigned int k1 = (unsigned int)((npix - 1) >> 1);
    const bool need_two = (npix & 1) == 0;
    const unsigned int a = hmin & (unsigned int)(kLogBins - 1);
    // ---- locate rank k1 along the cycle from the running totals (hist_prefix_inplace) ---------------------------
    constexpr int PT = kLogBins / BT, blk = kWave * PT;  // counters per thread; bins per wave block
    unsigned int woff[nw + 1];  // values before wave block w
    woff[0] = 0u;
#pragma unroll
    for (int w = 0; w < nw; ++w) woff[w + 1] = woff[w] + S.wave_tot[w];
    // values in the physical bins before the cycle's origin come LAST along the cycle
    unsigned int base = 0u;
    if (a != 0u) {
        const unsigned int am = a - 1u;
        unsigned int wo = 0u;
#pragma unroll
        for (int w = 1; w < nw; ++w) wo = (am >= (unsigned int)(w * blk)) ? woff[w] : wo;
        base = wo + S.hist[am];
    }
    const unsigned int n_hi = (unsigned int)npix - base;
    const unsigned int r = k1 < n_hi ? base + k1 : k1 - n_hi;  // rank k1 in PHYSICAL bin order
    int W = 0;
#pragma unroll
    for (int w = 1; w < nw; ++w) W = (r >= woff[w]) ? w : W;
    unsigned int wsel = 0u;
#pragma unroll
    for (int w = 1; w < nw; ++w) wsel = (W == w) ? woff[w] : wsel;
    const unsigned int rp = r - wsel;  // ... within wave block W
    const unsigned int T = S.hist[W * blk + lane * PT + PT - 1];  // running total at the end of lane's counters
    const int L = uni(__ffsll((long long)__ballot(T > rp)) - 1) & 63;
    const unsigned int prevT = L > 0 ? (unsigned int)__builtin_amdgcn_readlane((int)T, L - 1) : 0u;
    unsigned int cv[PT];
    {
        const uint4 *q4 = reinterpret_cast<const uint4 *>(&S.hist[W * blk + L * PT]);  // (uniform address: a broadcast)
        const uint4 x = q4[0];
        cv[0] = x.x; cv[1] = x.y; cv[2] = x.z; cv[3] = x.w;
        if (PT == 8) {
            const uint4 y = q4[1];
            cv[PT - 4] = y.x; cv[PT - 3] = y.y; cv[PT - 2] = y.z; cv[PT - 1] = y.w;
        }
    }
    int isel = 0;
    unsigned int below = prevT, upto = cv[0];  // running totals before / through the selected bin
#pragma unroll
    for (int i = 0; i + 1 < PT; ++i) {
        const bool past = cv[i] <= rp;  // (monotone: true for a prefix of the counters)
        isel = past ? i + 1 : isel;
        below = past ? cv[i] : below;
        upto = past ? cv[i + 1] : upto;
    }
    const unsigned int kk = rp - below;
    unsigned int cnt = upto - below;
    const unsigned int sel_p = (unsigned int)(W * blk + L * PT + isel);  // PHYSICAL bin of rank k1
    // Rank k1 + 1 (even npix) lies in the same bin unless rank k1 is that bin's last value; then it is the smallest
    // value of the NEXT non-empty bin along the cycle, and that bin's values are gathered too: in the union, ranks
    // kk and kk + 1 are the two middle values either way.
    unsigned int nxt_p = sel_p;
    if (need_two && kk + 1 == cnt) {  // (uniform; ~1 walker in `cnt`)
        unsigned int p0 = sel_p + 1u;
        for (int it = 0; it < kLogBins / kWave; ++it, p0 += (unsigned int)kWave) {  // (one exists: rank k1 + 1 < npix)
            const unsigned int bb = (p0 + (unsigned int)lane) & (unsigned int)(kLogBins - 1);
            const unsigned int hb = S.hist[bb], hp = (bb & (unsigned int)(blk - 1)) ? S.hist[bb - 1u] : 0u;
            const unsigned long long nz = __ballot(hb != hp);
            if (nz != 0ull) {
                const int J = uni(__ffsll((long long)nz) - 1);
                nxt_p = (p0 + (unsigned int)J) & (unsigned int)(kLogBins - 1);
                cnt += (unsigned int)__builtin_amdgcn_readlane((int)(hb - hp), J);
                break;
            }
        }
    }
    out->sel_p = sel_p; out->nxt_p = nxt_p; out->kk = kk; out->cnt = cnt;
    return cnt <= (unsigned int)kSelectFinish;  // (more: heavy duplication, the general path sorts it out)
}
// Rank the gathered candidates (S.cand[0 .. cnt), complete: a barrier has passed).  Up to 64 candidates (the usual
// case): wave `rank_wave` alone, in registers, and only that wave learns the median -- no further barrier.  More: the
// first waves through LDS, one barrier, every thread learns it.  All threads of the workgroup call it.
template <int BT>
__device__ __forceinline__ double logbin_rank(BlockScratch &S, const LogbinSel &Q, bool need_two, int rank_wave) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned int cnt = Q.cnt, kk = Q.kk;
    unsigned long long v1 = 0, v2 = 0;
    bool second = false;
    if (cnt <= (unsigned int)kWave) {
        if (wave == rank_wave) {
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
    (void)second;  // (even npix: rank kk + 1 is among the candidates by construction)
    return need_two ? (val_of(v1) + val_of(v2)) / 2.0 : val_of(v1);
}

// The same ranking for <= 64 candidates, SPLIT over several waves (the fused kernel's last lines: the pass is over, most
// waves have nothing left to do, and one wave comparing every candidate with every other is the longest chain left):
// wave `part` of `nparts` takes every nparts-th trip of eight candidates and adds its counts to S.rk (zero on entry);
// after a barrier, logbin_rank_pick (one wave) reads the sums.  The pad slots S.cand[cnt .. cnt + 8) must hold ~0.
template <int BT>
__device__ __forceinline__ void logbin_rank_part(BlockScratch &S, const LogbinSel &Q, int part, int nparts) {
    const int lane = threadIdx.x & 63;
    const unsigned int cnt = Q.cnt;
    const unsigned long long mine = S.cand[lane < (int)cnt ? lane : (int)cnt];
    unsigned int lt = 0, le = 0;
    const ulonglong2 *c2 = reinterpret_cast<const ulonglong2 *>(S.cand);
    for (int j = 8 * part; j < (int)cnt; j += 8 * nparts) {
        ulonglong2 x[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) x[u] = c2[(j >> 1) + u];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            lt += (x[u].x < mine) + (x[u].y < mine);
            le += (x[u].x <= mine) + (x[u].y <= mine);
        }
    }
    if (lane < (int)cnt && 8 * part < (int)cnt) {
        atomicAdd(&S.rk[0][lane], lt);
        atomicAdd(&S.rk[1][lane], le);
    }
}
template <int BT>
__device__ __forceinline__ double logbin_rank_pick(BlockScratch &S, const LogbinSel &Q, bool need_two) {
    const int lane = threadIdx.x & 63;
    const unsigned int cnt = Q.cnt, kk = Q.kk;
    const unsigned long long mine = S.cand[lane < (int)cnt ? lane : (int)cnt];
    const unsigned int lt = S.rk[0][lane], le = S.rk[1][lane];
    const unsigned long long b1 = __ballot(lane < (int)cnt && lt <= kk && kk < le);
    const unsigned long long b2 = __ballot(lane < (int)cnt && lt <= kk + 1 && kk + 1 < le);
    const unsigned long long v1 = readlane_u64(mine, uni(__ffsll((long long)b1) - 1));
    unsigned long long v2 = 0;
    if (b2 != 0ull) v2 = readlane_u64(mine, uni(__ffsll((long long)b2) - 1));
    return need_two ? (val_of(v1) + val_of(v2)) / 2.0 : val_of(v1);
}

// The value range as the early-histogram median wants it: the range [f0, f1] of the UNMASKED bin number
// F(x) = hi32(x) >> 12 over the vector.  All values positive normal numbers <=> f0 >= 1 and f1 < 0x7ff00 (zeros and
// subnormals have F = 0; infinities, NaNs and every negative number F >= 0x7ff00); spanning less than the histogram's
// cycle <=> f1 - f0 < 2048.  Integer min / max per pixel and one-instruction DPP steps per reduction, where the float64
// range (min, max, a NaN flag) costs five times that -- and the exact range is only needed by block_median.
__device__ __forceinline__ bool frange_applicable(unsigned int f0, unsigned int f1) {
    return f0 >= 1u && f1 < 0x7ff00u && f1 - f0 < (unsigned int)kLogBins;
}
// hmin = F of the vector's minimum; the caller has checked frange_applicable
// split != nullptr: with <= 64 candidates the ranking is left to the caller (logbin_rank_part / _pick; *split = true, the
// bins in *Qout, the pad slots written) -- it has idle waves to spread it over
template <int BT, bool FULL = false, class Elem>
__device__ __forceinline__ bool logbin_median(const double *model, int npix, unsigned int hmin, BlockScratch &S, Elem &elem,
                                              double *med_out, bool *split = nullptr, LogbinSel *Qout = nullptr) {
    const bool need_two = (npix & 1) == 0;
    MED_STAMP(0);
    MED_STAMP(1);
    elem.prime();  // the pass's first loads travel while the rank is located
    LogbinSel Q;
    if (!logbin_locate_h<BT>(npix, hmin, S, &Q)) return false;
    const unsigned int sel_p = Q.sel_p, nxt_p = Q.nxt_p;
    MED_STAMP(2);
    // ---- one pass: chi^2 terms + the candidates (the values of the one or two bins above) ----------------------
    pass_trips<BT, FULL>(model, npix, elem, [&](const int (&p)[4], const double (&xv)[4]) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const unsigned int pb = logbin(xv[u]);
            if ((FULL || p[u] < npix) && (pb == sel_p || pb == nxt_p)) S.cand[atomicAdd(&S.cand_n, 1u)] = key_of(xv[u]);
        }
    });
    if (split) {
        *split = Q.cnt <= (unsigned int)kWave;
        *Qout = Q;
        if (*split && threadIdx.x < 8) S.cand[Q.cnt + threadIdx.x] = ~0ull;  // (the slots behind the candidates: Q.cnt is exact)
    }
    elem.flush(S);
    __syncthreads();
    MED_STAMP(3);
    if (split && *split) return true;
    *med_out = logbin_rank<BT>(S, Q, need_two, 0);
    MED_STAMP(4);
#ifdef MSX_STAMPS
    if (threadIdx.x == 0) g_med_stamps[blockIdx.x * 8 + 6] = Q.cnt;
#endif
    return true;
}

}  // namespace

#endif  // MSX_MEDIAN_H
