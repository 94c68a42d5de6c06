// split_kernels.h -- part of the single translation unit msx.hip (included there, in this order).
// The SPLIT form of the hot path, for launches the one-workgroup-per-walker kernel serves badly: many walkers
// (every workgroup re-pulls the same few grid rows through its CU's L2 port) and few walkers with long spectra
// (half the CUs idle).  Four launches per batch, same arithmetic and the same bits as the fused kernel:
//
//   1. logprob_kernel<..., STAGE = 1>   the recipe alone: prior gate, A1, A2, A4 weights, prior + band terms
//                                       -> one WalkerRec per walker (and the final answer of rejected walkers)
//   2. plan_tiles_kernel                groups the live walkers by grid cell (their sorted node lists) and cuts
//                                       every group into tiles of <= kTileWalkers walkers
//   3. blend_tiles_kernel               work item = (tile, chunk of pixels): loads the tile's 4*NS pair rows ONCE
//                                       and blends them for every walker of the tile -> model[walker][pixel]
//                                       in a global scratch (A2, A4, A7, A8.1); pixels spread over all CUs
//   4. logprob_kernel<..., STAGE = 2>   per walker: reads its model vector back; fit sums, exact median,
//                                       continuum fit, chi^2, combine (A8.2, A8.3, A9) exactly as the fused kernel
//
// A walker's value depends on nothing but its own coordinates: which tile it lands in, and with whom, changes no
// bit (blend_pixel() is the fused kernel's own inner function and every sum is taken in the canonical order).
#ifndef MSX_SPLIT_KERNELS_H
#define MSX_SPLIT_KERNELS_H

namespace {

constexpr int kPlanSlots = 2048;    // hash slots of the planner (distinct grid-cell combinations it can group)
constexpr int kPlanThreads = 1024;
constexpr int kPlanProbes = 64;

// ------------------------------------------------------------------------------------------------
// Stage 2: one workgroup plans the batch.  hdr[0] = number of tiles, hdr[1] = number of live walkers.
// Groups beyond the table's capacity (or, never observed, two node lists with one hash) become
// one-walker tiles: slower, same values.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kPlanThreads)
plan_tiles_kernel(const WalkerRec *__restrict__ rec, int n, int nc, int32_t *__restrict__ perm, TileHdr *__restrict__ tiles,
                  int32_t *__restrict__ hdr, int32_t *__restrict__ tmp_slot, int32_t *__restrict__ tmp_pos) {
    __shared__ unsigned long long tkey[kPlanSlots];
    __shared__ int trep[kPlanSlots];
    __shared__ int tcnt[kPlanSlots];
    __shared__ int wbase[kPlanSlots];   // live walkers in earlier slots
    __shared__ int tbase[kPlanSlots];   // tiles of earlier slots
    __shared__ int wave_w[kPlanThreads / kWave], wave_t[kPlanThreads / kWave];
    __shared__ int nsingle, tot_w, tot_t;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < kPlanSlots; i += kPlanThreads) { tkey[i] = 0ull; trep[i] = 0x7fffffff; tcnt[i] = 0; }
    if (tid == 0) nsingle = 0;
    __syncthreads();
    // ---- insert every live walker's key ------------------------------------------------------------
    for (int i = tid; i < n; i += kPlanThreads) {
        int slot = -1;  // -1: not live
        if (rec[i].status == MSX_W_OK) {
            const unsigned long long h = rec[i].key;
            int s = (int)(h % (unsigned long long)kPlanSlots);
            slot = -2;  // -2: no slot found -> single
            for (int probe = 0; probe < kPlanProbes; ++probe) {
                const unsigned long long old = atomicCAS(&tkey[s], 0ull, h);
                if (old == 0ull || old == h) { slot = s; break; }
                s = (s + 1) & (kPlanSlots - 1);
            }
            if (slot >= 0) atomicMin(&trep[slot], i);
        }
        tmp_slot[i] = slot;
    }
    __syncthreads();
    // ---- verify against the slot's representative (a hash is not an identity), take a position ------
    for (int i = tid; i < n; i += kPlanThreads) {
        int slot = tmp_slot[i];
        if (slot >= 0) {
            const int r = trep[slot];
            bool same = true;
            for (int c = 0; c < nc; ++c) same = same && rec[i].node[c] == rec[r].node[c];
            if (!same) slot = -2;
        }
        int pos = 0;
        if (slot >= 0) pos = atomicAdd(&tcnt[slot], 1);
        else if (slot == -2) pos = atomicAdd(&nsingle, 1);
        tmp_slot[i] = slot;
        tmp_pos[i] = pos;
    }
    __syncthreads();
    // ---- exclusive prefix sums over the slots: walkers and tiles ------------------------------------
    constexpr int per = kPlanSlots / kPlanThreads;  // 2 slots per thread
    int cw[per], ct[per], sw = 0, st = 0;
#pragma unroll
    for (int k = 0; k < per; ++k) {
        cw[k] = tcnt[tid * per + k];
        ct[k] = (cw[k] + kTileWalkers - 1) / kTileWalkers;
        sw += cw[k];
        st += ct[k];
    }
    const int iw = (int)wave_scan_u32((unsigned int)sw), it = (int)wave_scan_u32((unsigned int)st);
    if (lane == 63) { wave_w[wave] = iw; wave_t[wave] = it; }
    __syncthreads();
    int bw = 0, bt = 0;
    for (int x = 0; x < wave; ++x) { bw += wave_w[x]; bt += wave_t[x]; }
    int ew = bw + iw - sw, et = bt + it - st;
#pragma unroll
    for (int k = 0; k < per; ++k) {
        wbase[tid * per + k] = ew;
        tbase[tid * per + k] = et;
        ew += cw[k];
        et += ct[k];
    }
    if (tid == kPlanThreads - 1) { tot_w = ew; tot_t = et; }
    __syncthreads();
    const int grouped_w = tot_w, grouped_t = tot_t;
    // ---- walker order, tile headers ---------------------------------------------------------------------
    for (int i = tid; i < n; i += kPlanThreads) {
        const int slot = tmp_slot[i], pos = tmp_pos[i];
        if (slot >= 0) {
            perm[wbase[slot] + pos] = i;
        } else if (slot == -2) {
            perm[grouped_w + pos] = i;
            TileHdr t;
            t.start = grouped_w + pos; t.count = 1; t.pad[0] = t.pad[1] = 0;
            for (int c = 0; c < kMaxCorners; ++c) t.node[c] = c < nc ? rec[i].node[c] : 0;
            tiles[grouped_t + pos] = t;
        }
    }
    for (int s = tid; s < kPlanSlots; s += kPlanThreads) {
        const int cnt = tcnt[s];
        if (cnt == 0) continue;
        const int r = trep[s];
        TileHdr t;
        t.pad[0] = t.pad[1] = 0;
        for (int c = 0; c < kMaxCorners; ++c) t.node[c] = c < nc ? rec[r].node[c] : 0;
        for (int j = 0; j * kTileWalkers < cnt; ++j) {
            t.start = wbase[s] + j * kTileWalkers;
            t.count = cnt - j * kTileWalkers < kTileWalkers ? cnt - j * kTileWalkers : kTileWalkers;
            tiles[tbase[s] + j] = t;
        }
    }
    if (tid == 0) { hdr[0] = grouped_t + nsingle; hdr[1] = grouped_w + nsingle; }
}

// ------------------------------------------------------------------------------------------------
// Stage 3: work item = (tile, chunk of 256 table elements = 512 pixels), grid-strided.  With a chunk count that is a
// multiple of 8 (4096 px / 512, 16384 px / 512) and a grid that is a multiple of 8, block b only ever sees chunks = b mod 8:
// under round-robin dispatch every XCD's L2 holds one eighth of each pair row (speed only).
// ------------------------------------------------------------------------------------------------
template <int NS>
__global__ void __launch_bounds__(256)
blend_tiles_kernel(const WalkerRec *__restrict__ rec, const int32_t *__restrict__ perm, const TileHdr *__restrict__ tiles,
                   const int32_t *__restrict__ hdr, const double2 *__restrict__ r2, const float2 *__restrict__ h2,
                   const double2 *__restrict__ kl2, const float2 *__restrict__ dk2, int npix, int npair,
                   double *__restrict__ model) {
    constexpr int NC = NS * 4;
    __shared__ __attribute__((aligned(16))) double sW[kTileWalkers][NC];
    __shared__ double sRedc[kTileWalkers];
    __shared__ int sWalker[kTileWalkers];
    __shared__ double e2tab[kExp2Tab];
    const int tid = threadIdx.x;
    fill_exp2_table(e2tab, tid);  // (published by the first item's barrier)
    const int ntiles = hdr[0];
    const int nchunk = npair / 256;  // one chunk = 256 elements = 512 pixels
    const long long nitems = (long long)ntiles * nchunk;
    for (long long item = blockIdx.x; item < nitems; item += gridDim.x) {
        const int chunk = (int)(item % nchunk);
        const int tile = (int)(item / nchunk);
        const TileHdr *th = tiles + tile;
        const int start = __builtin_amdgcn_readfirstlane(th->start), count = __builtin_amdgcn_readfirstlane(th->count);
        // rows first: the longest latency of the item
        const int e = chunk * 256 + tid;
        const int pa = chunk * 512 + tid, pb = pa + 256;
        double2 rr[NC];
        float2 hh[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int64_t off = (int64_t)__builtin_amdgcn_readfirstlane(th->node[c]) * npair;
            rr[c] = r2[off + e];
            hh[c] = h2[off + e];
        }
        const double2 kl = kl2[e];
        const float2 dk = dk2[e];
        // the tile's walkers: weights + reddening coefficient to LDS (one element per thread)
        if (tid < count * (NC + 1)) {
            const int wi = tid / (NC + 1), c = tid - wi * (NC + 1);
            const int wk = perm[start + wi];
            if (c < NC) sW[wi][c] = rec[wk].w[c];
            else { sRedc[wi] = rec[wk].redc; sWalker[wi] = wk; }
        }
        __syncthreads();
        double ra[NC], rb[NC];
        float ha[NC], hb[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) { ra[c] = rr[c].x; rb[c] = rr[c].y; ha[c] = hh[c].x; hb[c] = hh[c].y; }
        for (int wi = 0; wi < count; ++wi) {
            double w[NC];
            float wf[NC];
#pragma unroll
            for (int c = 0; c < NC; ++c) { w[c] = sW[wi][c]; wf[c] = (float)w[c]; }
            const double redc = sRedc[wi];
            const bool redden = redc != 0.0;
            double *out = model + (int64_t)sWalker[wi] * npix;
            const double ma = blend_pixel_rh<NC>(ra, ha, w, wf, kl.x, (double)dk.x, redc, redden, e2tab);
            const double mb = blend_pixel_rh<NC>(rb, hb, w, wf, kl.y, (double)dk.y, redc, redden, e2tab);
            if (pa < npix) out[pa] = ma;
            if (pb < npix) out[pb] = mb;
        }
        __syncthreads();  // the next item overwrites sW
    }
}

}  // namespace

#endif  // MSX_SPLIT_KERNELS_H
