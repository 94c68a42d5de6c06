// logprob_kernel.h -- part of the single translation unit msx.hip (included there, in this order).
// THE HOT KERNEL logprob_kernel<NS,U,MAXT,GM,CP,PF> and the walker's last lines (walker_done).
#ifndef MSX_LOGPROB_KERNEL_H
#define MSX_LOGPROB_KERNEL_H

namespace {

__device__ __forceinline__ double nan_with_status(int st) {
    return __longlong_as_double(0x7ff8000000000000ll | (long long)(st & 0xff));
}
__device__ __forceinline__ int status_of_nan(double v) {  // 0 for anything that is not one of the NaNs above
    const long long b = __double_as_longlong(v);
    return ((b & 0x7ff8000000000000ll) == 0x7ff8000000000000ll && (b >> 63) == 0) ? (int)(b & 0xff) : 0;
}

// linked form: a store that is handed to another workgroup -- agent scope, i.e. written through the XCD's L2
__device__ __forceinline__ void publish_u64(unsigned long long *p, unsigned long long v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- the walker's scalar arithmetic after the pixel loops, in ONE place: the fused kernel and the pair kernel
// (pair_kernel.h) must give a walker the same bits, so both call these and nothing else ---------------------------
// one pixel's contribution to the three fit sums of data / model against [1, u, u^2]      mft6.py:194-195
__device__ __forceinline__ void fit_accumulate(double m, double flux, double u, double &q0, double &q1, double &q2) {
    const double f = fast_div(flux, m);  // frac before the median scale, mft6.py:194
    const double f1 = f * u, f2 = f * (u * u);
    q0 += f; q1 += f1; q2 += f2;
}
// coefficients of the raw quadratic fit of data / model from the three fit sums (mft6.py:195; minv = inverse Gram matrix)
__device__ __forceinline__ void fit_coefs(const DevProblem &P, const double (&q)[3], double &c0, double &c1, double &c2) {
    c0 = P.minv[0] * q[0] + P.minv[1] * q[1] + P.minv[2] * q[2];
    c1 = P.minv[3] * q[0] + P.minv[4] * q[1] + P.minv[5] * q[2];
    c2 = P.minv[6] * q[0] + P.minv[7] * q[1] + P.minv[8] * q[2];
}
// one pixel's chi^2 term before the median's scale^2: (model - data / P(u))^2 / err^2      mft6.py:196,120
__device__ __forceinline__ double chi_term(double c0, double c1, double c2, double u, double f, double e, double xv, bool live) {
#pragma clang fp contract(off)
    // (no contraction: the term is a PRODUCT, added by the caller.  Where `live` is a compile-time true -- the pair kernel's
    // FULL variants -- nothing stands between this multiply and the caller's add any more, and a fused multiply-add there
    // would round differently from every variant that selects on `live`: a walker's bits must not depend on the variant.)
    const double poly = fma(fma(c2, u, c1), u, c0);
    const double r = xv - fast_div(f, poly);
    const double t = (r * r) * e;
    return live ? t : 0.0;
}
// the walker's value from its chi^2 sum (fused modes: before scale^2), the two medians and the recipe's scalars
__device__ __forceinline__ double fused_total(const DevProblem &P, double chi_sum, double med_data, double med_model, int npix,
                                              double chi_extra) {
    const double scale = fast_div(med_data, med_model);  // mft6.py:1173
    const double tot = chi_sum * (scale * scale);
    const double iic = fast_div(tot, (double)npix);                // mft6.py:1179
    return iic * (double)(P.nc + P.np) + chi_extra;                // mft6.py:1191
}
__device__ __forceinline__ double value_of_total(int mode, double total, double lp) {
    if (mode == MSX_MODE_CHISQ) return total;                      // mft6.py:1198-1199
    return isnan(total) ? -INFINITY : lp + (-0.5 * total);          // mft6.py:1202-1205, 1470
}

// Last lines of a walker (one lane): publish the value and, for the device-resident sampler, apply the
// stretch move's accept rule  log(u) < (ndim-1) ln z + ln p(q) - ln p(s)  (NaN differences compare false,
// like -inf - -inf on the host) and record the walker's row of the chain: a walker only changes in its
// own half-step, so its row after the step is written here.
// What the walker's last lines read of the problem, copied out of the by-value kernel argument ONCE per kernel (smp_view):
// walker_done is inlined at every exit of the kernel, and a by-value DevProblem with too many uses is no longer
// recognised as read-only by the compiler, which then keeps a private copy of all 1.2 KB of it in scratch
// (tests/test_abi.py watches the variants' scratch size).
struct SmpView {
    unsigned long long *clk_probe;  // non-null: a probe launch (see DevProblem)
    int32_t smp_on, smp_defer, smp_overlap, linked_fault;
    int64_t smp_stride;
    double *smp_coords, *smp_logp, *smp_chain_row, *smp_lp_row;
    int64_t *smp_naccept;
    int32_t *smp_worst;
    unsigned long long *smp_gran;
    int64_t smp_gwalkers;
};
__device__ __forceinline__ SmpView smp_view(const DevProblem &P, bool probe) {
    return {probe ? P.clk_probe : nullptr, P.smp_on, P.smp_defer, P.smp_overlap, P.linked_fault, P.smp_stride, P.smp_coords, P.smp_logp, P.smp_chain_row,
            P.smp_lp_row, P.smp_naccept, P.smp_worst, P.smp_gran, P.smp_gwalkers};
}
__device__ __forceinline__ void walker_done(const SmpView &P, const WalkerDesc &D, int64_t wk, int ndim, double out, int st,
                            double *__restrict__ logp, int32_t *__restrict__ status) {
    // an error status travels inside the NaN it produces (payload = MSX_W_*): the sharded sampler's all-gather
    // carries log-probabilities only, and every rank must learn of every rank's failures
    logp[wk] = (st > MSX_W_REJECT) ? nan_with_status(st) : out;
    status[wk] = st;
    if (P.clk_probe && wk < kProbeWalkers) {
        P.clk_probe[wk * 4 + 2] = wall_clock64();
        P.clk_probe[wk * 4 + 3] = (unsigned long long)__builtin_readcyclecounter();
    }
    if (!P.smp_on) return;
    if (st > MSX_W_REJECT) atomicMax(P.smp_worst, st);
    if (P.smp_defer) return;  // sharded: sampler_apply_kernel finishes the move after the all-gather
    const int64_t s = D.smp_s;
    const double lnpdiff = (D.smp_zfac + out) - D.smp_old;
    const bool acc = D.smp_logu < lnpdiff;
    if (P.smp_overlap) {
        // Overlapped half-steps: the next half-step's workgroups are already resident and poll for THIS walker's next
        // version.  What they read goes out as tagged granules (DevProblem::smp_gran): every word {32 bits | new version}
        // by one agent-scope store (through the L2, like the linked form's partials), into the buffer of the new version's
        // parity -- accepted or not: the other buffer still holds what workgroups of the half-steps in flight may be
        // reading.  No wait for the stores' acknowledgements and no flag behind them: a word that shows the version IS the
        // data.  (Rounds 1-3: data, vmcnt(0), then a version word; the reader polled the word, acquired and fetched the
        // data -- two more trips through the fabric per hand-over.)
        const unsigned int nv = D.smp_ver + 1u;
        const double newlp = acc ? out : D.smp_old;
        const long long nacc = D.smp_nacc + (acc ? 1 : 0);
        // (test hook: nobody publishes, so every wait of the following half-steps runs into its bound)
        if (!P.linked_fault) {
            unsigned long long *g = P.smp_gran + ((int64_t)(nv & 1u) * P.smp_gwalkers + s) * kGranPerWalker;
            for (int d = 0; d < ndim; ++d) {
                const unsigned long long b = (unsigned long long)__double_as_longlong(acc ? D.theta[d] : D.smp_sv[d]);
                publish_u64(g + 2 * d, granule((unsigned int)(b >> 32), nv));
                publish_u64(g + 2 * d + 1, granule((unsigned int)b, nv));
            }
            const unsigned long long lb = (unsigned long long)__double_as_longlong(newlp);
            publish_u64(g + kGranLogp, granule((unsigned int)(lb >> 32), nv));
            publish_u64(g + kGranLogp + 1, granule((unsigned int)lb, nv));
            publish_u64(g + kGranNacc, granule((unsigned int)nacc, nv));
        }
        // ... and the plain arrays the host reads when the chunk is over (nobody on the device waits for these)
        double *row = P.smp_coords + (int64_t)(nv & 1u) * P.smp_stride + s * ndim;
        for (int d = 0; d < ndim; ++d) {
            const double v = acc ? D.theta[d] : D.smp_sv[d];
            row[d] = v;
            P.smp_chain_row[s * ndim + d] = v;
        }
        P.smp_logp[s] = newlp;
        P.smp_naccept[s] = nacc;
        P.smp_lp_row[s] = newlp;
        return;
    }
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

// The chi^2 terms that ride along the median's pass over the model vector (see phase B in the kernel): u, data flux
// and 1/err^2 by table ELEMENT (two pixels 256 apart) -- the four pixels of a trip are the elements (base >> 1) + tid
// and + MAXT (pass_pixel).  u and flux come from LDS with PF, else from the tables; 1/err^2 always from its table.
// AHEAD = global loads run one trip ahead of their use, into the register set of the other parity (median.h,
//         pass_trips): the one-workgroup-per-CU variants, which have the registers and few waves to hide a load
//         behind.  Otherwise a trip's loads are issued at its start, before the trip's LDS reads.
// ALWAYS = the terms are wanted whatever `on` says (the early-histogram median only runs in the fused modes): no
//         run-time flag inside the pass, whose merge points would bring register copies back.
// Holds plain pointers, never a reference to the by-value kernel argument (see DevProblem).
// FLDS = the data flux (alone) comes from LDS: the linked form, whose workgroup has room for one more vector of its segment.
// FULL = every element / pixel of every trip is valid (the spectrum is whole trips: logprob_kernel's FULL): no clamps, no selects
template <int MAXT, bool PF, bool ALWAYS, bool AHEAD, bool FLDS = false, bool FULL = false>
struct ChiElem {
    static constexpr int VK = kMaxWaves / (MAXT / kWave);
    static constexpr int NSET = AHEAD ? 2 : 1;
    const double2 *u2, *f2, *iv2;
    int ne, npix;
    double c0, c1, c2;
    double acc[VK];  // one per slot this lane holds (see phase A and pass_pixel)
    bool on;
    double *red0;    // [MAXT] LDS: the lanes' partials of the chi^2 sum
    double2 nu[NSET][2], nf[NSET][2], nv[NSET][2];  // [register set][element of the trip]
    double tot_run;  // wave 0: the chi^2 sum over the segments finished so far (see phase A)
    template <int SET>
    __device__ __forceinline__ void load_trip(int base) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            int e = (base >> 1) + j * MAXT + (int)threadIdx.x;
            e = (FULL || e < ne) ? e : ne - 1;
            const unsigned int o16 = (unsigned int)e << 4;
            nv[SET][j] = ld_off(iv2, o16);
            if (!PF) { nu[SET][j] = ld_off(u2, o16); if (!FLDS) nf[SET][j] = ld_off(f2, o16); }
        }
    }
    __device__ __forceinline__ void prime_from(int base) {  // before a pass that starts at pixel `base`
        if (!ALWAYS && !on) return;
        if (AHEAD) load_trip<0>(base);
    }
    __device__ __forceinline__ void prime() { prime_from(0); }  // before the pass (and before whatever the caller does first)
    template <int PAR>
    __device__ __forceinline__ void begin_trip(int base) {
        if (!ALWAYS && !on) return;
        if (!AHEAD) load_trip<0>(base);
    }
    // the four pixels of one trip (pass_pixel order)
    template <int PAR>
    __device__ __forceinline__ void process4(int base, const int (&p)[4], const double (&xv)[4]) {
        if (!ALWAYS && !on) return;
        constexpr int SET = AHEAD ? PAR : 0;
        double2 cu[2], cf[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if (PF) {
                int e = (base >> 1) + j * MAXT + (int)threadIdx.x;
                e = (FULL || e < ne) ? e : ne - 1;
                cu[j] = u2[e]; cf[j] = f2[e];  // LDS
            } else {
                cu[j] = nu[SET][j]; cf[j] = nf[SET][j];
                if (FLDS) {
                    int e = (base >> 1) + j * MAXT + (int)threadIdx.x;
                    e = (FULL || e < ne) ? e : ne - 1;
                    cf[j] = f2[e];  // LDS (f2 is the workgroup's staged copy, indexed by the element's own number)
                }
            }
        }
        // (FULL has no clamp inside load_trip: the trip requested behind the LAST one would lie past the tables' end --
        // it asks for the current trip again instead; uniform, two scalar instructions)
        if (AHEAD) load_trip<AHEAD ? 1 - PAR : 0>((FULL && base + 4 * MAXT >= npix) ? base : base + 4 * MAXT);
        const double u[4] = {cu[0].x, cu[0].y, cu[1].x, cu[1].y}, f[4] = {cf[0].x, cf[0].y, cf[1].x, cf[1].y};
        const double e[4] = {nv[SET][0].x, nv[SET][0].y, nv[SET][1].x, nv[SET][1].y};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            acc[k & (VK - 1)] += chi_term(c0, c1, c2, u[k], f[k], e[k], xv[k], FULL || p[k] < npix);  // up to scale^2
        }
        // end of a segment of the canonical sum (8192 pixels), more pixels to come: fold it in.  (Uniform: every
        // thread of the workgroup walks the same trips.)
        const int next = base + 4 * MAXT;
        if ((next & (2 * kSegElems - 1)) == 0 && next < npix) {
            red0[threadIdx.x] = lane_partial<VK>(acc);
#pragma unroll
            for (int k = 0; k < VK; ++k) acc[k] = 0.0;
            __syncthreads();
            if ((threadIdx.x >> 6) == 0) tot_run += reduce_published<MAXT>(red0, (int)threadIdx.x & 63);
            __syncthreads();
        }
    }
    __device__ __forceinline__ void flush(BlockScratch &) {  // one partial per lane; wave 0 finishes at the very end
        if (!ALWAYS && !on) return;
        red0[threadIdx.x] = lane_partial<VK>(acc);
    }
};

// ------------------------------------------------------------------------------------------------
// THE HOT KERNEL: one workgroup per walker.
//   phase 0    one wave per star builds the walker's recipe from register-resident tables (prior gate, A1, A2,
//              A4); an idle wave computes the Gaussian prior terms (f1); with PF the remaining waves stage
//              pixel statics in LDS
//   phase A    blend + redden + resample into the model vector (LDS); fit sums, value range and the median's
//              logarithmic histogram are accumulated on the way                       (A2, A4, A7, A8.1)
//              wave 0 computes the contrast / photometry terms (A5/A6) while it waits at phase A's barrier
//   phase B+C  exact median (per-wave bin scan -> ONE pass that also carries the chi^2 terms and gathers the
//              median bin's candidates -> wave-0 rank); scale, chi^2 and the combine   (A8.2, A8.3, A9)
//              vectors the early histogram cannot handle take block_median (linear bins, radix fallback); the
//              pre-optimiser modes keep a separate chi^2 pass (phase C proper)
// ------------------------------------------------------------------------------------------------
extern __shared__ __attribute__((aligned(16))) unsigned char dyn_lds[];

// MAXT = the workgroup size the variant is compiled for and launched with: 256 (three workgroups per CU, capped at
//        168 VGPRs) or 512 (one per CU; with SH: two per CU, capped at 128 VGPRs).  U is the pixels per lane and trip:
//        always 2, the two pixels of one table element (below).
// GM = the walker's model vector lives in global memory (spectra longer than ~17k pixels) instead of LDS.
// SH = 512-thread variant that shares its CU with a second workgroup (MSX_BLOCK_512_SHARED).
// PF = the walker-independent pixel vectors u and data flux are staged in LDS, in the tables' own layout, by the waves
//      that idle during the recipe; the blend loop and the chi^2 pass read them there instead of pulling them through
//      the CU's L2 port, twice (one workgroup per CU only: 3 npix doubles of LDS).
// LK = the LINKED form of the same kernel, for few walkers x long spectra (one workgroup per walker leaves CUs idle
//      and is a chain of 16,384 pixels' latencies): one workgroup per (walker, SEGMENT of 8192 pixels), all of them
//      equals.  Each builds the recipe, blends ITS segment into LDS (fit sums, value range, histogram on the way) and
//      leaves those partials in P.segparts; the walker's workgroups MEET (an arrival counter per walker, agent-scope
//      release / acquire; the wait is bounded by kHandoverTicks, then the walker fails with MSX_W_HANDOVER and the
//      context's linked form is POISONED: see below); every one of them then adds up the partials in segment order --
//      exactly the fused kernel's association -- locates the median's bin from the summed histogram and makes the
//      chi^2 / candidates pass over its own segment; chi^2 sum and candidates go to P.segparts again, and whichever
//      workgroup ARRIVES LAST at the second meeting point (nobody waits there) ranks the candidates of all segments,
//      adds the chi^2 sums in segment order and finishes the walker.  The model vector never leaves the CUs; a
//      hand-over is 8 KB of counters and a few numbers.  (Vectors the early histogram cannot handle: every segment
//      goes to the scratch row and the last arrival runs block_median over it, as the GM variants do.)
//      Block = (walker / 8) * 8 S + segment * 8 + walker % 8: a walker's workgroups are 8 blocks apart -- blocks go
//      round-robin over the 8 XCDs, so they share an XCD (one L2: the hand-over's data never leaves it) and follow
//      each other in that XCD's dispatch queue: the workgroups of every walker dispatched earlier are resident or
//      done, so the wait needs no co-residency guarantee beyond in-order dispatch.
//      The counter is never reset: a launch adds 2 S to it, and a workgroup reads the launch's base off the value its
//      own first increment returns (old - old % 2S).  (64 bits: it does not wrap.)
//
// TABLE LAYOUT.  A CU pulls data from L2 at ~32 B per clock when every lane loads 16 bytes, and no faster per
// instruction when lanes load less -- so every per-pixel table the blend reads is stored in ELEMENTS of two pixels,
// element e = pixels {pa, pa + 256}, pa = (e >> 8) * 512 + (e & 255): a lane's 16-byte load (8-byte for the float32
// table) brings both of its pixels, for workgroups of 256 and of 512 threads alike.  Per grid node and pixel the
// tables hold R = lo + (hi - lo) t (float64) and H = hi t (float32): see blend_pixel_rh (blend.h).
// R32 = the R table is stored in FLOAT32 (msx_set_grid_storage(MSX_STORE_F32): 8 instead of 12 bytes per node-pixel through
//       the CU's L2 port, whose limit the blend runs at) and widened to float64 in the registers: the arithmetic is the
//       same float64 chain, the grid values carry 2^-24 instead of 2^-53.  A SEPARATELY LABELLED precision (SURVEY 8b's
//       store_dtype), never the default; fused binaries only.
// FULL = the spectrum is whole trips of the variant with no pad pixels (npix == 2 npair, npair a multiple of the trip:
//        BASELINE's 4096 pixels): every lane of every trip holds live pixels, so the clamps of element and pixel indices,
//        the per-pixel validity compares and their selects can be compiled out -- bit 0: of the blend, bit 1: of the chi^2
//        pass and the candidates' gather.  Same arithmetic on the same pixels: same bits (the launcher picks it; msx.hip,
//        choose_variant).  Which bits pay was measured per workgroup size (same box, alternating runs): the 256-thread
//        variants gain from both (2,048 walkers 63.3 -> 60.4 us), the 512-thread headline variant gains from the chi^2
//        pass's (14.63 -> 14.36 us per step) and LOSES with the blend's (14.67 -> 14.96: the loads' order changed); the
//        two-per-CU and the linked 512-thread variants likewise (config 4's share 26.2 -> 25.8 us with bit 1, 26.4 with both).
// GIVEN = the model values are not blended here: the in-path broadening kernels (inpath_kernels.h) have left them in
//        P.given[walker][pixel]; everything else -- recipe (for the walker's status, its prior and band terms), fit sums,
//        median, chi^2 pass -- is this kernel's.  One variant: 512 threads, quad trips.
template <int NS, int U, int MAXT, bool GM = false, bool SH = false, bool PF = false, bool LK = false, bool R32 = false, int FULL = 0,
          bool GIVEN = false>
// (second launch bound = waves per SIMD the register allocation must leave room for: k workgroups of T threads per
// CU <=> k T / 256.  256 threads: three per CU = 168 VGPRs; 512 threads sharing a CU: two per CU = four waves per
// SIMD = 128 VGPRs.)
__global__ void __launch_bounds__(MAXT, MAXT == 256 ? (SH ? 2 : 3) : (MAXT == 512 && SH) ? 4 : 1)
logprob_kernel(const double *theta, const unsigned char *__restrict__ rblk, int niso_nt, int ng_mode_fast, int64_t n,
               double gate_tmin, double gate_tmax, const SmpRec *__restrict__ smp_rec, DevProblem P,
               double *__restrict__ logp, int32_t *__restrict__ status) {
    // The leading arguments (14 dwords: all the preload takes) are compiled for KERNARG PRELOAD (-mllvm
    // -amdgpu-kernarg-preload-count): the command processor delivers them in SGPRs at wave start, so theta and the
    // recipe's small tables are requested in the first instructions, while the 1.2 KB DevProblem (fetched from the
    // kernel-argument segment like any argument: a memory round trip, then a scalar-cache access per field) is
    // still on its way.  Everything the walker's critical chain needs up to its weights is among them:
    //   rblk            the recipe's tables in one block (dev_types.h: isochrone Teff / logg, the grid's node lists and
    //                   per-Teff-node presence bits at fixed offsets; from P.iso_t, P.iso_g, P.teff_nodes, ...)
    //   niso_nt         niso | nt << 16
    //   ng_mode_fast    ng | mode << 8 | fast << 16 | sampler << 17 | dist_fit << 18 | use_av << 19 | overlap << 20 | probe << 21 | segments << 24
    //   n               the batch size (ndim is 2 NS + 2, checked by the host)
    //   gate_tmin/tmax  the Teff box of the prior's hard gates (= P.tmin, P.tmax)
    //   theta, smp_rec  device-resident sampler: `theta` is the resident ensemble (= P.smp_coords) and smp_rec the
    //                   half-step's records (= P.smp_rec): the proposal is two dependent loads away from wave start
    const GateArgs gates = {gate_tmin, gate_tmax, ((ng_mode_fast >> 18) & 1) != 0, ((ng_mode_fast >> 19) & 1) != 0};
    constexpr int ndim = 2 * NS + 2;
    const bool probe = (ng_mode_fast >> 21) & 1;  // msx_probe_launch: clock stamps at the walker's first and last line
    const SmpView V = smp_view(P, probe);  // (for walker_done)
    __shared__ WalkerDesc D;
    __shared__ BlockScratch S;
    __shared__ double red[3][MAXT / kWave][kWave];  // one partial per lane and quantity (wave_ops.h, canonical sum)
    __shared__ double e2tab[kExp2Tab];              // 2^(j/32) for the reddening factor (blend.h)
    const int niso = niso_nt & 0xffff, nt = niso_nt >> 16;
    const int ng = ng_mode_fast & 0xff, mode = (ng_mode_fast >> 8) & 0xff;
    const bool fast = (ng_mode_fast >> 16) & 1;  // register-resident tables fit one wave (the usual case)
    const bool smp_on = (ng_mode_fast >> 17) & 1;  // device-resident sampler: theta is a proposal built here (= P.smp_on)
    const bool overlap = (ng_mode_fast >> 20) & 1;  // ... with overlapped half-steps (= P.smp_overlap)
    const int nsegs = LK ? (ng_mode_fast >> 24) & 0xff : 1;  // linked: workgroups per walker
    const unsigned int lk_grp = LK ? blockIdx.x / (8u * (unsigned int)nsegs) : 0u, lk_r = LK ? blockIdx.x % (8u * (unsigned int)nsegs) : 0u;
    const int64_t wk = LK ? (int64_t)lk_grp * 8 + (lk_r & 7u) : blockIdx.x;
    const int myseg = LK ? (int)(lk_r >> 3) : 0;
#ifdef MSX_STAMPS
    if (threadIdx.x == 0) msx_stamp_off = LK && myseg != ((MSX_STAMPS == 2) ? 0 : nsegs - 1);  // (one workgroup's stamps per walker)
#endif
    // theta FIRST: the kernel's first vector load, requested before the recipe's tables (whose consumers -- the uniform
    // first / last entries below -- wait for them): the walker's critical chain starts when theta arrives, and a load
    // issued behind those waits would only leave then (rounds 1-3 did that: ~0.9 k cycles of the chain).  Lane k takes
    // coordinate k (one VECTOR load: a scalar load would share its counter with the kernel-argument fetches below and be
    // waited for together with them).  The sampler builds its proposal below instead -- from its record, requested here.
    double theta_lane = 0.0;
    if (fast && !smp_on && (threadIdx.x >> 6) < NS && (threadIdx.x & 63) < 2 * NS + 2 && wk < n)
        theta_lane = theta[wk * (2 * NS + 2) + (threadIdx.x & 63)];
    SmpRec rc = {0, 0, 0.0, 0u, 0u};
    if (smp_on && wk < n) rc = smp_rec[wk];
    RecipeRegs RR;
    if (fast && (threadIdx.x >> 6) < NS) load_recipe_regs(RR, rblk, niso, nt, ng, threadIdx.x & 63);
    if (wk >= n) return;
    if (probe && threadIdx.x == 0 && wk < kProbeWalkers && (!LK || myseg == 0)) {
        P.clk_probe[wk * 4 + 0] = wall_clock64();
        P.clk_probe[wk * 4 + 1] = (unsigned long long)__builtin_readcyclecounter();
    }
    if (LK) {
        // a poisoned context (an earlier launch's meeting timed out, see below): no counter is trusted,
        // every walker of every linked launch fails loudly until the problem is staged again
        if (__hip_atomic_load(P.linked_poison, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
            if (myseg == 0 && threadIdx.x == 0) {
                logp[wk] = nan_with_status(MSX_W_HANDOVER);
                status[wk] = MSX_W_HANDOVER;
                if (P.smp_on) atomicMax(P.smp_worst, MSX_W_HANDOVER);
            }
            return;
        }
    }
    // [npix]; linked: LDS holds this workgroup's segment only, indexed by the pixel's own number all the same
    double *model = GM ? P.model_scratch + wk * P.npix : reinterpret_cast<double *>(dyn_lds) - (LK ? myseg * (2 * kSegElems) : 0);
    const int tid = threadIdx.x;
    constexpr int B = MAXT;  // every variant is launched with exactly MAXT threads (msx_logprob_batch_dev)
    const int lane = tid & 63, wave = tid >> 6;
    constexpr int nw = B >> 6;
    const int npix = (int)P.npix;
    static_assert(U == 2 && (MAXT == 256 || MAXT == 512), "one two-pixel element per lane and trip; 256 or 512 threads");
    const int ne = (int)P.npair;  // table elements (pixel pairs), a multiple of 256
    // PF: u and data flux in LDS behind the model vector, in the tables' own pair layout (16-byte aligned)
    double2 *const lds_u2 = PF ? reinterpret_cast<double2 *>(reinterpret_cast<double *>(dyn_lds) + ((npix + 1) & ~1)) : nullptr;
    double2 *const lds_f2 = PF ? lds_u2 + ne : nullptr;
    // linked: the data flux of this workgroup's segment behind its model values, indexed by the element's own number
    double2 *const lds_lf2 = LK ? reinterpret_cast<double2 *>(reinterpret_cast<double *>(dyn_lds) + 2 * kSegElems) - myseg * kSegElems : nullptr;

    MED_WALL(5);
    MSX_STAMP(P, wk, 0);
    MSX_STAMP(P, wk, 8);
    const double *th_row = theta + wk * ndim;
    // Early-histogram path (logbin_median): the median's histogram is filled while phase A computes the model,
    // and the walker's prior terms move off phase 0.  Likelihood / posterior / chi^2 modes with the register-resident
    // recipe and the model vector in LDS; everything else keeps block_median.
    const bool early = !GM && fast && !P.no_spectrum &&
                       (mode == MSX_MODE_LOGLIKE || mode == MSX_MODE_LOGPOST || mode == MSX_MODE_CHISQ);
    // The prior terms (f1) depend on theta alone and only the walker's last lines read them: an idle wave computes
    // them beside the recipe waves -- or, where phase A follows (`early` modes of the blending stages), a wave that
    // idles while wave 0 ranks the median's candidates (rejected walkers never read them).
    const bool prior_late = early;
    if (smp_on) {  // stretch-move proposal q = c - (c - s) z for this walker (mft6.py:1494 drives emcee's move)
        // two dependent levels from wave start: the walker's record {own index, partner's index, z} (a preloaded
        // pointer) -> the two coordinate rows of the resident ensemble (the `theta` argument).  Every recipe wave
        // forms the proposal itself, lane k coordinate k, straight into the register the recipe reads: no LDS round
        // trip, no barrier.  Wave 0 also leaves it in LDS for the phases after phase 0; a non-recipe wave meanwhile
        // fetches what the accept step will need at the very end.
        // Overlapped half-steps: this workgroup may have been dispatched while the half-step(s) before it are still
        // running.  Every wave that reads the ensemble -- the recipe waves (both rows) and the wave that fetches the
        // accept step's inputs (the walker's own entries) -- first waits until the walkers it reads have reached the
        // versions the move is defined on (bounded: then the chunk reports MSX_W_HANDOVER), acquires, and takes version
        // v of a walker from coordinate buffer v & 1.
        if (overlap) {
            // TAGGED GRANULES (DevProblem::smp_gran): lane l of a recipe wave watches word l of the walker's own record
            // (l < 2 ndim: the two halves of coordinate l / 2) or of its partner's (the next 2 ndim lanes); the wave that
            // fetches the accept step's inputs watches the walker's log-probability and acceptance count.  A lane polls its
            // word until it carries the version the move is defined on -- version v of a walker lives in buffer v & 1 --
            // and then HAS the data: the words are put together with readlanes, no second load.
            if (wave <= NS) {
                constexpr int NG = 2 * ndim;
                const bool rw = wave < NS;
                const bool mine = rw ? lane < 2 * NG : lane < 3;
                const bool par = rw && lane >= NG;
                const int gi = rw ? (par ? lane - NG : lane) : (kGranLogp + lane);
                const unsigned int want = par ? rc.ver_partner : rc.ver_own;
                const unsigned long long *gp = P.smp_gran + ((int64_t)(want & 1u) * P.smp_gwalkers + (par ? rc.ci : rc.si)) * kGranPerWalker + (mine ? gi : 0);
                unsigned long long g = 0ull;
                const unsigned long long t0 = wall_clock64();
                for (;;) {
                    if (mine) g = __hip_atomic_load(gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    // (versions only grow, and a walker is never more than one version ahead of what this move wants: the
                    // buffer of that parity holds this version or the one two before it)
                    if (__ballot(!mine || (unsigned int)g == want) == ~0ull) break;
                    if (wall_clock64() - t0 > kHandoverTicks) {
                        if (lane == 0) atomicMax(P.smp_worst, MSX_W_HANDOVER);
                        break;
                    }
                }
                const int piece = (int)(unsigned int)(g >> 32);
                if (rw) {
#pragma clang fp contract(off)
                    // no FMA contraction: the proposal must have the bits NumPy's `c - (c - s) * z` produces so that
                    // the device-resident and the host-driven sampler stay in lock-step.  Lane k < ndim puts coordinate k
                    // together from the four lanes that hold its halves (one cross-lane permute each) and is then exactly
                    // where the plain load of the other paths leaves it.
                    const int s_hi = __shfl(piece, 2 * lane), s_lo = __shfl(piece, 2 * lane + 1);
                    const int c_hi = __shfl(piece, NG + 2 * lane), c_lo = __shfl(piece, NG + 2 * lane + 1);
                    if (lane < ndim) {
                        const double sv = __hiloint2double(s_hi, s_lo), cv = __hiloint2double(c_hi, c_lo);
                        const double diff = cv - sv;
                        const double prod = diff * rc.zz;
                        const double qv = cv - prod;
                        theta_lane = qv;
                        if (wave == 0) {
                            D.theta[lane] = qv;
                            D.smp_sv[lane] = sv;
                            P.smp_q[wk * ndim + lane] = qv;  // (kept for inspection; nothing reads it back)
                        }
                    }
                } else {
                    const double old = __hiloint2double(__builtin_amdgcn_readlane(piece, 0), __builtin_amdgcn_readlane(piece, 1));
                    const unsigned int na = (unsigned int)__builtin_amdgcn_readlane(piece, 2);
                    if (lane == 0) {
                        D.smp_s = rc.si;
                        D.smp_ver = rc.ver_own;
                        D.smp_old = old;
                        D.smp_nacc = (int64_t)na;   // (the low 32 bits travel; the host's 64-bit count is the plain array's)
                        D.smp_zfac = P.smp_zfac[wk];
                        D.smp_logu = P.smp_logu[wk];
                    }
                }
            }
        } else if (wave < NS && lane < ndim) {
#pragma clang fp contract(off)
            // no FMA contraction: the proposal must have the bits NumPy's `c - (c - s) * z` produces so that
            // the device-resident and the host-driven sampler stay in lock-step
            const double sv = theta[(int64_t)rc.si * ndim + lane];
            const double cv = theta[(int64_t)rc.ci * ndim + lane];
            const double diff = cv - sv;
            const double prod = diff * rc.zz;
            const double qv = cv - prod;
            theta_lane = qv;
            if (wave == 0) {
                D.theta[lane] = qv;
                D.smp_sv[lane] = sv;
                P.smp_q[wk * ndim + lane] = qv;  // (kept for inspection; nothing reads it back)
            }
        } else if (tid == NS * kWave) {
            const int64_t si = rc.si;
            D.smp_s = si;
            D.smp_ver = rc.ver_own;
            D.smp_old = P.smp_logp[si];
            D.smp_nacc = P.smp_naccept[si];
            D.smp_zfac = P.smp_zfac[wk];
            D.smp_logu = P.smp_logu[wk];
        }
        // (readers of D.theta before phase 0's barrier: the prior terms' wave where they are not late, and the
        // general recipe -- which runs in wave 0, the writer)
        if (!prior_late) __syncthreads();
        th_row = D.theta;
    }
    for (int i = tid; i < kLogBins; i += B) S.hist[i] = 0;
    if (tid == 0) { S.cand_n = 0; S.has_second = 0; }
    if (tid < 2 * kWave) (&S.rk[0][0])[tid] = 0u;
    fill_exp2_table(e2tab, tid - (B - kWave));  // the last wave (no recipe work); published by phase 0's barrier
    if (PF && wave > NS) {
        // The waves with no recipe work bring the walker-independent pixel vectors the blend loop and the chi^2 pass
        // read -- u and the data flux -- into LDS while the recipe waves work (64 KB through the CU's L2 port in the
        // recipe's 1.9 us; published by phase 0's barrier): the blend loop, which runs at that port's limit, then
        // requests 116 instead of 132 bytes per pixel and has four loads fewer per trip to wait for.
        // (256 walkers x 4096 px: 15.3 -> 13.7 us on the same box.  Round 1 staged them like this, round 2 let the
        // blend loop leave them in LDS "since it loads them anyway" -- it does not have to.  The extinction curve k too,
        // 16 more bytes per element: no further gain, 13.7 us; 1/err^2 for the chi^2 pass likewise.  The staging is hidden
        // entirely: a build that skips it is not faster.)
        const int nthr = B - (NS + 1) * kWave, id = tid - (NS + 1) * kWave;
#pragma unroll 4
        for (int e = id; e < ne; e += nthr) {
            lds_u2[e] = P.u2[e];
            lds_f2[e] = P.f2[e];
        }
    }
    if (LK && wave > NS) {  // linked: the same for the data flux of this workgroup's segment (the LDS has room for one vector)
        const int nthr = B - (NS + 1) * kWave, id = tid - (NS + 1) * kWave;
        const int e_hi = (myseg + 1) * kSegElems < ne ? (myseg + 1) * kSegElems : ne;
#pragma unroll 4
        for (int e = myseg * kSegElems + id; e < e_hi; e += nthr) lds_lf2[e] = P.f2[e];
    }
    constexpr int NC = NS * 4;
    const int nseg_all = (ne + kSegElems - 1) / kSegElems;
    const int seg_lo = LK ? myseg : 0, seg_hi = LK ? myseg + 1 : nseg_all;
    if (fast && wave == NS && !prior_late) recipe_prior_terms<NS>(P, mode, th_row, D, lane);
    if (fast) {
        if (wave < NS) {
            double tv[ndim];
#pragma unroll
            for (int k = 0; k < ndim; ++k) tv[k] = readlane_f64(theta_lane, k);
            recipe_part1_regs<NS>(P, gates, RR, niso, nt, ng, mode, theta_lane, tv, D, lane, wk, wave);
        }
    } else if (wave == 0) {
        const RecipeTabs T = {P.iso_t, P.iso_g, P.iso_l, P.av_edges, P.av_mu, P.av_sig, P.teff_nodes, P.logg_nodes};
        build_recipe_wave<NS>(P, T, mode, th_row, ndim, D, lane, wk);
    }
    __syncthreads();
    int wst = D.status;
    if (fast) {  // first star that failed decides, like the reference's star-by-star loop ...
        wst = D.stat[0];
#pragma unroll
        for (int k = 1; k < NS; ++k) wst = (wst == MSX_W_OK) ? D.stat[k] : wst;
        // ... except that every star's logg is interpolated before the first star's spectrum is built (mft6.py:1149):
        // a Teff outside the isochrone on a later star raises before an earlier star's bracket can
#pragma unroll
        for (int k = 1; k < NS; ++k) wst = (wst != MSX_W_REJECT && D.stat[k] == MSX_W_VALUEERROR) ? MSX_W_VALUEERROR : wst;
    }
    if (wst != MSX_W_OK) {
        if (tid == 0 && myseg == 0) walker_done(V, D, wk, ndim, (wst == MSX_W_REJECT) ? -INFINITY : NAN, wst, logp, status);
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
            walker_done(V, D, wk, ndim, chi_valued ? total : (isnan(total) ? -INFINITY : D.lp + (-0.5 * total)), MSX_W_OK, logp,
                        status);
        }
        return;
    }
    MSX_STAMP(P, wk, 1);
    // What only the walker's last line reads -- the contrast / photometry terms (A5/A6) and the Gaussian prior terms (f1)
    // -- starts HERE, in waves 1 and 2, before they join the pixel loop: the band jobs' magnitudes (a round trip to the
    // band table, then logarithms) are kept in a register until the last lines, the prior terms are final.  At the
    // end, where the median's candidates are ranked, they were the longest chain left (with the ranking split over the
    // idle waves: 128 walkers 13.2 -> 12.5 us, 256 walkers 13.7 -> 13.4; either change alone gains nothing).
    double side_val = 0.0;
    if (early && !LK) {
        if (wave == 1) side_val = recipe_band_values<NS>(P, D, lane);
        if (wave == 2 && prior_late) recipe_prior_terms<NS>(P, mode, th_row, D, lane);
    }

    // ---- phase A ------------------------------------------------------------------------------------
    // kQuad: the 512-thread fused variants walk the tables a QUAD (two elements, four pixels) per lane and trip and
    // take the float32 values from the quad tables: one 16-byte load where two elements need two 8-byte ones
    // (512 threads: not the <= 128-VGPR variant, which has no room for a quad's rows; 256 threads: only the variant
    // that runs two per CU instead of three -- SH there -- and so has 256 VGPRs)
    constexpr bool kQuad = (MAXT == 512 && !SH) || (MAXT == 256 && SH);
    constexpr bool FULLB = (FULL & 1) != 0, FULLC = (FULL & 2) != 0;  // whole trips: no clamps in the blend / in the chi^2 pass
    const double2 *rows_r[NC];  // R = lo + (hi - lo) t of each corner's grid node, two pixels per element
    const float2 *rows_rf[NC];  // ... the float32 copy (R32)
    const float4 *rows_r4f[NC]; // ... by quad
    const float2 *rows_h[NC];   // H = hi t
    const float4 *rows_h4[NC];  // ... by quad
    double w[NC];
    float wf[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int64_t off = (int64_t)__builtin_amdgcn_readfirstlane(D.node[c]) * ne;
        rows_r[c] = P.r2 + off;
        rows_rf[c] = R32 ? P.r2f + off : nullptr;
        rows_r4f[c] = (R32 && kQuad) ? (MAXT == 512 ? P.r4f : P.r4fb) + (int64_t)__builtin_amdgcn_readfirstlane(D.node[c]) * P.nquad : nullptr;
        rows_h[c] = P.h2 + off;
        rows_h4[c] = (MAXT == 512 ? P.h4 : P.h4b) + (!kQuad ? 0 : (int64_t)__builtin_amdgcn_readfirstlane(D.node[c]) * P.nquad);
        w[c] = uniform_f64(D.w[c]);
        wf[c] = uniform_f32((float)w[c]);
    }
    const double redc = uniform_f64(D.redc);
    const bool redden = redc != 0.0;
    // Sums are taken in an order that does not depend on the workgroup size.  Pixel p belongs to SLOT p mod 1024;
    // a slot accumulates its pixels in ascending order in ONE lane's register, the 64 slots of VIRTUAL wave v
    // (slots 64 v .. 64 v + 63, lane = slot mod 64) are reduced by one DPP tree and the 16 partials are added
    // serially.  Which real lane holds which slot differs with the workgroup size, the association does not:
    //   256 threads: lane tid walks elements tid + 256 j -> pixels (j & 1) * 512 + tid (+ 256) mod 1024: four slots,
    //                accumulator k = 2 (j & 1) + u, virtual wave 4 k + wave
    //   512 threads: elements tid + 512 j -> pixels (tid >> 8) * 512 + (tid & 255) (+ 256) mod 1024: two slots,
    //                accumulator k = u, virtual wave (wave >> 2) * 8 + 4 u + (wave & 3)
    // so a walker's log-probability has the same bits whatever launch (batch size, shard, rank) evaluates it.
    // Spectra longer than 8192 pixels are summed SEGMENT by segment (kSegElems elements): each segment's slots are
    // reduced as above and the segments' sums added serially -- one more level of the same fixed association, and
    // what lets the linked form give each segment to a workgroup of its own.
    constexpr int vk = kMaxWaves / (MAXT / kWave);  // slots per lane: 4 or 2
    double q[3];
    // value range of the model vector, as the range of the unmasked histogram bin number F(m) = hi32(m) >> 12 (median.h,
    // frange_applicable); the exact float64 range is worked out only where block_median needs it
    unsigned int fmin_ = ~0u, fmax_ = 0u;
    constexpr int SUB = vk / U;  // elements per lane and outer trip: 2 (256 threads) or 1
    double qrun = 0.0;  // waves 0..2: their fit sum over the segments so far
    for (int seg = seg_lo; seg < seg_hi; ++seg) {
      double qa[vk][3];
#pragma unroll
      for (int k = 0; k < vk; ++k) qa[k][0] = qa[k][1] = qa[k][2] = 0.0;
      const int e_end = (seg + 1) * kSegElems < ne ? (seg + 1) * kSegElems : ne;
      // What follows a pixel pair's model values: the model vector, the fit sums, the value range, the histogram.
      auto finish_elem = [&](const double2 m2, const double2 f2, const double2 u2, const int ec, const bool live,
                             auto sub_c) __attribute__((always_inline)) {
        constexpr int sub = decltype(sub_c)::value;
        const int pa = ((ec >> 8) << 9) | (ec & 255), pb = pa + 256;
        const bool ok[U] = {FULLB || (live && pa < npix), FULLB || (live && pb < npix)};
        const int pp[U] = {(FULLB || pa < npix) ? pa : npix - 1, (FULLB || pb < npix) ? pb : npix - 1};
        static_assert(!PF || kQuad, "PF: u and the data flux come from LDS (staged in phase 0; the quad trips read them there)");
        const double mm[U] = {m2.x, m2.y}, ff[U] = {f2.x, f2.y}, uu[U] = {u2.x, u2.y};
        unsigned int fxs[U] = {0u, 0u};
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (ok[u]) {
                const double m = mm[u];
                model[pp[u]] = m;
                constexpr int slot = sub * U;  // (+ u: both unrolled)
#pragma unroll
                for (int k = 0; k < vk; ++k)
                    if (slot + u == k) fit_accumulate(m, ff[u], uu[u], qa[k][0], qa[k][1], qa[k][2]);
                const unsigned int fx = (unsigned int)__double2hiint(m) >> 12;
                fmin_ = fx < fmin_ ? fx : fmin_;
                fmax_ = fx > fmax_ ? fx : fmax_;
                fxs[u] = fx;
            }
        }
        if (early) hist_add_pair(S.hist, fxs[0], ok[0], fxs[1], ok[1]);  // (odd lanes in the other order: median.h)
      };
      // The trips of this segment, compiled twice: with the reddening terms (H rows, k, dk, the exp2) and without
      // (A_V <= 0: R rows only).  `redden` is uniform over the workgroup; as a run-time flag inside the loop it cost a
      // scalar branch and a zero-fill per H load.
      auto trips = [&](auto red_c) __attribute__((always_inline)) {
      constexpr bool RED = decltype(red_c)::value;
      if constexpr (kQuad) {
      for (int e0 = seg * kSegElems; e0 < e_end; e0 += 2 * B) {  // (segments are whole numbers of quad trips)
        const int eA = e0 + tid, eB = eA + B;
        const bool liveA = FULLB || eA < e_end, liveB = FULLB || eB < e_end;
        const int ecA = liveA ? eA : e_end - 1, ecB = liveB ? eB : e_end - 1;
        const unsigned int oA = (unsigned int)ecA << 4, oB = (unsigned int)ecB << 4;
        const unsigned int oq = (unsigned int)((e0 >> 1) + tid) << 4;  // quad (e0 / 1024) * 512 + tid, 16 bytes each
        constexpr int G = NC;  // corners per group of loads: all (the quad variants have the registers)
        // data flux and u: with the rows at 512 threads (26 walkers 15.35 -> 14.65 us, 4 x 32,768 px 35.9 -> 34.7),
        // after the blend at 256 (two workgroups per CU: 512 walkers 16.2 us early against 15.95 late)
        constexpr bool kEarlyFUq = MAXT == 512;
        double2 fA = make_double2(0.0, 0.0), uA = fA, fB = fA, uB = fA;
        if (kEarlyFUq && !PF) {
            uA = ld_off(P.u2, oA); uB = ld_off(P.u2, oB);
            if (!LK) { fA = ld_off(P.f2, oA); fB = ld_off(P.f2, oB); }
        }
        double sr[4] = {0.0, 0.0, 0.0, 0.0};
        float sh[4] = {0.f, 0.f, 0.f, 0.f};
        double2 klA = make_double2(0.0, 0.0), klB = klA;
        float4 dk = make_float4(0.f, 0.f, 0.f, 0.f);
        double2 mA, mB;
        if constexpr (GIVEN) {
            // (pixel order; pad pixels of the last element repeat the last real one -- finish_elem does not look at them)
            const double *gv = P.given + wk * P.given_stride;
            const int paA = ((ecA >> 8) << 9) | (ecA & 255), paB = ((ecB >> 8) << 9) | (ecB & 255);
            const int last = npix - 1;
            mA = make_double2(gv[paA < last ? paA : last], gv[paA + 256 < last ? paA + 256 : last]);
            mB = make_double2(gv[paB < last ? paB : last], gv[paB + 256 < last ? paB + 256 : last]);
        } else {
#pragma unroll
        for (int c0 = 0; c0 < NC; c0 += G) {
            double2 rA[G], rB[G];
            float4 hq[G];
#pragma unroll
            for (int c = 0; c < G; ++c) {
                if constexpr (R32) {  // (one 16-byte load brings the quad's four float32 values: it is the load COUNT the blend pays for)
                    const float4 qv4 = ld_off(rows_r4f[c0 + c], oq);
                    rA[c] = make_double2((double)qv4.x, (double)qv4.y);
                    rB[c] = make_double2((double)qv4.z, (double)qv4.w);
                } else {
                    rA[c] = ld_off(rows_r[c0 + c], oA);
                    rB[c] = ld_off(rows_r[c0 + c], oB);
                }
                hq[c] = RED ? ld_off(rows_h4[c0 + c], oq) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
            if (c0 == 0 && RED) { klA = ld_off(P.kl2, oA); klB = ld_off(P.kl2, oB); dk = ld_off(MAXT == 512 ? P.dk4 : P.dk4b, oq); }
            double r0[G], r1[G], r2[G], r3[G];
            float h0[G], h1[G], h2[G], h3[G];
#pragma unroll
            for (int c = 0; c < G; ++c) {
                r0[c] = rA[c].x; r1[c] = rA[c].y; r2[c] = rB[c].x; r3[c] = rB[c].y;
                h0[c] = hq[c].x; h1[c] = hq[c].y; h2[c] = hq[c].z; h3[c] = hq[c].w;
            }
            blend_accumulate<G>(r0, h0, w + c0, wf + c0, RED, sr[0], sh[0]);
            blend_accumulate<G>(r1, h1, w + c0, wf + c0, RED, sr[1], sh[1]);
            blend_accumulate<G>(r2, h2, w + c0, wf + c0, RED, sr[2], sh[2]);
            blend_accumulate<G>(r3, h3, w + c0, wf + c0, RED, sr[3], sh[3]);
        }
        mA.x = blend_finish(sr[0], sh[0], klA.x, (double)dk.x, redc, RED, e2tab);
        mA.y = blend_finish(sr[1], sh[1], klA.y, (double)dk.y, redc, RED, e2tab);
        mB.x = blend_finish(sr[2], sh[2], klB.x, (double)dk.z, redc, RED, e2tab);
        mB.y = blend_finish(sr[3], sh[3], klB.y, (double)dk.w, redc, RED, e2tab);
        }
        if (!kEarlyFUq) { fA = ld_off(P.f2, oA); uA = ld_off(P.u2, oA); fB = ld_off(P.f2, oB); uB = ld_off(P.u2, oB); }
        if (PF) { fA = lds_f2[ecA]; uA = lds_u2[ecA]; fB = lds_f2[ecB]; uB = lds_u2[ecB]; }  // (staged in phase 0)
        if (LK) { fA = lds_lf2[ecA]; fB = lds_lf2[ecB]; }
        finish_elem(mA, fA, uA, ecA, liveA, std::integral_constant<int, 0>{});
        finish_elem(mB, fB, uB, ecB, liveB, std::integral_constant<int, SUB - 1>{});  // (256 threads: the trip's second element)
      }
      } else {
      for (int e0 = seg * kSegElems; e0 < e_end; e0 += B * SUB) {
        auto one = [&](auto sub_c) __attribute__((always_inline)) {
        constexpr int sub = decltype(sub_c)::value;
        const int e = e0 + sub * B + tid;
        const bool live = FULLB || e < e_end;
        const int ec = live ? e : e_end - 1;
        const unsigned int o16 = (unsigned int)ec << 4, o8 = (unsigned int)ec << 3;
        // data flux and u: requested with the rows by the 256-thread variant (one wait per trip instead of two:
        // 16,384 walkers 458 -> 438 us), after the blend by the 512-thread ones (17.45 against 17.57 us at 256 walkers)
        constexpr bool kEarlyFU = MAXT == 256;
        double2 f2v = make_double2(0.0, 0.0), u2v = make_double2(0.0, 0.0);
        if (kEarlyFU) { f2v = ld_off(P.f2, o16); u2v = ld_off(P.u2, o16); }
        double2 m2;
        {
            // the model values of the two pixels (blend.h).
            // All corners' loads are issued together (192 bytes in flight per lane) -- except in the variant that
            // shares its CU (128 VGPRs), which takes the rows one star at a time.
            constexpr int G = SH ? 4 : NC;  // corners per group of loads
            double2 kl2 = make_double2(0.0, 0.0);
            float2 dk2 = make_float2(0.f, 0.f);
            double sra = 0.0, srb = 0.0;
            float sha = 0.0f, shb = 0.0f;
#pragma unroll
            for (int c0 = 0; c0 < NC; c0 += G) {
                double2 rr[G];
                float2 hh[G];
#pragma unroll
                for (int c = 0; c < G; ++c) {
                    if constexpr (R32) {
                        const float2 a = ld_off(rows_rf[c0 + c], o8);
                        rr[c] = make_double2((double)a.x, (double)a.y);
                    } else {
                        rr[c] = ld_off(rows_r[c0 + c], o16);
                    }
                    hh[c] = RED ? ld_off(rows_h[c0 + c], o8) : make_float2(0.f, 0.f);
                }
                if (c0 == 0 && RED) { kl2 = ld_off(P.kl2, o16); dk2 = ld_off(P.dk2, o8); }
                double ra[G], rb[G];
                float ha[G], hb[G];
#pragma unroll
                for (int c = 0; c < G; ++c) { ra[c] = rr[c].x; rb[c] = rr[c].y; ha[c] = hh[c].x; hb[c] = hh[c].y; }
                blend_accumulate<G>(ra, ha, w + c0, wf + c0, RED, sra, sha);
                blend_accumulate<G>(rb, hb, w + c0, wf + c0, RED, srb, shb);
            }
            m2.x = blend_finish(sra, sha, kl2.x, (double)dk2.x, redc, RED, e2tab);
            m2.y = blend_finish(srb, shb, kl2.y, (double)dk2.y, redc, RED, e2tab);
        }
        if (!kEarlyFU) { f2v = ld_off(P.f2, o16); u2v = ld_off(P.u2, o16); }
        finish_elem(m2, f2v, u2v, ec, live, sub_c);
        };
        one(std::integral_constant<int, 0>{});
        if constexpr (SUB == 2) one(std::integral_constant<int, 1>{});
      }
      }
      };
      if (redden) trips(std::true_type{}); else trips(std::false_type{});
      // this segment's three fit sums: one partial per lane to LDS, one wave per quantity finishes (wave_ops.h)
#pragma unroll
      for (int i = 0; i < 3; ++i) {
          double a[vk];
#pragma unroll
          for (int k = 0; k < vk; ++k) a[k] = qa[k][i];
          red[i][wave][lane] = lane_partial<vk>(a);
      }
      __syncthreads();
      if (wave < 3) qrun += reduce_published<MAXT>(&red[wave][0][0], lane);
      if (seg + 1 < seg_hi) __syncthreads();  // the next segment rewrites red
    }
    MSX_STAMP(P, wk, 2);
    // The contrast / photometry terms (A5/A6) need the recipe's nodes and weights and nothing else, the Gaussian prior
    // terms (f1) theta alone, and only the walker's last line reads either: waves 1 and 2 compute them while wave 0 ranks
    // the median's candidates (phase B) -- in the linked form while thread 0 waits at the meeting point.  (Other modes:
    // inside block_median.)
    const bool late_side = early && !LK;
    // the early histogram is complete (the segment loop's barrier): its running totals, published by the barrier below
    // (linked: the counters of ONE segment -- they are exchanged first)
    if (early && !LK) hist_prefix_inplace<MAXT>(S);
    {
        const unsigned int lo = wave_min_u32(fmin_), hi = wave_max_u32(fmax_);
        if (lane == 0) {
            // (a wave whose pixels were all beyond the spectrum's end reports the empty range: ~0 / 0)
            S.kmin[wave] = lo;
            S.kmax[wave] = hi;
            if (wave < 3) S.q[0][wave] = qrun;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 3; ++i) q[i] = S.q[0][i];
        fmin_ = (unsigned int)S.kmin[0]; fmax_ = (unsigned int)S.kmax[0];
        for (int x = 1; x < nw; ++x) {
            fmin_ = (unsigned int)S.kmin[x] < fmin_ ? (unsigned int)S.kmin[x] : fmin_;
            fmax_ = (unsigned int)S.kmax[x] > fmax_ ? (unsigned int)S.kmax[x] : fmax_;
        }
    }
    if (LK) {
        // ==== the linked form: this workgroup holds ONE segment's model values, sums, range and counters ====
        SegPart *const sp = P.segparts + wk * nsegs;
        const unsigned long long period = 2ull * (unsigned long long)nsegs;
        static_assert(kLogBins == 4 * MAXT || !LK, "four counters per thread");
        // ---- first meeting: every segment's partials to every workgroup of the walker ----
        // What is handed over is written with AGENT-SCOPE stores (publish_u64: they write through this XCD's L2), every
        // wave waits for its own stores' acknowledgements (s_waitcnt vmcnt(0): a wave's wait covers its own stores only,
        // and a CU's requests to different L2 channels are not ordered among themselves), and after the barrier thread 0
        // signals with a RELAXED agent-scope increment.  That is a release without the release fence's buffer_wbl2: the
        // write-back of the whole L2 is there for plain stores that may sit dirty in it, and these are none of those --
        // with 32 workgroups per XCD arriving together the write-backs queue up (128 walkers x 2 segments: 14.5 k cycles
        // per meeting with the fence).  The acquire side is the compiler's own fence.
        {
            const uint4 c = reinterpret_cast<const uint4 *>(S.hist)[tid];
            unsigned long long *h = reinterpret_cast<unsigned long long *>(sp[myseg].hist) + 2 * tid;
            publish_u64(h, (unsigned long long)c.x | ((unsigned long long)c.y << 32));
            publish_u64(h + 1, (unsigned long long)c.z | ((unsigned long long)c.w << 32));
        }
        if (tid == 0) {
#pragma unroll
            for (int i = 0; i < 3; ++i) publish_u64(reinterpret_cast<unsigned long long *>(&sp[myseg].q[i]), (unsigned long long)__double_as_longlong(q[i]));
            publish_u64(&sp[myseg].kmin, fmin_);
            publish_u64(&sp[myseg].kmax, fmax_);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        MSX_STAMP(P, wk, 3);
        // the pass over this segment (below) starts with loads that depend on nothing the meeting brings: they travel now
        // (the element's "npix" is this segment's end: pixels beyond it are not this workgroup's, and the canonical sum
        // of ONE segment has no fold)
        const int p_lo = myseg * (2 * kSegElems), p_hi = (p_lo + 2 * kSegElems < npix) ? p_lo + 2 * kSegElems : npix;
        ChiElem<MAXT, false, true, true, true, (FULL & 2) != 0> ce{P.u2, lds_lf2, P.iv2, ne, p_hi, 0.0, 0.0, 0.0, {}, true, &red[0][0][0], {}, {}, {}, 0.0};
        ce.prime_from(p_lo);
        if (tid == 0) {
            // (test hook: nobody signals, so every wait below runs into its bound)
            const unsigned long long old = P.linked_fault
                                               ? __hip_atomic_load(P.seg_flag + wk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                               : __hip_atomic_fetch_add(P.seg_flag + wk, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned long long base = old - old % period, want = base + (unsigned long long)nsegs;
            const unsigned long long t0 = wall_clock64();
            bool met = !P.linked_fault && old + 1ull >= want;  // (whoever arrives last knows from its own increment)
            while (!met) {  // (no sleep between the looks: see the sampler's wait above)
                if (wall_clock64() - t0 > kHandoverTicks) break;
                met = __hip_atomic_load(P.seg_flag + wk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= want;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            if (!met) {
                // The others did not come in time.  They may still arrive later in this launch, and nobody can tell when
                // the last one has: the counters of this context are not to be trusted again.  POISON the linked form
                // (sticky, device side): every linked launch checks the word first and reports MSX_W_HANDOVER for all of
                // its walkers until msx_stage_problem clears counters and word together.
                __hip_atomic_store(P.linked_poison, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            S.meet_state = met ? 1u : 0u;
            S.meet_base = base;
        } else if (wave == 1) {
            recipe_band_terms<NS>(P, mode, th_row, D, lane);
        } else if (wave == 2) {
            recipe_prior_terms<NS>(P, mode, th_row, D, lane);
        }
        __syncthreads();
        MSX_STAMP(P, wk, 4);
        if (S.meet_state == 0u) {  // (every workgroup that gave up says so: whichever of them is the only one)
            if (tid == 0) walker_done(V, D, wk, ndim, NAN, MSX_W_HANDOVER, logp, status);
            return;
        }
        {   // the segments' counters added up, their fit sums in segment order -- the order the fused kernel adds them in
            uint4 tot = make_uint4(0u, 0u, 0u, 0u);
            double acc[3] = {0.0, 0.0, 0.0};
            unsigned int f0 = ~0u, f1 = 0u;
            for (int g = 0; g < nsegs; ++g) {
                const bool own = g == myseg;
                const uint4 c = own ? reinterpret_cast<const uint4 *>(S.hist)[tid] : reinterpret_cast<const uint4 *>(sp[g].hist)[tid];
                tot.x += c.x; tot.y += c.y; tot.z += c.z; tot.w += c.w;
#pragma unroll
                for (int i = 0; i < 3; ++i) acc[i] += own ? q[i] : sp[g].q[i];
                const unsigned int g0 = own ? fmin_ : (unsigned int)sp[g].kmin, g1 = own ? fmax_ : (unsigned int)sp[g].kmax;
                f0 = g0 < f0 ? g0 : f0;
                f1 = g1 > f1 ? g1 : f1;
            }
            reinterpret_cast<uint4 *>(S.hist)[tid] = tot;  // (each thread its own four counters)
#pragma unroll
            for (int i = 0; i < 3; ++i) q[i] = acc[i];
            fmin_ = f0; fmax_ = f1;
        }
        hist_prefix_inplace<MAXT>(S);  // (own counters again: no barrier in between)
        __syncthreads();
        MSX_STAMP(P, wk, 5);
        // ---- the pass over this segment: chi^2 terms and the candidates of the median's bin(s) ----
        fit_coefs(P, q, ce.c0, ce.c1, ce.c2);
        const bool need_two = (npix & 1) == 0;
        LogbinSel Q;
        // (uniform over the walker's workgroups: all of them hold the same totals)
        const bool direct = frange_applicable(fmin_, fmax_) && logbin_locate_h<MAXT>(npix, fmin_, S, &Q);
        if (direct) {
            const unsigned int sel_p = Q.sel_p, nxt_p = Q.nxt_p;
            pass_trips_range<MAXT, (FULL & 2) != 0>(model, p_lo, p_hi, ce, [&](const int (&p)[4], const double (&xv)[4]) __attribute__((always_inline)) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const unsigned int pb = logbin(xv[u]);
                    if (((FULL & 2) != 0 || p[u] < p_hi) && (pb == sel_p || pb == nxt_p)) S.cand[atomicAdd(&S.cand_n, 1u)] = key_of(xv[u]);
                }
            });
            ce.flush(S);
            __syncthreads();
            const unsigned int nc = S.cand_n;  // (<= Q.cnt <= kSelectFinish)
            if (wave == 0) {
                const double c = reduce_published<MAXT>(&red[0][0][0], lane);
                if (lane == 0) {
                    publish_u64(reinterpret_cast<unsigned long long *>(&sp[myseg].chi), (unsigned long long)__double_as_longlong(c));
                    publish_u64(reinterpret_cast<unsigned long long *>(&sp[myseg].ncand), (unsigned long long)nc);  // (and pad[0])
                }
            }
            if (tid < (int)nc) publish_u64(&sp[myseg].cand[tid], S.cand[tid]);
        } else {
            // not a positive vector spanning < 8 binades, or > 256 equal-bin candidates: block_median wants the whole
            // vector in one place -- the scratch row (plain stores: this arrival is a release with its fence)
            double *row = P.model_scratch + wk * P.npix;
            for (int i = p_lo + tid; i < p_hi; i += B) row[i] = model[i];
        }
        MSX_STAMP(P, wk, 6);
        // ---- second meeting: nobody waits; whoever arrives last finishes the walker ----
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            if (!direct) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            const unsigned long long old = __hip_atomic_fetch_add(P.seg_flag + wk, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const bool last = old == S.meet_base + period - 1ull;
            if (last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            S.meet_state = last ? 2u : 1u;
        }
        __syncthreads();
        if (S.meet_state != 2u) return;
#ifdef MSX_STAMPS
        if (tid == 0) msx_stamp_off = 0;  // (whoever finishes the walker stamps its last lines)
#endif
        MSX_STAMP(P, wk, 7);
        if (direct) {
            // the other segments' candidates behind this one's; the chi^2 sums in segment order
            unsigned int have = S.cand_n;
            double chi = 0.0;
            for (int g = 0; g < nsegs; ++g) {
                // (the candidate slot is requested with the count, not after it: one round trip, not two)
                const unsigned long long cg = (g != myseg && tid < kSelectFinish) ? sp[g].cand[tid] : 0ull;
                chi += sp[g].chi;
                if (g == myseg) continue;
                const unsigned int ngc = sp[g].ncand;
                if (tid < (int)ngc && have + (unsigned int)tid < (unsigned int)kSelectFinish) S.cand[have + tid] = cg;
                have += ngc;
            }
            __syncthreads();
            // (the histogram said how many there are: anything else means the segments did not see the same totals)
            const bool sane = have == Q.cnt;
            const double med = sane ? logbin_rank<MAXT>(S, Q, need_two, 0) : 0.0;
            if (tid == 0) {
                const double total = fused_total(P, chi, P.median_flux, med, npix, D.chi_extra);
                walker_done(V, D, wk, ndim, sane ? value_of_total(mode, total, D.lp) : NAN, sane ? MSX_W_OK : MSX_W_HANDOVER, logp, status);
                MSX_STAMP(P, wk, 15);
            }
            return;
        }
        // the whole vector is in the scratch row: on as the variants with the model vector in global memory
        model = P.model_scratch + wk * P.npix;
        for (int i = tid; i < kLogBins; i += B) S.hist[i] = 0;
        if (tid == 0) { S.cand_n = 0; S.has_second = 0; }
        __syncthreads();
    }
    MSX_STAMP(P, wk, 3);

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
    double pc0, pc1, pc2;
    fit_coefs(P, q, pc0, pc1, pc2);
    constexpr bool kAhead = MAXT == 512 && !SH;
    ChiElem<MAXT, PF, false, kAhead> chi_elem{PF ? lds_u2 : P.u2, PF ? lds_f2 : P.f2, P.iv2, ne, npix, pc0, pc1, pc2, {}, fused,
                                      &red[0][0][0], {}, {}, {}, 0.0};
    bool chi_done = false;
    double med_model = 0.0;
    bool solved = false;
    bool rank_split = false;  // (<= 64 candidates: ranked by several waves, picked by wave 0 behind the closing barrier)
    LogbinSel rank_Q;
    if (early && !LK) {  // (linked: only vectors the early histogram could not handle come this far)
        ChiElem<MAXT, PF, true, kAhead, false, FULLC> chi_fast{PF ? lds_u2 : P.u2, PF ? lds_f2 : P.f2, P.iv2, ne, npix, pc0, pc1, pc2, {}, true,
                                         &red[0][0][0], {}, {}, {}, 0.0};
        // positive normal values spanning < 8 binades (anything else -- zeros, negatives, infinities, NaNs, huge ranges --
        // takes block_median below); > 256 equal-bin candidates come back unsolved too
        if (frange_applicable(fmin_, fmax_)) solved = logbin_median<MAXT, FULLC>(model, npix, fmin_, S, chi_fast, &med_model, &rank_split, &rank_Q);
        // the ranking of <= 64 candidates, a few trips of eight per wave (the waves that have nothing else left to do);
        // wave 0 reads the sums behind the closing barrier
        constexpr int kRankWaves = MAXT / kWave > 4 ? 4 : 2, kRank0 = MAXT / kWave > 4 ? 3 : 2;
        if (solved && rank_split && wave >= kRank0 && wave < kRank0 + kRankWaves) logbin_rank_part<MAXT>(S, rank_Q, wave - kRank0, kRankWaves);
        chi_done = solved;
        // Only wave 0 is busy from here (it ranks the candidates; the others left logbin_median after its barrier):
        // waves 1 and 2 compute what only the walker's last line reads -- the contrast / photometry terms (A5/A6, which
        // start with a round trip to the band table) and the Gaussian prior terms (f1).  One closing barrier below.
        if (late_side) {
            if (wave == 1) recipe_band_finish<NS>(P, mode, th_row, D, lane, side_val);
        }
        if (solved) chi_elem.tot_run = chi_fast.tot_run;
        if (!solved) {  // not a positive vector spanning < 8 binades, or > 256 equal-bin candidates: start over
            __syncthreads();  // every wave decided from the counters by itself: none may still be reading them
            for (int i = tid; i < kLogBins; i += B) S.hist[i] = 0;
            __syncthreads();
        }
    }
    if (!solved) {
        unsigned long long kmin, kmax;
        exact_range<MAXT>(model, npix, S, &kmin, &kmax);
        // np.median of a vector holding a NaN is NaN -> total NaN -> -inf (mft6.py:1202-1203)
        if (kmax > key_of(INFINITY) || kmin < key_of(-INFINITY)) {
            if (tid == 0) {
                const bool chi_valued = mode == MSX_MODE_CHISQ || mode == MSX_MODE_OPT_STEP || mode == MSX_MODE_OPT_INIT;
                if (mode == MSX_MODE_OPT_INIT) P.opt_med[wk] = NAN;
                walker_done(V, D, wk, ndim, chi_valued ? NAN : -INFINITY, MSX_W_OK, logp, status);
            }
            return;
        }
        med_model = block_median<MAXT>(model, npix, kmin, kmax, S, side, chi_elem, &chi_done);
    }
    if (fused && !chi_done) {  // degenerate vectors (all equal): the median took no pass, do it here
        chi_elem.prime();
        pass_trips<MAXT>(model, npix, chi_elem, [](const int (&)[4], const double (&)[4]) {});
        chi_elem.flush(S);
        __syncthreads();
    }
    MSX_STAMP(P, wk, 4);
    if (!LK) MSX_STAMP(P, wk, 5);

    // ---- phase C: median scale, quadratic continuum fit, chi^2 (A8.2, A8.3, A9) ------------------
    // Pre-optimiser variants (fit_spec, mft6.py:856-1137): OPT_INIT normalises the data against the
    // chain's initial model like the hot path does and KEEPS the normalised vector + its median
    // (:888-889); OPT_STEP compares every proposal with that stored vector, with no per-proposal
    // continuum fit (:1011-1015).  Both weight the spectrum term by 3 (:893,:1015).
    const bool opt_step = mode == MSX_MODE_OPT_STEP, opt_init = mode == MSX_MODE_OPT_INIT;
    const int64_t chain = opt_step ? (int64_t)P.opt_chain[wk] : wk;
    const double *__restrict__ dflux = opt_step ? P.opt_flux + chain * npix : P.pix_flux;
    const double med_data = opt_step ? P.opt_med[chain] : P.median_flux;
    const double scale = fast_div(med_data, med_model);  // mft6.py:1173 / :1011
    double coef[3] = {0.0, 0.0, 0.0};
    if (!fused) {  // (the fused modes' chi^2 terms rode along the median's pass)
#pragma unroll
        for (int i = 0; i < 3; ++i)
            coef[i] = (P.minv[3 * i] * q[0] + P.minv[3 * i + 1] * q[1] + P.minv[3 * i + 2] * q[2]) / scale;
    }
    double chia[vk];  // per virtual wave, like phase A
#pragma unroll
    for (int k = 0; k < vk; ++k) chia[k] = 0.0;
    unsigned long long dmin = ~0ull, dmax = 0ull;
    for (int base = 0; base < npix && !fused; base += 4 * B) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int p = pass_pixel<MAXT>(base, u, tid);
            if (p >= npix) continue;
            const double ms = model[p] * scale;
            double dn;
            if (opt_step) {
                dn = dflux[p];
            } else {
                const double uu_ = P.pix_u[p];
                const double poly = fma(fma(coef[2], uu_, coef[1]), uu_, coef[0]);
                dn = fast_div(dflux[p], poly);  // mft6.py:196
            }
            const double r = ms - dn;
            chia[u & (vk - 1)] += (r * r) * P.pix_ivar[p];  // mft6.py:120
            if (opt_init) {
                P.opt_flux[wk * npix + p] = dn;
                model[p] = dn;  // the model value is dead now; reuse the LDS vector for median(data')
                const unsigned long long key = key_of(dn);
                dmin = key < dmin ? key : dmin;
                dmax = key > dmax ? key : dmax;
            }
        }
    }
    if (!LK) MSX_STAMP(P, wk, 6);
    if (!fused) red[0][wave][lane] = lane_partial<vk>(chia);
    if (opt_init) {
        const unsigned long long a = wave_min_u64(dmin), b = wave_max_u64(dmax);
        if (lane == 0) { S.kmin[wave] = a; S.kmax[wave] = b; }
        for (int i = tid; i < kBins; i += B) S.hist[i] = 0;
    }
    if (!fused) __syncthreads();  // (fused: the partials were published before the median's barrier)
    MSX_STAMP(P, wk, 7);
    // the chi^2 sum: wave 0 (whose lane 0 finishes the walker) combines the lanes' partials
    double tot = 0.0;
    if (wave == 0) tot = reduce_published<MAXT>(&red[0][0][0], lane);
    // (fused modes: the median's scale^2 is applied in fused_total below; the optimiser modes' pass used the scaled model)
    if (opt_init) {
        dmin = S.kmin[0]; dmax = S.kmax[0];
        for (int x = 1; x < nw; ++x) {
            dmin = S.kmin[x] < dmin ? S.kmin[x] : dmin;
            dmax = S.kmax[x] > dmax ? S.kmax[x] : dmax;
        }
        const bool bad = dmax > key_of(INFINITY) || dmin < key_of(-INFINITY);
        NoElem no_elem;
        bool unused = false;
        const double md = bad ? NAN : block_median<MAXT>(model, npix, dmin, dmax, S, NoSide(), no_elem, &unused);  // np.median(flux), :1011
        if (tid == 0) P.opt_med[wk] = md;
    }
    if (late_side) __syncthreads();  // D.chi_extra, D.lp (waves 1 and 2)
    if (rank_split && wave == 0) med_model = logbin_rank_pick<MAXT>(S, rank_Q, (npix & 1) == 0);
    if (tid == 0) {
        double out;
        if (fused) {
            const double total = fused_total(P, chi_elem.tot_run + tot, med_data, med_model, npix, D.chi_extra);
            out = value_of_total(mode, total, D.lp);
        } else {
            const double iic = fast_div(tot, (double)npix) * 3;  // mft6.py:1179; :893,1015
            out = iic * (double)(P.nc + P.np) + D.chi_extra;      // mft6.py:904 / :1028
        }
        walker_done(V, D, wk, ndim, out, MSX_W_OK, logp, status);
        MSX_STAMP(P, wk, 15);
        MED_WALL(7);
    }
}

// ------------------------------------------------------------------------------------------------
// Sharded device-resident sampler (SURVEY §8e + f2): every rank keeps the whole ensemble in HBM and evaluates its
// block of the half-step's proposals with smp_defer = 1; the only thing that crosses xGMI is ONE all-gather of
// log p(q) (ns / G float64 per rank).  This kernel then finishes the half-step on every rank, for every active
// walker: the proposal is rebuilt from the resident state (its inputs -- the walker itself and its partner in the
// complementary half -- are untouched until now, and the expression is the fused kernel's, contraction off), the
// accept rule is the fused kernel's, so all ranks stay bit-identical to each other and to the one-GPU chain.
// ------------------------------------------------------------------------------------------------
__global__ void sampler_apply_kernel(DevProblem P, const double *__restrict__ newlp, int64_t ns, int ndim) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ns) return;
    const int64_t si = P.smp_sidx[i], ci = P.smp_partner[i];
    const double zz = P.smp_zz[i];
    const double out = newlp[i];
    const int st = status_of_nan(out);
    if (st > MSX_W_REJECT) atomicMax(P.smp_worst, st);
    const double old = P.smp_logp[si];
    const double lnpdiff = (P.smp_zfac[i] + out) - old;
    const bool acc = P.smp_logu[i] < lnpdiff;  // NaN differences compare false, like the host loop
    for (int d = 0; d < ndim; ++d) {
#pragma clang fp contract(off)
        const double sv = P.smp_coords[si * ndim + d];
        const double cv = P.smp_coords[ci * ndim + d];
        const double diff = cv - sv;
        const double prod = diff * zz;
        const double qv = cv - prod;
        const double v = acc ? qv : sv;
        if (acc) P.smp_coords[si * ndim + d] = v;
        P.smp_chain_row[si * ndim + d] = v;
    }
    if (acc) {
        P.smp_logp[si] = out;
        P.smp_naccept[si] = P.smp_naccept[si] + 1;
    }
    P.smp_lp_row[si] = acc ? out : old;
}

// ------------------------------------------------------------------------------------------------
// The stretch move's randomness, drawn ON THE DEVICE (emcee's move as mft6.py:1491-1494 drives it): a
// COUNTER-BASED generator -- every number is a pure function of (seed, iteration, stream, index), so a chunk is one
// launch, any rank of a sharded run draws the same numbers without a broadcast, and the host can restate the stream
// (mcmc_spec_amd/sampler.py::counter_draws) to check it.  The generator is SplitMix64's output function over the
// counter sequence seed * K + (ctr + 1) * gamma (Steele, Lea & Flood 2014: the stream a SplitMix64 instance produces).
// One workgroup per iteration:
//   stream 0      one 64-bit key per walker; the walkers sorted by (key, index) are the iteration's random permutation,
//                 its first half the first half-step's walkers, its second half their complementary ensemble (and the
//                 other way round for the second half-step) -- emcee's randomised split
//   streams 1..6  per half-step h and position j: u_z -> z = ((a - 1) u_z + 1)^2 / a, u_p -> partner floor(u_p ns),
//                 u_a -> ln u_a of the accept draw
// and writes the chunk's arrays exactly as the host-fed entry point uploads them (msx.hip, chunk_prepare).
// ------------------------------------------------------------------------------------------------
__host__ __device__ inline unsigned long long counter_mix64(unsigned long long seed, unsigned long long it, unsigned int stream,
                                                           unsigned int index) {
    const unsigned long long ctr = (it << 28) + ((unsigned long long)stream << 24) + (unsigned long long)index;
    unsigned long long x = seed * 0xD1342543DE82EF95ull + (ctr + 1ull) * 0x9E3779B97F4A7C15ull;
    x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
    x ^= x >> 27; x *= 0x94D049BB133111EBull;
    x ^= x >> 31;
    return x;
}
__host__ __device__ inline double counter_uniform(unsigned long long seed, unsigned long long it, unsigned int stream, unsigned int index) {
    return (double)(counter_mix64(seed, it, stream, index) >> 11) * (1.0 / 9007199254740992.0);  // [0, 1), 53 bits
}
constexpr int kDrawMaxWalkers = 4096;  // (key, index) pairs of one iteration sorted in LDS: 48 KB
constexpr int kDrawThreads = 256;
// resolve != 0: `partner` receives cidx[partner] (the ensemble index of the complementary walker: what the kernels read);
// 0: the raw index into the complementary half (what the host loop consumes)
__global__ void __launch_bounds__(kDrawThreads)
sampler_draw_kernel(unsigned long long seed, double a, int64_t first_iter, int64_t nw, int32_t ndim, int32_t resolve, int32_t overlap,
                    int32_t *__restrict__ sidx, int32_t *__restrict__ cidx, int32_t *__restrict__ partner, double *__restrict__ zz,
                    double *__restrict__ zfac, double *__restrict__ logu, SmpRec *__restrict__ rec) {
    __shared__ unsigned long long key[kDrawMaxWalkers];
    __shared__ int32_t idx[kDrawMaxWalkers];
    const int64_t st = blockIdx.x;  // iteration of the chunk
    const unsigned long long it = (unsigned long long)(first_iter + st);
    const int ns = (int)(nw / 2);
    int npad = 1;
    while (npad < nw) npad <<= 1;
    for (int i = threadIdx.x; i < npad; i += kDrawThreads) {
        key[i] = i < nw ? counter_mix64(seed, it, 0u, (unsigned int)i) : ~0ull;  // (pads sort last: index >= nw breaks the tie)
        idx[i] = i;
    }
    __syncthreads();
    // bitonic sort of (key, index), ascending
    for (int k = 2; k <= npad; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < npad; i += kDrawThreads) {
                const int l = i ^ j;
                if (l > i) {
                    const unsigned long long ka = key[i], kb = key[l];
                    const int32_t ia = idx[i], ib = idx[l];
                    const bool a_gt_b = ka > kb || (ka == kb && ia > ib);
                    const bool up = (i & k) == 0;
                    if (a_gt_b == up) { key[i] = kb; key[l] = ka; idx[i] = ib; idx[l] = ia; }
                }
            }
            __syncthreads();
        }
    }
    // the two half-steps of this iteration
    const int64_t base = st * 2 * ns;
    for (int t = threadIdx.x; t < 2 * ns; t += kDrawThreads) {
#pragma clang fp contract(off)
        const int h = t / ns, j = t - h * ns;
        const int32_t s_w = idx[h * ns + j];                    // the moving walker
        const int32_t *comp = idx + (1 - h) * ns;               // the complementary half
        const double uz = counter_uniform(seed, it, 1u + 3u * (unsigned int)h, (unsigned int)j);
        const double up = counter_uniform(seed, it, 2u + 3u * (unsigned int)h, (unsigned int)j);
        const double ua = counter_uniform(seed, it, 3u + 3u * (unsigned int)h, (unsigned int)j);
        const double t1 = (a - 1.0) * uz + 1.0;
        const double z = (t1 * t1) / a;                         // emcee: ((a - 1) u + 1)^2 / a
        int pj = (int)(up * (double)ns);
        pj = pj < ns - 1 ? pj : ns - 1;
        const int64_t o = base + t;
        sidx[o] = s_w;
        cidx[o] = comp[j];
        partner[o] = resolve ? comp[pj] : pj;
        zz[o] = z;
        zfac[o] = ((double)ndim - 1.0) * log(z);
        logu[o] = log(ua);                                      // (u = 0: -inf, accepted by nothing -- like log(random()))
        if (rec) {
            SmpRec r;
            r.si = s_w; r.ci = comp[pj]; r.zz = z;
            r.ver_own = overlap ? (uint32_t)it : 0u;
            r.ver_partner = overlap ? (uint32_t)(it + (unsigned long long)h) : 0u;
            rec[o] = r;
        }
    }
}

}  // namespace

#endif  // MSX_LOGPROB_KERNEL_H
