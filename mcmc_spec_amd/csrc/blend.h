// blend.h -- part of the single translation unit msx.hip (included there, in this order).
// The per-pixel arithmetic of the model vector (A2 blend, A4 scale + sum, A7 reddening, A8.1 resample) as ONE inline
// function, shared by the fused hot kernel (logprob_kernel.h, its linked form included) and by the pair kernel
// (pair_kernel.h): every form must produce the same bits for a walker, so all of them call this and nothing else.
#ifndef MSX_BLEND_H
#define MSX_BLEND_H

namespace {

// Canonical corner order of a star's four grid nodes: ascending flat node index.  The reference blends
// "nearest node first" (mft6.py:439-477, 508-511) with a lerp-of-lerps formula; here the blend is a weighted sum
// of the four rows and its summation order is a convention of this library.  Sorting by node index makes the
// order a function of the walker's GRID CELL alone (not of which corner happens to be nearest), so that all the
// walkers of a cell read the same rows in the same order -- what the pair kernel shares loads on.
// Duplicated nodes (Teff or logg exactly on a node) carry weight 0 on one copy; the sum is the same either way.
__device__ __forceinline__ void sort4_by_node(int (&node)[4], double (&w)[4]) {
#define MSX_CSWAP(a, b)                                                             \
    do {                                                                            \
        const bool sw = node[a] > node[b];                                          \
        const int tn = sw ? node[b] : node[a], un = sw ? node[a] : node[b];         \
        const double tw = sw ? w[b] : w[a], uw = sw ? w[a] : w[b];                  \
        node[a] = tn; node[b] = un; w[a] = tw; w[b] = uw;                           \
    } while (0)
    MSX_CSWAP(0, 1); MSX_CSWAP(2, 3); MSX_CSWAP(0, 2); MSX_CSWAP(1, 3); MSX_CSWAP(1, 2);
#undef MSX_CSWAP
}

// One data pixel of one walker.  The reference blends the two model samples that bracket the pixel (lo, hi),
// reddens each and interpolates (mft6.py:508-511, 1161-1170): with y = sum_c w_c * sample_c,
//     m = e_lo y_lo + (e_hi y_hi - e_lo y_lo) t,        e = 10^(-0.4 A_V k) at the two samples.
// Reddening acts on the model grid BEFORE the resample, so the two samples get different factors and the resample
// cannot simply be folded into a table.  But e_hi / e_lo = 1 + eps with eps ~ 4e-5 A_V, and
//     m = e_lo y_lo + (e_hi y_hi - e_lo y_lo) t = e_lo [ (y_lo + (y_hi - y_lo) t)  +  eps (y_hi t) ]
// exactly.  Staging therefore stores, per grid node and pixel, R = lo + (hi - lo) t in float64 (8 B) and
// H = hi t in float32 (4 B): H (and dk) only ever enter multiplied by eps, so their 2^-24 rounding perturbs m by
// < 2.4e-12 A_V relative (measured on config 2, half the walkers at A_V ~ 1: log-posteriors within 5e-15 of the
// 16-byte {lo, hi} form) -- 12 bytes per node-pixel instead of 16, eight float64 FMAs instead of sixteen.
// r[c], h[c]: corner c's R and H; wf = w rounded to float32; kl = k[lo], dk = k[lo+1] - k[lo].
// With no reddening (A_V <= 0, mft6.py:1161) m = sum_c w_c R_c: the unreddened interpolation itself.
// (the sums may be taken in pieces -- corners C0 .. C0 + N - 1 at a time, always in ascending corner order -- so
// that a register-starved variant can load one star's rows, fold them in and load the next: same chain, same bits)
template <int N>
__device__ __forceinline__ void blend_accumulate(const double *r, const float *h, const double *w, const float *wf,
                                                 const bool redden, double &sr, float &sh) {
#pragma unroll
    for (int c = 0; c < N; ++c) sr = fma(w[c], r[c], sr);
    if (redden) {
#pragma unroll
        for (int c = 0; c < N; ++c) sh = fmaf(wf[c], h[c], sh);
    }
}
// 2^x for the reddening factor, x = -0.4 log2(10) A_V k <= 0.  The library exp2 spends 33 vector instructions per
// call (full-range argument handling, an 11-term polynomial whose coefficients each need a register move); here
// x = n + j/64 + z with |z| <= 1/128, 2^(j/64) from a 64-entry table in LDS (kExp2Tab, filled once per workgroup by
// fill_exp2_table) and 2^z - 1 = sum_{k=1..5} (ln 2)^k z^k / k! (truncation < 4e-17 relative), the coefficients
// wave-uniform in SGPRs: twelve instructions, the result within ~1.5 ulp.  Underflow goes through v_ldexp_f64; a
// NaN argument gives NaN.
constexpr int kExp2Tab = 64;
constexpr double kLn2Pow1 = 0.6931471805599453, kLn2Pow2 = 0.2402265069591007, kLn2Pow3 = 0.055504108664821576,
                 kLn2Pow4 = 0.009618129107628477, kLn2Pow5 = 0.0013333558146428441;  // (ln 2)^k / k!
__device__ __forceinline__ void fill_exp2_table(double *tab, int idx) {  // call with idx = 0 .. kExp2Tab - 1, then barrier
    if ((unsigned)idx < (unsigned)kExp2Tab) tab[idx] = exp2((double)idx * (1.0 / kExp2Tab));
}
__device__ __forceinline__ double table_exp2(const double x, const double *__restrict__ tab) {
    const double nf = rint(x * (double)kExp2Tab);
    const int ni = (int)nf;
    const double z = fma(nf, -1.0 / kExp2Tab, x);  // exact
    double p = fma_sc(z, kLn2Pow5, kLn2Pow4);
    p = fma_sc(p, z, kLn2Pow3);
    p = fma_sc(p, z, kLn2Pow2);
    p = fma_sc(p, z, kLn2Pow1);
    p = p * z;  // 2^z - 1
    const double t = tab[ni & (kExp2Tab - 1)];
    return ldexp(fma(t, p, t), ni >> 6);
}
static_assert(kExp2Tab == 64, "table_exp2 shifts by 6");

__device__ __forceinline__ double blend_finish(const double sr, const float sh, const double kl, const double dk,
                                               const double redc, const bool redden, const double *__restrict__ e2tab) {
    if (!redden) return sr;
    const double elo = table_exp2(redc * kl, e2tab);  // 10^(-0.4 A_V k)     mft6.py:62-63
    const double s = redc * dk;
    // eps = 2^s - 1: four series terms (exact to < 1e-14 relative for |s| < 1e-3, and eps only ever scales the
    // H term, itself <~ 1e-3 of the pixel), else through the table (the library exp2 would park its eleven
    // coefficients in vector registers for a branch that real extinction curves never take)
    double eps;
    if (fabs(s) < 1e-3) {
        double p = fma_sc(s, kLn2Pow4, kLn2Pow3);
        p = fma_sc(p, s, kLn2Pow2);
        p = fma_sc(p, s, kLn2Pow1);
        eps = p * s;
    } else {
        eps = table_exp2(s, e2tab) - 1.0;
    }
    return elo * fma(eps, (double)sh, sr);
}
template <int NC>
__device__ __forceinline__ double blend_pixel_rh(const double (&r)[NC], const float (&h)[NC], const double (&w)[NC],
                                                 const float (&wf)[NC], const double kl, const double dk, const double redc,
                                                 const bool redden, const double *__restrict__ e2tab) {
    double sr = 0.0;
    float sh = 0.0f;
    blend_accumulate<NC>(r, h, w, wf, redden, sr, sh);
    return blend_finish(sr, sh, kl, dk, redc, redden, e2tab);
}

}  // namespace

#endif  // MSX_BLEND_H
