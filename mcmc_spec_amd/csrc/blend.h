// blend.h -- part of the single translation unit msx.hip (included there, in this order).
// The per-pixel arithmetic of the model vector (A2 blend, A4 scale + sum, A7 reddening, A8.1 resample) as ONE inline
// function, shared by the fused hot kernel (logprob_kernel.h) and by the walker-tiled blend kernel of the split
// path (split_kernels.h): both must produce the same bits for a walker, so both call this and nothing else.
#ifndef MSX_BLEND_H
#define MSX_BLEND_H

namespace {

// Canonical corner order of a star's four grid nodes: ascending flat node index.  The reference blends
// "nearest node first" (mft6.py:439-477, 508-511) with a lerp-of-lerps formula; here the blend is a weighted sum
// of the four rows and its summation order is a convention of this library.  Sorting by node index makes the
// order a function of the walker's GRID CELL alone (not of which corner happens to be nearest), so that all the
// walkers of a cell read the same rows in the same order -- what the walker-tiled kernel shares loads on.
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

// One data pixel of one walker.  v[c] = {flux[lo], flux[lo+1]} of corner c's grid node at this pixel (compact
// pairs: .y holds the float32 DIFFERENCE), w[c] = bilinear weight x (R/d)^2, kk = CCM89 k at the two samples,
// t = resample weight, redc = -0.4 log2(10) A_V (0: no reddening).
template <int NC, bool CP>
__device__ __forceinline__ double blend_pixel(const double2 (&v)[NC], const double (&w)[NC], const double2 kk, const double t,
                                              const double redc, const bool redden) {
    double ylo = 0.0, yhi = 0.0;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        ylo = fma(w[c], v[c].x, ylo);
        yhi = fma(w[c], v[c].y, yhi);
    }
    if (CP) yhi += ylo;  // blended difference -> blended upper sample
    if (redden) {
        const double elo = exp2(redc * kk.x);  // 10^(-0.4 A_V k)     mft6.py:62-63
        // neighbouring grid samples: y = ln2 * c * (k_hi - k_lo) is tiny, so e^y from four series
        // terms is exact to < 1e-17 for |y| < 1e-3; anything larger takes the full exp2
        const double y = 0.6931471805599453 * (redc * (kk.y - kk.x));
        const double ehi = (fabs(y) < 1e-3) ? elo * fma(y, fma(y, fma(y, fma(y, 1.0 / 24, 1.0 / 6), 0.5), 1.0), 1.0)
                                            : exp2(redc * kk.y);
        ylo *= elo;
        yhi *= ehi;
    }
    return fma(yhi - ylo, t, ylo);  // mft6.py:1169-1170
}

}  // namespace

#endif  // MSX_BLEND_H
