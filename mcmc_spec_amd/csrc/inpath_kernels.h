// In-path broadening (SURVEY A3 placement (ii); MSX_PATH_INPATH): the instrumental broadening applied PER WALKER to the
// unreddened composite inside the data window, instead of once per grid node at staging (the reference's live path,
// mft6.py:366-378 -- the default here too).  Broadening is linear, so the two placements give the same model values up to
// the order of the sums; this form exists for callers who want the resolution inside the evaluation (north_star's
// "broadening kernel" in the path) and to show the equivalence on the device.  It is NOT the headline path: a walker reads
// eight raw window rows (nwin samples each) and convolves nwin samples with lx taps where the table form reads resampled
// pixel tables.
//
//   inpath_recipe_kernel    one thread per walker: the recipe (recipe.h, recipe_scalar2 -- the planner's) -> InpathRec
//   inpath_conv_kernel      one workgroup per (tile of the window, walker): composite of the RAW rows with the recipe's
//                           weights, staged through LDS, convolved with the Gaussian taps (pyasl.instrBroadGaussFast as
//                           restated by broaden_conv_kernel: np.convolve(y, e, 'same'), zero padding) -> tmp[walker][nwin]
//   inpath_resample_kernel  one thread per (pixel, walker): the edge patches broad[0:5] = broad[5], broad[n-10:n] =
//                           broad[n-11] (mft6.py:129-130), reddening at the two model samples that bracket the pixel and
//                           the linear resample (blend.h, blend_finish: the table form's own last step) -> given[walker][pixel]
//   logprob_kernel<GIVEN>   everything else (logprob_kernel.h)
#ifndef MSX_INPATH_KERNELS_H
#define MSX_INPATH_KERNELS_H

namespace {

__global__ void __launch_bounds__(kPlanThreads)
inpath_recipe_kernel(const double *__restrict__ theta, const unsigned char *__restrict__ rblk, int niso_nt, int ng_mode_fast, int64_t n,
                     double gate_tmin, double gate_tmax, InpathRec *__restrict__ rec, DevProblem P) {
    constexpr int NS = 2, ndim = 6;
    const int niso = niso_nt & 0xffff, nt = niso_nt >> 16;
    const int ng = ng_mode_fast & 0xff, mode = (ng_mode_fast >> 8) & 0xff;
    const GateArgs gates = {gate_tmin, gate_tmax, ((ng_mode_fast >> 18) & 1) != 0, ((ng_mode_fast >> 19) & 1) != 0};
    const int tid = threadIdx.x;
    const int64_t i = (int64_t)blockIdx.x * kPlanThreads + tid;
    const bool mine = i < n;
    double t[ndim];
#pragma unroll
    for (int k = 0; k < ndim; ++k) t[k] = mine ? theta[i * ndim + k] : 0.0;
    ScalarTabs T;
    ScalarPriorTabs TP;
    stage_scalar_tables(rblk, P, niso, nt, ng, T, TP);
    __syncthreads();
    if (!mine) return;
    int node[NS * 4], iso_lo[NS], av_bin;
    double w[NS * 4], redc;
    const int st = recipe_scalar2(gates, T, mode, t, node, w, &redc, iso_lo, &av_bin);
    InpathRec R;
#pragma unroll
    for (int c = 0; c < NS * 4; ++c) { R.w[c] = w[c]; R.node[c] = node[c]; }
    R.redc = redc;
    R.ok = st == MSX_W_OK ? 1 : 0;  // (a walker the recipe fails gets its value and status from logprob_kernel<GIVEN>'s own recipe)
    R.pad = 0;
    rec[i] = R;
}

// out[walker][n] = sum_k e[k] * comp[n + c - k], comp = sum_corner w_corner * raw[node_corner][i], zero outside [0, N)
// (broaden_conv_kernel with the composite in place of a stored row; the taps are built the same way)
__global__ void __launch_bounds__(256)
inpath_conv_kernel(const double *__restrict__ raw, int64_t raw_stride, const InpathRec *__restrict__ rec, double *__restrict__ out,
                   int64_t out_stride, int64_t n, int lx, double dx, double sigma) {
    double *taps = reinterpret_cast<double *>(dyn_lds);  // [lx]
    double *tile = taps + lx;                             // [(kConvTile + lx - 1) * 5 / 4 + 1]
    __shared__ double part[4];
    const int tid = threadIdx.x;
    const InpathRec &R = rec[blockIdx.y];
    if (!R.ok) return;  // (uniform)
    const int c = (lx - 1) / 2;
    const int off0 = lx / 2 + lx % 2 - 1;
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
    const double *rows[8];
    double w[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { rows[k] = raw + (int64_t)R.node[k] * raw_stride; w[k] = R.w[k]; }
    const int64_t t0 = (int64_t)blockIdx.x * kConvTile;
    const int64_t g0 = t0 + c - (lx - 1);
    // (the tile is stored with one pad word per four -- index i at i + i / 4 -- so that the lanes' reads below, four samples
    // apart, fall on different banks)
    for (int j = tid; j < kConvTile + lx - 1; j += 256) {
        const int64_t g = g0 + j;
        double v = 0.0;
        if (g >= 0 && g < n) {
            // star by star, like the reference's sum (mft6.py:744,751); within a star in the canonical corner order
            double s0 = 0.0, s1 = 0.0;
#pragma unroll
            for (int k = 0; k < 4; ++k) { s0 = fma(w[k], rows[k][g], s0); s1 = fma(w[4 + k], rows[4 + k][g], s1); }
            v = s0 + s1;
        }
        tile[j + (j >> 2)] = v;
    }
    for (int k = tid; k < lx; k += 256) taps[k] = taps[k] / norm;
    __syncthreads();
    // Four ADJACENT outputs per thread: a tap is read once for the four, and of the four samples it meets three are the
    // previous tap's, kept in registers -- two LDS reads per four multiply-adds where one output at a time takes eight.
    // Every output's sum runs over the taps in ascending order, as in broaden_conv_kernel.
    static_assert(kConvTile == 4 * 256, "four outputs per thread");
    double *orow = out + (int64_t)blockIdx.y * out_stride;
    const int base = 4 * tid + (lx - 1);
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    double v1 = tile[(base + 1) + ((base + 1) >> 2)], v2 = tile[(base + 2) + ((base + 2) >> 2)], v3 = tile[(base + 3) + ((base + 3) >> 2)];
    for (int k = 0; k < lx; ++k) {
        const int i = base - k;
        const double v0 = tile[i + (i >> 2)], tk = taps[k];
        s0 = fma(tk, v0, s0); s1 = fma(tk, v1, s1); s2 = fma(tk, v2, s2); s3 = fma(tk, v3, s3);
        v3 = v2; v2 = v1; v1 = v0;
    }
    const int64_t o = t0 + 4 * tid;
    if (o < n) orow[o] = s0;
    if (o + 1 < n) orow[o + 1] = s1;
    if (o + 2 < n) orow[o + 2] = s2;
    if (o + 3 < n) orow[o + 3] = s3;
}

// given[walker][p] = (1 - t) c'[lo] e_lo + t c'[lo + 1] e_hi with c' the broadened composite (edge patches applied on the
// way), e = 10^(-0.4 A_V k) at the two model samples -- through blend_finish, i.e. in the table form's own arithmetic
// (R = lo + (hi - lo) t in float64, H = hi t in float32, the series for e_hi / e_lo - 1)
__global__ void __launch_bounds__(256)
inpath_resample_kernel(const double *__restrict__ tmp, int64_t tmp_stride, int64_t n, int64_t win_i0, const InpathRec *__restrict__ rec,
                       const int64_t *__restrict__ pix_lo, const double *__restrict__ pix_t, const double *__restrict__ kgrid,
                       int64_t npix, double *__restrict__ given, int64_t given_stride) {
    __shared__ double e2tab[kExp2Tab];
    fill_exp2_table(e2tab, (int)threadIdx.x);
    __syncthreads();
    const InpathRec &R = rec[blockIdx.y];
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (!R.ok || p >= given_stride) return;
    const int64_t pc = p < npix ? p : npix - 1;  // (pad pixels repeat the last real one)
    const int64_t jl = pix_lo[pc];
    int64_t a = jl - win_i0, b = a + 1;          // (both inside the window: checked at msx_stage_problem)
    a = a < 5 ? 5 : (a >= n - 10 ? n - 11 : a);  // broad[0:5] = broad[5]; broad[n-10:n] = broad[n-11]
    b = b < 5 ? 5 : (b >= n - 10 ? n - 11 : b);
    const double *row = tmp + (int64_t)blockIdx.y * tmp_stride;
    const double lo = row[a], hi = row[b], t = pix_t[pc];
    const double sr = fma(hi - lo, t, lo);
    const float sh = (float)(hi * t);
    const double kl = kgrid[jl];
    const float dk = (float)(kgrid[jl + 1] - kgrid[jl]);
    given[(int64_t)blockIdx.y * given_stride + p] = blend_finish(sr, sh, kl, (double)dk, R.redc, R.redc != 0.0, e2tab);
}

}  // namespace

#endif  // MSX_INPATH_KERNELS_H
