// staging_kernels.h -- part of the single translation unit msx.hip (included there, in this order).
// staging / measurement kernels: CCM89 curve, pair gather, band integrals, broadening FIR, resample, make_composite, stream copy.
#ifndef MSX_STAGING_KERNELS_H
#define MSX_STAGING_KERNELS_H

namespace {

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

// The blend's tables (logprob_kernel.h "TABLE LAYOUT"; blend.h blend_pixel_rh).  Element e holds pixels
// {pa, pa + 256}, pa = (e >> 8) * 512 + (e & 255); pad pixels (>= npix) repeat the last real pixel.
__device__ __forceinline__ void element_pixels(int64_t e, int64_t npix, int64_t *pa, int64_t *pb) {
    const int64_t a = ((e >> 8) << 9) | (e & 255), b = a + 256;
    *pa = a < npix ? a : npix - 1;
    *pb = b < npix ? b : npix - 1;
}
// the R table once more in float32 (msx_set_grid_storage(MSX_STORE_F32): a separately labelled storage precision)
__global__ void narrow_r_kernel(const double2 *__restrict__ r2, float2 *__restrict__ r2f, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) r2f[i] = make_float2((float)r2[i].x, (float)r2[i].y);
}

// R = lo + (hi - lo) t, H = hi t of one grid node (blockIdx.y) at both pixels of element e
__global__ void gather_rh_kernel(const double *__restrict__ grid, int64_t nwl, const int64_t *__restrict__ lo,
                                 const double *__restrict__ t, int64_t npix, int64_t npair, double2 *__restrict__ R,
                                 float2 *__restrict__ H) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= npair) return;
    int64_t pa, pb;
    element_pixels(e, npix, &pa, &pb);
    const double *row = grid + (int64_t)blockIdx.y * nwl;
    const double a0 = row[lo[pa]], a1 = row[lo[pa] + 1], b0 = row[lo[pb]], b1 = row[lo[pb] + 1];
    R[(int64_t)blockIdx.y * npair + e] = make_double2(fma(a1 - a0, t[pa], a0), fma(b1 - b0, t[pb], b0));
    H[(int64_t)blockIdx.y * npair + e] = make_float2((float)(a1 * t[pa]), (float)(b1 * t[pb]));
}
// CCM89 k[lo], k[lo+1] - k[lo], and element copies of the data flux, the mapped wavelength and 1/err^2
__global__ void gather_statics_kernel(const double *__restrict__ kgrid, const int64_t *__restrict__ lo,
                                      const double *__restrict__ flux, const double *__restrict__ u,
                                      const double *__restrict__ ivar, int64_t npix, int64_t npair,
                                      double2 *__restrict__ kl2, float2 *__restrict__ dk2, double2 *__restrict__ f2,
                                      double2 *__restrict__ u2, double2 *__restrict__ iv2) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= npair) return;
    int64_t pa, pb;
    element_pixels(e, npix, &pa, &pb);
    const int64_t ja = lo[pa], jb = lo[pb];
    kl2[e] = make_double2(kgrid[ja], kgrid[jb]);
    dk2[e] = make_float2((float)(kgrid[ja + 1] - kgrid[ja]), (float)(kgrid[jb + 1] - kgrid[jb]));
    f2[e] = make_double2(flux[pa], flux[pb]);
    u2[e] = make_double2(u[pa], u[pb]);
    iv2[e] = make_double2(ivar[pa], ivar[pb]);
}

// The float32 tables once more in QUADS of four pixels: quad q = 512 J + t holds the elements eA = 1024 J + t and
// eB = eA + 512 (pixels {x, x + 256, x + 1024, x + 1280}), so that a lane of a 512-thread workgroup fetches the H
// (and dk) values of the two elements it takes per quad trip with ONE 16-byte load instead of two 8-byte ones.
// src has `rows` rows of npair float2 (H: one row per grid node; dk: one row); dst rows of nquad float4.
// (stride = the workgroup size the table is for: 512, or 256 -- quad q = stride J + t holds elements eA = 2 stride J + t
// and eB = eA + stride, the two elements a lane of such a workgroup takes per quad trip)
__global__ void gather_quads_kernel(const float2 *__restrict__ src, int64_t npair, int64_t nquad, int64_t stride,
                                    float4 *__restrict__ dst) {
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nquad) return;
    const int64_t eA = (q / stride) * 2 * stride + (q % stride), eB = eA + stride;
    const float2 *row = src + (int64_t)blockIdx.y * npair;
    const float2 a = eA < npair ? row[eA] : make_float2(0.f, 0.f), b = eB < npair ? row[eB] : make_float2(0.f, 0.f);
    dst[(int64_t)blockIdx.y * nquad + q] = make_float4(a.x, a.y, b.x, b.y);
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

// Stream copy (the measured-HBM-bandwidth figure quoted beside the roofline): every lane keeps UNROLL independent
// 16-byte loads in flight, then stores them; workgroups stride over tiles of 256 * UNROLL float4.  NT = nontemporal
// loads and stores (no reuse: keep the lines out of the way).  msx_stream_copy_gbps times a few (UNROLL, NT, grid)
// combinations and reports the best: it is a property of the chip that is being measured, not of one kernel shape.
template <int UNROLL, bool NT>
__global__ void __launch_bounds__(256) copy_float4_kernel(const float4 *__restrict__ src, float4 *__restrict__ dst, int64_t n4) {
    const int64_t tile = 256 * UNROLL;
    const int64_t stride = (int64_t)gridDim.x * tile;
    typedef float vf4 __attribute__((ext_vector_type(4)));  // (the nontemporal builtins take native vectors)
    const vf4 *s4 = reinterpret_cast<const vf4 *>(src);
    vf4 *d4 = reinterpret_cast<vf4 *>(dst);
    for (int64_t base = (int64_t)blockIdx.x * tile + threadIdx.x; base < n4; base += stride) {
        vf4 v[UNROLL];
#pragma unroll
        for (int k = 0; k < UNROLL; ++k) {
            const int64_t i = base + 256 * k;
            if (i < n4) v[k] = NT ? __builtin_nontemporal_load(s4 + i) : s4[i];
        }
#pragma unroll
        for (int k = 0; k < UNROLL; ++k) {
            const int64_t i = base + 256 * k;
            if (i < n4) { if (NT) __builtin_nontemporal_store(v[k], d4 + i); else d4[i] = v[k]; }
        }
    }
}

}  // namespace

#endif  // MSX_STAGING_KERNELS_H
