// wave_ops.h -- part of the single translation unit msx.hip (included there, in this order).
// small device helpers: reddening rule, DPP cross-lane reductions / scans, order-preserving keys of doubles.
#ifndef MSX_WAVE_OPS_H
#define MSX_WAVE_OPS_H

namespace {

// ------------------------------------------------------------------------------------------------
// small device helpers
// ------------------------------------------------------------------------------------------------
// When is the composite (and its photometry) reddened?
//   loglikelihood / logposterior : `av == True and extinct_guess > 0`                 mft6.py:1161
//   fit_spec, proposals          : `var_par[1] > 0`                                    mft6.py:1002
//   fit_spec, initial guess      : never (the extinct() call is commented out and the chi^2 uses the
//                                  un-reddened `phot`)                                 mft6.py:880,901
__device__ __forceinline__ bool redden_rule(int mode, int use_av, double a_v) {
    if (mode == MSX_MODE_OPT_INIT) return false;
    if (mode == MSX_MODE_OPT_STEP) return a_v > 0.0;
    return use_av && a_v > 0.0;
}

// ---- cross-lane reductions on the DPP path (VALU speed) instead of ds_bpermute shuffles (an LDS round
// trip, ~50-100 cycles each, per 32-bit half, per step).  Four DPP steps leave every lane of a 16-lane
// row with its row's result; the four rows are then combined through v_readlane in a fixed order, so
// every lane returns the same, run-to-run reproducible value.
template <int CTRL>
__device__ __forceinline__ int dpp_i32(int v) {
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false);
}
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
    return __hiloint2double(dpp_i32<CTRL>(__double2hiint(v)), dpp_i32<CTRL>(__double2loint(v)));
}
template <int CTRL>
__device__ __forceinline__ unsigned long long dpp_u64(unsigned long long v) {
    const unsigned int lo = (unsigned int)dpp_i32<CTRL>((int)(unsigned int)v);
    const unsigned int hi = (unsigned int)dpp_i32<CTRL>((int)(unsigned int)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}
constexpr int kDppQuadSwap1 = 0xB1;  // quad_perm:[1,0,3,2]
constexpr int kDppQuadSwap2 = 0x4E;  // quad_perm:[2,3,0,1]
constexpr int kDppRowRor4 = 0x124;   // row_ror:4
constexpr int kDppRowRor8 = 0x128;   // row_ror:8

__device__ __forceinline__ double lane_f64(double v, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l),
                            __builtin_amdgcn_readlane(__double2loint(v), l));
}
__device__ __forceinline__ unsigned long long lane_u64(unsigned long long v, int l) {
    const unsigned int lo = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)v, l);
    const unsigned int hi = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)(v >> 32), l);
    return ((unsigned long long)hi << 32) | lo;
}

__device__ __forceinline__ double wave_sum(double v) {
    v += dpp_f64<kDppQuadSwap1>(v);
    v += dpp_f64<kDppQuadSwap2>(v);
    v += dpp_f64<kDppRowRor4>(v);
    v += dpp_f64<kDppRowRor8>(v);
    return ((lane_f64(v, 0) + lane_f64(v, 16)) + lane_f64(v, 32)) + lane_f64(v, 48);
}
__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v) {
    unsigned long long t;
    t = dpp_u64<kDppQuadSwap1>(v); v = t < v ? t : v;
    t = dpp_u64<kDppQuadSwap2>(v); v = t < v ? t : v;
    t = dpp_u64<kDppRowRor4>(v); v = t < v ? t : v;
    t = dpp_u64<kDppRowRor8>(v); v = t < v ? t : v;
    const unsigned long long a = lane_u64(v, 0), b = lane_u64(v, 16), c = lane_u64(v, 32), d = lane_u64(v, 48);
    const unsigned long long ab = a < b ? a : b, cd = c < d ? c : d;
    return ab < cd ? ab : cd;
}
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
    unsigned long long t;
    t = dpp_u64<kDppQuadSwap1>(v); v = t > v ? t : v;
    t = dpp_u64<kDppQuadSwap2>(v); v = t > v ? t : v;
    t = dpp_u64<kDppRowRor4>(v); v = t > v ? t : v;
    t = dpp_u64<kDppRowRor8>(v); v = t > v ? t : v;
    const unsigned long long a = lane_u64(v, 0), b = lane_u64(v, 16), c = lane_u64(v, 32), d = lane_u64(v, 48);
    const unsigned long long ab = a > b ? a : b, cd = c > d ? c : d;
    return ab > cd ? ab : cd;
}
// inclusive prefix sum over the 64 lanes: DPP row_shr steps inside each row of 16, then the three row
// carries through readlane
__device__ __forceinline__ unsigned int wave_scan_u32(unsigned int v) {
    const int lane = threadIdx.x & 63;
    unsigned int x = v;
    x += (unsigned int)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, true);  // row_shr:1
    x += (unsigned int)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, true);  // row_shr:2
    x += (unsigned int)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, true);  // row_shr:4
    x += (unsigned int)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, true);  // row_shr:8
    const unsigned int r0 = (unsigned int)__builtin_amdgcn_readlane((int)x, 15);
    const unsigned int r1 = (unsigned int)__builtin_amdgcn_readlane((int)x, 31);
    const unsigned int r2 = (unsigned int)__builtin_amdgcn_readlane((int)x, 47);
    const int row = lane >> 4;
    return x + (row > 0 ? r0 : 0u) + (row > 1 ? r1 : 0u) + (row > 2 ? r2 : 0u);
}

// order-preserving map double -> uint64 (NaN with sign bit clear sorts above +inf, like np.sort)
__device__ __forceinline__ unsigned long long key_of(double x) {
    unsigned long long b = (unsigned long long)__double_as_longlong(x);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double val_of(unsigned long long k) {
    unsigned long long b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    return __longlong_as_double((long long)b);
}

}  // namespace

#endif  // MSX_WAVE_OPS_H
