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
// (32-bit: ONE DPP-operand instruction per step)
__device__ __forceinline__ unsigned int wave_min_u32(unsigned int v) {
    unsigned int t;
    t = (unsigned int)dpp_i32<kDppQuadSwap1>((int)v); v = t < v ? t : v;
    t = (unsigned int)dpp_i32<kDppQuadSwap2>((int)v); v = t < v ? t : v;
    t = (unsigned int)dpp_i32<kDppRowRor4>((int)v); v = t < v ? t : v;
    t = (unsigned int)dpp_i32<kDppRowRor8>((int)v); v = t < v ? t : v;
    const unsigned int a = (unsigned int)__builtin_amdgcn_readlane((int)v, 0), b = (unsigned int)__builtin_amdgcn_readlane((int)v, 16);
    const unsigned int c = (unsigned int)__builtin_amdgcn_readlane((int)v, 32), d = (unsigned int)__builtin_amdgcn_readlane((int)v, 48);
    const unsigned int ab = a < b ? a : b, cd = c < d ? c : d;
    return ab < cd ? ab : cd;
}
__device__ __forceinline__ unsigned int wave_max_u32(unsigned int v) {
    unsigned int t;
    t = (unsigned int)dpp_i32<kDppQuadSwap1>((int)v); v = t > v ? t : v;
    t = (unsigned int)dpp_i32<kDppQuadSwap2>((int)v); v = t > v ? t : v;
    t = (unsigned int)dpp_i32<kDppRowRor4>((int)v); v = t > v ? t : v;
    t = (unsigned int)dpp_i32<kDppRowRor8>((int)v); v = t > v ? t : v;
    const unsigned int a = (unsigned int)__builtin_amdgcn_readlane((int)v, 0), b = (unsigned int)__builtin_amdgcn_readlane((int)v, 16);
    const unsigned int c = (unsigned int)__builtin_amdgcn_readlane((int)v, 32), d = (unsigned int)__builtin_amdgcn_readlane((int)v, 48);
    const unsigned int ab = a > b ? a : b, cd = c > d ? c : d;
    return ab > cd ? ab : cd;
}
__device__ __forceinline__ double wave_min_f64(double v) {  // fmin / fmax: a NaN operand is ignored (callers flag NaNs apart)
    v = fmin(v, dpp_f64<kDppQuadSwap1>(v));
    v = fmin(v, dpp_f64<kDppQuadSwap2>(v));
    v = fmin(v, dpp_f64<kDppRowRor4>(v));
    v = fmin(v, dpp_f64<kDppRowRor8>(v));
    return fmin(fmin(lane_f64(v, 0), lane_f64(v, 16)), fmin(lane_f64(v, 32), lane_f64(v, 48)));
}
__device__ __forceinline__ double wave_max_f64(double v) {
    v = fmax(v, dpp_f64<kDppQuadSwap1>(v));
    v = fmax(v, dpp_f64<kDppQuadSwap2>(v));
    v = fmax(v, dpp_f64<kDppRowRor4>(v));
    v = fmax(v, dpp_f64<kDppRowRor8>(v));
    return fmax(fmax(lane_f64(v, 0), lane_f64(v, 16)), fmax(lane_f64(v, 32), lane_f64(v, 48)));
}

// ---- instruction-level helpers for the per-pixel loops (these run at the VALU issue limit with large batches: every
// instruction the compiler adds around an operation is paid 4096 times per walker) ---------------------------------
// A wave-uniform value, moved to SGPRs: the blend weights, the reddening coefficient.  As scalar operands of the
// per-pixel FMAs they occupy no vector registers (24 for a binary's eight corners).
__device__ __forceinline__ double uniform_f64(double v) {
    const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v)), hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ float uniform_f32(float v) {
    return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v)));
}
// a * b + c with c a wave-uniform constant held in SGPRs.  The compiler's own choice for `fma(p, y, c)` with a
// loop-invariant c is v_fmac (destination = addend), i.e. one register-pair COPY of c per use.
__device__ __forceinline__ double fma_sc(double a, double b, double c) {
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(c));
    return r;
}
// min / max as the bare instruction (a quiet-NaN operand is ignored, like fmin / fmax): the compiler brackets
// fmin / fmax with a canonicalising v_max_f64 x, x per operand -- three instructions for one.  (No signalling NaN
// reaches these: every operand is the result of arithmetic.)
__device__ __forceinline__ double min_nc(double a, double b) {
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ double max_nc(double a, double b) {
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// base[byte_off / sizeof(T)] with a wave-uniform base and a 32-bit byte offset: selects the SGPR-base + VGPR-offset
// addressing of global loads, so a lane spends no instruction per load on a 64-bit address (one shared offset
// register instead of a v_lshl_add_u64 per table row).
template <class T>
__device__ __forceinline__ T ld_off(const T *__restrict__ base, unsigned int byte_off) {
    return *reinterpret_cast<const T *>(reinterpret_cast<const char *>(base) + byte_off);
}

// a / b for NORMAL b (model fluxes, continuum polynomials): v_rcp_f64 (relative error ~2^-24), ONE Newton step
// (-> 2^-48) and one residual correction of the quotient (q + (a - b q) r: relative error 2^-96 before the final
// rounding) -- the compiler's own division sequence without its second Newton step and its v_div_scale /
// v_div_fmas / v_div_fixup wrapping (denormal / overflow scaling, special values), six instructions shorter.
// b = 0 gives NaN where IEEE gives +-inf: either way the walker's sums stop being finite and its log-probability
// ends as -inf.
__device__ __forceinline__ double fast_div(double a, double b) {
    double r = __builtin_amdgcn_rcp(b);
    const double e = fma(-b, r, 1.0);
    r = fma(r, e, r);
    const double q = a * r;
    return fma(fma(-b, q, a), r, q);
}

// ---- the canonical sum over pixels (logprob_kernel.h, phase A) --------------------------------------------------
// Pixel p belongs to slot p mod 1024 = 64 v + l (virtual wave v, lane l); a slot adds its pixels in ascending order
// in one lane's register.  The 1024 slot sums A[v][l] are combined as
//     TREE_l ( sum_{r = 0..3} ( (A[r][l] + A[r+4][l]) + (A[r+8][l] + A[r+12][l]) ) )
// (sum over r serial, TREE = wave_sum's DPP tree).  With 256 threads wave w's lanes hold A[w + 4k][l], k = 0..3: the
// bracket is a lane-local sum; with 512 threads waves r and r + 4 each hold one of its halves.  Every lane publishes
// ONE partial per quantity to LDS (red[wave][lane]) and one wave finishes: one DPP tree per quantity and walker
// instead of one per virtual wave.
template <int VK>
__device__ __forceinline__ double lane_partial(const double (&a)[VK]) {
    static_assert(VK == 2 || VK == 4, "256 or 512 threads");
    if (VK == 4) return (a[0] + a[1]) + (a[2] + a[3]);
    return a[0] + a[1];
}
template <int BT>
__device__ __forceinline__ double reduce_published(const double *red /* [BT / 64][64] */, int lane) {
    double c;
    if (BT == 256) {
        c = ((red[lane] + red[64 + lane]) + red[128 + lane]) + red[192 + lane];
    } else {
        const double c0 = red[lane] + red[256 + lane], c1 = red[64 + lane] + red[320 + lane];
        const double c2 = red[128 + lane] + red[384 + lane], c3 = red[192 + lane] + red[448 + lane];
        c = ((c0 + c1) + c2) + c3;
    }
    return wave_sum(c);
}
// The four pixels a lane takes per trip of a PASS over the model vector (a trip = 4 BT pixels from `base`), and the
// slot accumulator each belongs to: 256 threads -- base + 256 u + tid, accumulator u; 512 threads -- the two pixels
// of elements tid and tid + 512 of the trip (pixel pairs 256 apart, like phase A), accumulator u & 1.
template <int BT>
__device__ __forceinline__ int pass_pixel(int base, int u, int tid) {
    if (BT == 256) return base + u * 256 + tid;
    const int e = (base >> 1) + (u >> 1) * 512 + tid;
    return (((e >> 8) << 9) | (e & 255)) + (u & 1) * 256;
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
