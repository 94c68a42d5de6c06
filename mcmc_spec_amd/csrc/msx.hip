// msx.hip -- gfx950 (CDNA4) kernels + C ABI for mcmc_spec's per-walker log-likelihood path.
//
// What is in here (rows of SURVEY.md §8a; reference = /root/reference/mft6.py):
//   A0  staged grid [nt][ng][nwl] in HBM                      (specs dict, :342-383)
//   A1  isochrone Teff -> logg / luminosity lookup            (get_logg :87-98, get_radius :66-85)
//   A2  nearest-node bracket + bilinear blend                 (get_spec :387-563)
//   A3  Gaussian instrumental broadening, LDS-tiled FIR       (broaden :124-152 -> instrBroadGaussFast)
//   A4  flux scaling + component sum                          (make_composite :687-707,:740-751)
//   A5  contrast magnitudes via per-node band integrals       (:713-741)
//   A6  unresolved photometry via per-node band integrals     (:755-783)
//   A7  CCM89 reddening                                       (extinct :46-64)
//   A8  resample to data pixels, median scale, quadratic fit  (:1169-1174, norm_spec :193-196)
//   A9  chi^2 + combine                                       (chisq :115-122, :1178-1205)
//   f1  prior box + Gaussian terms                            (logprior :1207-1272, logposterior :1459-1470)
//
// Layout of the translation unit (device code in headers, included below in this order):
//   dev_types.h        constants, DevProblem (by-value kernel argument), WalkerDesc (LDS)
//   wave_ops.h         DPP reductions / scans, order-preserving keys
//   blend.h            the per-pixel model arithmetic shared by the fused, linked and pair forms
//   recipe.h           phase 0: gates, isochrone, brackets, weights, prior and band terms
//   median.h           exact median selects (block_median, logbin_median, radix fallback)
//   logprob_kernel.h   the hot kernel and its variants (fused; linked = several workgroups per walker in one launch)
//   pair_kernel.h      the pair form: two walkers of one grid cell per workgroup, one set of row loads
//   staging_kernels.h  CCM89, pair gather, band integrals, broadening, resample, composite, stream copy
//   msx.hip            host context + the C ABI of include/msx.h
//
// Design notes (details in DESIGN.md):
//   * one workgroup per walker; the walker's Npix-long model vector lives in LDS for the exact
//     median (radix select), the 3-term fit and the chi^2 pass; all sums are float64 with a fixed
//     reduction order (wave shuffles, then LDS across waves) -> bit-reproducible run to run.
//   * the resample (A8) only ever touches the two model samples bracketing each data pixel, and
//     those indices are static per dataset, so staging gathers them once into a pixel-major
//     "pair table" pairs[node][pix] = {flux[lo], flux[lo+1]}: every hot-loop load is a 16-byte
//     per-lane, fully coalesced read (1 KiB per wave instruction).
//   * wave = 64 lanes everywhere; no MFMA (nothing here is a GEMM); no CUDA-compat shims.

#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <mutex>
#include <type_traits>
#include <string>
#include <vector>

#include "../../include/msx.h"

// The device code lives in the headers below (one translation unit; every kernel variant is instantiated here).
#include "dev_types.h"
#include "wave_ops.h"
#include "blend.h"
#include "recipe.h"
#include "median.h"
#include "logprob_kernel.h"
#include "pair_kernel.h"
#include "staging_kernels.h"
#include "inpath_kernels.h"

// ================================================================================================
// host side: context + C ABI
// ================================================================================================
struct msx_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    hipDeviceProp_t prop;
    // grid
    int64_t nwl = 0;
    int nt = 0, ng = 0;
    std::vector<double> h_wl, h_teff, h_logg;  // host copies of the wavelength axis and the node lists
    double *d_grid = nullptr, *d_wl = nullptr, *d_kgrid = nullptr, *d_teff = nullptr, *d_logg = nullptr;
    uint8_t *d_present = nullptr;
    bool grid_staged = false;
    // problem
    bool problem_staged = false;
    DevProblem P;
    std::vector<void *> prob_allocs;
    // scratch for host-pointer entry points
    double *d_theta = nullptr, *d_logp = nullptr;
    int32_t *d_status = nullptr;  // points into d_logp's allocation
    void *h_pin = nullptr;        // pinned host staging for the host-pointer entry points
    int64_t cap_walkers = 0;
    double *d_misc = nullptr;  // composite args / desc / small outputs
    double *d_spec = nullptr;
    int64_t cap_spec = 0;
    double *d_opt_flux = nullptr, *d_opt_med = nullptr;
    int32_t *d_opt_chain = nullptr;
    int64_t opt_chains = 0, cap_chain = 0;
    int max_dyn_lds = 0;
    bool pf_ok = false;   // the LDS-staged-statics variants fit (msx_stage_problem)
    bool pf256_ok = false; // ... the 256-thread two-per-CU one (two workgroups of it in a CU's LDS)
    bool use_pf = true;   // MSX_NO_PF=1 in the environment turns them off (A/B measurements)
    bool use_full = true; // MSX_NO_FULL=1: never the FULL (no-clamp) variants of the fused kernel
    int q256 = -1;           // 256-thread launches: the two-per-CU quad-trip variant always (1) / never (0) / up to two walkers per CU (-1); MSX_Q256
    bool force_sh2 = false;  // MSX_NO_SH2=0: binaries take the <= 128-VGPR variant even with a CU to themselves (A/B measurements)
    bool zero_copy = true;   // host-pointer entry point without copy commands; MSX_ZERO_COPY=0 restores them
    int64_t pad_lds = 0;     // MSX_PAD_LDS=bytes: extra dynamic LDS per workgroup (occupancy experiments only)
    bool model_in_global = false;
    // RCCL all-gather of log-probabilities: communicator + its own stream + per-slot events
    void *rccl_comm = nullptr;
    hipStream_t comm_stream = nullptr;
    hipEvent_t ev_ready = nullptr;
    hipEvent_t ev_done[4] = {nullptr, nullptr, nullptr, nullptr};
    int comm_world = 0, comm_rank = 0;
    // LOOPBACK group (msx_comm_init_loopback): `comm_world` contexts of this process stand for the ranks of one job;
    // the all-gather of the sharded sampler becomes same-process device copies between their gathered vectors, driven
    // in lock-step by msx_sampler_enqueue_group.  No RCCL: the rank >= 1 paths run on a one-GPU box (tests).
    std::vector<msx_ctx *> loop_peers;   // all members in rank order (this context at [comm_rank]); empty = none
    hipEvent_t loop_eval_done = nullptr; // this rank's block of log p(q) is in its gathered vector
    hipEvent_t loop_copied = nullptr;    // this rank has copied every peer's block (peers may overwrite theirs)
    // Scratch rows (walkers per sub-batch), sized once at msx_stage_problem and only for the forms that can run on
    // the staged spectrum: model vectors of the GM variants (> 17,152 pixels) and of the linked form's scratch-row exit
    // (2..8 segments of 8192 pixels), the segments' partials and arrival counters.  No launch allocates.
    double *d_model_scratch = nullptr;
    SegPart *d_segparts = nullptr;  // linked form: [scratch_rows][segments]
    unsigned long long *d_seg_flag = nullptr;  // linked form: [scratch_rows] arrival counters (a multiple of 2 x segments between launches), + the poison word
    int nseg = 1;                   // segments of the staged spectrum (8192 pixels each)
    int64_t scratch_rows = 0;       // 0 = neither form applies: launches are never cut into sub-batches
    int32_t path = 0;               // MSX_PATH_AUTO / _FUSED / _PAIR / _LINKED (msx_set_path)
    // in-path broadening (inpath_kernels.h; msx_set_broadening, MSX_PATH_INPATH): the raw window rows kept by
    // msx_broaden_grid, the taps' parameters, and the form's scratch (sized at msx_stage_problem)
    int32_t broaden_placement = 0;  // MSX_BROADEN_STAGING / MSX_BROADEN_IN_PATH (takes effect at the next msx_broaden_grid)
    double *d_raw_win = nullptr;    // [nt * ng][raw_n]: the window before the broadening
    int64_t raw_i0 = 0, raw_n = 0;
    int raw_lx = 0;
    double raw_dx = 0.0, raw_sigma = 0.0;
    InpathRec *d_inp_rec = nullptr;
    double *d_inp_tmp = nullptr, *d_inp_given = nullptr;
    const int64_t *d_pix_lo = nullptr;
    int64_t inp_rows = 0, inp_gstride = 0;  // walkers per sub-batch (0: the staged problem has no in-path form)
    // pair form (pair_kernel.h): binaries of <= 4096 pixels with the register-resident recipe
    int64_t pair_rows = 0;          // walkers per sub-batch = capacity of the planner's item lists (0: no pair form here)
    // MSX_PATH_AUTO takes the pair form from this many walkers on (MSX_PAIR_MIN; 0 = never); set per problem by
    // msx_stage_problem, with the measurements behind the rule
    int64_t pair_min_walkers = 4096;
    int32_t *d_pair_plan = nullptr; // the planner's output (pair_kernel.h: header, pairs, singles)
    PairItem *d_pair_items = nullptr;   // ... and its items: the pairs' recipes, [pair_rows / 2]
    PairRec *d_pair_singles = nullptr;  // ... the singles', [pair_rows]
    // {pairs, singles} of the planner's last run, written by the device into host memory and read here WITHOUT waiting
    // for it (so possibly a launch or two old): MSX_PATH_AUTO's only evidence of whether pairing pays (pair_worth_it)
    int32_t *h_pair_stats = nullptr;
    int64_t pair_auto_launches = 0;
    int32_t linked = -1;            // MSX_LINKED: -1 = automatic (walkers x segments <= #CUs), 0 never, 1 whenever possible
    bool linked_poisoned = false;   // a hand-over of the linked form timed out on this context (seen by a synchronous
                                    // entry point): MSX_PATH_AUTO takes the fused form until the problem is staged again
    bool recipe_fast = false;       // the register-resident recipe applies (small tables)
    unsigned char *d_recipe_block = nullptr;  // ... and its tables in one block (dev_types.h), freed with the problem
    struct SamplerRun *smp = nullptr;  // device-resident sampler in flight (msx_sampler_begin .. _end)
    bool smp_overlap_launch = false;   // the launch being queued is a half-step of an overlapped run: fused form, bit 20
    bool probe_launch = false;         // ... is msx_probe_launch's: the kernel leaves clock stamps (bit 21)
    int32_t last_form = 0;             // MSX_FORM_* of the last launch queued (msx_last_form)
    int32_t smp_overlap_policy = -1;   // msx_sampler_policy: -1 = overlap half-steps when the rule allows, 0 = never
    int32_t store_dtype = MSX_STORE_F64;  // msx_set_grid_storage: what the NEXT msx_stage_problem builds the R table in
    bool store_f32 = false;               // ... and what the staged problem holds
};
static void sampler_free(msx_ctx *c);

namespace {

// RCCL is resolved at run time from the copy PyTorch-ROCm already mapped (same SONAME librccl.so.1 as
// /opt/rocm's), so the process never holds two RCCL instances; nothing is linked at build time.
struct RcclUniqueId { char internal[128]; };
struct RcclApi {
    void *handle = nullptr;
    int (*GetUniqueId)(RcclUniqueId *) = nullptr;
    int (*CommInitRank)(void **, int, RcclUniqueId, int) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    bool ok = false;
};
RcclApi &rccl() {
    static RcclApi api;
    if (api.handle) return api;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *nm : names) {
        api.handle = dlopen(nm, RTLD_NOW | RTLD_NOLOAD);
        if (api.handle) break;
    }
    for (int i = 0; !api.handle && i < 3; ++i) api.handle = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
    if (!api.handle) return api;
    api.GetUniqueId = (int (*)(RcclUniqueId *))dlsym(api.handle, "ncclGetUniqueId");
    api.CommInitRank = (int (*)(void **, int, RcclUniqueId, int))dlsym(api.handle, "ncclCommInitRank");
    api.AllGather = (int (*)(const void *, void *, size_t, int, void *, hipStream_t))dlsym(api.handle, "ncclAllGather");
    api.CommDestroy = (int (*)(void *))dlsym(api.handle, "ncclCommDestroy");
    api.GetErrorString = (const char *(*)(int))dlsym(api.handle, "ncclGetErrorString");
    api.ok = api.GetUniqueId && api.CommInitRank && api.AllGather && api.CommDestroy && api.GetErrorString;
    return api;
}
constexpr int kNcclFloat64 = 8;  // ncclFloat64 / ncclDouble (rccl.h)

int raise_dynamic_lds_limits(msx_ctx *c);  // (defined next to the launchers)

int fail(msx_ctx *c, int code, const std::string &msg) {
    if (c) c->err = msg;
    return code;
}

#define HIP_TRY(ctx, expr)                                                                         \
    do {                                                                                           \
        hipError_t e__ = (expr);                                                                   \
        if (e__ != hipSuccess)                                                                     \
            return fail(ctx, MSX_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e__));     \
    } while (0)

template <typename T>
int dev_alloc_copy(msx_ctx *c, std::vector<void *> *track, const T *host, int64_t count, T **out) {
    *out = nullptr;
    if (count <= 0) {
        // keep kernels simple: always a valid pointer
        HIP_TRY(c, hipMalloc((void **)out, 16));
        if (track) track->push_back(*out);
        return MSX_OK;
    }
    HIP_TRY(c, hipMalloc((void **)out, sizeof(T) * count));
    if (track) track->push_back(*out);
    HIP_TRY(c, hipMemcpy(*out, host, sizeof(T) * count, hipMemcpyHostToDevice));
    return MSX_OK;
}

void free_problem(msx_ctx *c) {
    sampler_free(c);  // a sampler in flight holds pointers into the problem's tables
    for (void *p : c->prob_allocs) (void)hipFree(p);
    c->prob_allocs.clear();
    c->problem_staged = false;
    if (c->d_opt_flux) (void)hipFree(c->d_opt_flux);
    if (c->d_opt_med) (void)hipFree(c->d_opt_med);
    c->d_opt_flux = c->d_opt_med = nullptr;
    c->opt_chains = 0;
    if (c->h_pair_stats) (void)hipHostFree(c->h_pair_stats);
    c->h_pair_stats = nullptr;
    void *sp[] = {c->d_model_scratch, c->d_segparts, c->d_seg_flag, c->d_pair_plan, c->d_pair_items, c->d_pair_singles,
                  c->d_inp_rec, c->d_inp_tmp, c->d_inp_given};
    for (void *p : sp)
        if (p) (void)hipFree(p);
    c->d_segparts = nullptr; c->d_seg_flag = nullptr; c->d_model_scratch = nullptr; c->scratch_rows = 0;
    c->d_pair_plan = nullptr; c->d_pair_items = nullptr; c->d_pair_singles = nullptr; c->pair_rows = 0;
    c->linked_poisoned = false;
    c->d_inp_rec = nullptr; c->d_inp_tmp = c->d_inp_given = nullptr; c->d_pix_lo = nullptr; c->inp_rows = 0;
}

void free_grid(msx_ctx *c) {
    void *ptrs[] = {c->d_grid, c->d_wl, c->d_kgrid, c->d_teff, c->d_logg, c->d_present};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    c->d_grid = c->d_wl = c->d_kgrid = c->d_teff = c->d_logg = nullptr;
    c->d_present = nullptr;
    if (c->d_raw_win) (void)hipFree(c->d_raw_win);
    c->d_raw_win = nullptr; c->raw_n = 0;
    c->grid_staged = false;
}

int conv_taps(double mean_wl, double dx, double resolution, double maxsig, double *sigma_out, int *lx_out) {
    // pyasl.instrBroadGaussFast: fwhm = mean(wl)/R; sigma = fwhm/(2 sqrt(2 ln 2)); broadGaussFast:
    // lx = int(((sigma*maxsig)/dx)*2.0) + 1
    const double fwhm = 1.0 / resolution * mean_wl;
    const double sigma = fwhm / (2.0 * sqrt(2.0 * log(2.0)));
    const int lx = (int)(((sigma * maxsig) / dx) * 2.0) + 1;
    *sigma_out = sigma;
    *lx_out = lx;
    return lx;
}

int check_even_spacing(msx_ctx *c, const double *wl, int64_t n) {
    // broadGaussFast: abs(max(dxs) - min(dxs)) > mean(dxs)*1e-6 -> error
    double mx = -INFINITY, mn = INFINITY, sum = 0.0;
    for (int64_t i = 1; i < n; ++i) {
        const double d = wl[i] - wl[i - 1];
        mx = d > mx ? d : mx;
        mn = d < mn ? d : mn;
        sum += d;
    }
    if (fabs(mx - mn) > (sum / (double)(n - 1)) * 1e-6)
        return fail(c, MSX_ERR_RANGE, "broaden: the wavelength axis is not equidistant");
    return MSX_OK;
}

int launch_conv(msx_ctx *c, const double *d_in, int64_t in_stride, double *d_tmp, int64_t rows, int64_t n, int lx,
                double dx, double sigma) {
    const size_t lds = sizeof(double) * ((size_t)lx + kConvTile + lx - 1);
    if (lds > 150 * 1024) return fail(c, MSX_ERR_RANGE, "broaden: kernel too long for the LDS tile");
    if ((int)lds > 64 * 1024)
        HIP_TRY(c, hipFuncSetAttribute((const void *)broaden_conv_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)lds));
    dim3 g((unsigned)((n + kConvTile - 1) / kConvTile), (unsigned)rows);
    hipLaunchKernelGGL(broaden_conv_kernel, g, dim3(256), lds, c->stream, d_in, in_stride, d_tmp, n, n, lx, dx, sigma);
    HIP_TRY(c, hipGetLastError());
    return MSX_OK;
}

// A synchronous entry point has seen the walkers' statuses: MSX_W_HANDOVER anywhere means the linked form's flags are
// no longer trustworthy on this context (the device-side poison word says the same to every later linked launch).
// (The pair form's bounded wait -- a spill row's lease never granted -- reports the same status: the launch is over when a
// synchronous entry point reads it, nobody holds a row, so the leases are cleared for the launches that follow.)
void note_handover(msx_ctx *c, const int32_t *status, int64_t n) {
    for (int64_t i = 0; i < n; ++i)
        if (status[i] == MSX_W_HANDOVER) {
            c->linked_poisoned = true;
            if (c->d_pair_plan) (void)hipMemsetAsync(c->d_pair_plan + kPairHdrInts, 0, sizeof(int32_t) * kPairSpillRows, c->stream);
            return;
        }
}

int pick_block(const msx_ctx *c, int64_t n, int64_t npix) {
    // Measured at 4096 px: up to one walker per CU, 512 threads owning the CU with the pixel statics in
    // LDS; up to 2 per CU, 512 threads sharing the CU two by two (<= 128 VGPRs); beyond, 256 threads three per CU
    // (512 / 1024 / 2048 walkers: 28.3 / 48.2 / 87.8 us shared-512 against 29.2 / 47.2 / 77.6 us with 256 threads).
    // Long spectra (model vector > half the LDS): 512 threads, one workgroup per CU.
    // The choice only affects speed: every variant sums in the same order (see phase A), so a walker's value
    // has the same bits whichever one evaluates it.
    const int64_t cus = c->prop.multiProcessorCount > 0 ? c->prop.multiProcessorCount : 256;
    if (npix >= 8192) return 512;
    if (n <= cus) return 512;
    if (npix <= 2048) return 256;  // short spectra (512 walkers x 1194 px: 17.8 us with 256 threads, 20.2 shared-512)
    if (n <= 2 * cus) return 512;
    return 256;
}

}  // namespace

// ---- launchers ------------------------------------------------------------------------------------------------
namespace {

struct LaunchArgs {
    const double *theta;
    double *logp;
    int32_t *status;
    int64_t n;
    int32_t ndim, mode;
    hipStream_t s;
    int niso_nt, ng_mode_fast;
};

// A copy of the staged problem whose per-walker pointers start at walker `off` of the caller's batch (sub-batches)
DevProblem problem_at(const DevProblem &P0, int64_t off, int mode, int ndim) {
    DevProblem P = P0;
    if (off == 0) return P;
    if (P.opt_chain) P.opt_chain += off;
    if (mode == MSX_MODE_OPT_INIT) { P.opt_flux += off * P.npix; P.opt_med += off; }
    if (P.smp_on) {
        P.smp_sidx += off; P.smp_cidx += off; P.smp_partner += off;
        P.smp_zz += off; P.smp_zfac += off; P.smp_logu += off; P.smp_rec += off;
        P.smp_q += off * ndim;
    }
    return P;
}

// a 512-thread workgroup of a launch of n walkers has its CU to itself (long spectra: one per CU anyway)
bool owns_cu(const msx_ctx *c, int64_t n) {
    return n <= c->prop.multiProcessorCount || sizeof(double) * (size_t)c->P.npix > 70 * 1024;
}
// ... and takes the variant that keeps u and the data flux in LDS for the chi^2 pass (PF)
bool takes_pf(const msx_ctx *c, int64_t n) {
    return !c->model_in_global && !(c->P.nspec == 2 && c->force_sh2 && sizeof(double) * (size_t)c->P.npix <= 70 * 1024) &&
           owns_cu(c, n) && c->pf_ok && c->use_pf;
}

// ---- the variants of logprob_kernel, as a TABLE: the launcher, msx_launch_info (what bench.py prints as the roofline's
// kernel) and the dynamic-LDS limits all read this one list -- nobody mirrors the choice -------------------------------
struct Variant {
    const void *fn;
    int ns, threads;
    bool gm, sh, pf, lk, r32;
    int full;  // FULL bits of the variant (logprob_kernel.h): 1 blend, 2 chi^2 pass
    const char *what;
    bool given = false;  // model values given (in-path broadening): the one such entry at the table's end
};
#define MSX_V(NS_, T_, GM_, SH_, PF_, LK_, WHAT_) \
    {(const void *)logprob_kernel<NS_, 2, T_, GM_, SH_, PF_, LK_>, NS_, T_, GM_, SH_, PF_, LK_, false, 0, WHAT_}
#define MSX_V32(T_, SH_, PF_, WHAT_) \
    {(const void *)logprob_kernel<2, 2, T_, false, SH_, PF_, false, true>, 2, T_, false, SH_, PF_, false, true, 0, WHAT_}
#define MSX_VF(T_, SH_, PF_, F_, WHAT_) \
    {(const void *)logprob_kernel<2, 2, T_, false, SH_, PF_, false, false, F_>, 2, T_, false, SH_, PF_, false, false, F_, WHAT_}
constexpr int kFull256 = 3;  // (measured against 1 and 2 as well: profiles/r4_ab_full.txt)
const Variant kVariants[] = {
    MSX_V(2, 256, false, false, false, false, "three workgroups per CU"),
    MSX_V(2, 256, false, true, false, false, "two per CU, four pixels per lane and trip"),
    MSX_V(2, 256, false, true, true, false, "two per CU, u / flux staged in LDS during the recipe, four pixels per lane and trip"),
    MSX_V(2, 512, false, false, false, false, "one workgroup per CU, four pixels per lane and trip"),
    MSX_V(2, 512, false, true, false, false, "<= 128 VGPRs: two workgroups fit a CU; rows one star at a time"),
    MSX_V(2, 512, false, false, true, false, "one workgroup per CU, u / flux staged in LDS during the recipe, four pixels per lane and trip"),
    MSX_V(3, 256, false, false, false, false, "three workgroups per CU"),
    MSX_V(3, 512, false, false, false, false, "one workgroup per CU, four pixels per lane and trip"),
    MSX_V(3, 512, false, false, true, false, "one workgroup per CU, u / flux staged in LDS during the recipe, four pixels per lane and trip"),
    MSX_V(2, 512, false, false, false, true, "one workgroup per walker and 8192-pixel segment; partial sums and histogram counters exchanged inside the launch; the segment's data flux staged in LDS; four pixels per lane and trip"),
    MSX_V(3, 512, false, false, false, true, "one workgroup per walker and 8192-pixel segment; partial sums and histogram counters exchanged inside the launch; the segment's data flux staged in LDS; four pixels per lane and trip"),
    MSX_V(2, 512, true, false, false, false, "model vector in global memory (spectra beyond the LDS), sub-batched"),
    MSX_V(3, 512, true, false, false, false, "model vector in global memory (spectra beyond the LDS), sub-batched"),
    // the fused binary variants once more over the FLOAT32 copy of the R table (msx_set_grid_storage(MSX_STORE_F32))
    MSX_V32(256, false, false, "three workgroups per CU; R table stored in float32"),
    MSX_V32(256, true, false, "two per CU, four pixels per lane and trip; R table stored in float32"),
    MSX_V32(256, true, true, "two per CU, u / flux staged in LDS during the recipe, four pixels per lane and trip; R table stored in float32"),
    MSX_V32(512, false, false, "one workgroup per CU, four pixels per lane and trip; R table stored in float32"),
    MSX_V32(512, true, false, "<= 128 VGPRs: two workgroups fit a CU; rows one star at a time; R table stored in float32"),
    MSX_V32(512, false, true, "one workgroup per CU, u / flux staged in LDS during the recipe, four pixels per lane and trip; R table stored in float32"),
    // the binary variants once more for spectra that fill their trips exactly (FULL: no clamps, no validity selects; bit 0
    // in the blend, bit 1 in the chi^2 pass).  Same box, general / FULL, us per batch of 4096 px -- 256 threads, both bits:
    // 512 walkers 24.2 / 23.1, 1,024: 39.5 / 37.8, 2,048: 63.3 / 60.4, 2,304: 68.9 / 66.7.  512 threads, one workgroup per
    // CU (the 256-walker headline), per step: both bits 14.35 -> 14.55 (slower), the blend's alone 14.67 -> 14.96 (slower:
    // without the clamps the scheduler orders the trip's loads differently), the chi^2 pass's alone 14.63 -> 14.36.
    MSX_VF(256, false, false, kFull256, "three workgroups per CU; whole trips, no clamps"),
    MSX_VF(256, true, false, kFull256, "two per CU, four pixels per lane and trip; whole trips, no clamps"),
    MSX_VF(256, true, true, kFull256, "two per CU, u / flux staged in LDS during the recipe, four pixels per lane and trip; whole trips, no clamps"),
    MSX_VF(512, false, true, 2, "one workgroup per CU, u / flux staged in LDS during the recipe, four pixels per lane and trip; whole trips, no clamps in the chi^2 pass"),
    MSX_VF(512, true, false, 2, "<= 128 VGPRs: two workgroups fit a CU; rows one star at a time; whole trips, no clamps in the chi^2 pass"),
    {(const void *)logprob_kernel<2, 2, 512, false, false, true, false, true, 2>, 2, 512, false, false, true, false, true, 2,
     "float32-stored grid table R; one workgroup per CU, u / flux staged in LDS; whole trips, no clamps in the chi^2 pass"},
    // the linked form for whole-trip segments: no clamps in the segment's chi^2 pass and candidates' gather
    {(const void *)logprob_kernel<2, 2, 512, false, false, false, true, false, 2>, 2, 512, false, false, false, true, false, 2,
     "linked: one workgroup per walker and 8192-pixel segment; whole trips, no clamps in the chi^2 pass"},
    // in-path broadening (inpath_kernels.h): the model values are given, the blend is compiled out
    {(const void *)logprob_kernel<2, 2, 512, false, false, false, false, false, 0, true>, 2, 512, false, false, false, false, false, 0,
     "model values given by the in-path broadening kernels; four pixels per lane and trip", true},
};
#undef MSX_V
#undef MSX_V32
#undef MSX_VF
const Variant *find_variant(int ns, int threads, bool gm, bool sh, bool pf, bool lk, bool r32 = false, int full = 0) {
    for (const Variant &v : kVariants)
        if (!v.given && v.ns == ns && v.threads == threads && v.gm == gm && v.sh == sh && v.pf == pf && v.lk == lk && v.r32 == r32 && v.full == full) return &v;
    return nullptr;
}
const Variant *given_variant() {
    for (const Variant &v : kVariants)
        if (v.given) return &v;
    return nullptr;
}
struct VariantChoice {
    const Variant *v;
    size_t dyn_lds;
};
// Which variant a launch of n walkers with workgroups of B threads takes (LK: the linked form, one workgroup per walker
// and segment; else the fused kernel in the variant the table in pick_block() names), and its dynamic LDS.
VariantChoice choose_variant(const msx_ctx *c, const DevProblem &P, int64_t n, int B, bool shared512, bool LK) {
    if (P.given) return {given_variant(), sizeof(double) * (size_t)P.npix + (size_t)c->pad_lds};  // (in-path form: launch_inpath)
    // dynamic LDS: the model vector (linked: one segment of it, and the segment's data flux behind it)
    const size_t lds = LK ? sizeof(double) * (size_t)(2 * kSegElems) + sizeof(double2) * (size_t)kSegElems
                          : sizeof(double) * (size_t)P.npix + (size_t)c->pad_lds;
    // PF adds u and the data flux in the tables' pair layout
    const size_t lds_pf = sizeof(double) * (size_t)((P.npix + 1) & ~1ll) + 2 * sizeof(double2) * (size_t)P.npair;
    const int ns = P.nspec == 2 ? 2 : 3;
    if (LK) {
        const bool full_lk = ns == 2 && c->use_full && P.npix == 2 * P.npair && P.npair % 1024 == 0;
        return {find_variant(ns, 512, false, false, false, true, false, full_lk ? 2 : 0), lds};
    }
    // spectra longer than the LDS: the model vector lives in the global scratch (the kernel writes it there itself)
    if (c->model_in_global) return {find_variant(ns, 512, true, false, false, false), 0};
    // pixel statics staged in LDS (PF): 512-thread workgroups that own their CU and whose 3 npix doubles fit
    const bool own_cu = owns_cu(c, n);
    // binaries between one and two walkers per CU: the <= 128-VGPR variant, two workgroups per CU.  With a CU to
    // itself a workgroup takes the quad-walking variants (pixel statics staged in LDS when they fit): 256 walkers x
    // 4096 px 16.7-16.9 us against 17.0-17.1 for the <= 128-VGPR variant; MSX_NO_SH2=0 in the environment forces the latter
    const bool sh2 = B == 512 && P.nspec == 2 && lds <= 70 * 1024 && c->force_sh2;
    const bool pf = B == 512 && !shared512 && !sh2 && takes_pf(c, n);
    const bool sh = B == 512 && !pf && (sh2 || shared512 || !own_cu);
    const bool r32 = c->store_f32;  // (msx_stage_problem has checked that the problem has such variants: fused binaries)
    if (ns == 2 && r32) {
        const bool q256 = c->q256 > 0 || (c->q256 < 0 && n <= 2 * (int64_t)c->prop.multiProcessorCount);
        if (B == 256 && q256 && c->pf256_ok && c->use_pf) return {find_variant(2, 256, false, true, true, false, true), lds_pf};
        if (B == 256 && q256) return {find_variant(2, 256, false, true, false, false, true), lds};
        if (B == 256) return {find_variant(2, 256, false, false, false, false, true), lds};
        if (pf && c->use_full && P.npix == 2 * P.npair && P.npair % 1024 == 0)
            return {find_variant(2, 512, false, false, true, false, true, 2), lds_pf};
        if (pf) return {find_variant(2, 512, false, false, true, false, true), lds_pf};
        if (sh) return {find_variant(2, 512, false, true, false, false, true), lds};
        return {find_variant(2, 512, false, false, false, false, true), lds};
    }
    // FULL: no pad pixels and whole trips (a trip of the 256-thread variants is 512 elements = 1024 pixels): the variants
    // without clamps (MSX_NO_FULL=1 in the environment keeps the general ones: A/B measurements)
    const bool full = ns == 2 && B == 256 && c->use_full && P.npix == 2 * P.npair && P.npair % 512 == 0;
    if (full) {
        const bool q256 = c->q256 > 0 || (c->q256 < 0 && n <= 2 * (int64_t)c->prop.multiProcessorCount);
        if (q256 && c->pf256_ok && c->use_pf) return {find_variant(2, 256, false, true, true, false, false, kFull256), lds_pf};
        if (q256) return {find_variant(2, 256, false, true, false, false, false, kFull256), lds};
        return {find_variant(2, 256, false, false, false, false, false, kFull256), lds};
    }
    if (ns == 2 && B == 512 && pf && c->use_full && P.npix == 2 * P.npair && P.npair % 1024 == 0)
        return {find_variant(2, 512, false, false, true, false, false, 2), lds_pf};
    if (ns == 2 && B == 512 && sh && c->use_full && P.npix == 2 * P.npair && P.npair % 1024 == 0)
        return {find_variant(2, 512, false, true, false, false, false, 2), lds};
    if (ns == 2) {
        // 256 threads, at most two walkers per CU (config 5's 512 x 1194 px): the variant compiled for two workgroups per
        // CU has the registers for quad trips (16.0 against 16.3 us); beyond, three per CU matter more (MSX_Q256=1 / 0 forces)
        const bool q256 = c->q256 > 0 || (c->q256 < 0 && n <= 2 * (int64_t)c->prop.multiProcessorCount);
        // (... with u and the data flux staged in LDS when two such workgroups still fit a CU)
        if (B == 256 && q256 && c->pf256_ok && c->use_pf) return {find_variant(2, 256, false, true, true, false), lds_pf};
        if (B == 256 && q256) return {find_variant(2, 256, false, true, false, false), lds};
        if (B == 256) return {find_variant(2, 256, false, false, false, false), lds};
        if (pf) return {find_variant(2, 512, false, false, true, false), lds_pf};
        if (sh) return {find_variant(2, 512, false, true, false, false), lds};   // two workgroups per CU
        return {find_variant(2, 512, false, false, false, false), lds};
    }
    // triples: twelve corners do not fit the shared variant's 128 VGPRs -- it is the plain one
    if (B == 256) return {find_variant(3, 256, false, false, false, false), lds};
    if (pf) return {find_variant(3, 512, false, false, true, false), lds_pf};
    return {find_variant(3, 512, false, false, false, false), lds};
}

// One launch of the chosen variant over A.n walkers.
template <bool LK>
int launch_logprob(msx_ctx *c, const DevProblem &P, const LaunchArgs &A, int B, bool shared512) {
    const VariantChoice ch = choose_variant(c, P, A.n, B, shared512, LK);
    if (!ch.v) return fail(c, MSX_ERR_STATE, "no kernel variant for this launch");
    // (linked: block = (walker / 8) * 8 segments + segment * 8 + walker % 8, see the kernel)
    const dim3 g((unsigned)(LK ? ((A.n + 7) & ~7ll) * c->nseg : A.n));
    // the kernel's arguments, in its own order (the leading 14 dwords arrive preloaded in SGPRs: logprob_kernel)
    const double *a_theta = P.smp_on ? (const double *)P.smp_coords : A.theta;
    const unsigned char *a_rblk = (const unsigned char *)c->d_recipe_block;
    int a_niso_nt = A.niso_nt, a_word = A.ng_mode_fast;
    int64_t a_n = A.n;
    double a_tmin = P.tmin, a_tmax = P.tmax;
    const SmpRec *a_rec = P.smp_rec;
    DevProblem a_P = P;
    double *a_logp = A.logp;
    int32_t *a_status = A.status;
    void *args[] = {&a_theta, &a_rblk, &a_niso_nt, &a_word, &a_n, &a_tmin, &a_tmax, &a_rec, &a_P, &a_logp, &a_status};
    HIP_TRY(c, hipLaunchKernel(ch.v->fn, g, dim3((unsigned)ch.v->threads), args, ch.dyn_lds, A.s));
    return MSX_OK;
}

// The in-path broadening form over A.n <= inp_rows walkers (inpath_kernels.h): recipe -> composite of the raw window rows,
// convolved -> edge patches, reddening, resample -> logprob_kernel<GIVEN>.
int launch_inpath(msx_ctx *c, const DevProblem &P, const LaunchArgs &A) {
    const int64_t m = A.n, nwin = c->raw_n;
    hipLaunchKernelGGL(inpath_recipe_kernel, dim3((unsigned)((m + kPlanThreads - 1) / kPlanThreads)), dim3(kPlanThreads), 0, A.s, A.theta,
                       (const unsigned char *)c->d_recipe_block, A.niso_nt, A.ng_mode_fast, m, P.tmin, P.tmax, c->d_inp_rec, P);
    HIP_TRY(c, hipGetLastError());
    const size_t lds = sizeof(double) * ((size_t)c->raw_lx + ((size_t)(kConvTile + c->raw_lx - 1) * 5) / 4 + 2);
    // (the tile fits: checked, and the kernel's dynamic-LDS limit raised, at msx_stage_problem -- a launch neither allocates nor
    // changes function attributes, so that it can be captured like the others)
    hipLaunchKernelGGL(inpath_conv_kernel, dim3((unsigned)((nwin + kConvTile - 1) / kConvTile), (unsigned)m), dim3(256), lds, A.s,
                       c->d_raw_win, nwin, c->d_inp_rec, c->d_inp_tmp, nwin, nwin, c->raw_lx, c->raw_dx, c->raw_sigma);
    HIP_TRY(c, hipGetLastError());
    hipLaunchKernelGGL(inpath_resample_kernel, dim3((unsigned)((c->inp_gstride + 255) / 256), (unsigned)m), dim3(256), 0, A.s, c->d_inp_tmp,
                       nwin, nwin, c->raw_i0, c->d_inp_rec, c->d_pix_lo, P.pix_t, c->d_kgrid, (int64_t)P.npix, c->d_inp_given, c->inp_gstride);
    HIP_TRY(c, hipGetLastError());
    DevProblem Pg = P;
    Pg.given = c->d_inp_given;
    Pg.given_stride = c->inp_gstride;
    return launch_logprob<false>(c, Pg, A, 512, false);
}

// Does MSX_PATH_AUTO take the linked form for n walkers (of a problem and mode that have one)?  While every workgroup
// gets a CU of its own: the walker's workgroups wait for each other.
bool auto_takes_linked(const msx_ctx *c, int64_t n) {
    const int64_t cus = c->prop.multiProcessorCount > 0 ? c->prop.multiProcessorCount : 256;
    return c->d_seg_flag != nullptr && c->recipe_fast && !c->P.no_spectrum && !c->linked_poisoned && c->linked != 0 &&
           c->path != MSX_PATH_FUSED && c->path != MSX_PATH_PAIR && (c->linked > 0 || c->path == MSX_PATH_LINKED || n * c->nseg <= cus);
}

// Does MSX_PATH_AUTO take the pair form for the next large batch?  A walker the planner cannot pair costs the pair kernel
// a whole workgroup (two per CU: 0.042 us per item at 16,384 walkers) where the fused kernel runs three per CU (0.027 us
// per walker), so pairing pays while singles < 0.75 pairs -- an ensemble in a handful of grid cells, the normal state of
// a chain -- and does not for one spread over the grid (burn-in from a wide start: 16,384 walkers 521 us against 440).
// The evidence is the planner's count of its last run, which the device leaves in host memory; it is read without
// synchronising, so it may lag.  While it says "spread" the fused kernel runs, and every 32nd qualifying launch goes
// through the pair form anyway to look again.  (The choice is frozen into a captured hipGraph like any launch
// parameter.)  Values never depend on it.
bool pair_worth_it(msx_ctx *c, bool peek = false) {
    const int32_t np = ((volatile int32_t *)c->h_pair_stats)[0], ns = ((volatile int32_t *)c->h_pair_stats)[1];
    const bool pays = 4 * (int64_t)ns < 3 * (int64_t)np;
    if (peek) return pays;  // (msx_launch_info: what the next launch would take, without counting as one)
    return pays || (++c->pair_auto_launches % 32) == 0;
}

// Which FORM of the path a launch of n walkers in `mode` takes (msx_logprob_batch_dev decides with this; msx_launch_info
// and msx_bytes_per_eval ask with peek = true).  err != MSX_OK: an explicit msx_set_path that the staged problem cannot take.
struct FormChoice {
    bool linked = false, pair = false, inpath = false;
    int err = MSX_OK;
    const char *msg = "";
};
FormChoice decide_form(msx_ctx *c, int64_t n, int mode, bool peek) {
    FormChoice f;
    const DevProblem &Pc = c->P;
    const bool fast = c->recipe_fast;
    const bool lp_mode = mode == MSX_MODE_LOGLIKE || mode == MSX_MODE_LOGPOST || mode == MSX_MODE_CHISQ;
    // linked (logprob_kernel<..., LK>): few walkers x long spectrum -- one workgroup per (walker, 8192-pixel segment); the
    // walker's workgroups exchange their segments' partial sums and counters inside the kernel and each makes the chi^2 /
    // median pass over its own segment, so a launch of <= #CUs / segments walkers uses segments x as many CUs and every
    // workgroup's chain of latencies is 8192 pixels long.  Only the likelihood / posterior / chi^2 modes of a problem
    // with a spectrum term and the register-resident recipe.  MSX_PATH_AUTO takes it while walkers x segments <= #CUs
    // (one workgroup per CU: they wait for each other).  MSX_LINKED=0 / 1 in the environment: never / whenever possible.
    // A context on which a meeting has once timed out is POISONED until the problem is staged again: AUTO takes the
    // fused form, an explicit MSX_PATH_LINKED is refused (and the kernel itself fails every walker, for callers of
    // the device entry point who never looked at the statuses).
    const bool can_link = c->d_seg_flag != nullptr && fast && !Pc.no_spectrum && lp_mode;
    f.linked = can_link && auto_takes_linked(c, n) && !c->smp_overlap_launch;
    if (c->path == MSX_PATH_LINKED) {
        if (!can_link) { f.err = MSX_ERR_STATE; f.msg = "msx_set_path(LINKED): needs a spectrum of 2..8 segments of 8192 pixels, the register-resident recipe and a likelihood / posterior / chi^2 mode"; return f; }
        if (c->linked_poisoned) { f.err = MSX_ERR_STATE; f.msg = "msx_set_path(LINKED): a hand-over timed out on this context (MSX_W_HANDOVER); stage the problem again"; return f; }
        f.linked = true;
    }
    // pair (pair_kernel.h): many walkers -- two walkers of one grid cell per workgroup share one set of row loads
    const bool can_pair = c->pair_rows > 0 && fast && !Pc.smp_on && Pc.nspec == 2 && lp_mode;
    f.pair = can_pair && n >= c->pair_min_walkers && pair_worth_it(c, peek);
    if (c->path == MSX_PATH_FUSED || c->path == MSX_PATH_LINKED) f.pair = false;
    if (c->path == MSX_PATH_PAIR) {
        if (!can_pair) { f.err = MSX_ERR_STATE; f.msg = "msx_set_path(PAIR): needs a binary of <= 4096 pixels, the register-resident recipe and a likelihood / posterior / chi^2 mode"; return f; }
        f.pair = true;
    }
    if (c->probe_launch) f.pair = false;  // (the pair form carries no clock stamps)
    if (c->store_f32) {  // float32-stored R table: the fused variants compiled for it, nothing else
        if (c->path == MSX_PATH_PAIR || c->path == MSX_PATH_LINKED) { f.err = MSX_ERR_STATE; f.msg = "float32 grid storage (msx_set_grid_storage): the fused form only"; return f; }
        f.pair = false; f.linked = false;
    }
    if (f.pair) f.linked = false;
    // in-path broadening (inpath_kernels.h): never taken by MSX_PATH_AUTO -- the per-node placement is the reference's
    if (c->path == MSX_PATH_INPATH) {
        if (c->inp_rows <= 0 || !fast || Pc.smp_on || !lp_mode || c->store_f32 || c->probe_launch) {
            f.err = MSX_ERR_STATE;
            f.msg = "msx_set_path(INPATH): needs msx_set_broadening(MSX_BROADEN_IN_PATH) before msx_broaden_grid, a binary whose data pixels lie inside that window, float64 tables, the register-resident recipe and a likelihood / posterior / chi^2 mode";
            return f;
        }
        f.inpath = true; f.pair = false; f.linked = false;
    }
    return f;
}

// The pair form over A.n walkers (pair_kernel.h): the variant compiled for the smallest trip count that covers the
// spectrum (2 / 4 element trips per lane = up to 2048 / 4096 pixels).
int launch_pair(msx_ctx *c, const DevProblem &P, const LaunchArgs &A) {
    // 1. the planner: every walker's recipe (one thread per walker), final values of the rejected / failed ones, and
    //    who shares a workgroup
    const int32_t *plan = c->d_pair_plan;
    hipLaunchKernelGGL(pair_plan_kernel, dim3((unsigned)((A.n + kPlanThreads - 1) / kPlanThreads)), dim3(kPlanThreads), 0, A.s, A.theta,
                       (const unsigned char *)c->d_recipe_block, A.niso_nt, A.ng_mode_fast, (int64_t)A.n, P.tmin, P.tmax,
                       c->d_pair_plan, c->d_pair_items, c->d_pair_singles, A.logp, A.status, c->h_pair_stats, P);
    HIP_TRY(c, hipGetLastError());
    // 2. the planner's items: singles + pairs <= n workgroups; those beyond the planner's count leave after one load
    const dim3 g((unsigned)A.n);
    const int64_t ne = P.npair;
#define MSX_PAIR_GO2(T_, NT_, RED_)                                                                                   \
    hipLaunchKernelGGL((logprob_pair_kernel<T_, NT_, RED_>), g, dim3(T_), 0, A.s, A.theta, (const unsigned char *)c->d_recipe_block, \
                       A.niso_nt, A.ng_mode_fast, (int64_t)A.n, P.tmin, P.tmax, plan, P, A.logp, A.status)
    // (always the variant that loads the extinction terms: a problem staged without extinction -- every walker at
    // redc = 0 -- takes the unreddened values from it all the same and pays for the H rows; the variant without them
    // spilled registers at 4096 pixels and is not built)
#define MSX_PAIR_GO(T_, NT_) do { if (full) MSX_PAIR_GOF(T_, NT_); else MSX_PAIR_GO2(T_, NT_, true); } while (0)
#define MSX_PAIR_GOF(T_, NT_)                                                                                                \
    hipLaunchKernelGGL((logprob_pair_kernel<T_, NT_, true, true>), g, dim3(T_), 0, A.s, A.theta, (const unsigned char *)c->d_recipe_block, \
                       A.niso_nt, A.ng_mode_fast, (int64_t)A.n, P.tmin, P.tmax, plan, P, A.logp, A.status)
    // (FULL: the spectrum fills the variant exactly -- no clamps, no validity selects; pair_kernel.h)
    const bool full = c->use_full && P.npix == 2 * ne && (ne == 2 * 512 || ne == 4 * 512);
    // (512 threads, two workgroups per CU at <= 128 VGPRs: 16 waves per CU.  The 256-thread variants -- two per CU at
    // 256 VGPRs, 8 waves -- measured 411.8 us against 344.8 at 16,384 walkers and are not built.)
    if (ne <= 2 * 512) MSX_PAIR_GO(512, 2); else MSX_PAIR_GO(512, 4);
#undef MSX_PAIR_GO
#undef MSX_PAIR_GOF
#undef MSX_PAIR_GO2
    HIP_TRY(c, hipGetLastError());
    return MSX_OK;
}

// Every variant that takes dynamic LDS may be launched with up to the CU's 160 KiB minus its own static LDS.
// The limit is a property of the FUNCTION in this process, not of a context: it is raised once, to the maximum,
// so that contexts staged with different spectrum lengths can never lower it under one another.
hipError_t raise_one(const void *kernel) {
    hipFuncAttributes at;
    hipError_t e = hipFuncGetAttributes(&at, kernel);
    if (e != hipSuccess) return e;
    const int room = (160 * 1024 - (int)at.sharedSizeBytes) & ~15;
    return hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, room);
}
hipError_t raise_all() {
    hipError_t e = hipSuccess;
    for (const Variant &v : kVariants)
        if (e == hipSuccess && !v.gm) e = raise_one(v.fn);
    if (e == hipSuccess) e = raise_one((const void *)broaden_conv_kernel);
    return e;
}
constexpr int kMaxDevices = 64;
int raise_dynamic_lds_limits(msx_ctx *c) {
    // once per device and process; contexts may be created and staged from several host threads
    static std::mutex mu;
    static bool done[kMaxDevices] = {};
    if (c->device < 0 || c->device >= kMaxDevices)
        return fail(c, MSX_ERR_INVALID, "device index beyond the per-device table of dynamic-LDS limits (64 devices)");
    std::lock_guard<std::mutex> lock(mu);
    if (done[c->device]) return MSX_OK;
    HIP_TRY(c, raise_all());
    done[c->device] = true;
    return MSX_OK;
}

}  // namespace

extern "C" {

int msx_create(int device, msx_ctx **out) {
    if (!out) return MSX_ERR_INVALID;
    *out = nullptr;
    msx_ctx *c = new msx_ctx();
    c->device = device;
    memset(&c->P, 0, sizeof(c->P));
    if (const char *e = getenv("MSX_NO_PF")) c->use_pf = !(e[0] == '1');
    if (const char *e = getenv("MSX_NO_FULL")) c->use_full = !(e[0] == '1');
    if (const char *e = getenv("MSX_NO_SH2")) c->force_sh2 = e[0] == '0';
    if (const char *e = getenv("MSX_Q256")) c->q256 = e[0] == '1' ? 1 : 0;
    if (const char *e = getenv("MSX_LINKED")) c->linked = e[0] == '1' ? 1 : 0;
    if (const char *e = getenv("MSX_ZERO_COPY")) c->zero_copy = !(e[0] == '0');
    if (const char *e = getenv("MSX_PAD_LDS")) c->pad_lds = std::max<int64_t>(0, atoll(e));
    *out = c;  // returned even on failure so the caller can read msx_last_error
    HIP_TRY(c, hipSetDevice(device));
    HIP_TRY(c, hipGetDeviceProperties(&c->prop, device));
    HIP_TRY(c, hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    HIP_TRY(c, hipMalloc((void **)&c->d_misc, 4096));
    return MSX_OK;
}

void msx_destroy(msx_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    sampler_free(c);
    free_problem(c);
    free_grid(c);
    void *ptrs[] = {c->d_theta, c->d_logp, c->d_misc, c->d_spec, c->d_opt_flux, c->d_opt_med, c->d_opt_chain};
    if (c->h_pin) (void)hipHostFree(c->h_pin);
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    if (c->rccl_comm && rccl().ok) (void)rccl().CommDestroy(c->rccl_comm);
    for (msx_ctx *p : c->loop_peers)  // a loopback group ends with its first member
        if (p != c) { p->loop_peers.clear(); p->comm_world = 0; p->comm_rank = 0; }
    if (c->loop_eval_done) (void)hipEventDestroy(c->loop_eval_done);
    if (c->loop_copied) (void)hipEventDestroy(c->loop_copied);
    if (c->comm_stream) (void)hipStreamDestroy(c->comm_stream);
    if (c->ev_ready) (void)hipEventDestroy(c->ev_ready);
    for (hipEvent_t e : c->ev_done)
        if (e) (void)hipEventDestroy(e);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

const char *msx_last_error(msx_ctx *c) { return c ? c->err.c_str() : "null ctx"; }

int msx_device_info(msx_ctx *c, int64_t *out3, char *name, int name_len) {
    if (!c || !out3) return MSX_ERR_INVALID;
    out3[0] = c->prop.multiProcessorCount;
    out3[1] = (int64_t)c->prop.totalGlobalMem;
    out3[2] = c->prop.clockRate;
    if (name && name_len > 0) {
        strncpy(name, c->prop.name, name_len - 1);
        name[name_len - 1] = 0;
    }
    return MSX_OK;
}

int msx_stage_grid(msx_ctx *c, const double *wl, int64_t nwl, const double *teff_nodes, int32_t nt,
                   const double *logg_nodes, int32_t ng, const double *flux, const uint8_t *present) {
    if (!c || !wl || !teff_nodes || !logg_nodes || !flux || nwl < 2 || nt < 1 || ng < 1)
        return fail(c, MSX_ERR_INVALID, "msx_stage_grid: bad arguments");
    HIP_TRY(c, hipSetDevice(c->device));
    free_problem(c);
    free_grid(c);
    const int64_t nn = (int64_t)nt * ng;
    c->nwl = nwl; c->nt = nt; c->ng = ng;
    c->h_wl.assign(wl, wl + nwl);
    c->h_teff.assign(teff_nodes, teff_nodes + nt);
    c->h_logg.assign(logg_nodes, logg_nodes + ng);
    HIP_TRY(c, hipMalloc((void **)&c->d_grid, sizeof(double) * nn * nwl));
    HIP_TRY(c, hipMemcpy(c->d_grid, flux, sizeof(double) * nn * nwl, hipMemcpyHostToDevice));
    int rc;
    if ((rc = dev_alloc_copy(c, nullptr, wl, nwl, &c->d_wl))) return rc;
    if ((rc = dev_alloc_copy(c, nullptr, teff_nodes, (int64_t)nt, &c->d_teff))) return rc;
    if ((rc = dev_alloc_copy(c, nullptr, logg_nodes, (int64_t)ng, &c->d_logg))) return rc;
    std::vector<uint8_t> pres(nn, 1);
    if (present) pres.assign(present, present + nn);
    if ((rc = dev_alloc_copy(c, nullptr, pres.data(), nn, &c->d_present))) return rc;
    HIP_TRY(c, hipMalloc((void **)&c->d_kgrid, sizeof(double) * nwl));
    hipLaunchKernelGGL(ccm89_kernel, dim3((unsigned)((nwl + 255) / 256)), dim3(256), 0, c->stream, c->d_wl, nwl, 3.1,
                       c->d_kgrid);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->grid_staged = true;
    return MSX_OK;
}

int msx_ccm89_k(msx_ctx *c, const double *wl, int64_t n, double rv, double *out) {
    if (!c || !wl || !out || n < 0) return fail(c, MSX_ERR_INVALID, "msx_ccm89_k: bad arguments");
    if (n == 0) return MSX_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    double *d_in = nullptr, *d_out = nullptr;
    HIP_TRY(c, hipMalloc((void **)&d_in, sizeof(double) * n));
    HIP_TRY(c, hipMalloc((void **)&d_out, sizeof(double) * n));
    HIP_TRY(c, hipMemcpy(d_in, wl, sizeof(double) * n, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(ccm89_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, d_in, n, rv, d_out);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipMemcpy(out, d_out, sizeof(double) * n, hipMemcpyDeviceToHost));
    (void)hipFree(d_in);
    (void)hipFree(d_out);
    return MSX_OK;
}

int msx_resample_linear(msx_ctx *c, const double *x, const double *y, int64_t n, const double *xq, int64_t m,
                        double *out) {
    if (!c || !x || !y || !xq || !out || n < 2 || m < 0) return fail(c, MSX_ERR_INVALID, "msx_resample_linear: bad arguments");
    if (m == 0) return MSX_OK;
    for (int64_t i = 1; i < n; ++i)
        if (!(x[i] >= x[i - 1])) return fail(c, MSX_ERR_INVALID, "msx_resample_linear: x must be sorted ascending");
    double qmin = INFINITY, qmax = -INFINITY;
    for (int64_t i = 0; i < m; ++i) { qmin = xq[i] < qmin ? xq[i] : qmin; qmax = xq[i] > qmax ? xq[i] : qmax; }
    if (qmin < x[0]) return fail(c, MSX_ERR_RANGE, "A value in x_new is below the interpolation range's minimum value.");
    if (qmax > x[n - 1]) return fail(c, MSX_ERR_RANGE, "A value in x_new is above the interpolation range's maximum value.");
    HIP_TRY(c, hipSetDevice(c->device));
    double *d_x = nullptr, *d_y = nullptr, *d_q = nullptr, *d_o = nullptr;
    HIP_TRY(c, hipMalloc((void **)&d_x, sizeof(double) * n));
    HIP_TRY(c, hipMalloc((void **)&d_y, sizeof(double) * n));
    HIP_TRY(c, hipMalloc((void **)&d_q, sizeof(double) * m));
    HIP_TRY(c, hipMalloc((void **)&d_o, sizeof(double) * m));
    hipError_t e = hipMemcpy(d_x, x, sizeof(double) * n, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_y, y, sizeof(double) * n, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_q, xq, sizeof(double) * m, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(resample_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, c->stream, d_x, d_y, n, d_q, m, d_o);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) e = hipMemcpy(out, d_o, sizeof(double) * m, hipMemcpyDeviceToHost);
    (void)hipFree(d_x); (void)hipFree(d_y); (void)hipFree(d_q); (void)hipFree(d_o);
    if (e != hipSuccess) return fail(c, MSX_ERR_HIP, hipGetErrorString(e));
    return MSX_OK;
}

int msx_broaden(msx_ctx *c, const double *wl, const double *flux, int64_t n, double resolution, double maxsig,
                double *out) {
    if (!c || !wl || !flux || !out || n < 16 || !(resolution > 0) || !(maxsig > 0))
        return fail(c, MSX_ERR_INVALID, "msx_broaden: bad arguments (need n >= 16)");
    HIP_TRY(c, hipSetDevice(c->device));
    int rc = check_even_spacing(c, wl, n);
    if (rc) return rc;
    double mean = 0.0;
    for (int64_t i = 0; i < n; ++i) mean += wl[i];
    mean /= (double)n;
    double sigma;
    int lx;
    conv_taps(mean, wl[1] - wl[0], resolution, maxsig, &sigma, &lx);
    if (lx < 1) return fail(c, MSX_ERR_RANGE, "msx_broaden: empty kernel");
    double *d_in = nullptr, *d_tmp = nullptr, *d_out = nullptr;
    HIP_TRY(c, hipMalloc((void **)&d_in, sizeof(double) * n));
    HIP_TRY(c, hipMalloc((void **)&d_tmp, sizeof(double) * n));
    HIP_TRY(c, hipMalloc((void **)&d_out, sizeof(double) * n));
    HIP_TRY(c, hipMemcpy(d_in, flux, sizeof(double) * n, hipMemcpyHostToDevice));
    rc = launch_conv(c, d_in, n, d_tmp, 1, n, lx, wl[1] - wl[0], sigma);
    if (rc == MSX_OK) {
        hipLaunchKernelGGL(broaden_patch_kernel, dim3((unsigned)((n + 255) / 256), 1), dim3(256), 0, c->stream, d_tmp, n,
                           d_out, n, n);
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e == hipSuccess) e = hipMemcpy(out, d_out, sizeof(double) * n, hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = fail(c, MSX_ERR_HIP, hipGetErrorString(e));
    }
    (void)hipFree(d_in);
    (void)hipFree(d_tmp);
    (void)hipFree(d_out);
    return rc;
}

int msx_broaden_grid(msx_ctx *c, int64_t i0, int64_t n, double resolution, double maxsig) {
    if (!c) return MSX_ERR_INVALID;
    if (!c->grid_staged) return fail(c, MSX_ERR_STATE, "msx_broaden_grid: no grid staged");
    if (i0 < 0 || n < 16 || i0 + n > c->nwl || !(resolution > 0) || !(maxsig > 0))
        return fail(c, MSX_ERR_INVALID, "msx_broaden_grid: bad window");
    HIP_TRY(c, hipSetDevice(c->device));
    const double *wl = c->h_wl.data() + i0;
    int rc = check_even_spacing(c, wl, n);
    if (rc) return rc;
    double mean = 0.0;
    for (int64_t i = 0; i < n; ++i) mean += wl[i];
    mean /= (double)n;
    double sigma;
    int lx;
    conv_taps(mean, wl[1] - wl[0], resolution, maxsig, &sigma, &lx);
    const int64_t rows = (int64_t)c->nt * c->ng;
    // in-path placement (msx_set_broadening): the window's rows as they are NOW are kept for the per-walker form; the grid
    // is broadened in place all the same, so that every other form -- and the band tables -- see the reference's live path
    if (c->d_raw_win) { (void)hipFree(c->d_raw_win); c->d_raw_win = nullptr; c->raw_n = 0; }
    if (c->broaden_placement == MSX_BROADEN_IN_PATH) {
        HIP_TRY(c, hipMalloc((void **)&c->d_raw_win, sizeof(double) * rows * n));
        HIP_TRY(c, hipMemcpy2D(c->d_raw_win, sizeof(double) * n, c->d_grid + i0, sizeof(double) * c->nwl, sizeof(double) * n, (size_t)rows,
                               hipMemcpyDeviceToDevice));
        c->raw_i0 = i0; c->raw_n = n; c->raw_lx = lx; c->raw_dx = wl[1] - wl[0]; c->raw_sigma = sigma;
    }
    double *d_tmp = nullptr;
    HIP_TRY(c, hipMalloc((void **)&d_tmp, sizeof(double) * rows * n));
    rc = launch_conv(c, c->d_grid + i0, c->nwl, d_tmp, rows, n, lx, wl[1] - wl[0], sigma);
    if (rc == MSX_OK) {
        hipLaunchKernelGGL(broaden_patch_kernel, dim3((unsigned)((n + 255) / 256), (unsigned)rows), dim3(256), 0,
                           c->stream, d_tmp, n, c->d_grid + i0, c->nwl, n);
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) rc = fail(c, MSX_ERR_HIP, hipGetErrorString(e));
    }
    (void)hipFree(d_tmp);
    // any staged problem was derived from the pre-broadening grid
    free_problem(c);
    return rc;
}

int msx_read_node(msx_ctx *c, int32_t it, int32_t ig, double *out) {
    if (!c || !out) return MSX_ERR_INVALID;
    if (!c->grid_staged) return fail(c, MSX_ERR_STATE, "msx_read_node: no grid staged");
    if (it < 0 || it >= c->nt || ig < 0 || ig >= c->ng) return fail(c, MSX_ERR_INVALID, "msx_read_node: bad node");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipMemcpy(out, c->d_grid + ((int64_t)it * c->ng + ig) * c->nwl, sizeof(double) * c->nwl,
                         hipMemcpyDeviceToHost));
    return MSX_OK;
}

int msx_stage_problem(msx_ctx *c, const msx_problem *p) {
    if (!c || !p) return MSX_ERR_INVALID;
    if (p->struct_size != (int32_t)sizeof(msx_problem))
        return fail(c, MSX_ERR_INVALID, "msx_stage_problem: struct_size mismatch (header/library skew)");
    if (!c->grid_staged) return fail(c, MSX_ERR_STATE, "msx_stage_problem: stage the grid first");
    if (p->nspec < 2 || p->nspec > MSX_MAX_SPEC) return fail(c, MSX_ERR_INVALID, "nspec must be 2 or 3");
    if (p->npix < 4) return fail(c, MSX_ERR_INVALID, "npix too small");
    if (p->n_contrast < 0 || p->n_contrast > MSX_MAX_BANDS || p->n_phot < 0 || p->n_phot > MSX_MAX_BANDS)
        return fail(c, MSX_ERR_INVALID, "too many bands");
    if (p->niso < 2) return fail(c, MSX_ERR_INVALID, "isochrone table needs >= 2 rows");
    if (p->win_j0 < 0 || p->win_n < 1 || p->win_j0 + p->win_n > c->nwl)
        return fail(c, MSX_ERR_RANGE, "composite window is outside the staged grid");
    const int64_t need_lds = (int64_t)sizeof(double) * p->npix;
    const bool model_in_global = need_lds > 134 * 1024;  // > 17,152 pixels (160 KiB - ~25 KiB of static LDS): the GM variants
    for (int64_t i = 0; i < p->npix; ++i)
        if (p->pix_lo[i] < 0 || p->pix_lo[i] + 1 >= c->nwl)
            return fail(c, MSX_ERR_RANGE, "A value in x_new is outside the interpolation range (data pixel vs model grid)");
    const int nb = p->n_contrast + p->n_phot;
    for (int b = 0; b < nb; ++b)
        if (p->band_i0[b] < 0 || p->band_len[b] < 0 || p->band_i0[b] + p->band_len[b] > c->nwl)
            return fail(c, MSX_ERR_RANGE, "band weights run outside the staged grid");
    HIP_TRY(c, hipSetDevice(c->device));
    free_problem(c);
    std::vector<void *> &tr = c->prob_allocs;
    DevProblem &P = c->P;
    memset(&P, 0, sizeof(P));
    P.grid = c->d_grid; P.kgrid = c->d_kgrid; P.nwl = c->nwl; P.nt = c->nt; P.ng = c->ng;
    P.teff_nodes = c->d_teff; P.logg_nodes = c->d_logg; P.present = c->d_present;
    P.npix = p->npix; P.median_flux = p->median_flux; P.nspec = p->nspec;
    memcpy(P.minv, p->fit_minv, sizeof(P.minv));
    P.nc = p->n_contrast; P.np = p->n_phot;
    for (int i = 0; i < P.nc; ++i) { P.cmag[i] = p->cmag[i]; P.cerr[i] = p->cerr[i]; P.civar[i] = 1.0 / (p->cerr[i] * p->cerr[i]); }
    for (int i = 0; i < P.np; ++i) {
        P.pmag[i] = p->pmag[i]; P.perr[i] = p->perr[i]; P.pzero[i] = p->phot_zero[i]; P.pk[i] = p->phot_k[i];
        P.pivar[i] = 1.0 / (p->perr[i] * p->perr[i]);
    }
    P.win_j0 = p->win_j0; P.win_n = p->win_n;
    P.niso = p->niso; P.nav = p->nav; P.tmin = p->tmin; P.tmax = p->tmax;
    memcpy(P.pmean, p->prior_mean, sizeof(P.pmean));
    memcpy(P.psig, p->prior_sig, sizeof(P.psig));
    P.use_av = p->use_av; P.dist_fit = p->dist_fit; P.rad_prior = p->rad_prior; P.has_prior = p->has_prior_list;
    P.no_spectrum = p->no_spectrum;
    int rc;
    double *d;
    if ((rc = dev_alloc_copy(c, &tr, p->pix_t, p->npix, &d))) return rc; P.pix_t = d;
    if ((rc = dev_alloc_copy(c, &tr, p->pix_u, p->npix, &d))) return rc; P.pix_u = d;
    if ((rc = dev_alloc_copy(c, &tr, p->pix_flux, p->npix, &d))) return rc; P.pix_flux = d;
    {
        std::vector<double> ivar(p->npix);
        for (int64_t i = 0; i < p->npix; ++i) ivar[i] = 1.0 / (p->pix_err[i] * p->pix_err[i]);
        if ((rc = dev_alloc_copy(c, &tr, ivar.data(), p->npix, &d))) return rc;
        P.pix_ivar = d;
    }
    if ((rc = dev_alloc_copy(c, &tr, p->iso_teff, (int64_t)p->niso, &d))) return rc; P.iso_t = d;
    if ((rc = dev_alloc_copy(c, &tr, p->iso_logg, (int64_t)p->niso, &d))) return rc; P.iso_g = d;
    if ((rc = dev_alloc_copy(c, &tr, p->iso_lum, (int64_t)p->niso, &d))) return rc; P.iso_l = d;
    if ((rc = dev_alloc_copy(c, &tr, p->av_edges_pc, (int64_t)(p->nav > 0 ? p->nav + 1 : 0), &d))) return rc; P.av_edges = d;
    if ((rc = dev_alloc_copy(c, &tr, p->av_mu, (int64_t)p->nav, &d))) return rc; P.av_mu = d;
    if ((rc = dev_alloc_copy(c, &tr, p->av_sig, (int64_t)p->nav, &d))) return rc; P.av_sig = d;

    const int64_t nn = (int64_t)c->nt * c->ng;
    // the blend's tables in two-pixel elements (logprob_kernel.h "TABLE LAYOUT"): per node R (f64) + H (f32),
    // per pixel k[lo], dk, data flux, u
    int64_t *d_lo = nullptr;
    if ((rc = dev_alloc_copy(c, &tr, p->pix_lo, p->npix, &d_lo))) return rc;
    const int64_t npair = ((p->npix + 511) / 512) * 256;
    double2 *d_r2 = nullptr, *d_kl2 = nullptr, *d_f2 = nullptr, *d_u2 = nullptr, *d_iv2 = nullptr;
    float2 *d_h2 = nullptr, *d_dk2 = nullptr;
    HIP_TRY(c, hipMalloc((void **)&d_r2, sizeof(double2) * nn * npair)); tr.push_back(d_r2);
    HIP_TRY(c, hipMalloc((void **)&d_h2, sizeof(float2) * nn * npair)); tr.push_back(d_h2);
    HIP_TRY(c, hipMalloc((void **)&d_kl2, sizeof(double2) * npair)); tr.push_back(d_kl2);
    HIP_TRY(c, hipMalloc((void **)&d_dk2, sizeof(float2) * npair)); tr.push_back(d_dk2);
    HIP_TRY(c, hipMalloc((void **)&d_f2, sizeof(double2) * npair)); tr.push_back(d_f2);
    HIP_TRY(c, hipMalloc((void **)&d_u2, sizeof(double2) * npair)); tr.push_back(d_u2);
    HIP_TRY(c, hipMalloc((void **)&d_iv2, sizeof(double2) * npair)); tr.push_back(d_iv2);
    dim3 gg((unsigned)((npair + 255) / 256), (unsigned)nn);
    hipLaunchKernelGGL(gather_rh_kernel, gg, dim3(256), 0, c->stream, c->d_grid, c->nwl, d_lo, P.pix_t, p->npix, npair, d_r2, d_h2);
    HIP_TRY(c, hipGetLastError());
    hipLaunchKernelGGL(gather_statics_kernel, dim3(gg.x), dim3(256), 0, c->stream, c->d_kgrid, d_lo, P.pix_flux, P.pix_u,
                       P.pix_ivar, p->npix, npair, d_kl2, d_dk2, d_f2, d_u2, d_iv2);
    HIP_TRY(c, hipGetLastError());
    P.r2 = d_r2; P.h2 = d_h2; P.kl2 = d_kl2; P.dk2 = d_dk2; P.f2 = d_f2; P.u2 = d_u2; P.iv2 = d_iv2; P.npair = npair;
    c->d_pix_lo = d_lo;
    c->inp_rows = 0;
    P.given = nullptr; P.given_stride = 0;
    if (c->d_raw_win && p->nspec == 2 && !model_in_global && !p->no_spectrum) {
        // the in-path form needs both model samples of every data pixel inside the window that msx_broaden_grid kept raw
        bool inside = true;
        for (int64_t i = 0; i < p->npix && inside; ++i) inside = p->pix_lo[i] >= c->raw_i0 && p->pix_lo[i] + 1 < c->raw_i0 + c->raw_n;
        if (inside) {
            const int64_t gstride = 2 * npair;
            int64_t budget = 256ll << 20;
            if (const char *e = getenv("MSX_INPATH_MB")) budget = std::max<int64_t>(1, atoll(e)) << 20;
            int64_t rows_i = budget / (int64_t)(sizeof(double) * (c->raw_n + gstride) + sizeof(InpathRec));
            rows_i = std::max<int64_t>(16, std::min<int64_t>(rows_i, 16384));
            HIP_TRY(c, hipMalloc((void **)&c->d_inp_rec, sizeof(InpathRec) * rows_i));
            HIP_TRY(c, hipMalloc((void **)&c->d_inp_tmp, sizeof(double) * rows_i * c->raw_n));
            HIP_TRY(c, hipMalloc((void **)&c->d_inp_given, sizeof(double) * rows_i * gstride));
            const size_t lds_conv = sizeof(double) * ((size_t)c->raw_lx + ((size_t)(kConvTile + c->raw_lx - 1) * 5) / 4 + 2);
            if (lds_conv > 150 * 1024) return fail(c, MSX_ERR_RANGE, "in-path broadening: kernel too long for the LDS tile");
            if (lds_conv > 48 * 1024)
                HIP_TRY(c, hipFuncSetAttribute((const void *)inpath_conv_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
            c->inp_rows = rows_i; c->inp_gstride = gstride;
        }
    }
    c->store_f32 = false;
    if (c->store_dtype == MSX_STORE_F32) {
        // A separately labelled storage precision (store_dtype): the R table rounded to float32, read by the
        // fused binary variants compiled for it.  Everything else keeps float64 -- and the forms that have no such variant
        // (pair, linked, triples, spectra beyond the LDS) are refused rather than mixed in: one staged problem, one precision.
        if (p->nspec != 2 || model_in_global)
            return fail(c, MSX_ERR_STATE, "msx_set_grid_storage(MSX_STORE_F32): binaries of at most 17,152 pixels only");
        float2 *d_r2f = nullptr;
        HIP_TRY(c, hipMalloc((void **)&d_r2f, sizeof(float2) * nn * npair)); tr.push_back(d_r2f);
        const int64_t tot = nn * npair;
        hipLaunchKernelGGL(narrow_r_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, c->stream, d_r2, d_r2f, tot);
        HIP_TRY(c, hipGetLastError());
        P.r2f = d_r2f;
        c->store_f32 = true;
    }
    float2 *const d_r2f_for_quads = const_cast<float2 *>(P.r2f);
    {   // the float32 tables in quads, for the 512-thread variants
        const int64_t nquad = (npair + 1023) / 1024 * 512;
        float4 *d_h4 = nullptr, *d_dk4 = nullptr, *d_h4b = nullptr, *d_dk4b = nullptr;
        HIP_TRY(c, hipMalloc((void **)&d_h4, sizeof(float4) * nn * nquad)); tr.push_back(d_h4);
        HIP_TRY(c, hipMalloc((void **)&d_dk4, sizeof(float4) * nquad)); tr.push_back(d_dk4);
        HIP_TRY(c, hipMalloc((void **)&d_h4b, sizeof(float4) * nn * nquad)); tr.push_back(d_h4b);
        HIP_TRY(c, hipMalloc((void **)&d_dk4b, sizeof(float4) * nquad)); tr.push_back(d_dk4b);
        const unsigned gq = (unsigned)((nquad + 255) / 256);
        hipLaunchKernelGGL(gather_quads_kernel, dim3(gq, (unsigned)nn), dim3(256), 0, c->stream, d_h2, npair, nquad, (int64_t)512, d_h4);
        hipLaunchKernelGGL(gather_quads_kernel, dim3(gq, 1), dim3(256), 0, c->stream, d_dk2, npair, nquad, (int64_t)512, d_dk4);
        hipLaunchKernelGGL(gather_quads_kernel, dim3(gq, (unsigned)nn), dim3(256), 0, c->stream, d_h2, npair, nquad, (int64_t)256, d_h4b);
        hipLaunchKernelGGL(gather_quads_kernel, dim3(gq, 1), dim3(256), 0, c->stream, d_dk2, npair, nquad, (int64_t)256, d_dk4b);
        HIP_TRY(c, hipGetLastError());
        P.h4 = d_h4; P.dk4 = d_dk4; P.h4b = d_h4b; P.dk4b = d_dk4b; P.nquad = nquad;
        if (d_r2f_for_quads) {  // float32 storage: the R table by quad too
            float4 *d_r4f = nullptr, *d_r4fb = nullptr;
            HIP_TRY(c, hipMalloc((void **)&d_r4f, sizeof(float4) * nn * nquad)); tr.push_back(d_r4f);
            HIP_TRY(c, hipMalloc((void **)&d_r4fb, sizeof(float4) * nn * nquad)); tr.push_back(d_r4fb);
            hipLaunchKernelGGL(gather_quads_kernel, dim3(gq, (unsigned)nn), dim3(256), 0, c->stream, d_r2f_for_quads, npair, nquad, (int64_t)512, d_r4f);
            hipLaunchKernelGGL(gather_quads_kernel, dim3(gq, (unsigned)nn), dim3(256), 0, c->stream, d_r2f_for_quads, npair, nquad, (int64_t)256, d_r4fb);
            HIP_TRY(c, hipGetLastError());
            P.r4f = d_r4f; P.r4fb = d_r4fb;
        }
    }
    // band integrals
    double *d_tab = nullptr;
    HIP_TRY(c, hipMalloc((void **)&d_tab, sizeof(double) * std::max<int64_t>(1, nn * nb))); tr.push_back(d_tab);
    if (nb > 0) {
        std::vector<int64_t> woff(nb);
        int64_t tot = 0;
        for (int b = 0; b < nb; ++b) { woff[b] = tot; tot += p->band_len[b]; }
        double *d_w = nullptr;
        int64_t *d_woff = nullptr, *d_i0 = nullptr, *d_len = nullptr;
        if ((rc = dev_alloc_copy(c, &tr, p->band_w, tot, &d_w))) return rc;
        if ((rc = dev_alloc_copy(c, &tr, woff.data(), (int64_t)nb, &d_woff))) return rc;
        if ((rc = dev_alloc_copy(c, &tr, p->band_i0, (int64_t)nb, &d_i0))) return rc;
        if ((rc = dev_alloc_copy(c, &tr, p->band_len, (int64_t)nb, &d_len))) return rc;
        hipLaunchKernelGGL(band_integral_kernel, dim3((unsigned)nb, (unsigned)nn), dim3(256), 0, c->stream, c->d_grid,
                           c->nwl, d_w, d_woff, d_i0, d_len, nb, d_tab);
        HIP_TRY(c, hipGetLastError());
    }
    P.band_tab = d_tab;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->max_dyn_lds = (int)need_lds;
    c->model_in_global = model_in_global;
    {   // the PF variants' dynamic LDS -- model + u2 + f2 -- must fit beside their static LDS (asked of the functions
        // themselves: the static part has grown over the rounds, a constant here once let 5,800..6,270 pixels through)
        hipFuncAttributes a2, a3;
        HIP_TRY(c, hipFuncGetAttributes(&a2, (const void *)logprob_kernel<2, 2, 512, false, false, true>));
        HIP_TRY(c, hipFuncGetAttributes(&a3, (const void *)logprob_kernel<3, 2, 512, false, false, true>));
        const int64_t room = (160 * 1024 - (int64_t)std::max(a2.sharedSizeBytes, a3.sharedSizeBytes)) & ~15ll;
        c->pf_ok = !model_in_global && (int64_t)sizeof(double) * ((p->npix + 1) & ~1ll) + 32 * npair + (int64_t)c->pad_lds <= room;
        hipFuncAttributes a256;
        HIP_TRY(c, hipFuncGetAttributes(&a256, (const void *)logprob_kernel<2, 2, 256, false, true, true>));
        c->pf256_ok = 2 * ((int64_t)sizeof(double) * ((p->npix + 1) & ~1ll) + 32 * npair + (int64_t)a256.sharedSizeBytes) <= 160 * 1024;
    }
    if ((rc = raise_dynamic_lds_limits(c))) return rc;
    c->recipe_fast = P.niso <= 4 * kWave && P.nt <= kWave && P.ng <= 32 && P.nav + 1 <= 2 * kWave;
    c->d_recipe_block = nullptr;
    if (c->recipe_fast) {  // the recipe's tables behind one (preloaded) pointer
        void *blk = nullptr;
        HIP_TRY(c, hipMalloc(&blk, kRecipeBlockBytes));
        tr.push_back(blk);
        c->d_recipe_block = (unsigned char *)blk;
        HIP_TRY(c, hipMemsetAsync(blk, 0, kRecipeBlockBytes, c->stream));
        const hipMemcpyKind dd = hipMemcpyDeviceToDevice;
        HIP_TRY(c, hipMemcpyAsync(c->d_recipe_block + kRbIsoT, P.iso_t, sizeof(double) * P.niso, dd, c->stream));
        HIP_TRY(c, hipMemcpyAsync(c->d_recipe_block + kRbIsoG, P.iso_g, sizeof(double) * P.niso, dd, c->stream));
        HIP_TRY(c, hipMemcpyAsync(c->d_recipe_block + kRbTeff, P.teff_nodes, sizeof(double) * P.nt, dd, c->stream));
        HIP_TRY(c, hipMemcpyAsync(c->d_recipe_block + kRbLogg, P.logg_nodes, sizeof(double) * P.ng, dd, c->stream));
        std::vector<uint8_t> pres((size_t)(P.nt * P.ng));
        HIP_TRY(c, hipMemcpyAsync(pres.data(), P.present, pres.size(), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        std::vector<uint32_t> pmask((size_t)P.nt, 0u);
        for (int64_t t = 0; t < P.nt; ++t)
            for (int64_t g = 0; g < P.ng; ++g)
                if (pres[(size_t)(t * P.ng + g)]) pmask[(size_t)t] |= 1u << g;
        HIP_TRY(c, hipMemcpyAsync(c->d_recipe_block + kRbPresent, pmask.data(), sizeof(uint32_t) * pmask.size(),
                                  hipMemcpyHostToDevice, c->stream));
        // the packed copies for the wave form's lanes (dev_types.h): entry + right neighbour, slopes, pads
        std::vector<unsigned char> pack((size_t)(kRecipeBlockBytes - kRbIsoPack), 0);
        {
            double *iso = reinterpret_cast<double *>(pack.data());
            for (int i = 0; i < 4 * kWave; ++i) {
                const bool in = i < P.niso, nx = i + 1 < P.niso;
                const double x = in ? p->iso_teff[i] : INFINITY, y = in ? p->iso_logg[i] : 0.0;
                const double xn = nx ? p->iso_teff[i + 1] : INFINITY, yn = nx ? p->iso_logg[i + 1] : 0.0;
                iso[4 * i] = x; iso[4 * i + 1] = xn; iso[4 * i + 2] = y;
                iso[4 * i + 3] = nx ? (yn - y) / (xn - x) : 0.0;  // np.interp's slope (IEEE division, like the device's)
            }
            double *tp = reinterpret_cast<double *>(pack.data() + (kRbTeffPack - kRbIsoPack));
            double *gp = reinterpret_cast<double *>(pack.data() + (kRbLoggPack - kRbIsoPack));
            uint32_t *mp = reinterpret_cast<uint32_t *>(pack.data() + (kRbMaskPack - kRbIsoPack));
            for (int l = 0; l < kWave; ++l) {
                tp[2 * l] = l < P.nt ? c->h_teff[(size_t)l] : INFINITY;
                tp[2 * l + 1] = l + 1 < P.nt ? c->h_teff[(size_t)l + 1] : INFINITY;
                gp[2 * l] = l < P.ng ? c->h_logg[(size_t)l] : INFINITY;
                gp[2 * l + 1] = l + 1 < P.ng ? c->h_logg[(size_t)l + 1] : INFINITY;
                mp[2 * l] = l < P.nt ? pmask[(size_t)l] : 0u;
                mp[2 * l + 1] = l + 1 < P.nt ? pmask[(size_t)l + 1] : 0u;
            }
        }
        HIP_TRY(c, hipMemcpyAsync(c->d_recipe_block + kRbIsoPack, pack.data(), pack.size(), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    // Scratch, sized once here so that no launch ever allocates or synchronises -- and only for the forms the staged
    // spectrum can take: the GM variants (model vectors beyond the LDS) and the linked form (2..8 segments).  A batch
    // beyond `scratch_rows` walkers is then cut into sub-batches.  Everything else (config 2, 3, 5) allocates nothing.
    c->nseg = (int)((npair + kSegElems - 1) / kSegElems);
    const bool can_link = c->nseg >= 2 && c->nseg <= 8;  // (a workgroup of the linked form holds one segment: up to 65,536 pixels)
    P.linked_fault = 0;
    if (const char *e = getenv("MSX_LINKED_FAULT")) P.linked_fault = e[0] == '1';  // (tests: the bounded wait)
    if (model_in_global || can_link) {
        int64_t budget = 96ll << 20;
        if (const char *e = getenv("MSX_SCRATCH_MB")) budget = std::max<int64_t>(1, atoll(e)) << 20;
        int64_t sb = budget / (int64_t)(sizeof(double) * p->npix);
        sb = std::max<int64_t>(256, std::min<int64_t>(sb, 16384));
        if (const char *e = getenv("MSX_SCRATCH_ROWS")) sb = std::max<int64_t>(1, atoll(e));
        HIP_TRY(c, hipMalloc((void **)&c->d_model_scratch, sizeof(double) * sb * p->npix));
        c->scratch_rows = sb;
        P.model_scratch = c->d_model_scratch;
        if (can_link) {
            HIP_TRY(c, hipMalloc((void **)&c->d_segparts, sizeof(SegPart) * sb * c->nseg));
            P.segparts = c->d_segparts;
            // hand-over flags, and behind them the poison word (cleared together, here and nowhere else)
            HIP_TRY(c, hipMalloc((void **)&c->d_seg_flag, sizeof(unsigned long long) * (sb + 1)));
            HIP_TRY(c, hipMemset(c->d_seg_flag, 0, sizeof(unsigned long long) * (sb + 1)));
            P.seg_flag = c->d_seg_flag;
            P.linked_poison = reinterpret_cast<int32_t *>(c->d_seg_flag + sb);
        }
    }
    c->linked_poisoned = false;
    // the pair form: binaries of <= 4096 pixels (model values in registers), register-resident recipe.  Its spill path
    // (vectors the early histogram cannot handle) leases one of kPairSpillRows scratch rows (32 MB at 4096 pixels); the
    // planner's items take 256 bytes per walker of a sub-batch.
    if (p->nspec == 2 && p->npix <= kPairMaxPix && c->recipe_fast && !p->no_spectrum) {
        int64_t rows = 16384;
        if (const char *e = getenv("MSX_PAIR_ROWS")) rows = std::max<int64_t>(2, atoll(e));
        HIP_TRY(c, hipMalloc((void **)&c->d_model_scratch, sizeof(double) * kPairSpillRows * p->npix));
        P.model_scratch = c->d_model_scratch;
        // the planner's header and, behind it, the leases of the spill rows
        const size_t plan_bytes = sizeof(int32_t) * (size_t)(kPairHdrInts + kPairSpillRows);
        HIP_TRY(c, hipMalloc((void **)&c->d_pair_plan, plan_bytes));
        HIP_TRY(c, hipMemset(c->d_pair_plan, 0, plan_bytes));
        P.pair_lease = c->d_pair_plan + kPairHdrInts;
        HIP_TRY(c, hipMalloc((void **)&c->d_pair_items, sizeof(PairItem) * (size_t)((rows + 1) / 2)));
        HIP_TRY(c, hipMalloc((void **)&c->d_pair_singles, sizeof(PairRec) * (size_t)rows));
        P.pair_items = c->d_pair_items;
        P.pair_singles = c->d_pair_singles;
        HIP_TRY(c, hipHostMalloc((void **)&c->h_pair_stats, 2 * sizeof(int32_t), hipHostMallocDefault));
        c->h_pair_stats[0] = 1; c->h_pair_stats[1] = 0;  // (nothing known yet: try)
        c->pair_auto_launches = 0;
        c->pair_rows = rows;
        // From where the pair form pays (the planner costs 10.4 us whatever the batch; 14.6 before its searches went 4-ary
        // and side by side, round 4): measured on 256 CUs in one process, fused (FULL variants) against pair, us per batch
        // -- 4096 px (profiles/r4_crossover_4096px.jsonl): 1,024 walkers 39.6 / 42.2, 1,536: 49.7 / 48.9, 2,048: 61.4 / 57.2,
        // 2,304: 66.9 / 61.6, 3,072: 85.2 / 73.5, 4,096: 109.5 / 89.6.  1194 px (r4_crossover_1194px.jsonl): 2,304 walkers
        // 45.8 / 45.7, 3,072: 55.4 / 53.2, 4,096: 72.2 / 64.0, 6,144: 104.5 / 85.7.  In walkers per CU: 8 for the long
        // spectra, 12 for the short ones.
        {
            const int64_t cus = c->prop.multiProcessorCount > 0 ? c->prop.multiProcessorCount : 256;
            c->pair_min_walkers = (p->npix > 3072 ? 8 : 12) * cus;
        }
        if (const char *e = getenv("MSX_PAIR_MIN")) c->pair_min_walkers = atoll(e) > 0 ? std::max<int64_t>(2, atoll(e)) : INT64_MAX;
    }
#ifdef MSX_STAMPS
    {   // diagnostic build only: per-walker shader-clock stamps
        unsigned long long *st = nullptr;
        HIP_TRY(c, hipMalloc((void **)&st, sizeof(unsigned long long) * 16 * 65536)); tr.push_back(st);
        HIP_TRY(c, hipMemset(st, 0, sizeof(unsigned long long) * 16 * 65536));
        P.stamps = st;
    }
#endif
    {   // the clock probe's stamps (msx_probe_launch)
        unsigned long long *cp = nullptr;
        HIP_TRY(c, hipMalloc((void **)&cp, sizeof(unsigned long long) * 4 * kProbeWalkers)); tr.push_back(cp);
        HIP_TRY(c, hipMemset(cp, 0, sizeof(unsigned long long) * 4 * kProbeWalkers));
        P.clk_probe = cp;
    }
    c->problem_staged = true;
    return MSX_OK;
}

#ifdef MSX_STAMPS
int msx_diag_read_med_stamps(msx_ctx *c, int64_t n, unsigned long long *out) {
    if (!c || n > 65536) return MSX_ERR_INVALID;
    HIP_TRY(c, hipDeviceSynchronize());
    HIP_TRY(c, hipMemcpyFromSymbol(out, HIP_SYMBOL(g_med_stamps), sizeof(unsigned long long) * 8 * n));
    return MSX_OK;
}
int msx_diag_read_stamps(msx_ctx *c, int64_t n, unsigned long long *out) {
    if (!c || !c->problem_staged || n > 65536) return MSX_ERR_INVALID;
    HIP_TRY(c, hipDeviceSynchronize());
    HIP_TRY(c, hipMemcpy(out, c->P.stamps, sizeof(unsigned long long) * 16 * n, hipMemcpyDeviceToHost));
    return MSX_OK;
}
#endif

int msx_set_grid_storage(msx_ctx *c, int32_t store_dtype) {
    if (!c || (store_dtype != MSX_STORE_F64 && store_dtype != MSX_STORE_F32))
        return fail(c, MSX_ERR_INVALID, "msx_set_grid_storage: MSX_STORE_F64 or MSX_STORE_F32");
    c->store_dtype = store_dtype;
    return MSX_OK;
}

int msx_set_path(msx_ctx *c, int32_t path) {
    if (!c || (path != MSX_PATH_AUTO && path != MSX_PATH_FUSED && path != MSX_PATH_LINKED && path != MSX_PATH_PAIR && path != MSX_PATH_INPATH))
        return fail(c, MSX_ERR_INVALID, "msx_set_path: bad path (MSX_PATH_AUTO, _FUSED, _PAIR, _LINKED or _INPATH)");
    c->path = path;
    return MSX_OK;
}

int msx_set_broadening(msx_ctx *c, int32_t placement) {
    if (!c || (placement != MSX_BROADEN_STAGING && placement != MSX_BROADEN_IN_PATH))
        return fail(c, MSX_ERR_INVALID, "msx_set_broadening: MSX_BROADEN_STAGING or MSX_BROADEN_IN_PATH");
    c->broaden_placement = placement;
    return MSX_OK;
}

int msx_logprob_batch_dev(msx_ctx *c, int32_t mode, const double *d_theta, int64_t n, int32_t ndim, double *d_logp,
                          int32_t *d_status, void *hip_stream, int32_t block_threads) {
    if (!c) return MSX_ERR_INVALID;
    if (!c->problem_staged) return fail(c, MSX_ERR_STATE, "msx_logprob_batch: no problem staged");
    if (n < 0 || !d_theta || !d_logp || !d_status) return fail(c, MSX_ERR_INVALID, "msx_logprob_batch: bad arguments");
    if (ndim != 2 * c->P.nspec + 2)
        return fail(c, MSX_ERR_INVALID, "P0 doesn't match what I was expecting (ndim must be 2*nspec+2)");
    if (mode < 0 || mode > 5) return fail(c, MSX_ERR_INVALID, "msx_logprob_batch: bad mode");
    if ((mode == MSX_MODE_OPT_STEP || mode == MSX_MODE_OPT_INIT) && !c->P.opt_flux)
        return fail(c, MSX_ERR_STATE, "optimiser modes go through msx_opt_init / msx_opt_step");
    if (n == 0) return MSX_OK;
    const bool shared512 = block_threads == MSX_BLOCK_512_SHARED;  // 512 threads, the <= 128-VGPR variant that
    if (shared512) block_threads = 512;                            // shares a CU with another workgroup
    if (block_threads != 0 && block_threads != 256 && block_threads != 512)
        return fail(c, MSX_ERR_INVALID, "block_threads must be 0, 256, 512 or MSX_BLOCK_512_SHARED");
    hipStream_t s = (hipStream_t)hip_stream;
    const DevProblem &Pc = c->P;
    // the leading, preloaded kernel arguments (see logprob_kernel): theta, the recipe's small tables and three
    // packed ints; `fast` = the tables fit the register-resident recipe
    const bool fast = c->recipe_fast;
    LaunchArgs A;
    A.ndim = ndim; A.mode = mode; A.s = s;
    A.niso_nt = (int)(std::min<int64_t>(Pc.niso, 0xffff) | ((int64_t)std::min<int64_t>(Pc.nt, 0x7fff) << 16));
    A.ng_mode_fast = (int)std::min<int64_t>(Pc.ng, 0xff) | (mode << 8) | ((fast ? 1 : 0) << 16) | ((Pc.smp_on ? 1 : 0) << 17) |
                     ((Pc.dist_fit ? 1 : 0) << 18) | ((Pc.use_av ? 1 : 0) << 19);

    // ---- which form of the path (decide_form) ---------------------------------------------------------------
    const FormChoice form = decide_form(c, n, mode, false);
    if (form.err != MSX_OK) return fail(c, form.err, form.msg);
    const bool linked = form.linked, pair = form.pair, inpath = form.inpath;
    if (c->smp_overlap_launch) A.ng_mode_fast |= 1 << 20;
    if (c->probe_launch) A.ng_mode_fast |= 1 << 21;
    c->last_form = inpath ? MSX_FORM_INPATH : pair ? MSX_FORM_PAIR : linked ? MSX_FORM_LINKED : MSX_FORM_FUSED;
    // sub-batches: the linked form's scratch, and the fused kernel's global model vectors for spectra beyond the LDS,
    // hold scratch_rows walkers; the pair form's spill rows pair_rows
    const int64_t step = inpath ? c->inp_rows : pair ? c->pair_rows : (linked || c->model_in_global) ? c->scratch_rows : n;
    for (int64_t off = 0; off < n; off += step) {
        const int64_t m = std::min<int64_t>(step, n - off);
        A.theta = d_theta + off * ndim; A.logp = d_logp + off; A.status = d_status + off; A.n = m;
        const DevProblem P = problem_at(Pc, off, mode, ndim);
        const int B = block_threads > 0 ? block_threads : pick_block(c, m, Pc.npix);
        int rc;
        if (inpath) {
            if ((rc = launch_inpath(c, P, A))) return rc;
            continue;
        }
        if (pair) {
            if ((rc = launch_pair(c, P, A))) return rc;
            continue;
        }
        if (linked) {
            LaunchArgs A5 = A;
            A5.ng_mode_fast |= c->nseg << 24;
            if ((rc = launch_logprob<true>(c, P, A5, 512, false))) return rc;
        } else {
            if ((rc = launch_logprob<false>(c, P, A, B, shared512))) return rc;
        }
    }
    return MSX_OK;
}

int msx_probe_launch(msx_ctx *c, int32_t mode, const double *d_theta, int64_t n, int32_t ndim, double *d_logp,
                     int32_t *d_status, void *hip_stream, int32_t block_threads, double *out4) {
    if (!c || !out4) return MSX_ERR_INVALID;
    if (!c->problem_staged) return fail(c, MSX_ERR_STATE, "msx_probe_launch: no problem staged");
    HIP_TRY(c, hipSetDevice(c->device));
    const int64_t m = std::min<int64_t>(n, kProbeWalkers);
    HIP_TRY(c, hipMemsetAsync(c->P.clk_probe, 0, sizeof(unsigned long long) * 4 * (size_t)m, (hipStream_t)hip_stream));
    c->probe_launch = true;
    const int rc = msx_logprob_batch_dev(c, mode, d_theta, n, ndim, d_logp, d_status, hip_stream, block_threads);
    c->probe_launch = false;
    if (rc != MSX_OK) return rc;
    HIP_TRY(c, hipStreamSynchronize((hipStream_t)hip_stream));
    std::vector<unsigned long long> h((size_t)(4 * m));
    HIP_TRY(c, hipMemcpy(h.data(), c->P.clk_probe, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost));
    std::vector<double> mhz, chain;
    unsigned long long w0 = ~0ull, w1 = 0ull;
    for (int64_t i = 0; i < m; ++i) {
        const unsigned long long a = h[(size_t)(4 * i)], ca = h[(size_t)(4 * i + 1)], b = h[(size_t)(4 * i + 2)], cb = h[(size_t)(4 * i + 3)];
        if (a == 0ull || b <= a) continue;  // (a walker the pair planner or an early exit finished: no stamps)
        mhz.push_back((double)(cb - ca) / (double)(b - a) * 100.0);  // wall clock: 100 MHz
        chain.push_back((double)(b - a) / 100.0);
        w0 = std::min(w0, a); w1 = std::max(w1, b);
    }
    out4[0] = out4[1] = out4[2] = out4[3] = 0.0;
    if (mhz.empty()) return MSX_OK;
    std::sort(mhz.begin(), mhz.end()); std::sort(chain.begin(), chain.end());
    out4[0] = mhz[mhz.size() / 2];
    out4[1] = chain[chain.size() / 2];
    out4[2] = chain.back();
    out4[3] = (double)(w1 - w0) / 100.0;
    return MSX_OK;
}

int msx_logprob_batch(msx_ctx *c, int32_t mode, const double *theta, int64_t n, int32_t ndim, double *logp_out,
                      int32_t *status_out) {
    if (!c) return MSX_ERR_INVALID;
    if (!theta || !logp_out || !status_out || n < 0) return fail(c, MSX_ERR_INVALID, "msx_logprob_batch: bad arguments");
    // everything the copy below relies on is checked BEFORE the pinned staging buffer is sized or written
    if (!c->problem_staged) return fail(c, MSX_ERR_STATE, "msx_logprob_batch: no problem staged");
    if (ndim != 2 * c->P.nspec + 2 || ndim > MSX_MAX_DIM)
        return fail(c, MSX_ERR_INVALID, "P0 doesn't match what I was expecting (ndim must be 2*nspec+2)");
    if (n == 0) return MSX_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    if (n > c->cap_walkers) {
        if (c->d_theta) (void)hipFree(c->d_theta);
        if (c->d_logp) (void)hipFree(c->d_logp);
        if (c->h_pin) (void)hipHostFree(c->h_pin);
        c->d_theta = c->d_logp = nullptr; c->d_status = nullptr; c->h_pin = nullptr; c->cap_walkers = 0;
        const int64_t cap = std::max<int64_t>(n, 1024);
        HIP_TRY(c, hipMalloc((void **)&c->d_theta, sizeof(double) * cap * MSX_MAX_DIM));
        // log-probs and statuses share one device allocation
        HIP_TRY(c, hipMalloc((void **)&c->d_logp, (sizeof(double) + sizeof(int32_t)) * cap));
        c->d_status = reinterpret_cast<int32_t *>(c->d_logp + cap);
        // pinned staging: async copies from / to pageable memory are staged synchronously by the runtime and
        // cost ~15 us each; through pinned memory the whole call is launch + ~12 us
        HIP_TRY(c, hipHostMalloc((void **)&c->h_pin, (sizeof(double) * (MSX_MAX_DIM + 1) + sizeof(int32_t)) * cap, hipHostMallocDefault));
        c->cap_walkers = cap;
    }
    const int64_t cap = c->cap_walkers;
    double *h_theta = reinterpret_cast<double *>(c->h_pin);
    double *h_out = h_theta + cap * MSX_MAX_DIM;  // [cap] log-probs followed by [cap] int32 statuses
    memcpy(h_theta, theta, sizeof(double) * n * ndim);
    if (c->zero_copy) {
        // the kernel reads theta from, and writes its n results to, the pinned (device-mapped, coherent) staging
        // buffer itself -- 48 + 12 bytes per walker over PCIe instead of two copy commands (41 -> 38 us per
        // 256-walker call, 122 -> 111 us at 2,048)
        int32_t *h_st = reinterpret_cast<int32_t *>(h_out + n);
        int rc = msx_logprob_batch_dev(c, mode, h_theta, n, ndim, h_out, h_st, c->stream, 0);
        if (rc) return rc;
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        memcpy(logp_out, h_out, sizeof(double) * n);
        memcpy(status_out, h_st, sizeof(int32_t) * n);
        note_handover(c, status_out, n);
        return MSX_OK;
    }
    HIP_TRY(c, hipMemcpyAsync(c->d_theta, h_theta, sizeof(double) * n * ndim, hipMemcpyHostToDevice, c->stream));
    // the statuses go right behind this call's n log-probs, so that one copy brings both back
    int32_t *d_st = reinterpret_cast<int32_t *>(c->d_logp + n);
    int rc = msx_logprob_batch_dev(c, mode, c->d_theta, n, ndim, c->d_logp, d_st, c->stream, 0);
    if (rc) return rc;
    HIP_TRY(c, hipMemcpyAsync(h_out, c->d_logp, (sizeof(double) + sizeof(int32_t)) * n, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    memcpy(logp_out, h_out, sizeof(double) * n);
    memcpy(status_out, reinterpret_cast<int32_t *>(h_out + n), sizeof(int32_t) * n);
    note_handover(c, status_out, n);
    return MSX_OK;
}

int msx_opt_init(msx_ctx *c, const double *theta0, int64_t nchains, int32_t ndim, double *chi2_out,
                 int32_t *status_out) {
    if (!c) return MSX_ERR_INVALID;
    if (!c->problem_staged) return fail(c, MSX_ERR_STATE, "msx_opt_init: no problem staged");
    if (!theta0 || !chi2_out || !status_out || nchains < 1) return fail(c, MSX_ERR_INVALID, "msx_opt_init: bad arguments");
    HIP_TRY(c, hipSetDevice(c->device));
    if (c->d_opt_flux) (void)hipFree(c->d_opt_flux);
    if (c->d_opt_med) (void)hipFree(c->d_opt_med);
    c->d_opt_flux = c->d_opt_med = nullptr; c->opt_chains = 0;
    c->P.opt_flux = c->P.opt_med = nullptr;
    HIP_TRY(c, hipMalloc((void **)&c->d_opt_flux, sizeof(double) * nchains * c->P.npix));
    HIP_TRY(c, hipMalloc((void **)&c->d_opt_med, sizeof(double) * nchains));
    c->opt_chains = nchains;
    c->P.opt_flux = c->d_opt_flux; c->P.opt_med = c->d_opt_med; c->P.opt_chain = nullptr;
    return msx_logprob_batch(c, MSX_MODE_OPT_INIT, theta0, nchains, ndim, chi2_out, status_out);
}

int msx_opt_step(msx_ctx *c, const double *theta, const int32_t *chain, int64_t n, int32_t ndim, double *chi2_out,
                 int32_t *status_out) {
    if (!c) return MSX_ERR_INVALID;
    if (!c->problem_staged || !c->P.opt_flux) return fail(c, MSX_ERR_STATE, "msx_opt_step: call msx_opt_init first");
    if (!theta || !chain || !chi2_out || !status_out || n < 0) return fail(c, MSX_ERR_INVALID, "msx_opt_step: bad arguments");
    if (n == 0) return MSX_OK;
    for (int64_t i = 0; i < n; ++i)
        if (chain[i] < 0 || chain[i] >= c->opt_chains) return fail(c, MSX_ERR_INVALID, "msx_opt_step: chain index out of range");
    HIP_TRY(c, hipSetDevice(c->device));
    if (n > c->cap_chain) {
        if (c->d_opt_chain) (void)hipFree(c->d_opt_chain);
        c->d_opt_chain = nullptr; c->cap_chain = 0;
        HIP_TRY(c, hipMalloc((void **)&c->d_opt_chain, sizeof(int32_t) * std::max<int64_t>(n, 1024)));
        c->cap_chain = std::max<int64_t>(n, 1024);
    }
    HIP_TRY(c, hipMemcpyAsync(c->d_opt_chain, chain, sizeof(int32_t) * n, hipMemcpyHostToDevice, c->stream));
    c->P.opt_chain = c->d_opt_chain;
    return msx_logprob_batch(c, MSX_MODE_OPT_STEP, theta, n, ndim, chi2_out, status_out);
}

// ---- device-resident sampler, pipelined -------------------------------------------------------------------
// begin: ensemble state + two slots (device chunk buffers, pinned host staging, events).  enqueue(slot): the
// chunk's randomness goes host -> pinned -> device on the copy stream while the previous chunk's kernels run,
// its 2*nsteps fused launches go on the compute stream, its chain comes back on the copy stream.  collect(slot)
// waits for that slot only.  With two slots the compute stream never drains between chunks.
struct SamplerRun {
    int32_t mode = 0, ndim = 0;
    int64_t nw = 0, ns = 0, cap_steps = 0;
    bool failed = false;  // an enqueue returned an error after queuing part of its launches
    // sharded run (msx_sampler_shard): this rank evaluates block `rank` of every half-step's ns proposals
    bool sharded = false;
    int32_t world = 1, rank = 0;
    int64_t shard_m = 0;            // ceil(ns / world): walkers per rank, and the all-gather's count
    double *d_newlp_all = nullptr;  // [shard_m * world] gathered log p(q) of the half-step
    char *d_state = nullptr;
    double *d_coords = nullptr, *d_logp = nullptr, *d_q = nullptr, *d_newlp = nullptr;
    int64_t *d_nacc = nullptr;
    int32_t *d_wst = nullptr;
    hipStream_t copy = nullptr, up = nullptr;  // downloads / uploads: separate queues, or chunk i+1's upload
                                               // would wait behind chunk i's download (which waits for its kernels)
    // Overlapped half-steps (one GPU, a half-step that fills at most half the CUs): half-step j goes to stream j & 1
    // and starts while j - 1 is still running; its workgroups wait per walker for the versions they read (SmpRec,
    // logprob_kernel.h), the coordinates are double-buffered by version parity, and an event keeps j behind j - 3 --
    // the last launch that may still read what j overwrites (j - 2 shares j's stream).
    int overlap = -1;               // -1: decided at the first chunk; 0 / 1
    hipStream_t s2 = nullptr;
    hipEvent_t hs_done[4] = {nullptr, nullptr, nullptr, nullptr}, chunk_open = nullptr, s2_done = nullptr;
    int64_t steps_done = 0;         // iterations queued so far = every walker's version when they are done
    uint32_t *d_ver = nullptr;
    unsigned long long *d_gran = nullptr;  // [2][nw][kGranPerWalker] the hand-over's tagged granules (dev_types.h)
    struct Slot {
        char *d_in = nullptr, *d_out = nullptr, *h_in = nullptr, *h_out = nullptr;
        hipEvent_t in_ready = nullptr, kernels_done = nullptr, out_ready = nullptr;
        int64_t nsteps = 0;
        bool busy = false;
    } slot[2];
    double *coords_now() const { return d_coords + (overlap == 1 ? (steps_done & 1) * nw * ndim : 0); }
    size_t in_bytes(int64_t st) const {  // [zz | zfac | logu | sidx | cidx | partner | records]
        return (size_t)(st * 2 * ns) * (3 * sizeof(double) + 3 * sizeof(int32_t) + sizeof(SmpRec));
    }
    size_t out_bytes(int64_t st) const {
        return sizeof(double) * (size_t)(st * nw * ndim + st * nw) + sizeof(int64_t) * (size_t)nw + 16;
    }
};

static void sampler_free(msx_ctx *c) {
    SamplerRun *r = c->smp;
    if (!r) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    if (r->copy) (void)hipStreamSynchronize(r->copy);
    if (r->up) (void)hipStreamSynchronize(r->up);
    for (auto &sl : r->slot) {
        if (sl.d_in) (void)hipFree(sl.d_in);
        if (sl.d_out) (void)hipFree(sl.d_out);
        if (sl.h_in) (void)hipHostFree(sl.h_in);
        if (sl.h_out) (void)hipHostFree(sl.h_out);
        if (sl.in_ready) (void)hipEventDestroy(sl.in_ready);
        if (sl.kernels_done) (void)hipEventDestroy(sl.kernels_done);
        if (sl.out_ready) (void)hipEventDestroy(sl.out_ready);
    }
    if (r->s2) (void)hipStreamSynchronize(r->s2);
    for (auto &ev : r->hs_done)
        if (ev) (void)hipEventDestroy(ev);
    if (r->chunk_open) (void)hipEventDestroy(r->chunk_open);
    if (r->s2_done) (void)hipEventDestroy(r->s2_done);
    if (r->s2) (void)hipStreamDestroy(r->s2);
    if (r->d_state) (void)hipFree(r->d_state);
    if (r->d_newlp_all) (void)hipFree(r->d_newlp_all);
    if (r->copy) (void)hipStreamDestroy(r->copy);
    if (r->up) (void)hipStreamDestroy(r->up);
    delete r;
    c->smp = nullptr;
    c->P.smp_on = 0;
    c->P.smp_defer = 0;
    c->P.smp_overlap = 0;
    c->smp_overlap_launch = false;
}

int msx_sampler_begin(msx_ctx *c, int32_t mode, int64_t nw, int32_t ndim, int64_t max_chunk_steps, const double *coords,
                      const double *logp, const int64_t *naccept) {
    if (!c) return MSX_ERR_INVALID;
    if (!c->problem_staged) return fail(c, MSX_ERR_STATE, "msx_sampler_begin: no problem staged");
    if (!coords || !logp || nw < 2 || (nw & 1) || max_chunk_steps < 1)
        return fail(c, MSX_ERR_INVALID, "msx_sampler_begin: bad arguments (need an even number of walkers)");
    if (ndim != 2 * c->P.nspec + 2) return fail(c, MSX_ERR_INVALID, "P0 doesn't match what I was expecting");
    if (mode != MSX_MODE_LOGPOST && mode != MSX_MODE_LOGLIKE) return fail(c, MSX_ERR_INVALID, "msx_sampler_begin: bad mode");
    HIP_TRY(c, hipSetDevice(c->device));
    sampler_free(c);
    SamplerRun *r = new SamplerRun;
    c->smp = r;
    r->mode = mode; r->ndim = ndim; r->nw = nw; r->ns = nw / 2; r->cap_steps = max_chunk_steps;
    const int64_t ns = r->ns;
    // (two coordinate buffers, two half-steps' worth of per-launch outputs: overlapped half-steps)
    const size_t gran_words = (size_t)(2 * nw * kGranPerWalker);
    std::vector<unsigned long long> hg;  // (lives until the stream has been synchronised below)
    const size_t state_bytes = sizeof(double) * (size_t)(2 * nw * ndim + nw + 2 * ns * ndim + 2 * ns) + sizeof(int64_t) * (size_t)nw +
                               sizeof(int32_t) * (size_t)(2 * ns) + sizeof(uint32_t) * (size_t)nw + 64 + sizeof(unsigned long long) * gran_words;
    hipError_t e = hipMalloc((void **)&r->d_state, state_bytes);
    if (e == hipSuccess) e = hipMemsetAsync(r->d_state, 0, state_bytes, c->stream);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&r->copy, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&r->up, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&r->s2, hipStreamNonBlocking);
    for (auto &ev : r->hs_done)
        if (e == hipSuccess) e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&r->chunk_open, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&r->s2_done, hipEventDisableTiming);
    for (auto &sl : r->slot) {
        if (e == hipSuccess) e = hipMalloc((void **)&sl.d_in, r->in_bytes(max_chunk_steps));
        if (e == hipSuccess) e = hipMalloc((void **)&sl.d_out, r->out_bytes(max_chunk_steps));
        if (e == hipSuccess) e = hipHostMalloc((void **)&sl.h_in, r->in_bytes(max_chunk_steps), hipHostMallocDefault);
        if (e == hipSuccess) e = hipHostMalloc((void **)&sl.h_out, r->out_bytes(max_chunk_steps), hipHostMallocDefault);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&sl.in_ready, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&sl.kernels_done, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&sl.out_ready, hipEventDisableTiming);
    }
    if (e == hipSuccess) {
        r->d_coords = (double *)r->d_state; r->d_logp = r->d_coords + 2 * nw * ndim; r->d_q = r->d_logp + nw;
        r->d_newlp = r->d_q + 2 * ns * ndim; r->d_nacc = (int64_t *)(r->d_newlp + 2 * ns); r->d_wst = (int32_t *)(r->d_nacc + nw);
        r->d_ver = (uint32_t *)(r->d_wst + 2 * ns);
        r->d_gran = (unsigned long long *)(((uintptr_t)(r->d_ver + nw) + 15) & ~(uintptr_t)15);
        e = hipMemcpyAsync(r->d_coords, coords, sizeof(double) * nw * ndim, hipMemcpyHostToDevice, c->stream);
    }
    if (e == hipSuccess) {
        // the granules of version 0 (buffer 0); buffer 1 carries a version nobody asks for until it is written
        hg.assign(gran_words, granule(0u, 0xffffffffu));
        for (int64_t w = 0; w < nw; ++w) {
            unsigned long long *g = hg.data() + (size_t)w * kGranPerWalker;
            for (int d = 0; d < ndim; ++d) {
                unsigned long long b;
                memcpy(&b, &coords[w * ndim + d], 8);
                g[2 * d] = granule((unsigned int)(b >> 32), 0u);
                g[2 * d + 1] = granule((unsigned int)b, 0u);
            }
            unsigned long long lb;
            memcpy(&lb, &logp[w], 8);
            g[kGranLogp] = granule((unsigned int)(lb >> 32), 0u);
            g[kGranLogp + 1] = granule((unsigned int)lb, 0u);
            g[kGranNacc] = granule((unsigned int)(naccept ? naccept[w] : 0), 0u);
        }
        e = hipMemcpyAsync(r->d_gran, hg.data(), sizeof(unsigned long long) * gran_words, hipMemcpyHostToDevice, c->stream);  // (behind the memset)
    }
    if (e == hipSuccess) e = hipMemcpyAsync(r->d_logp, logp, sizeof(double) * nw, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess)
        e = naccept ? hipMemcpyAsync(r->d_nacc, naccept, sizeof(int64_t) * nw, hipMemcpyHostToDevice, c->stream)
                    : hipMemsetAsync(r->d_nacc, 0, sizeof(int64_t) * nw, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);  // the caller's arrays are consumed on return
    if (e != hipSuccess) {
        sampler_free(c);
        return fail(c, MSX_ERR_HIP, std::string("msx_sampler_begin: ") + hipGetErrorString(e));
    }
    return MSX_OK;
}

int msx_sampler_shard(msx_ctx *c, int32_t rank, int32_t world) {
    if (!c) return MSX_ERR_INVALID;
    SamplerRun *r = c->smp;
    if (!r) return fail(c, MSX_ERR_STATE, "msx_sampler_shard: call msx_sampler_begin first");
    if (world < 1 || rank < 0 || rank >= world) return fail(c, MSX_ERR_INVALID, "msx_sampler_shard: bad rank / world");
    const bool have_comm = c->rccl_comm || !c->loop_peers.empty();
    if (world > 1 && (!have_comm || c->comm_world != world || c->comm_rank != rank))
        return fail(c, MSX_ERR_STATE, "msx_sampler_shard: msx_comm_init(rank, world) or msx_comm_init_loopback must come first");
    for (auto &sl : r->slot)
        if (sl.busy) return fail(c, MSX_ERR_STATE, "msx_sampler_shard: chunks are already in flight");
    HIP_TRY(c, hipSetDevice(c->device));
    r->world = world; r->rank = rank;
    r->shard_m = (r->ns + world - 1) / world;
    if (r->d_newlp_all) (void)hipFree(r->d_newlp_all);
    r->d_newlp_all = nullptr;
    HIP_TRY(c, hipMalloc((void **)&r->d_newlp_all, sizeof(double) * (size_t)(r->shard_m * world)));
    HIP_TRY(c, hipMemset(r->d_newlp_all, 0, sizeof(double) * (size_t)(r->shard_m * world)));
    r->sharded = true;
    return MSX_OK;
}

// ---- one chunk of the device-resident sampler, in pieces ------------------------------------------------------
// msx_sampler_enqueue = prepare; { eval; gather; apply } per half-step; finish.  The loopback group's entry point
// (msx_sampler_enqueue_group) runs the same pieces for all its ranks in lock-step, with device copies for the gather.
struct ChunkPtrs {
    int64_t nh = 0;
    double *d_zz = nullptr, *d_zfac = nullptr, *d_logu = nullptr;
    int32_t *d_sidx = nullptr, *d_cidx = nullptr, *d_partner = nullptr;
    const SmpRec *d_rec = nullptr;
    double *d_chain = nullptr, *d_lpchain = nullptr;
    int64_t *d_nacc_snap = nullptr;
    int32_t *d_worst = nullptr;
};

// (draw != nullptr: the chunk's randomness is drawn on the device -- sampler_draw_kernel, keyed by draw->seed and the
// run's absolute iteration numbers -- instead of coming from the host's arrays)
struct DeviceDraw { unsigned long long seed; double a; };
static int chunk_prepare(msx_ctx *c, int32_t slot, int64_t nsteps, const int32_t *sidx, const int32_t *cidx,
                         const int32_t *partner, const double *zz, const double *zfac, const double *logu, ChunkPtrs *cp,
                         const DeviceDraw *draw = nullptr) {
    SamplerRun *r = c->smp;
    if (!r) return fail(c, MSX_ERR_STATE, "msx_sampler_enqueue: call msx_sampler_begin first");
    if (slot < 0 || slot > 1 || nsteps < 1 || nsteps > r->cap_steps || (!draw && (!sidx || !cidx || !partner || !zz || !zfac || !logu)))
        return fail(c, MSX_ERR_INVALID, "msx_sampler_enqueue: bad arguments");
    if (draw && (r->nw > kDrawMaxWalkers || !(draw->a > 1.0)))
        return fail(c, MSX_ERR_INVALID, "msx_sampler_enqueue_drawn: the device generator takes up to 4096 walkers and a stretch scale a > 1");
    if (r->failed)
        return fail(c, MSX_ERR_STATE, "msx_sampler_enqueue: an earlier enqueue failed part-way; end this run (msx_sampler_end) and begin again");
    SamplerRun::Slot &sl = r->slot[slot];
    if (sl.busy) return fail(c, MSX_ERR_STATE, "msx_sampler_enqueue: slot not collected yet");
    HIP_TRY(c, hipSetDevice(c->device));
    const int64_t ns = r->ns, nw = r->nw, nh = nsteps * 2 * ns;
    const int ndim = r->ndim;
    // Overlapped half-steps?  Decided once per run, here (sharding is set up after msx_sampler_begin): an unsharded run
    // whose half-step takes the fused kernel (one workgroup per walker) and fills at most HALF the CUs -- two half-steps
    // are resident together, and a workgroup that waits for a walker of the half-step before it must never keep that
    // walker's workgroup off the chip.  MSX_SMP_OVERLAP=0 in the environment: never.
    if (r->overlap < 0) {
        const int64_t cus = c->prop.multiProcessorCount > 0 ? c->prop.multiProcessorCount : 256;
        const char *e = getenv("MSX_SMP_OVERLAP");
        // (Not on a context that holds a communicator: RCCL's kernels take CUs this rule counts on.  Two half-steps of
        // config 2's ensemble need every CU of the chip -- one workgroup each -- so anything else resident on the device
        // (another process, another context's launches) can keep a waiting workgroup's producer off the chip: the wait is
        // bounded, the chunk then ends with MSX_W_HANDOVER and the run is lost -- msx_sampler_policy(ctx, 0) or
        // MSX_SMP_OVERLAP=0 gives the plain launches on a device that is shared.)
        r->overlap = !(e && e[0] == '0') && c->smp_overlap_policy != 0 && !r->sharded && !c->rccl_comm && c->loop_peers.empty() &&
                     2 * ns <= cus && !c->model_in_global && c->recipe_fast &&
                     c->path != MSX_PATH_LINKED && c->path != MSX_PATH_PAIR && !auto_takes_linked(c, ns);
    }
    if (draw) {
        // the chunk's arrays, written where the host-fed path uploads them: [zz | zfac | logu | sidx | cidx | partner | records]
        double *g_zz = (double *)sl.d_in, *g_zfac = g_zz + nh, *g_logu = g_zfac + nh;
        int32_t *g_sidx = (int32_t *)(g_logu + nh), *g_cidx = g_sidx + nh, *g_partner = g_cidx + nh;
        SmpRec *g_rec = (SmpRec *)(g_partner + nh);
        hipLaunchKernelGGL(sampler_draw_kernel, dim3((unsigned)nsteps), dim3(kDrawThreads), 0, c->stream, draw->seed, draw->a, r->steps_done,
                           nw, (int32_t)ndim, 1, (int32_t)(r->overlap == 1), g_sidx, g_cidx, g_partner, g_zz, g_zfac, g_logu, g_rec);
        HIP_TRY(c, hipGetLastError());
    } else {
    // pinned staging, doubles first: [zz | zfac | logu | sidx | cidx | partner]
    double *hz = (double *)sl.h_in;
    int32_t *hi = (int32_t *)(hz + 3 * nh);
    memcpy(hz, zz, sizeof(double) * nh); memcpy(hz + nh, zfac, sizeof(double) * nh); memcpy(hz + 2 * nh, logu, sizeof(double) * nh);
    memcpy(hi, sidx, sizeof(int32_t) * nh); memcpy(hi + nh, cidx, sizeof(int32_t) * nh);
    memcpy(hi + 2 * nh, partner, sizeof(int32_t) * nh);
    // every index is dereferenced on the device: check them here
    for (int64_t i = 0; i < nh; ++i)
        if ((uint32_t)hi[i] >= (uint32_t)nw || (uint32_t)hi[nh + i] >= (uint32_t)nw || (uint32_t)hi[2 * nh + i] >= (uint32_t)ns)
            return fail(c, MSX_ERR_INVALID, "msx_sampler_enqueue: walker / partner index out of range");
    // resolve partner -> ensemble index of the complementary walker here, so that the kernel's proposal needs
    // two dependent loads (index, coordinates) instead of three
    for (int64_t i = 0; i < nh; ++i) hi[2 * nh + i] = hi[nh + (i / ns) * ns + hi[2 * nh + i]];
    if (r->overlap == 1) {
        // the version protocol rests on every walker moving exactly once per iteration: the two half-steps' walkers must
        // be a permutation of the ensemble (emcee's random split is; checked here because a violation would not fail
        // until a workgroup's wait runs out on the device)
        std::vector<int64_t> seen((size_t)nw, -1);
        for (int64_t i = 0; i < nh; ++i) {
            const int64_t it = i / (2 * ns);
            if (seen[(size_t)hi[i]] == it)
                return fail(c, MSX_ERR_INVALID, "msx_sampler_enqueue: a walker appears twice in one iteration's two half-steps");
            seen[(size_t)hi[i]] = it;
        }
    }
    // ... and the proposal's inputs once more as one record per walker (the kernel's first load), with the versions of
    // the two walkers the move reads: before iteration k every walker has version k; the second half-step's partners
    // were updated by the first
    SmpRec *hr = (SmpRec *)(hi + 3 * nh);
    for (int64_t i = 0; i < nh; ++i) {
        hr[i].si = hi[i]; hr[i].ci = hi[2 * nh + i]; hr[i].zz = hz[i];
        const int64_t k = r->steps_done + i / (2 * ns), half = (i / ns) & 1;
        hr[i].ver_own = r->overlap == 1 ? (uint32_t)k : 0u;
        hr[i].ver_partner = r->overlap == 1 ? (uint32_t)(k + half) : 0u;
    }
    HIP_TRY(c, hipMemcpyAsync(sl.d_in, sl.h_in, r->in_bytes(nsteps), hipMemcpyHostToDevice, r->up));
    HIP_TRY(c, hipEventRecord(sl.in_ready, r->up));
    HIP_TRY(c, hipStreamWaitEvent(c->stream, sl.in_ready, 0));
    }
    cp->nh = nh;
    cp->d_zz = (double *)sl.d_in; cp->d_zfac = cp->d_zz + nh; cp->d_logu = cp->d_zfac + nh;
    cp->d_sidx = (int32_t *)(cp->d_logu + nh); cp->d_cidx = cp->d_sidx + nh; cp->d_partner = cp->d_cidx + nh;
    cp->d_rec = (const SmpRec *)(cp->d_partner + nh);
    cp->d_chain = (double *)sl.d_out; cp->d_lpchain = cp->d_chain + nsteps * nw * ndim;
    cp->d_nacc_snap = (int64_t *)(cp->d_lpchain + nsteps * nw);
    cp->d_worst = (int32_t *)(cp->d_nacc_snap + nw);
    HIP_TRY(c, hipMemsetAsync(cp->d_worst, 0, sizeof(int32_t), c->stream));
    DevProblem &P = c->P;
    P.smp_on = 1;
    P.smp_coords = r->d_coords; P.smp_logp = r->d_logp; P.smp_q = r->d_q; P.smp_naccept = r->d_nacc; P.smp_worst = cp->d_worst;
    P.smp_overlap = r->overlap == 1; P.smp_stride = nw * ndim; P.smp_ver = r->d_ver;
    P.smp_gran = r->d_gran; P.smp_gwalkers = nw;
    if (r->overlap == 1) {  // the second stream's launches of this chunk come after the chunk's inputs and the cleared status
        HIP_TRY(c, hipEventRecord(r->chunk_open, c->stream));
        HIP_TRY(c, hipStreamWaitEvent(r->s2, r->chunk_open, 0));
    }
    return MSX_OK;
}

// the half-step's pointers; then this rank's evaluation: the whole half-step fused (unsharded), or log p(q) of its
// block of the proposals only, accept deferred (sharded)
static int chunk_half_eval(msx_ctx *c, const ChunkPtrs &cp, int64_t st, int half) {
    SamplerRun *r = c->smp;
    DevProblem &P = c->P;
    const int64_t ns = r->ns, nw = r->nw;
    const int ndim = r->ndim;
    const int64_t off = (st * 2 + half) * ns;
    P.smp_sidx = cp.d_sidx + off; P.smp_cidx = cp.d_cidx + off; P.smp_partner = cp.d_partner + off;
    P.smp_zz = cp.d_zz + off; P.smp_zfac = cp.d_zfac + off; P.smp_logu = cp.d_logu + off; P.smp_rec = cp.d_rec + off;
    P.smp_chain_row = cp.d_chain + st * nw * ndim; P.smp_lp_row = cp.d_lpchain + st * nw;
    if (r->overlap == 1) {
        // half-step j on stream j & 1, behind j - 3 (see SamplerRun); per-launch outputs nobody reads go to the
        // stream's own half of their arrays
        const int64_t j = (r->steps_done + st) * 2 + half;
        hipStream_t s = (j & 1) ? r->s2 : c->stream;
        if (j >= 3) HIP_TRY(c, hipStreamWaitEvent(s, r->hs_done[(j - 3) & 3], 0));
        P.smp_q = r->d_q + (j & 1) * ns * ndim;
        c->smp_overlap_launch = true;
        const int rc = msx_logprob_batch_dev(c, r->mode, P.smp_q, ns, ndim, r->d_newlp + (j & 1) * ns, r->d_wst + (j & 1) * ns, s, 0);
        c->smp_overlap_launch = false;
        if (rc != MSX_OK) return rc;
        HIP_TRY(c, hipEventRecord(r->hs_done[j & 3], s));
        return MSX_OK;
    }
    if (!r->sharded) return msx_logprob_batch_dev(c, r->mode, r->d_q, ns, ndim, r->d_newlp, r->d_wst, c->stream, 0);
    // sharded: (1) this rank's block of proposals -> log p(q) only; (2) ONE all-gather of shard_m float64 per rank, in
    // place in the gathered vector; (3) every rank finishes the half-step for all ns walkers
    const int64_t lo = std::min<int64_t>(r->rank * r->shard_m, ns), hi = std::min<int64_t>(lo + r->shard_m, ns);
    int rc = MSX_OK;
    if (hi > lo) {
        const DevProblem keep = P;
        P.smp_defer = 1;
        P.smp_sidx += lo; P.smp_cidx += lo; P.smp_partner += lo; P.smp_zz += lo; P.smp_zfac += lo; P.smp_logu += lo;
        P.smp_rec += lo;
        P.smp_q = r->d_q + lo * ndim;
        rc = msx_logprob_batch_dev(c, r->mode, r->d_q + lo * ndim, hi - lo, ndim, r->d_newlp_all + lo, r->d_wst + lo, c->stream, 0);
        P = keep;
    }
    return rc;
}

static int chunk_half_gather_rccl(msx_ctx *c) {
    SamplerRun *r = c->smp;
    if (!(c->rccl_comm && c->comm_world == r->world)) return MSX_OK;  // (world 1 without a communicator: nothing to do)
    // (a one-rank communicator still runs the collective)
    const int nrc = rccl().AllGather(r->d_newlp_all + r->rank * r->shard_m, r->d_newlp_all, (size_t)r->shard_m, kNcclFloat64,
                                     c->rccl_comm, c->stream);
    if (nrc != 0) return fail(c, MSX_ERR_HIP, std::string("ncclAllGather: ") + rccl().GetErrorString(nrc));
    return MSX_OK;
}

static int chunk_half_apply(msx_ctx *c) {
    SamplerRun *r = c->smp;
    if (!r->sharded) return MSX_OK;
    hipLaunchKernelGGL(sampler_apply_kernel, dim3((unsigned)((r->ns + 255) / 256)), dim3(256), 0, c->stream, c->P,
                       r->d_newlp_all, r->ns, r->ndim);
    if (hipGetLastError() != hipSuccess) return fail(c, MSX_ERR_HIP, "sampler_apply_kernel launch failed");
    return MSX_OK;
}

static int chunk_finish(msx_ctx *c, int32_t slot, int64_t nsteps, const ChunkPtrs &cp, int rc) {
    SamplerRun *r = c->smp;
    SamplerRun::Slot &sl = r->slot[slot];
    DevProblem &P = c->P;
    P.smp_on = 0;
    P.smp_defer = 0;
    if (rc != MSX_OK) {
        // some of this chunk's half-steps may already be queued: the resident state is no longer the state any
        // host-side bookkeeping expects.  Refuse everything but msx_sampler_end from here on.
        r->failed = true;
        return rc;
    }
    if (r->overlap == 1) {  // the chunk's last half-step ran on the second stream
        HIP_TRY(c, hipEventRecord(r->s2_done, r->s2));
        HIP_TRY(c, hipStreamWaitEvent(c->stream, r->s2_done, 0));
    }
    r->steps_done += nsteps;
    // acceptance counters keep running while this chunk's results travel: snapshot them in stream order
    HIP_TRY(c, hipMemcpyAsync(cp.d_nacc_snap, r->d_nacc, sizeof(int64_t) * r->nw, hipMemcpyDeviceToDevice, c->stream));
    HIP_TRY(c, hipEventRecord(sl.kernels_done, c->stream));
    HIP_TRY(c, hipStreamWaitEvent(r->copy, sl.kernels_done, 0));
    HIP_TRY(c, hipMemcpyAsync(sl.h_out, sl.d_out, r->out_bytes(nsteps), hipMemcpyDeviceToHost, r->copy));
    HIP_TRY(c, hipEventRecord(sl.out_ready, r->copy));
    sl.nsteps = nsteps;
    sl.busy = true;
    return MSX_OK;
}

int msx_sampler_enqueue(msx_ctx *c, int32_t slot, int64_t nsteps, const int32_t *sidx, const int32_t *cidx,
                        const int32_t *partner, const double *zz, const double *zfac, const double *logu) {
    if (!c) return MSX_ERR_INVALID;
    if (c->smp && c->smp->sharded && c->smp->world > 1) {
        if (!c->loop_peers.empty())
            return fail(c, MSX_ERR_STATE, "msx_sampler_enqueue: the ranks of a loopback group advance together (msx_sampler_enqueue_group)");
        if (!c->rccl_comm || c->comm_world != c->smp->world)
            return fail(c, MSX_ERR_STATE, "msx_sampler_enqueue: the run is sharded over a communicator that no longer exists");
    }
    ChunkPtrs cp;
    int rc = chunk_prepare(c, slot, nsteps, sidx, cidx, partner, zz, zfac, logu, &cp);
    if (rc != MSX_OK) return rc;
    for (int64_t st = 0; st < nsteps && rc == MSX_OK; ++st)
        for (int half = 0; half < 2 && rc == MSX_OK; ++half) {
            rc = chunk_half_eval(c, cp, st, half);
            if (rc == MSX_OK && c->smp->sharded) rc = chunk_half_gather_rccl(c);
            if (rc == MSX_OK) rc = chunk_half_apply(c);
        }
    return chunk_finish(c, slot, nsteps, cp, rc);
}

int msx_sampler_enqueue_drawn(msx_ctx *c, int32_t slot, int64_t nsteps, uint64_t seed, double a) {
    if (!c) return MSX_ERR_INVALID;
    if (c->smp && c->smp->sharded && c->smp->world > 1) {
        if (!c->loop_peers.empty())
            return fail(c, MSX_ERR_STATE, "msx_sampler_enqueue_drawn: the ranks of a loopback group advance together (msx_sampler_enqueue_group)");
        if (!c->rccl_comm || c->comm_world != c->smp->world)
            return fail(c, MSX_ERR_STATE, "msx_sampler_enqueue_drawn: the run is sharded over a communicator that no longer exists");
    }
    ChunkPtrs cp;
    const DeviceDraw dd = {(unsigned long long)seed, a};
    int rc = chunk_prepare(c, slot, nsteps, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, &cp, &dd);
    if (rc != MSX_OK) return rc;
    for (int64_t st = 0; st < nsteps && rc == MSX_OK; ++st)
        for (int half = 0; half < 2 && rc == MSX_OK; ++half) {
            rc = chunk_half_eval(c, cp, st, half);
            if (rc == MSX_OK && c->smp->sharded) rc = chunk_half_gather_rccl(c);
            if (rc == MSX_OK) rc = chunk_half_apply(c);
        }
    return chunk_finish(c, slot, nsteps, cp, rc);
}

int msx_sampler_draw(msx_ctx *c, uint64_t seed, double a, int64_t first_iter, int64_t nsteps, int64_t nw, int32_t ndim,
                     int32_t *sidx, int32_t *cidx, int32_t *partner, double *zz, double *zfac, double *logu) {
    if (!c || !sidx || !cidx || !partner || !zz || !zfac || !logu || nsteps < 1 || nw < 2 || (nw & 1) || nw > kDrawMaxWalkers ||
        first_iter < 0 || ndim < 1 || !(a > 1.0))
        return fail(c, MSX_ERR_INVALID, "msx_sampler_draw: bad arguments (an even number of walkers, at most 4096; a > 1)");
    HIP_TRY(c, hipSetDevice(c->device));
    const int64_t nh = nsteps * nw;
    char *d = nullptr;
    HIP_TRY(c, hipMalloc((void **)&d, (size_t)nh * (3 * sizeof(double) + 3 * sizeof(int32_t))));
    double *g_zz = (double *)d, *g_zfac = g_zz + nh, *g_logu = g_zfac + nh;
    int32_t *g_sidx = (int32_t *)(g_logu + nh), *g_cidx = g_sidx + nh, *g_partner = g_cidx + nh;
    hipLaunchKernelGGL(sampler_draw_kernel, dim3((unsigned)nsteps), dim3(kDrawThreads), 0, c->stream, (unsigned long long)seed, a, first_iter, nw,
                       ndim, 0, 0, g_sidx, g_cidx, g_partner, g_zz, g_zfac, g_logu, (SmpRec *)nullptr);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) e = hipMemcpy(zz, g_zz, sizeof(double) * nh, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(zfac, g_zfac, sizeof(double) * nh, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(logu, g_logu, sizeof(double) * nh, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(sidx, g_sidx, sizeof(int32_t) * nh, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(cidx, g_cidx, sizeof(int32_t) * nh, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(partner, g_partner, sizeof(int32_t) * nh, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(c, MSX_ERR_HIP, std::string("msx_sampler_draw: ") + hipGetErrorString(e));
    return MSX_OK;
}

int msx_sampler_enqueue_group(msx_ctx **ctxs, int32_t world, int32_t slot, int64_t nsteps, const int32_t *sidx,
                              const int32_t *cidx, const int32_t *partner, const double *zz, const double *zfac,
                              const double *logu) {
    if (!ctxs || world < 1 || !ctxs[0]) return MSX_ERR_INVALID;
    msx_ctx *c0 = ctxs[0];
    for (int r = 0; r < world; ++r) {
        msx_ctx *c = ctxs[r];
        if (!c || (int)c->loop_peers.size() != world || c->loop_peers[(size_t)r] != c || c->comm_rank != r)
            return fail(c0, MSX_ERR_STATE, "msx_sampler_enqueue_group: the contexts are not the ranks 0..world-1 of one loopback group");
        if (!c->smp || !c->smp->sharded || c->smp->world != world || c->smp->rank != r)
            return fail(c0, MSX_ERR_STATE, "msx_sampler_enqueue_group: every rank needs msx_sampler_begin + msx_sampler_shard(rank, world) first");
        if (c->smp->ns != c0->smp->ns || c->smp->ndim != c0->smp->ndim)
            return fail(c0, MSX_ERR_STATE, "msx_sampler_enqueue_group: the ranks hold different ensembles");
    }
    std::vector<ChunkPtrs> cp((size_t)world);
    std::vector<int> rcs((size_t)world, MSX_OK);
    int rc = MSX_OK;
    for (int r = 0; r < world && rc == MSX_OK; ++r)   // every rank is fed the same randomness
        rc = rcs[(size_t)r] = chunk_prepare(ctxs[r], slot, nsteps, sidx, cidx, partner, zz, zfac, logu, &cp[(size_t)r]);
    if (rc != MSX_OK) {  // nothing is queued yet on the ranks after the failing one; the prepared ones are unwound
        for (int r = 0; r < world; ++r) { ctxs[r]->P.smp_on = 0; ctxs[r]->P.smp_defer = 0; }
        if (ctxs[0] != c0 || c0->err.empty()) c0->err = "msx_sampler_enqueue_group: a rank refused the chunk";
        return rc;
    }
    const int64_t m = c0->smp->shard_m;
    for (int64_t st = 0; st < nsteps && rc == MSX_OK; ++st)
        for (int half = 0; half < 2 && rc == MSX_OK; ++half) {
            // (1) every rank evaluates its block into its own gathered vector -- once the peers have taken the
            //     previous half-step's block out of it
            for (int r = 0; r < world && rc == MSX_OK; ++r) {
                msx_ctx *c = ctxs[r];
                for (int p = 0; p < world; ++p)
                    if (p != r && hipStreamWaitEvent(c->stream, ctxs[p]->loop_copied, 0) != hipSuccess) rc = fail(c0, MSX_ERR_HIP, "loopback: hipStreamWaitEvent");
                if (rc == MSX_OK) rc = chunk_half_eval(c, cp[(size_t)r], st, half);
                if (rc == MSX_OK && hipEventRecord(c->loop_eval_done, c->stream) != hipSuccess) rc = fail(c0, MSX_ERR_HIP, "loopback: hipEventRecord");
            }
            // (2) the all-gather: rank r copies block p out of rank p's vector, in place at p * m; (3) apply
            for (int r = 0; r < world && rc == MSX_OK; ++r) {
                msx_ctx *c = ctxs[r];
                for (int p = 0; p < world && rc == MSX_OK; ++p) {
                    if (p == r) continue;
                    hipError_t e = hipStreamWaitEvent(c->stream, ctxs[p]->loop_eval_done, 0);
                    if (e == hipSuccess)
                        e = hipMemcpyAsync(c->smp->d_newlp_all + p * m, ctxs[p]->smp->d_newlp_all + p * m, sizeof(double) * (size_t)m,
                                           hipMemcpyDeviceToDevice, c->stream);
                    if (e != hipSuccess) rc = fail(c0, MSX_ERR_HIP, std::string("loopback all-gather: ") + hipGetErrorString(e));
                }
                if (rc == MSX_OK && hipEventRecord(c->loop_copied, c->stream) != hipSuccess) rc = fail(c0, MSX_ERR_HIP, "loopback: hipEventRecord");
                if (rc == MSX_OK) rc = chunk_half_apply(c);
            }
        }
    int out = rc;
    for (int r = 0; r < world; ++r) {
        const int f = chunk_finish(ctxs[r], slot, nsteps, cp[(size_t)r], rc);
        if (out == MSX_OK) out = f;
    }
    return out;
}

int msx_sampler_collect(msx_ctx *c, int32_t slot, double *chain_out, double *logp_out, int64_t *naccept,
                        int32_t *worst_status) {
    if (!c) return MSX_ERR_INVALID;
    SamplerRun *r = c->smp;
    if (!r) return fail(c, MSX_ERR_STATE, "msx_sampler_collect: call msx_sampler_begin first");
    if (slot < 0 || slot > 1 || !chain_out || !logp_out || !naccept || !worst_status)
        return fail(c, MSX_ERR_INVALID, "msx_sampler_collect: bad arguments");
    SamplerRun::Slot &sl = r->slot[slot];
    if (!sl.busy) return fail(c, MSX_ERR_STATE, "msx_sampler_collect: nothing enqueued in this slot");
    HIP_TRY(c, hipEventSynchronize(sl.out_ready));
    const int64_t st = sl.nsteps, nw = r->nw;
    const double *h_chain = (const double *)sl.h_out, *h_lp = h_chain + st * nw * r->ndim;
    const int64_t *h_nacc = (const int64_t *)(h_lp + st * nw);
    memcpy(chain_out, h_chain, sizeof(double) * st * nw * r->ndim);
    memcpy(logp_out, h_lp, sizeof(double) * st * nw);
    memcpy(naccept, h_nacc, sizeof(int64_t) * nw);
    *worst_status = *(const int32_t *)(h_nacc + nw);
    sl.busy = false;
    return MSX_OK;
}

int msx_sampler_end(msx_ctx *c, double *coords, double *logp) {
    if (!c) return MSX_ERR_INVALID;
    SamplerRun *r = c->smp;
    if (!r) return MSX_OK;
    hipError_t e = hipSetDevice(c->device);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess && r->s2) e = hipStreamSynchronize(r->s2);
    if (e == hipSuccess && coords) e = hipMemcpy(coords, r->coords_now(), sizeof(double) * r->nw * r->ndim, hipMemcpyDeviceToHost);
    if (e == hipSuccess && logp) e = hipMemcpy(logp, r->d_logp, sizeof(double) * r->nw, hipMemcpyDeviceToHost);
    sampler_free(c);
    if (e != hipSuccess) return fail(c, MSX_ERR_HIP, std::string("msx_sampler_end: ") + hipGetErrorString(e));
    return MSX_OK;
}

int msx_sampler_policy(msx_ctx *c, int32_t overlap) {
    if (!c || (overlap != -1 && overlap != 0)) return fail(c, MSX_ERR_INVALID, "msx_sampler_policy: -1 (automatic) or 0 (never overlap half-steps)");
    c->smp_overlap_policy = overlap;
    return MSX_OK;
}

int msx_sampler_overlapped(msx_ctx *c, int32_t *out) {
    if (!c || !out) return MSX_ERR_INVALID;
    if (!c->smp) return fail(c, MSX_ERR_STATE, "msx_sampler_overlapped: call msx_sampler_begin first");
    *out = c->smp->overlap;
    return MSX_OK;
}

// one synchronous chunk (the pipelined entry points above, used back to back)
int msx_sampler_run(msx_ctx *c, int32_t mode, int64_t nw, int32_t ndim, int64_t nsteps, double *coords, double *logp,
                    const int32_t *sidx, const int32_t *cidx, const int32_t *partner, const double *zz,
                    const double *zfac, const double *logu, double *chain_out, double *logp_out, int64_t *naccept,
                    int32_t *worst_status) {
    if (!c) return MSX_ERR_INVALID;
    if (!coords || !logp || !chain_out || !logp_out || !naccept || !worst_status || nsteps < 1)
        return fail(c, MSX_ERR_INVALID, "msx_sampler_run: bad arguments (need an even number of walkers)");
    int rc = msx_sampler_begin(c, mode, nw, ndim, nsteps, coords, logp, naccept);
    if (rc == MSX_OK) rc = msx_sampler_enqueue(c, 0, nsteps, sidx, cidx, partner, zz, zfac, logu);
    if (rc == MSX_OK) rc = msx_sampler_collect(c, 0, chain_out, logp_out, naccept, worst_status);
    if (rc == MSX_OK) return msx_sampler_end(c, coords, logp);
    const std::string keep = c->err;
    sampler_free(c);
    c->err = keep;
    return rc;
}

int msx_make_composite(msx_ctx *c, const double *teff, const double *logg, const double *rad, int32_t use_distance,
                       double plx, double *spec_out, double *contrast_out, double *phot_out, int32_t *status_out) {
    if (!c) return MSX_ERR_INVALID;
    if (!c->problem_staged) return fail(c, MSX_ERR_STATE, "msx_make_composite: no problem staged");
    if (!teff || !logg || !rad || !spec_out || !status_out) return fail(c, MSX_ERR_INVALID, "msx_make_composite: bad arguments");
    HIP_TRY(c, hipSetDevice(c->device));
    const int ns = c->P.nspec;
    double args[3 * MSX_MAX_SPEC + 1];
    for (int i = 0; i < ns; ++i) { args[i] = teff[i]; args[ns + i] = logg[i]; args[2 * ns + i] = rad[i]; }
    args[3 * ns] = plx;
    if (c->P.win_n > c->cap_spec) {
        if (c->d_spec) (void)hipFree(c->d_spec);
        c->d_spec = nullptr; c->cap_spec = 0;
        HIP_TRY(c, hipMalloc((void **)&c->d_spec, sizeof(double) * c->P.win_n));
        c->cap_spec = c->P.win_n;
    }
    double *d_args = c->d_misc;
    WalkerDesc *d_desc = reinterpret_cast<WalkerDesc *>(c->d_misc + 64);
    static_assert(sizeof(WalkerDesc) + 64 * sizeof(double) <= 4096, "misc buffer too small");
    HIP_TRY(c, hipMemcpyAsync(d_args, args, sizeof(double) * (3 * ns + 1), hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(composite_setup_kernel, dim3(1), dim3(64), 0, c->stream, c->P, d_args, (int)use_distance, d_desc);
    HIP_TRY(c, hipGetLastError());
    hipLaunchKernelGGL(composite_kernel, dim3((unsigned)((c->P.win_n + 255) / 256)), dim3(256), 0, c->stream, c->P, d_desc,
                       c->d_spec);
    HIP_TRY(c, hipGetLastError());
    WalkerDesc h;
    HIP_TRY(c, hipMemcpyAsync(&h, d_desc, sizeof(WalkerDesc), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    *status_out = h.status;
    if (h.status != MSX_W_OK) return MSX_OK;
    HIP_TRY(c, hipMemcpy(spec_out, c->d_spec, sizeof(double) * c->P.win_n, hipMemcpyDeviceToHost));
    for (int i = 0; i < c->P.nc && contrast_out; ++i) contrast_out[i] = h.contrast[i];
    for (int i = 0; i < c->P.np && phot_out; ++i) phot_out[i] = h.phot[i];
    return MSX_OK;
}

int msx_comm_unique_id(msx_ctx *c, uint8_t *out128) {
    if (!c || !out128) return MSX_ERR_INVALID;
    if (!rccl().ok) return fail(c, MSX_ERR_STATE, "RCCL (librccl.so.1) could not be resolved in this process");
    RcclUniqueId id;
    const int rc = rccl().GetUniqueId(&id);
    if (rc != 0) return fail(c, MSX_ERR_HIP, std::string("ncclGetUniqueId: ") + rccl().GetErrorString(rc));
    memcpy(out128, id.internal, 128);
    return MSX_OK;
}

int msx_comm_init(msx_ctx *c, const uint8_t *id128, int32_t rank, int32_t world) {
    if (!c || !id128 || world < 1 || rank < 0 || rank >= world) return fail(c, MSX_ERR_INVALID, "msx_comm_init: bad arguments");
    if (!rccl().ok) return fail(c, MSX_ERR_STATE, "RCCL (librccl.so.1) could not be resolved in this process");
    if (c->rccl_comm) return fail(c, MSX_ERR_STATE, "msx_comm_init: communicator already initialised");
    HIP_TRY(c, hipSetDevice(c->device));
    RcclUniqueId id;
    memcpy(id.internal, id128, 128);
    const int rc = rccl().CommInitRank(&c->rccl_comm, world, id, rank);
    if (rc != 0) {
        c->rccl_comm = nullptr;
        return fail(c, MSX_ERR_HIP, std::string("ncclCommInitRank: ") + rccl().GetErrorString(rc));
    }
    HIP_TRY(c, hipStreamCreateWithFlags(&c->comm_stream, hipStreamNonBlocking));
    HIP_TRY(c, hipEventCreateWithFlags(&c->ev_ready, hipEventDisableTiming));
    for (hipEvent_t &e : c->ev_done) HIP_TRY(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    c->comm_world = world;
    c->comm_rank = rank;
    return MSX_OK;
}

int msx_comm_init_loopback(msx_ctx **ctxs, int32_t world) {
    if (!ctxs || world < 1 || !ctxs[0]) return MSX_ERR_INVALID;
    msx_ctx *c0 = ctxs[0];
    for (int r = 0; r < world; ++r) {
        if (!ctxs[r]) return fail(c0, MSX_ERR_INVALID, "msx_comm_init_loopback: null context");
        if (ctxs[r]->rccl_comm || !ctxs[r]->loop_peers.empty())
            return fail(c0, MSX_ERR_STATE, "msx_comm_init_loopback: a context already belongs to a communicator");
        if (ctxs[r]->device != c0->device)
            return fail(c0, MSX_ERR_INVALID, "msx_comm_init_loopback: the ranks of a loopback group share one device");
        for (int q = 0; q < r; ++q)
            if (ctxs[q] == ctxs[r]) return fail(c0, MSX_ERR_INVALID, "msx_comm_init_loopback: one context per rank");
    }
    HIP_TRY(c0, hipSetDevice(c0->device));
    for (int r = 0; r < world; ++r) {
        msx_ctx *c = ctxs[r];
        HIP_TRY(c0, hipEventCreateWithFlags(&c->loop_eval_done, hipEventDisableTiming));
        HIP_TRY(c0, hipEventCreateWithFlags(&c->loop_copied, hipEventDisableTiming));
        // (recorded once so that the first half-step's waits find completed events)
        HIP_TRY(c0, hipEventRecord(c->loop_eval_done, c->stream));
        HIP_TRY(c0, hipEventRecord(c->loop_copied, c->stream));
        c->loop_peers.assign(ctxs, ctxs + world);
        c->comm_world = world;
        c->comm_rank = r;
    }
    return MSX_OK;
}

int msx_comm_allgather_dev(msx_ctx *c, const double *d_send, double *d_recv, int64_t count, void *compute_stream,
                           int32_t slot) {
    if (!c || !d_send || !d_recv || count < 1 || slot < 0 || slot > 3) return fail(c, MSX_ERR_INVALID, "msx_comm_allgather_dev: bad arguments");
    if (!c->rccl_comm) return fail(c, MSX_ERR_STATE, "msx_comm_allgather_dev: call msx_comm_init first");
    // the collective starts once everything queued so far on the compute stream is done, runs on the
    // communicator's own stream (so the next launch overlaps it) and signals the slot's event
    HIP_TRY(c, hipEventRecord(c->ev_ready, (hipStream_t)compute_stream));
    HIP_TRY(c, hipStreamWaitEvent(c->comm_stream, c->ev_ready, 0));
    const int rc = rccl().AllGather(d_send, d_recv, (size_t)count, kNcclFloat64, c->rccl_comm, c->comm_stream);
    if (rc != 0) return fail(c, MSX_ERR_HIP, std::string("ncclAllGather: ") + rccl().GetErrorString(rc));
    HIP_TRY(c, hipEventRecord(c->ev_done[slot], c->comm_stream));
    return MSX_OK;
}

int msx_comm_wait_slot(msx_ctx *c, int32_t slot, void *compute_stream) {
    if (!c || slot < 0 || slot > 3) return MSX_ERR_INVALID;
    if (!c->rccl_comm) return fail(c, MSX_ERR_STATE, "msx_comm_wait_slot: call msx_comm_init first");
    HIP_TRY(c, hipStreamWaitEvent((hipStream_t)compute_stream, c->ev_done[slot], 0));
    return MSX_OK;
}

int msx_stream_copy_gbps(msx_ctx *c, int64_t bytes, int32_t iters, double *gbps_out) {
    if (!c || !gbps_out || bytes < 4096 || iters < 1) return fail(c, MSX_ERR_INVALID, "msx_stream_copy_gbps: bad arguments");
    HIP_TRY(c, hipSetDevice(c->device));
    const int64_t n4 = bytes / 16;
    float4 *a = nullptr, *b = nullptr;
    HIP_TRY(c, hipMalloc((void **)&a, n4 * 16));
    HIP_TRY(c, hipMalloc((void **)&b, n4 * 16));
    HIP_TRY(c, hipMemsetAsync(a, 1, n4 * 16, c->stream));
    hipEvent_t e0, e1;
    HIP_TRY(c, hipEventCreate(&e0));
    HIP_TRY(c, hipEventCreate(&e1));
    const int cus = c->prop.multiProcessorCount > 0 ? c->prop.multiProcessorCount : 256;
    double best = 0.0;
    for (int variant = 0; variant < 4; ++variant) {
        for (int per_cu : {8, 16, 32}) {
            const dim3 g((unsigned)(cus * per_cu)), bthreads(256);
            auto go = [&]() {
                switch (variant) {
                    case 0: hipLaunchKernelGGL((copy_float4_kernel<4, false>), g, bthreads, 0, c->stream, a, b, n4); break;
                    case 1: hipLaunchKernelGGL((copy_float4_kernel<8, false>), g, bthreads, 0, c->stream, a, b, n4); break;
                    case 2: hipLaunchKernelGGL((copy_float4_kernel<4, true>), g, bthreads, 0, c->stream, a, b, n4); break;
                    default: hipLaunchKernelGGL((copy_float4_kernel<8, true>), g, bthreads, 0, c->stream, a, b, n4); break;
                }
            };
            go();
            HIP_TRY(c, hipEventRecord(e0, c->stream));
            for (int i = 0; i < iters; ++i) go();
            HIP_TRY(c, hipEventRecord(e1, c->stream));
            HIP_TRY(c, hipEventSynchronize(e1));
            float ms = 0.f;
            HIP_TRY(c, hipEventElapsedTime(&ms, e0, e1));
            const double gbps = (2.0 * (double)(n4 * 16) * iters) / ((double)ms * 1e-3) / 1e9;
            best = gbps > best ? gbps : best;
        }
    }
    *gbps_out = best;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipFree(a);
    (void)hipFree(b);
    return MSX_OK;
}

// bytes the form / variant an automatic launch of n walkers takes requests from the memory system, per walker
static int64_t requested_bytes_of(msx_ctx *c, int64_t n, const FormChoice &f, const Variant *v) {
    const int64_t npix = c->P.npix;
    // the pair form: two walkers per set of loads -- rows, extinction terms, the fit sweep's data flux / u, the pass's three
    // vectors -- + the planner's record
    if (f.pair) return npix * (12 * 8 + 12 + 16 + 24) / 2 + (int64_t)sizeof(PairRec) + 8 * 6 + 12;
    //   blend: 12-B {R f64, H f32} per corner + {k_lo f64, dk f32} + data flux, u (f64)        per pixel
    //   chi^2 pass: 1/err^2, and -- unless the variant kept them in LDS (PF) -- u and data flux again
    const bool pf = v && v->pf;
    const int64_t per_corner = (v && v->r32) ? 8 : 12;  // {R f64 | f32, H f32}
    int64_t b = npix * (per_corner * (int64_t)c->P.nspec * 4 + 12 + 16 + (pf ? 8 : 24)) + 8 * (2 * c->P.nspec + 2) + 12;
    // the linked form: every segment's workgroup reads theta and writes its partials (counters, sums, range; chi^2 sum
    // and candidates: <= 64 of them as a rule), reads the other segments' partials, and one of them their candidates
    if (f.linked) {
        const int64_t S = c->nseg, part = 4 * kSegBins + 64, fin = 16 + 8 * 64;
        b += (S - 1) * (8 * (2 * c->P.nspec + 2)) + S * (part + fin) + S * (S - 1) * part + (S - 1) * fin;
    }
    return b;
}

int msx_bytes_per_eval(msx_ctx *c, int64_t n, int64_t *requested_bytes) {
    if (!c || !requested_bytes || n < 1) return MSX_ERR_INVALID;
    if (!c->problem_staged) return fail(c, MSX_ERR_STATE, "msx_bytes_per_eval: no problem staged");
    const FormChoice f = decide_form(c, n, MSX_MODE_LOGPOST, true);
    if (f.err != MSX_OK) return fail(c, f.err, f.msg);
    const int64_t m = f.pair ? std::min<int64_t>(n, c->pair_rows) : (f.linked || c->model_in_global) && c->scratch_rows ? std::min<int64_t>(n, c->scratch_rows) : n;
    const VariantChoice ch = choose_variant(c, c->P, m, f.linked ? 512 : pick_block(c, m, c->P.npix), false, f.linked);
    *requested_bytes = requested_bytes_of(c, n, f, ch.v);
    return MSX_OK;
}

int msx_launch_info(msx_ctx *c, int32_t mode, int64_t n, int32_t block_threads, char *name, int32_t name_len, int64_t *out8) {
    if (!c || !out8 || n < 1) return MSX_ERR_INVALID;
    if (!c->problem_staged) return fail(c, MSX_ERR_STATE, "msx_launch_info: no problem staged");
    HIP_TRY(c, hipSetDevice(c->device));
    const bool shared512 = block_threads == MSX_BLOCK_512_SHARED;
    if (shared512) block_threads = 512;
    if (block_threads != 0 && block_threads != 256 && block_threads != 512)
        return fail(c, MSX_ERR_INVALID, "block_threads must be 0, 256, 512 or MSX_BLOCK_512_SHARED");
    const FormChoice f = decide_form(c, n, mode, true);
    if (f.err != MSX_OK) return fail(c, f.err, f.msg);
    // (the first sub-batch stands for the launch: sub-batches only differ in their walker count)
    const int64_t m = f.inpath ? std::min<int64_t>(n, c->inp_rows) : f.pair ? std::min<int64_t>(n, c->pair_rows) : (f.linked || c->model_in_global) && c->scratch_rows ? std::min<int64_t>(n, c->scratch_rows) : n;
    std::string nm;
    const void *fn = nullptr;
    int64_t threads = 0, dyn = 0, grid = 0;
    const Variant *v = nullptr;
    if (f.pair) {
        const bool nt2 = c->P.npair <= 2 * 512;
        const bool full = c->use_full && c->P.npix == 2 * c->P.npair && (c->P.npair == 2 * 512 || c->P.npair == 4 * 512);
        fn = full ? (nt2 ? (const void *)logprob_pair_kernel<512, 2, true, true> : (const void *)logprob_pair_kernel<512, 4, true, true>)
                  : (nt2 ? (const void *)logprob_pair_kernel<512, 2, true> : (const void *)logprob_pair_kernel<512, 4, true>);
        nm = std::string("pair_plan_kernel + logprob_pair_kernel<512 threads, ") + (nt2 ? "2" : "4") +
             " element trips per lane" + (full ? ", FULL" : "") + "> (planner: one thread per walker; two walkers of one grid cell per workgroup, one set of row loads, model values in registers; two workgroups per CU)";
        threads = 512; grid = m;
    } else {
        const int B = (f.linked || f.inpath) ? 512 : block_threads > 0 ? block_threads : pick_block(c, m, c->P.npix);
        DevProblem Pq = c->P;
        if (f.inpath) { Pq.given = c->d_inp_given; Pq.given_stride = c->inp_gstride; }
        const VariantChoice ch = choose_variant(c, Pq, m, B, shared512, f.linked);
        if (!ch.v) return fail(c, MSX_ERR_STATE, "no kernel variant for this launch");
        v = ch.v;
        fn = v->fn;
        threads = v->threads; dyn = (int64_t)ch.dyn_lds;
        grid = f.linked ? ((m + 7) & ~7ll) * c->nseg : m;
        nm = std::string("logprob_kernel<NS=") + std::to_string(v->ns) + ", " + std::to_string(v->threads) + " threads" +
             (v->lk ? ", linked" : v->gm ? ", GM" : v->pf && v->sh ? ", SH, PF" : v->pf ? ", PF" : v->sh ? ", SH" : "") + (v->r32 ? ", R32" : "") + (v->full == 3 ? ", FULL" : v->full == 2 ? ", FULL(chi2 pass)" : "") + (v->given ? ", GIVEN" : "") + "> (" + v->what + ")";
        if (f.inpath) nm = "inpath_recipe_kernel + inpath_conv_kernel + inpath_resample_kernel + " + nm;
    }
    hipFuncAttributes at;
    HIP_TRY(c, hipFuncGetAttributes(&at, fn));
    out8[0] = f.inpath ? MSX_FORM_INPATH : f.pair ? MSX_FORM_PAIR : f.linked ? MSX_FORM_LINKED : MSX_FORM_FUSED;
    out8[1] = threads;
    out8[2] = at.numRegs;
    out8[3] = (int64_t)at.sharedSizeBytes;
    out8[4] = dyn;
    out8[5] = requested_bytes_of(c, n, f, v);
    out8[6] = grid;
    out8[7] = m;
    if (name && name_len > 0) {
        strncpy(name, nm.c_str(), (size_t)name_len - 1);
        name[name_len - 1] = 0;
    }
    return MSX_OK;
}

int msx_last_form(msx_ctx *c, int32_t *form) {
    if (!c || !form) return MSX_ERR_INVALID;
    *form = c->last_form;
    return MSX_OK;
}

int msx_pair_stats(msx_ctx *c, int64_t *out2) {
    if (!c || !out2) return MSX_ERR_INVALID;
    if (!c->d_pair_plan) return fail(c, MSX_ERR_STATE, "msx_pair_stats: the staged problem has no pair form");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipDeviceSynchronize());
    int32_t h[2];
    HIP_TRY(c, hipMemcpy(h, c->d_pair_plan, sizeof(h), hipMemcpyDeviceToHost));
    out2[0] = h[0];
    out2[1] = h[1];
    return MSX_OK;
}

int msx_test_hook(msx_ctx *c, int32_t what, int32_t value) {
    if (!c) return MSX_ERR_INVALID;
    if (what == MSX_HOOK_LINKED_FAULT) {
        if (!c->problem_staged) return fail(c, MSX_ERR_STATE, "msx_test_hook: no problem staged");
        c->P.linked_fault = value != 0;
        return MSX_OK;
    }
    if (what == MSX_HOOK_PAIR_LEASES) {
        if (!c->problem_staged || !c->d_pair_plan) return fail(c, MSX_ERR_STATE, "msx_test_hook: the staged problem has no pair form");
        HIP_TRY(c, hipSetDevice(c->device));
        HIP_TRY(c, hipDeviceSynchronize());
        std::vector<int32_t> v((size_t)kPairSpillRows, value != 0 ? 1 : 0);
        HIP_TRY(c, hipMemcpy(c->d_pair_plan + kPairHdrInts, v.data(), sizeof(int32_t) * v.size(), hipMemcpyHostToDevice));
        return MSX_OK;
    }
    return fail(c, MSX_ERR_INVALID, "msx_test_hook: unknown hook");
}

}  // extern "C"
