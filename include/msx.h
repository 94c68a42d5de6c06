/* msx.h -- C ABI of the MI355X (gfx950) implementation of mcmc_spec's per-walker log-likelihood path.
 *
 * The reference (kendallsullivan/mcmc_spec, mft6.py) is pure Python and exposes no FFI of its own
 * (SURVEY.md §8b); this header is therefore the boundary a maintainer would bind with ctypes (the
 * stub is shown in INTEGRATION.md).  Each entry point names the reference code it stands in for.
 *
 * Conventions
 *   - every call returns MSX_OK (0) or a negative MSX_ERR_* code; nothing throws across the ABI;
 *     msx_last_error(ctx) returns a human-readable message for the last failing call on that ctx.
 *   - the caller owns every host buffer; they are consumed before the call returns.
 *   - the ctx owns every device buffer.  One ctx per device; calls on one ctx must be serialised by
 *     the caller, different ctxs may be driven concurrently from different host threads.
 *   - all floating point data are IEEE float64, arrays are C-contiguous.
 *   - "*_dev" entry points take DEVICE pointers and a hipStream_t (as void*) and do not synchronise.
 */
#ifndef MSX_H
#define MSX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MSX_OK 0
#define MSX_ERR_INVALID (-1) /* bad argument / inconsistent sizes                          */
#define MSX_ERR_HIP (-2)     /* a HIP runtime call failed (message has the hipError string) */
#define MSX_ERR_STATE (-3)   /* call order: grid or problem not staged yet                  */
#define MSX_ERR_RANGE (-4)   /* a value is outside the staged tables (Python: ValueError)   */

/* per-walker status written next to each log-probability (reference error conventions) */
#define MSX_W_OK 0
#define MSX_W_REJECT 1     /* prior box / non-finite -> log-prob = -inf        (mft6.py:1228,1230) */
#define MSX_W_KEYERROR 2   /* a needed grid node is not staged                 (mft6.py:489-500)   */
#define MSX_W_INDEXERROR 3 /* logg/Teff bracket runs past the last node        (mft6.py:453,477)   */
#define MSX_W_VALUEERROR 4 /* Teff outside the isochrone table                 (mft6.py:95)        */
#define MSX_W_HANDOVER 5   /* device fault, not a reference error: a linked launch's workgroups did not meet (MSX_PATH_LINKED) */

/* evaluation modes for msx_logprob_batch* (modes 4 and 5 are reached through msx_opt_step / msx_opt_init) */
#define MSX_MODE_LOGLIKE 0      /* loglikelihood   (mft6.py:1139-1205)                              */
#define MSX_MODE_LOGPOST 1      /* logposterior = logprior gate + loglikelihood (mft6.py:1459-1470) */
#define MSX_MODE_CHISQ 2        /* loglikelihood(optimize=True): returns total chi^2 (mft6.py:1198) */
#define MSX_MODE_LOGPRIOR 3     /* logprior alone (mft6.py:1207-1272); needs no spectrum pass       */
#define MSX_MODE_OPT_STEP 4     /* fit_spec proposal chi^2 (mft6.py:997-1028); via msx_opt_step      */
#define MSX_MODE_OPT_INIT 5     /* fit_spec initial guess (mft6.py:871-907); via msx_opt_init        */

#define MSX_MAX_SPEC 3
#define MSX_MAX_BANDS 8
#define MSX_MAX_DIM 8

typedef struct msx_ctx msx_ctx;

/* Everything that is static per dataset.  Built on the host by mcmc_spec_amd/staging.py from the
 * reference's own arguments (data, err, fr, ctm, ptm, matrix, prior ...); copied at stage time. */
typedef struct msx_problem {
    int32_t struct_size; /* sizeof(msx_problem), ABI check */
    int32_t nspec;       /* 2 (binary, ndim 6) or 3 (triple, ndim 8)          mft6.py:1145,1153 */
    int64_t npix;        /* data pixels                                                          */
    /* A8 resample tables: for pixel p the model samples wl[lo], wl[lo+1] bracket the pixel and
     * t = (x - x_lo)/(x_hi - x_lo)                                              mft6.py:1169-1170 */
    const int64_t *pix_lo;
    const double *pix_t;
    const double *pix_u;    /* wavelength mapped to [-1,1] for the quadratic fit   mft6.py:195     */
    const double *pix_flux; /* data (already median-normalised by the caller)      mft6.py:3507    */
    const double *pix_err;  /* sigma per pixel                                     mft6.py:120     */
    double median_flux;     /* np.median(data)                                     mft6.py:1173    */
    double fit_minv[9];     /* inverse Gram matrix of [1,u,u^2] (row major)        mft6.py:195     */
    /* A5/A6 band integrals are linear in the node spectrum: out = sum_i w[i]*flux[node][i0+i]    */
    int32_t n_contrast;     /* contrast filters (first in the band list)           mft6.py:717     */
    int32_t n_phot;         /* photometric bands                                   mft6.py:771     */
    const int64_t *band_i0; /* [n_contrast+n_phot] first grid index of each band                   */
    const int64_t *band_len;
    const double *band_w;   /* concatenated weights                                                */
    const double *cmag, *cerr;       /* [n_contrast] observed contrasts            mft6.py:1182    */
    const double *pmag, *perr;       /* [n_phot] observed magnitudes               mft6.py:1188    */
    const double *phot_zero;         /* [n_phot] zero-point flux per band          mft6.py:780-782 */
    const double *phot_k;            /* [n_phot] CCM89 a+b/Rv at phot_cwl          mft6.py:1163    */
    int64_t win_j0, win_n;  /* make_composite's window into the grid wavelength    mft6.py:687,542 */
    /* A1 isochrone, sorted by Teff (stable)                                        mft6.py:87-98   */
    int32_t niso;
    const double *iso_teff, *iso_logg, *iso_lum;
    /* prior                                                         mft6.py:1207-1272 */
    int32_t nav;            /* A_V(distance) table bins; 0 = no A_V prior term                     */
    const double *av_edges_pc; /* [nav+1] */
    const double *av_mu, *av_sig; /* [nav] */
    double tmin, tmax;      /* Teff box                                             mft6.py:1227    */
    double prior_mean[MSX_MAX_DIM]; /* Gaussian priors; mean == 0 -> unused         mft6.py:1257-1260 */
    double prior_sig[MSX_MAX_DIM];
    int32_t use_av;         /* `a` / `av` flag                                      mft6.py:1161,1229 */
    int32_t dist_fit;       /* 1: the parallax is a fitted parameter with its box / Gaussian terms (mft6.py:1212-1272);
                             * 0: the `dist_fit=False` branch, radius-ratio scaling only (mft6.py:1275-1327)  */
    int32_t rad_prior;      /*                                                      mft6.py:1262    */
    int32_t has_prior_list; /* `prior != 0`                                         mft6.py:1241    */
    int32_t no_spectrum;    /* 1 = the mft6_nospec.py variant: total = contrast + photometry chi^2 only
                             * (mft6_nospec.py:1163-1196); the spectral phases are skipped            */
} msx_problem;

/* ---- lifecycle ------------------------------------------------------------------------------- */
int msx_create(int device, msx_ctx **out);
void msx_destroy(msx_ctx *ctx);
const char *msx_last_error(msx_ctx *ctx);
/* device facts for reports: out[0]=CUs, out[1]=total global memory bytes, out[2]=clock kHz */
int msx_device_info(msx_ctx *ctx, int64_t *out3, char *name, int name_len);

/* ---- A0: the staged grid (replaces the `specs` dict, mft6.py:342-383 / consumed :481-500) ----- */
/* flux is [nt][ng][nwl]; present is [nt][ng] (0 = node absent -> KeyError when touched) or NULL.  */
int msx_stage_grid(msx_ctx *ctx, const double *wl, int64_t nwl, const double *teff_nodes, int32_t nt,
                   const double *logg_nodes, int32_t ng, const double *flux, const uint8_t *present);

/* ---- A7: CCM89 k(lambda) = a(x) + b(x)/R_V at arbitrary wavelengths [A] (extinction.ccm89 with
 * a_v = 1, mft6.py:62); the per-grid-sample table is built inside msx_stage_grid with R_V = 3.1.  */
int msx_ccm89_k(msx_ctx *ctx, const double *wl, int64_t n, double rv, double *out);

/* ---- f3: the resample step of the grid loader: interp1d(x, y)(xq), x sorted ascending, host buffers
 * (mft6.py:369-371).  MSX_ERR_RANGE when a query lies outside [x[0], x[n-1]] (scipy raises ValueError). */
int msx_resample_linear(msx_ctx *ctx, const double *x, const double *y, int64_t n, const double *xq, int64_t m,
                        double *out);

/* ---- A3: instrumental broadening (pyasl.instrBroadGaussFast + edge patches, mft6.py:124-152) -- */
/* one spectrum, host buffers; used by the drop-in `broaden()`                                     */
int msx_broaden(msx_ctx *ctx, const double *wl, const double *flux, int64_t n, double resolution,
                double maxsig, double *out);
/* every staged node, in place, over grid samples [i0, i0+n): the staging step mft6.py:366-378     */
int msx_broaden_grid(msx_ctx *ctx, int64_t i0, int64_t n, double resolution, double maxsig);
/* Where the broadening is PLACED: MSX_BROADEN_STAGING (default) -- once per grid node, by msx_broaden_grid: the
 * reference's live path; MSX_BROADEN_IN_PATH -- msx_broaden_grid additionally keeps the window's rows as they were, and
 * the problems staged afterwards get the per-walker form MSX_PATH_INPATH (below) beside all the others.  Takes effect at the
 * next msx_broaden_grid. */
#define MSX_BROADEN_STAGING 0
#define MSX_BROADEN_IN_PATH 1
int msx_set_broadening(msx_ctx *ctx, int32_t placement);
/* copy one staged node back to the host (tests / the drop-in `specs` view)                        */
int msx_read_node(msx_ctx *ctx, int32_t it, int32_t ig, double *out_nwl);

/* ---- problem staging -------------------------------------------------------------------------- */
int msx_stage_problem(msx_ctx *ctx, const msx_problem *p);

/* ---- the hot path: log-probability of a whole ensemble in one launch -------------------------- */
/* theta is [n][ndim] row-major (ndim = 2*nspec+2); logp_out [n]; status_out [n] (MSX_W_*).        */
int msx_logprob_batch(msx_ctx *ctx, int32_t mode, const double *theta, int64_t n, int32_t ndim,
                      double *logp_out, int32_t *status_out);
/* same with device pointers on a caller stream; does not synchronise (no launch allocates: every scratch buffer is
 * sized by msx_stage_problem).  block_threads: 0 = auto (up to #CUs walkers, spectra of >= 8192 pixels, triples: 512
 * threads, one workgroup per CU; binaries below 8192 pixels between #CUs and 2 x #CUs walkers: 512 threads in the
 * <= 128-VGPR variant, two of which fit a CU; beyond, and from #CUs walkers on for spectra of <= 2048 pixels: 256
 * threads, three per CU), or 256 / 512, or MSX_BLOCK_512_SHARED = 512 threads in the <= 128-VGPR variant whatever
 * the batch size (what a launch wants when another kernel, e.g. a collective, holds CUs at the same time).  The
 * choice affects speed only: every variant produces the same bits.                                               */
#define MSX_BLOCK_512_SHARED 1512
int msx_logprob_batch_dev(msx_ctx *ctx, int32_t mode, const double *d_theta, int64_t n, int32_t ndim,
                          double *d_logp, int32_t *d_status, void *hip_stream, int32_t block_threads);

/* Three forms of the same path, same bits.
 * FUSED: one launch, one workgroup per walker.
 * PAIR: for large batches of a binary with <= 4096 pixels: a planner kernel (one thread per walker: the recipe, the
 * prior and band terms, who shares a grid cell with whom) and a kernel that evaluates TWO walkers of one grid cell
 * per workgroup from one set of loads, model values in registers (16,384 walkers 316 us against 419 fused).
 * MSX_PATH_AUTO takes it from 8 walkers per CU on (2,048; spectra of <= 3,072 pixels: 12 per CU; MSX_PAIR_MIN in the
 * environment) while the planner's last count says the ensemble pairs (msx_pair_stats); DESIGN.md section 5.1.
 * LINKED: for few walkers x long spectra (2..8 segments of 8192 pixels), one workgroup per (walker, segment) in ONE
 * launch, so that e.g. 128 walkers x 16,384 pixels use 256 CUs instead of 128 and each workgroup's chain of latencies
 * is 8192 pixels long.  A walker's workgroups are equals: each blends its segment, they exchange the segments' fit
 * sums / value ranges / histogram counters inside the kernel (a bounded wait: 20 ms), each makes the chi^2 / median-
 * candidates pass over its own segment, and whichever finishes last ranks the candidates and completes the walker.
 * MSX_PATH_AUTO takes it while walkers x segments <= #CUs (MSX_LINKED=0 in the environment: never; =1: whenever the
 * spectrum has 2..8 segments): 128 walkers x 16,384 px 27.2 us against 31.9 fused, 8..64 walkers 23.5..24.5 us
 * (DESIGN.md); also under the device-resident sampler.
 * A meeting that times out fails its walker with MSX_W_HANDOVER and POISONS the context's linked form: a device-side
 * word makes every later linked launch fail ALL its walkers with MSX_W_HANDOVER (never a value computed from stale
 * counters), MSX_PATH_AUTO takes the fused form once a synchronous entry point has seen the status, an explicit
 * MSX_PATH_LINKED is refused with MSX_ERR_STATE -- until msx_stage_problem clears counters and word together.
 * (Values 2 and 3 were the split and wide forms of rounds 1-2: measured slower than FUSED at every size and removed;
 * the measurements are kept in DESIGN.md.)                                                                        */
#define MSX_PATH_AUTO 0
#define MSX_PATH_FUSED 1
#define MSX_PATH_PAIR 2
#define MSX_PATH_LINKED 4
/* In-path broadening (SURVEY A3 placement (ii); mft6.py:124-152 applied per evaluation, the call the reference keeps
 * commented out at :550): the instrumental broadening is applied PER WALKER to the unreddened composite inside the data
 * window -- composite of the raw window rows with the recipe's weights, Gaussian FIR, the two edge patches, then reddening
 * and the resample to the data pixels -- instead of once per grid node at staging.  Broadening is linear: the values agree
 * with every other form to the order of the sums (~1e-13 relative), not bit for bit.  Never taken by MSX_PATH_AUTO (the
 * per-node placement is the reference's live path); needs msx_set_broadening(MSX_BROADEN_IN_PATH) BEFORE
 * msx_broaden_grid, a binary whose data pixels all lie inside that window, float64 tables, a likelihood / posterior /
 * chi^2 mode.  Costs a walker eight raw rows of the window and a convolution where the table form reads resampled pixels:
 * several times the headline path's time. */
#define MSX_PATH_INPATH 8
int msx_set_path(msx_ctx *ctx, int32_t path);

/* ---- f4: the pre-optimiser's chi^2 (fit_spec, mft6.py:856-1137) on the same kernel --------------- */
/* msx_opt_init: one chain per row of theta0 [nchains][ndim].  Normalises the data against each chain's
 * initial (un-reddened) model like mft6.py:884-889, keeps the normalised vector and its median on the
 * device, and returns the initial likelihood chi^2 = 3*iic*(n_c+n_p) + contrast + phot (mft6.py:893-904;
 * the opt_prior terms of :910-929 are added by the host driver).                                        */
int msx_opt_init(msx_ctx *ctx, const double *theta0, int64_t nchains, int32_t ndim, double *chi2_out,
                 int32_t *status_out);
/* msx_opt_step: proposal i belongs to chain[i]; returns the chi^2 of mft6.py:1011-1028 (no per-proposal
 * continuum fit, spectrum weight 3) against that chain's stored data vector.                            */
int msx_opt_step(msx_ctx *ctx, const double *theta, const int32_t *chain, int64_t n, int32_t ndim,
                 double *chi2_out, int32_t *status_out);

/* ---- f2 on the device: nsteps iterations of the affine-invariant stretch move (Goodman & Weare 2010, the
 * default move of emcee 3; the loop the reference drives at mft6.py:1494-1524) with the walker state resident in
 * HBM.  Per half-step there is ONE launch of the fused log-probability kernel: its first lines build each active
 * walker's proposal q = c - (c - s) z from the resident coordinates, its last lines apply the accept rule and write
 * the walker's row of the chain -- no separate proposal / accept kernels and no host round trip.  The host supplies
 * the randomness of every half-step h = 2*step + half, each an array of nw/2 entries: the active walkers sidx, the
 * complementary half cidx, partner (index into cidx), z = ((a-1)u+1)^2/a, zfac = (ndim-1) ln z and logu = ln(u')
 * for the accept test logu < zfac + lp(q) - lp(s).
 * coords/logp are updated in place; chain_out [nsteps][nw][ndim] and logp_out [nsteps][nw] hold the state after
 * every step; naccept[nw] accumulates; worst_status returns the largest MSX_W_* error seen (0 = none).        */
int msx_sampler_run(msx_ctx *ctx, int32_t mode, int64_t nw, int32_t ndim, int64_t nsteps, double *coords, double *logp,
                    const int32_t *sidx, const int32_t *cidx, const int32_t *partner, const double *zz,
                    const double *zfac, const double *logu, double *chain_out, double *logp_out, int64_t *naccept,
                    int32_t *worst_status);

/* The same loop, pipelined (what mcmc_spec_amd.sampler.DeviceEnsembleSampler drives): begin uploads the ensemble
 * state and sets up two slots; enqueue(slot) stages one chunk of randomness (arrays of nsteps*2*(nw/2) entries, laid
 * out as for msx_sampler_run; indices are range-checked on the host) and queues its 2*nsteps fused launches WITHOUT
 * waiting -- uploads, launches and chain downloads run on three HIP streams; collect(slot) waits for that chunk only
 * and returns its chain rows, the cumulative acceptance counts as of its last step and its worst status.  Enqueue
 * chunk i+1 before collecting chunk i and the GPU never idles between chunks.  end() optionally returns the final
 * state (coords/logp may be NULL) and frees everything; restaging the problem or destroying the ctx ends a run too. */
int msx_sampler_begin(msx_ctx *ctx, int32_t mode, int64_t nw, int32_t ndim, int64_t max_chunk_steps, const double *coords,
                      const double *logp, const int64_t *naccept /* NULL = zeros */);
/* Sharded form of the same run: call once between msx_sampler_begin and the first enqueue, on every
 * rank, after msx_comm_init(rank, world).  Every rank holds the whole ensemble in HBM and must be fed the same
 * randomness; per half-step rank r evaluates block r of the nw/2 proposals (ceil((nw/2)/world) walkers), ONE RCCL
 * all-gather of that many float64 log-probabilities per rank crosses xGMI on the compute stream, and a small kernel
 * applies the accept rule for all walkers on every rank -- so every rank's chain is bit-identical to the one-GPU
 * chain and nothing returns to the host between half-steps.  world = 1 is allowed (no collective): the same three
 * device steps on one GPU.  A walker error (MSX_W_*) on any rank reaches every rank's worst_status: it travels as
 * the payload of the NaN log-probability it produces.                                                            */
int msx_sampler_shard(msx_ctx *ctx, int32_t rank, int32_t world);
int msx_sampler_enqueue(msx_ctx *ctx, int32_t slot /* 0|1 */, int64_t nsteps, const int32_t *sidx, const int32_t *cidx,
                        const int32_t *partner, const double *zz, const double *zfac, const double *logu);
/* The chunk's randomness drawn ON THE DEVICE instead of coming from the host: a counter-based generator (SplitMix64's
 * output function over counters made of (seed, absolute iteration of the run, stream, index)) yields, per iteration, the
 * random split of the ensemble into two halves (walkers sorted by a 64-bit key) and, per half-step and walker, the stretch
 * factor z = ((a - 1) u + 1)^2 / a, the partner floor(u ns) in the complementary half and ln u of the accept draw -- the
 * quantities msx_sampler_enqueue takes from the host (emcee's stretch move; mft6.py:1491-1494 drives it).  One launch per
 * chunk, no upload; every rank of a sharded run draws the same numbers from the same seed.  Up to 4096 walkers.
 * msx_sampler_draw returns the same stream to the host (arrays [nsteps][2][nw/2] as for msx_sampler_enqueue; `partner`
 * indexes the complementary half): the host loop fed with it reproduces the device-drawn chain bit for bit
 * (tests/test_gpu_overlap.py), and mcmc_spec_amd/sampler.py::counter_draws restates the generator in NumPy.           */
int msx_sampler_enqueue_drawn(msx_ctx *ctx, int32_t slot, int64_t nsteps, uint64_t seed, double a);
int msx_sampler_draw(msx_ctx *ctx, uint64_t seed, double a, int64_t first_iter, int64_t nsteps, int64_t nw, int32_t ndim,
                     int32_t *sidx, int32_t *cidx, int32_t *partner, double *zz, double *zfac, double *logu);
int msx_sampler_collect(msx_ctx *ctx, int32_t slot, double *chain_out, double *logp_out, int64_t *naccept,
                        int32_t *worst_status);
int msx_sampler_end(msx_ctx *ctx, double *coords, double *logp);
/* OVERLAPPED half-steps (one GPU, unsharded, a half-step of at most #CUs / 2 walkers through the fused kernel with one
 * workgroup per CU -- config 2's 256 walkers): consecutive half-steps go to two streams and run concurrently; a walker's
 * workgroup waits inside the kernel (bounded: 20 ms, then the chunk's worst_status is MSX_W_HANDOVER) until the two
 * walkers its move reads have reached the versions the move is defined on; what a move reads of a walker is handed over
 * as 8-byte words {32 bits of payload | version}, each written by one agent-scope store and double-buffered by version
 * parity, so the load that sees the version has the data -- a half-step's launch, start-up and slowest walker no longer
 * sit between two dependent evaluations (256 walkers x 4096 px: 28 us per iteration; 31.3 with a version word behind the
 * data, 38.0 with plain launches).
 * Same chain, bit for bit.  Chosen by the first msx_sampler_enqueue of a run; MSX_SMP_OVERLAP=0 in the environment:
 * never.  *out = 1 if the run begun on ctx takes it, 0 if not, -1 before its first chunk.                           */
int msx_sampler_overlapped(msx_ctx *ctx, int32_t *out);
/* ... and the caller's say: overlap = -1 (default) lets the rule above decide, 0 never overlaps (plain launches, one
 * half-step after the other).  The rule counts on this context's launches having the device to themselves -- two
 * half-steps of 128 walkers need all 256 CUs at once -- and never takes the overlap on a context that holds a
 * communicator.  On a device shared with other work use 0: a waiting workgroup whose producer cannot get a CU ends, after
 * the 20 ms bound, as MSX_W_HANDOVER for the chunk, and the run has to be started again (the chain up to the last
 * collected chunk stands).  Takes effect at the next msx_sampler_begin.                                                */
int msx_sampler_policy(msx_ctx *ctx, int32_t overlap);

/* ---- A4-A6: make_composite (mft6.py:651-831, plot=False) -------------------------------------- */
/* teff/logg/rad are [nspec]; use_distance = 0 mirrors `distance=False` (mft6.py:701-703).         */
/* spec_out [win_n], contrast_out [n_contrast], phot_out [n_phot] (unreddened magnitudes).         */
int msx_make_composite(msx_ctx *ctx, const double *teff, const double *logg, const double *rad,
                       int32_t use_distance, double plx, double *spec_out, double *contrast_out,
                       double *phot_out, int32_t *status_out);

/* ---- the one collective of the sharded path -- an RCCL all-gather of `count` float64 log-probs per
 * rank.  The reference has no counterpart (its parallelism is multiprocessing.Pool, mft6.py:1744).  RCCL is
 * resolved at run time from the librccl.so.1 already mapped in the process (PyTorch-ROCm's).  The id comes
 * from rank 0 (msx_comm_unique_id) and reaches the other ranks by any side channel (bench.py: a
 * torch.distributed broadcast).  msx_comm_allgather_dev runs the collective on the ctx's communication stream
 * after everything queued on `compute_stream`, then signals event `slot` (0..3); msx_comm_wait_slot makes
 * `compute_stream` wait for that event before a buffer is reused -- so step i's all-gather overlaps step i+1. */
int msx_comm_unique_id(msx_ctx *ctx, uint8_t *out128);
int msx_comm_init(msx_ctx *ctx, const uint8_t *id128, int32_t rank, int32_t world);
int msx_comm_allgather_dev(msx_ctx *ctx, const double *d_send, double *d_recv, int64_t count, void *compute_stream,
                           int32_t slot);
int msx_comm_wait_slot(msx_ctx *ctx, int32_t slot, void *compute_stream);

/* ---- the "fake collective" -- a LOOPBACK group.  `world` contexts of ONE process on ONE device stand
 * for ranks 0..world-1 (ctxs[r] becomes rank r); the sharded sampler's all-gather becomes device copies between the
 * ranks' gathered vectors (rank r takes block p out of rank p's vector, in place at p * ceil(ns / world), exactly
 * where RCCL's in-place all-gather puts it).  Every rank-dependent line of the sharded path -- block offsets, ragged
 * and empty shards, the NaN-payload status, the apply kernel over the gathered vector -- then runs on a one-GPU box,
 * no RCCL involved.  After msx_sampler_begin + msx_sampler_shard(r, world) on every rank, msx_sampler_enqueue_group
 * queues one chunk on all ranks in lock-step (same randomness for all, as the sharded form requires) in place of the
 * per-rank msx_sampler_enqueue; collect / end stay per rank.  The group ends when its first member is destroyed.    */
int msx_comm_init_loopback(msx_ctx **ctxs, int32_t world);
int msx_sampler_enqueue_group(msx_ctx **ctxs, int32_t world, int32_t slot /* 0|1 */, int64_t nsteps, const int32_t *sidx,
                              const int32_t *cidx, const int32_t *partner, const double *zz, const double *zfac,
                              const double *logu);

/* ---- measurement helpers ----------------------------------------------------------------------- */
/* float4 device-to-device copy of `bytes` bytes, `iters` times; returns GB/s (read+write counted)  */
int msx_stream_copy_gbps(msx_ctx *ctx, int64_t bytes, int32_t iters, double *gbps_out);
/* bytes the hot kernel requests from the memory system per walker, for the variant an automatic launch of n walkers
 * of the staged problem takes (the one-workgroup-per-CU variants keep u and the data flux in LDS for the chi^2 pass:
 * 132 instead of 148 bytes per pixel of a binary)                                                                  */
int msx_bytes_per_eval(msx_ctx *ctx, int64_t n, int64_t *requested_bytes);

/* One launch of the fused / linked form over n walkers like msx_logprob_batch_dev -- with clock stamps: thread 0 of every
 * walker's workgroup (the first 4096 walkers) reads the 100 MHz wall clock and the shader-cycle counter at its first and
 * last line.  Synchronises; out4 = {median shader clock in MHz while the walkers ran, median and maximum of a walker's own
 * time in us, first walker's start -> last walker's end in us}.  What a short timed region cannot tell apart by itself: a
 * box whose GPU clocks lower under this load from a launch that waits.                                              */
int msx_probe_launch(msx_ctx *ctx, int32_t mode, const double *d_theta, int64_t n, int32_t ndim, double *d_logp,
                     int32_t *d_status, void *hip_stream, int32_t block_threads, double *out4);

/* STORAGE precision of the staged grid tables (SURVEY 8b's `store_dtype`).  MSX_STORE_F64 (default): the per-node pixel
 * table R = lo + (hi - lo) t in float64.  MSX_STORE_F32: R rounded to float32 (8 instead of 12 bytes per node-pixel through
 * the CU's L2 port, which the blend runs at the limit of) and widened in the registers -- the arithmetic stays float64, the
 * grid values carry 2^-24.  A SEPARATELY LABELLED precision: log-probabilities then agree with the reference to ~1e-7
 * relative at S/N 100 (tests/test_gpu_parity.py), inside BASELINE's 1e-6 but not the 1e-9 the float64 tables are held to;
 * bench.py reports it under its own label and never as the headline.  Takes effect at the next msx_stage_problem; fused
 * binaries of at most 17,152 pixels only (the pair and linked forms, triples and longer spectra are refused, not mixed in). */
#define MSX_STORE_F64 0
#define MSX_STORE_F32 1
int msx_set_grid_storage(msx_ctx *ctx, int32_t store_dtype);

/* What an automatic launch of n walkers in `mode` (block_threads as for msx_logprob_batch_dev) WOULD take, asked of the
 * library's own launcher (nothing is queued): `name` receives the kernel's name and description, out8 = {form (MSX_FORM_*),
 * threads per workgroup, VGPRs, static LDS bytes, dynamic LDS bytes, bytes requested from the memory system per walker
 * (as msx_bytes_per_eval), workgroups of the first sub-batch, walkers of the first sub-batch}.                          */
#define MSX_FORM_FUSED 0
#define MSX_FORM_PAIR 1
#define MSX_FORM_LINKED 2
#define MSX_FORM_INPATH 3
int msx_launch_info(msx_ctx *ctx, int32_t mode, int64_t n, int32_t block_threads, char *name, int32_t name_len, int64_t *out8);
/* the form (MSX_FORM_*) the last launch queued on this context took (MSX_PATH_AUTO looks at the planner's lagging counts) */
int msx_last_form(msx_ctx *ctx, int32_t *form);

/* the planner's counts for the pair form's last launch (a sub-batch): out2[0] = pairs, out2[1] = walkers evaluated alone;
 * synchronises */
int msx_pair_stats(msx_ctx *ctx, int64_t *out2);

/* ---- test hooks (used by tests/ only) ------------------------------------------------------------ */
/* MSX_HOOK_LINKED_FAULT: value != 0 makes the workgroups of the linked form skip their signal -- and the walkers of an
 * overlapped sampler run the publication of their new version -- so that every in-kernel wait runs into its bound;
 * takes effect at the next launch, without restaging                                                              */
#define MSX_HOOK_LINKED_FAULT 1
/* MSX_HOOK_PAIR_LEASES: value != 0 marks every scratch row of the pair form's spill path as leased (what a launch torn
 * down mid-spill would leave behind), 0 clears them: the next pair launch's spilling walkers must end with
 * MSX_W_HANDOVER inside the bound instead of hanging; synchronises                                                  */
#define MSX_HOOK_PAIR_LEASES 2
int msx_test_hook(msx_ctx *ctx, int32_t what, int32_t value);

#ifdef __cplusplus
}
#endif
#endif /* MSX_H */
