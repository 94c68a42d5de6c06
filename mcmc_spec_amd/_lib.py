"""ctypes binding of ``libmsx.so`` (the C ABI declared in ``include/msx.h``).

There is no CPU fallback: if the HIP library has not been built the import of any compute entry
point fails loudly with the build command.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# MSX_LIB overrides the library path (diagnostic builds such as -DMSX_STAMPS; never a CPU fallback)
LIB_PATH = os.environ.get('MSX_LIB') or os.path.join(_HERE, 'libmsx.so')

MSX_OK = 0
MSX_ERR_INVALID, MSX_ERR_HIP, MSX_ERR_STATE, MSX_ERR_RANGE = -1, -2, -3, -4
W_OK, W_REJECT, W_KEYERROR, W_INDEXERROR, W_VALUEERROR, W_HANDOVER = 0, 1, 2, 3, 4, 5
MODE_LOGLIKE, MODE_LOGPOST, MODE_CHISQ, MODE_LOGPRIOR = 0, 1, 2, 3
BLOCK_512_SHARED = 1512  # include/msx.h MSX_BLOCK_512_SHARED: 512 threads, two workgroups per CU
PATH_AUTO, PATH_FUSED, PATH_PAIR, PATH_LINKED, PATH_INPATH = 0, 1, 2, 4, 8  # include/msx.h MSX_PATH_*
BROADEN_STAGING, BROADEN_IN_PATH = 0, 1  # include/msx.h MSX_BROADEN_*
FORM_FUSED, FORM_PAIR, FORM_LINKED, FORM_INPATH = 0, 1, 2, 3  # include/msx.h MSX_FORM_*
STORE_F64, STORE_F32 = 0, 1  # include/msx.h MSX_STORE_*
FORM_NAMES = {0: 'fused', 1: 'pair (planner + two walkers of one grid cell per workgroup)', 2: 'linked (one workgroup per walker and 8192-pixel segment)',
              3: 'in-path broadening (recipe, composite + convolution, resample, then the fused kernel on the given model values)'}
HOOK_LINKED_FAULT, HOOK_PAIR_LEASES = 1, 2  # include/msx.h MSX_HOOK_*
MAX_SPEC, MAX_BANDS, MAX_DIM = 3, 8, 8

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int64)


class MsxProblem(C.Structure):
    """Mirror of ``struct msx_problem`` (include/msx.h) -- keep the field order identical."""
    _fields_ = [
        ('struct_size', C.c_int32), ('nspec', C.c_int32), ('npix', C.c_int64),
        ('pix_lo', _ip), ('pix_t', _dp), ('pix_u', _dp), ('pix_flux', _dp), ('pix_err', _dp),
        ('median_flux', C.c_double), ('fit_minv', C.c_double * 9),
        ('n_contrast', C.c_int32), ('n_phot', C.c_int32),
        ('band_i0', _ip), ('band_len', _ip), ('band_w', _dp),
        ('cmag', _dp), ('cerr', _dp), ('pmag', _dp), ('perr', _dp), ('phot_zero', _dp), ('phot_k', _dp),
        ('win_j0', C.c_int64), ('win_n', C.c_int64),
        ('niso', C.c_int32), ('iso_teff', _dp), ('iso_logg', _dp), ('iso_lum', _dp),
        ('nav', C.c_int32), ('av_edges_pc', _dp), ('av_mu', _dp), ('av_sig', _dp),
        ('tmin', C.c_double), ('tmax', C.c_double),
        ('prior_mean', C.c_double * MAX_DIM), ('prior_sig', C.c_double * MAX_DIM),
        ('use_av', C.c_int32), ('dist_fit', C.c_int32), ('rad_prior', C.c_int32), ('has_prior_list', C.c_int32),
        ('no_spectrum', C.c_int32),
    ]


class MsxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__('msx error {}: {}'.format(code, msg))
        self.code = code
        self.msg = msg


_lib = None


def load():
    """Load ``libmsx.so`` once.  Raises ImportError with the build recipe if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            'mcmc_spec_amd: the HIP library {} is missing. Build it with\n'
            '  python -c "import __graft_entry__ as g; g.build()"   (or: make -C mcmc_spec_amd/csrc)\n'
            'There is no CPU fallback for the log-likelihood path.'.format(LIB_PATH))
    # One HIP runtime per process: PyTorch-ROCm ships its own libamdhip64 (same SONAME as
    # /opt/rocm's).  If torch were imported *after* this library had bound to /opt/rocm's copy the
    # process would hold two runtimes and torch's streams / device pointers would be foreign to
    # ours.  Importing torch first makes the loader resolve our NEEDED entry to the copy torch
    # already mapped.  (torch is plumbing here: device memory, streams, torch.distributed.)
    if os.environ.get('MSX_SKIP_TORCH_IMPORT') is None:
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    lib = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    sig = {
        'msx_create': (C.c_int, [C.c_int, C.POINTER(vp)]),
        'msx_destroy': (None, [vp]),
        'msx_last_error': (C.c_char_p, [vp]),
        'msx_device_info': (C.c_int, [vp, _ip, C.c_char_p, C.c_int]),
        'msx_stage_grid': (C.c_int, [vp, _dp, C.c_int64, _dp, C.c_int32, _dp, C.c_int32, _dp, C.POINTER(C.c_uint8)]),
        'msx_ccm89_k': (C.c_int, [vp, _dp, C.c_int64, C.c_double, _dp]),
        'msx_resample_linear': (C.c_int, [vp, _dp, _dp, C.c_int64, _dp, C.c_int64, _dp]),
        'msx_broaden': (C.c_int, [vp, _dp, _dp, C.c_int64, C.c_double, C.c_double, _dp]),
        'msx_broaden_grid': (C.c_int, [vp, C.c_int64, C.c_int64, C.c_double, C.c_double]),
        'msx_read_node': (C.c_int, [vp, C.c_int32, C.c_int32, _dp]),
        'msx_stage_problem': (C.c_int, [vp, C.POINTER(MsxProblem)]),
        'msx_logprob_batch': (C.c_int, [vp, C.c_int32, _dp, C.c_int64, C.c_int32, _dp, C.POINTER(C.c_int32)]),
        'msx_logprob_batch_dev': (C.c_int, [vp, C.c_int32, vp, C.c_int64, C.c_int32, vp, vp, vp, C.c_int32]),
        'msx_probe_launch': (C.c_int, [vp, C.c_int32, vp, C.c_int64, C.c_int32, vp, vp, vp, C.c_int32, _dp]),
        'msx_set_path': (C.c_int, [vp, C.c_int32]),
        'msx_set_grid_storage': (C.c_int, [vp, C.c_int32]),
        'msx_set_broadening': (C.c_int, [vp, C.c_int32]),
        'msx_opt_init': (C.c_int, [vp, _dp, C.c_int64, C.c_int32, _dp, C.POINTER(C.c_int32)]),
        'msx_opt_step': (C.c_int, [vp, _dp, C.POINTER(C.c_int32), C.c_int64, C.c_int32, _dp, C.POINTER(C.c_int32)]),
        'msx_sampler_run': (C.c_int, [vp, C.c_int32, C.c_int64, C.c_int32, C.c_int64, _dp, _dp, C.POINTER(C.c_int32),
                                      C.POINTER(C.c_int32), C.POINTER(C.c_int32), _dp, _dp, _dp, _dp, _dp, _ip,
                                      C.POINTER(C.c_int32)]),
        'msx_sampler_begin': (C.c_int, [vp, C.c_int32, C.c_int64, C.c_int32, C.c_int64, _dp, _dp, _ip]),
        'msx_sampler_shard': (C.c_int, [vp, C.c_int32, C.c_int32]),
        'msx_sampler_enqueue': (C.c_int, [vp, C.c_int32, C.c_int64, C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                          C.POINTER(C.c_int32), _dp, _dp, _dp]),
        'msx_sampler_collect': (C.c_int, [vp, C.c_int32, _dp, _dp, _ip, C.POINTER(C.c_int32)]),
        'msx_sampler_enqueue_drawn': (C.c_int, [vp, C.c_int32, C.c_int64, C.c_uint64, C.c_double]),
        'msx_sampler_draw': (C.c_int, [vp, C.c_uint64, C.c_double, C.c_int64, C.c_int64, C.c_int64, C.c_int32, C.POINTER(C.c_int32),
                                       C.POINTER(C.c_int32), C.POINTER(C.c_int32), _dp, _dp, _dp]),
        'msx_sampler_end': (C.c_int, [vp, _dp, _dp]),
        'msx_make_composite': (C.c_int, [vp, _dp, _dp, _dp, C.c_int32, C.c_double, _dp, _dp, _dp,
                                         C.POINTER(C.c_int32)]),
        'msx_comm_unique_id': (C.c_int, [vp, C.POINTER(C.c_uint8)]),
        'msx_comm_init': (C.c_int, [vp, C.POINTER(C.c_uint8), C.c_int32, C.c_int32]),
        'msx_comm_allgather_dev': (C.c_int, [vp, vp, vp, C.c_int64, vp, C.c_int32]),
        'msx_comm_wait_slot': (C.c_int, [vp, C.c_int32, vp]),
        'msx_comm_init_loopback': (C.c_int, [C.POINTER(vp), C.c_int32]),
        'msx_sampler_enqueue_group': (C.c_int, [C.POINTER(vp), C.c_int32, C.c_int32, C.c_int64, C.POINTER(C.c_int32),
                                                C.POINTER(C.c_int32), C.POINTER(C.c_int32), _dp, _dp, _dp]),
        'msx_stream_copy_gbps': (C.c_int, [vp, C.c_int64, C.c_int32, _dp]),
        'msx_bytes_per_eval': (C.c_int, [vp, C.c_int64, _ip]),
        'msx_test_hook': (C.c_int, [vp, C.c_int32, C.c_int32]),
        'msx_launch_info': (C.c_int, [vp, C.c_int32, C.c_int64, C.c_int32, C.c_char_p, C.c_int32, _ip]),
        'msx_last_form': (C.c_int, [vp, C.POINTER(C.c_int32)]),
        'msx_pair_stats': (C.c_int, [vp, _ip]),
        'msx_sampler_overlapped': (C.c_int, [vp, C.POINTER(C.c_int32)]),
        'msx_sampler_policy': (C.c_int, [vp, C.c_int32]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)  # AttributeError here = header/library skew, fail loudly
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


EXPORTED = ['msx_create', 'msx_destroy', 'msx_last_error', 'msx_device_info', 'msx_stage_grid', 'msx_ccm89_k',
            'msx_resample_linear',
            'msx_broaden', 'msx_broaden_grid', 'msx_read_node', 'msx_stage_problem', 'msx_logprob_batch',
            'msx_logprob_batch_dev', 'msx_probe_launch', 'msx_set_path', 'msx_set_grid_storage', 'msx_set_broadening', 'msx_opt_init', 'msx_opt_step', 'msx_sampler_run', 'msx_sampler_begin',
            'msx_sampler_shard', 'msx_sampler_enqueue', 'msx_sampler_enqueue_drawn', 'msx_sampler_draw', 'msx_sampler_collect', 'msx_sampler_end', 'msx_make_composite', 'msx_comm_unique_id', 'msx_comm_init', 'msx_comm_allgather_dev', 'msx_comm_wait_slot',
            'msx_comm_init_loopback', 'msx_sampler_enqueue_group',
            'msx_stream_copy_gbps', 'msx_bytes_per_eval', 'msx_launch_info', 'msx_last_form', 'msx_test_hook', 'msx_pair_stats', 'msx_sampler_overlapped', 'msx_sampler_policy']


def as_f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def dptr(a):
    return a.ctypes.data_as(_dp)


def iptr(a):
    return a.ctypes.data_as(_ip)


class Context:
    """One ``msx_ctx`` (one device).  Thin, typed wrapper; all numerics happen in the library."""

    def __init__(self, device=0):
        self.lib = load()
        h = C.c_void_p()
        rc = self.lib.msx_create(int(device), C.byref(h))
        self.h = h
        if rc != MSX_OK:
            msg = self.lib.msx_last_error(h).decode() if h else 'allocation failed'
            if h:
                self.lib.msx_destroy(h)
                self.h = None
            raise MsxError(rc, msg)
        self.device = int(device)

    def close(self):
        if getattr(self, 'h', None):
            self.lib.msx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def check(self, rc):
        if rc != MSX_OK:
            msg = self.lib.msx_last_error(self.h).decode()
            if rc == MSX_ERR_RANGE:
                raise ValueError(msg)
            raise MsxError(rc, msg)

    # ---- device ---------------------------------------------------------------------------------
    def device_info(self):
        out = np.zeros(3, dtype=np.int64)
        name = C.create_string_buffer(256)
        self.check(self.lib.msx_device_info(self.h, iptr(out), name, 256))
        return {'name': name.value.decode(), 'cus': int(out[0]), 'mem_bytes': int(out[1]), 'clock_khz': int(out[2])}

    def stream_copy_gbps(self, nbytes=1 << 30, iters=10):
        out = C.c_double()
        self.check(self.lib.msx_stream_copy_gbps(self.h, int(nbytes), int(iters), C.byref(out)))
        return out.value

    # ---- grid -----------------------------------------------------------------------------------
    def stage_grid(self, wl, teff_nodes, logg_nodes, flux, present=None):
        wl, teff_nodes, logg_nodes = as_f64(wl), as_f64(teff_nodes), as_f64(logg_nodes)
        flux = as_f64(flux)
        nt, ng, nwl = len(teff_nodes), len(logg_nodes), len(wl)
        if flux.shape != (nt, ng, nwl):
            raise ValueError('flux must be [nt][ng][nwl]')
        pp = None
        if present is not None:
            present = np.ascontiguousarray(present, dtype=np.uint8)
            pp = present.ctypes.data_as(C.POINTER(C.c_uint8))
        self.check(self.lib.msx_stage_grid(self.h, dptr(wl), nwl, dptr(teff_nodes), nt, dptr(logg_nodes), ng,
                                           dptr(flux), pp))
        self.nwl = nwl

    def ccm89_k(self, wl, rv=3.1):
        wl = as_f64(np.atleast_1d(wl))
        out = np.empty_like(wl)
        self.check(self.lib.msx_ccm89_k(self.h, dptr(wl), len(wl), float(rv), dptr(out)))
        return out

    def resample_linear(self, x, y, xq):
        x, y, xq = as_f64(x), as_f64(y), as_f64(xq)
        out = np.empty_like(xq)
        self.check(self.lib.msx_resample_linear(self.h, dptr(x), dptr(y), len(x), dptr(xq), len(xq), dptr(out)))
        return out

    def broaden(self, wl, flux, resolution, maxsig=5.0):
        wl, flux = as_f64(wl), as_f64(flux)
        out = np.empty_like(flux)
        self.check(self.lib.msx_broaden(self.h, dptr(wl), dptr(flux), len(wl), float(resolution), float(maxsig),
                                        dptr(out)))
        return out

    def broaden_grid(self, i0, n, resolution, maxsig=5.0):
        self.check(self.lib.msx_broaden_grid(self.h, int(i0), int(n), float(resolution), float(maxsig)))

    def read_node(self, it, ig):
        out = np.empty(self.nwl)
        self.check(self.lib.msx_read_node(self.h, int(it), int(ig), dptr(out)))
        return out

    # ---- problem + hot path -----------------------------------------------------------------------
    def stage_problem(self, prob: MsxProblem):
        self.check(self.lib.msx_stage_problem(self.h, C.byref(prob)))

    def logprob_batch(self, theta, mode=MODE_LOGPOST):
        theta = as_f64(theta)
        n, ndim = theta.shape
        logp = np.empty(n)
        status = np.empty(n, dtype=np.int32)
        self.check(self.lib.msx_logprob_batch(self.h, int(mode), dptr(theta), n, ndim, dptr(logp),
                                              status.ctypes.data_as(C.POINTER(C.c_int32))))
        return logp, status

    def logprob_batch_dev(self, d_theta_ptr, n, ndim, d_logp_ptr, d_status_ptr, stream_ptr, mode=MODE_LOGPOST,
                          block_threads=0):
        self.check(self.lib.msx_logprob_batch_dev(self.h, int(mode), C.c_void_p(d_theta_ptr), int(n), int(ndim),
                                                  C.c_void_p(d_logp_ptr), C.c_void_p(d_status_ptr),
                                                  C.c_void_p(stream_ptr), int(block_threads)))

    def set_grid_storage(self, store):
        """'f64' (default) or 'f32': the precision the NEXT stage_problem stores the per-node R table in (msx_set_grid_storage;
        a separately labelled precision -- fused binaries only)."""
        self.check(self.lib.msx_set_grid_storage(self.h, {'f64': STORE_F64, 'f32': STORE_F32}[store]))

    def probe_launch(self, d_theta_ptr, n, ndim, d_logp_ptr, d_status_ptr, stream_ptr, mode=MODE_LOGPOST, block_threads=0):
        """One launch with clock stamps (msx_probe_launch): {'shader_mhz', 'walker_us_median', 'walker_us_max', 'span_us'}."""
        out = np.zeros(4)
        self.check(self.lib.msx_probe_launch(self.h, int(mode), C.c_void_p(d_theta_ptr), int(n), int(ndim), C.c_void_p(d_logp_ptr),
                                             C.c_void_p(d_status_ptr), C.c_void_p(stream_ptr), int(block_threads), dptr(out)))
        return {'shader_mhz': float(out[0]), 'walker_us_median': float(out[1]), 'walker_us_max': float(out[2]), 'span_us': float(out[3])}

    def set_broadening(self, placement):
        """'staging' (default: once per grid node, the reference's live path) or 'in_path' (msx_set_broadening): the next
        broaden_grid also keeps the raw window, and problems staged afterwards have the per-walker form PATH_INPATH."""
        self.check(self.lib.msx_set_broadening(self.h, {'staging': BROADEN_STAGING, 'in_path': BROADEN_IN_PATH}[placement]))

    def set_path(self, path):
        """PATH_AUTO / PATH_FUSED / PATH_PAIR / PATH_LINKED: which form of the hot path launches take (same bits either
        way).  PATH_PAIR (two walkers of one grid cell per workgroup) needs a binary of <= 4096 pixels, PATH_LINKED
        (one workgroup per walker and 8192-pixel segment) a spectrum of 2..8 segments; AUTO picks by batch size."""
        self.check(self.lib.msx_set_path(self.h, int(path)))

    def opt_init(self, theta0):
        theta0 = as_f64(theta0)
        n, ndim = theta0.shape
        chi, status = np.empty(n), np.empty(n, dtype=np.int32)
        self.check(self.lib.msx_opt_init(self.h, dptr(theta0), n, ndim, dptr(chi),
                                         status.ctypes.data_as(C.POINTER(C.c_int32))))
        return chi, status

    def opt_step(self, theta, chain):
        theta = as_f64(theta)
        chain = np.ascontiguousarray(chain, dtype=np.int32)
        n, ndim = theta.shape
        chi, status = np.empty(n), np.empty(n, dtype=np.int32)
        self.check(self.lib.msx_opt_step(self.h, dptr(theta), chain.ctypes.data_as(C.POINTER(C.c_int32)), n, ndim,
                                         dptr(chi), status.ctypes.data_as(C.POINTER(C.c_int32))))
        return chi, status

    def sampler_run(self, mode, coords, logp, sidx, cidx, partner, zz, zfac, logu):
        """Run len(zz) stretch-move steps on the device.  coords [nw][ndim] and logp [nw] are updated in place;
        returns (chain [nsteps][nw][ndim], logp_chain [nsteps][nw], naccept [nw], worst_status)."""
        i32p = C.POINTER(C.c_int32)
        nsteps = zz.shape[0]
        nw, ndim = coords.shape
        arrs = [np.ascontiguousarray(a, dtype=np.int32) for a in (sidx, cidx, partner)]
        dbl = [as_f64(a) for a in (zz, zfac, logu)]
        chain = np.empty((nsteps, nw, ndim))
        lpc = np.empty((nsteps, nw))
        nacc = np.zeros(nw, dtype=np.int64)
        worst = C.c_int32()
        self.check(self.lib.msx_sampler_run(self.h, int(mode), nw, ndim, nsteps, dptr(coords), dptr(logp),
                                            arrs[0].ctypes.data_as(i32p), arrs[1].ctypes.data_as(i32p),
                                            arrs[2].ctypes.data_as(i32p), dptr(dbl[0]), dptr(dbl[1]), dptr(dbl[2]),
                                            dptr(chain), dptr(lpc), iptr(nacc), C.byref(worst)))
        return chain, lpc, nacc, worst.value

    def sampler_begin(self, mode, coords, logp, max_chunk_steps, naccept=None):
        """Start a pipelined device-resident run (msx_sampler_begin): uploads the ensemble state."""
        coords, logp = as_f64(coords), as_f64(logp)
        nw, ndim = coords.shape
        nacc = None if naccept is None else np.ascontiguousarray(naccept, dtype=np.int64)
        self.check(self.lib.msx_sampler_begin(self.h, int(mode), nw, ndim, int(max_chunk_steps), dptr(coords), dptr(logp),
                                              None if nacc is None else iptr(nacc)))
        self._smp_shape = (nw, ndim)

    def sampler_overlapped(self):
        """1 if the run in flight overlaps its half-steps (msx_sampler_overlapped), 0 if not, -1 before its first chunk."""
        out = C.c_int32()
        self.check(self.lib.msx_sampler_overlapped(self.h, C.byref(out)))
        return out.value

    def sampler_policy(self, overlap=-1):
        """-1: overlap consecutive half-steps when the library's rule allows (default); 0: never (a device shared with other work)."""
        self.check(self.lib.msx_sampler_policy(self.h, int(overlap)))

    def sampler_shard(self, rank, world):
        """Shard the run begun by sampler_begin over `world` ranks (msx_sampler_shard); world > 1 needs comm_init."""
        self.check(self.lib.msx_sampler_shard(self.h, int(rank), int(world)))

    def sampler_enqueue(self, slot, sidx, cidx, partner, zz, zfac, logu):
        """Queue one chunk (arrays of shape (nsteps, 2, nw/2)) without waiting for it."""
        i32p = C.POINTER(C.c_int32)
        arrs = [np.ascontiguousarray(a, dtype=np.int32) for a in (sidx, cidx, partner)]
        dbl = [as_f64(a) for a in (zz, zfac, logu)]
        nsteps = dbl[0].shape[0]
        for a in arrs + dbl:
            if a.shape != (nsteps, 2, self._smp_shape[0] // 2):
                raise ValueError('sampler_enqueue: arrays must have shape (nsteps, 2, nwalkers/2)')
        self.check(self.lib.msx_sampler_enqueue(self.h, int(slot), nsteps, arrs[0].ctypes.data_as(i32p),
                                                arrs[1].ctypes.data_as(i32p), arrs[2].ctypes.data_as(i32p),
                                                dptr(dbl[0]), dptr(dbl[1]), dptr(dbl[2])))
        return nsteps

    def sampler_enqueue_drawn(self, slot, nsteps, seed, a=2.0):
        """Queue one chunk whose randomness the device draws itself (msx_sampler_enqueue_drawn)."""
        self.check(self.lib.msx_sampler_enqueue_drawn(self.h, int(slot), int(nsteps), int(seed) & 0xffffffffffffffff, float(a)))
        return int(nsteps)

    def sampler_draw(self, seed, a, first_iter, nsteps, nw, ndim):
        """The device generator's stream for iterations [first_iter, first_iter + nsteps): (sidx, cidx, partner, zz, zfac,
        logu), each (nsteps, 2, nw/2) -- what EnsembleSampler._draw_steps returns."""
        i32p = C.POINTER(C.c_int32)
        shp = (int(nsteps), 2, int(nw) // 2)
        ints = [np.empty(shp, dtype=np.int32) for _ in range(3)]
        dbl = [np.empty(shp) for _ in range(3)]
        self.check(self.lib.msx_sampler_draw(self.h, int(seed) & 0xffffffffffffffff, float(a), int(first_iter), int(nsteps), int(nw), int(ndim),
                                             ints[0].ctypes.data_as(i32p), ints[1].ctypes.data_as(i32p), ints[2].ctypes.data_as(i32p),
                                             dptr(dbl[0]), dptr(dbl[1]), dptr(dbl[2])))
        return tuple(ints) + tuple(dbl)

    def sampler_collect(self, slot, nsteps):
        """Wait for the chunk in `slot`: (chain [nsteps][nw][ndim], logp [nsteps][nw], naccept [nw], worst)."""
        nw, ndim = self._smp_shape
        chain, lpc = np.empty((nsteps, nw, ndim)), np.empty((nsteps, nw))
        nacc = np.zeros(nw, dtype=np.int64)
        worst = C.c_int32()
        self.check(self.lib.msx_sampler_collect(self.h, int(slot), dptr(chain), dptr(lpc), iptr(nacc), C.byref(worst)))
        return chain, lpc, nacc, worst.value

    def sampler_end(self, want_state=False):
        if not want_state:
            self.check(self.lib.msx_sampler_end(self.h, None, None))
            return None
        nw, ndim = self._smp_shape
        coords, logp = np.empty((nw, ndim)), np.empty(nw)
        self.check(self.lib.msx_sampler_end(self.h, dptr(coords), dptr(logp)))
        return coords, logp

    def make_composite(self, teff, logg, rad, use_distance, plx, win_n, nc, nph):
        teff, logg, rad = as_f64(teff), as_f64(logg), as_f64(rad)
        spec = np.empty(win_n)
        con = np.empty(max(nc, 1))
        ph = np.empty(max(nph, 1))
        st = C.c_int32()
        self.check(self.lib.msx_make_composite(self.h, dptr(teff), dptr(logg), dptr(rad), int(bool(use_distance)),
                                               float(plx), dptr(spec), dptr(con), dptr(ph), C.byref(st)))
        return spec, con[:nc], ph[:nph], st.value

    # ---- RCCL all-gather ------------------------------------------------------------------------------
    def comm_unique_id(self):
        buf = (C.c_uint8 * 128)()
        self.check(self.lib.msx_comm_unique_id(self.h, buf))
        return bytes(buf)

    def comm_init(self, id128, rank, world):
        buf = (C.c_uint8 * 128).from_buffer_copy(bytes(id128))
        self.check(self.lib.msx_comm_init(self.h, buf, int(rank), int(world)))

    # ---- loopback group: the ranks of a sharded run as contexts of one process (include/msx.h) ---------
    @staticmethod
    def comm_init_loopback(contexts):
        """``contexts[r]`` becomes rank r of a loopback group of ``len(contexts)`` ranks (all on one device)."""
        arr = (C.c_void_p * len(contexts))(*[c.h for c in contexts])
        contexts[0].check(contexts[0].lib.msx_comm_init_loopback(arr, len(contexts)))

    @staticmethod
    def sampler_enqueue_group(contexts, slot, sidx, cidx, partner, zz, zfac, logu):
        """One chunk on every rank of a loopback group, in lock-step (arrays as for sampler_enqueue)."""
        i32p = C.POINTER(C.c_int32)
        arrs = [np.ascontiguousarray(a, dtype=np.int32) for a in (sidx, cidx, partner)]
        dbl = [as_f64(a) for a in (zz, zfac, logu)]
        nsteps = dbl[0].shape[0]
        harr = (C.c_void_p * len(contexts))(*[c.h for c in contexts])
        contexts[0].check(contexts[0].lib.msx_sampler_enqueue_group(
            harr, len(contexts), int(slot), nsteps, arrs[0].ctypes.data_as(i32p), arrs[1].ctypes.data_as(i32p),
            arrs[2].ctypes.data_as(i32p), dptr(dbl[0]), dptr(dbl[1]), dptr(dbl[2])))
        return nsteps

    def bytes_per_eval(self, n=256):
        """Bytes the variant an automatic launch of ``n`` walkers takes requests from the memory system, per walker."""
        out = C.c_int64()
        self.check(self.lib.msx_bytes_per_eval(self.h, int(n), C.byref(out)))
        return out.value

    def launch_info(self, n, mode=MODE_LOGPOST, block_threads=0):
        """What an automatic launch of ``n`` walkers would take, from the library's own launcher (msx_launch_info)."""
        out = np.zeros(8, dtype=np.int64)
        name = C.create_string_buffer(512)
        self.check(self.lib.msx_launch_info(self.h, int(mode), int(n), int(block_threads), name, 512, iptr(out)))
        return {'kernel': name.value.decode(), 'form': FORM_NAMES[int(out[0])], 'form_id': int(out[0]), 'threads': int(out[1]),
                'vgprs': int(out[2]), 'static_lds_bytes': int(out[3]), 'dynamic_lds_bytes': int(out[4]),
                'requested_bytes_per_eval': int(out[5]), 'workgroups': int(out[6]), 'walkers_per_sub_batch': int(out[7])}

    def last_form(self):
        """The form (FORM_*) the last launch queued on this context took."""
        out = C.c_int32()
        self.check(self.lib.msx_last_form(self.h, C.byref(out)))
        return out.value

    def pair_stats(self):
        """(pairs, singles) of the pair form's last launch (its planner's counts)."""
        out = np.zeros(2, dtype=np.int64)
        self.check(self.lib.msx_pair_stats(self.h, iptr(out)))
        return int(out[0]), int(out[1])

    def test_hook(self, what, value):
        self.check(self.lib.msx_test_hook(self.h, int(what), int(value)))
