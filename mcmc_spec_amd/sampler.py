"""A minimal ensemble sampler speaking the emcee protocol the reference drives.

The reference's sampler block (mft6.py:1490-1529, commented out in the snapshot) needs:
``EnsembleSampler(nwalkers, ndim, log_prob_fn, args=, kwargs=)``, ``sample(pos, iterations=)`` yielding
states with ``.coords``, ``get_last_sample()``, ``reset()``, ``get_autocorr_time(quiet=True)``,
``acceptance_fraction`` and ``chain``.  emcee is not installed here, so this module restates the
published algorithm -- Goodman & Weare (2010) affine-invariant stretch move as emcee 3 applies it by
default: the ensemble is split into two random halves that are updated in turn, each walker proposing
``q = c - z (c - s)`` against a random walker ``c`` of the other half with
``z = ((a-1) u + 1)^2 / a`` and acceptance ``(ndim-1) ln z + ln p(q) - ln p(s)``.

With ``vectorize=True`` the log-probability function receives the whole half-ensemble as an
``(n, ndim)`` array -- one fused GPU launch per half-step with ``mcmc_spec_amd.mft6.logposterior``.
"""
from __future__ import annotations

import warnings

import numpy as np


class State:
    def __init__(self, coords, log_prob, random_state=None):
        self.coords = np.array(coords, dtype=float)
        self.log_prob = np.array(log_prob, dtype=float)
        self.random_state = random_state

    def __iter__(self):  # emcee allows `pos, lnp, rstate = state`
        return iter((self.coords, self.log_prob, self.random_state))


def counter_draws(seed, a, ndim, first_iter, m, nwalkers):
    """NumPy restatement of the DEVICE generator (csrc/logprob_kernel.h, sampler_draw_kernel): the randomness of
    iterations ``first_iter .. first_iter + m - 1`` of a run keyed by ``seed`` -- every number a pure function of (seed,
    iteration, stream, index) through SplitMix64's output function.  Returns what ``EnsembleSampler._draw_steps`` returns:
    ``sidx, cidx, partner`` (int32) and ``zz, zfac, logu`` (float64), each (m, 2, nwalkers / 2).  Indices and ``zz`` equal
    the device's bit for bit; ``zfac`` / ``logu`` go through NumPy's ``log`` instead of the device's (last-place
    differences), which is why the device-drawn chain is checked against the host loop fed with ``msx_sampler_draw``'s
    arrays, and this restatement against those arrays (tests/test_gpu_overlap.py)."""
    nw, ns = int(nwalkers), int(nwalkers) // 2
    u64 = np.uint64
    seed = u64(int(seed) & 0xffffffffffffffff)

    def mix(it, stream, index):
        with np.errstate(over='ignore'):
            ctr = (it.astype(u64) << u64(28)) + (u64(stream) << u64(24)) + index.astype(u64)
            x = seed * u64(0xD1342543DE82EF95) + (ctr + u64(1)) * u64(0x9E3779B97F4A7C15)
            x ^= x >> u64(30); x *= u64(0xBF58476D1CE4E5B9)
            x ^= x >> u64(27); x *= u64(0x94D049BB133111EB)
            x ^= x >> u64(31)
        return x

    def uniform(it, stream, index):
        return (mix(it, stream, index) >> u64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)

    it = (np.arange(m, dtype=np.int64) + int(first_iter))[:, None]
    keys = mix(np.broadcast_to(it, (m, nw)), 0, np.broadcast_to(np.arange(nw)[None, :], (m, nw)))
    perm = np.argsort(keys, axis=1, kind='stable').astype(np.int32)      # sorted by (key, index)
    halves = perm.reshape(m, 2, ns)
    sidx, cidx = halves, halves[:, ::-1]
    j = np.broadcast_to(np.arange(ns)[None, :], (m, ns))
    itb = np.broadcast_to(it, (m, ns))
    uz, up, ua = (np.stack([uniform(itb, k + 3 * h, j) for h in (0, 1)], axis=1) for k in (1, 2, 3))
    t1 = (float(a) - 1.0) * uz + 1.0
    zz = (t1 * t1) / float(a)
    partner = np.minimum((up * ns).astype(np.int32), ns - 1)
    with np.errstate(divide='ignore'):
        logu = np.log(ua)
    zfac = (float(ndim) - 1.0) * np.log(zz)
    return np.ascontiguousarray(sidx), np.ascontiguousarray(cidx), partner, zz, zfac, logu


class EnsembleSampler:
    def __init__(self, nwalkers, ndim, log_prob_fn, args=None, kwargs=None, a=2.0, vectorize=False, pool=None,
                 threads=None, seed=None, draws=None):
        """``draws(first_iteration, m)`` (optional) replaces the sampler's own generators: it returns the randomness of
        ``m`` iterations in ``_draw_steps``' layout -- e.g. ``lambda i, m: ctx.sampler_draw(seed, a, i, m, nwalkers, ndim)``
        makes this host loop walk the chain the device-resident sampler draws for itself (``rng='device'``)."""
        self._draws = draws
        self._drawn = 0
        if nwalkers < 2 * ndim or nwalkers % 2:
            raise ValueError('the stretch move needs an even number of walkers >= 2*ndim')
        if threads is not None:
            warnings.warn('threads= is an emcee-2 argument and is ignored (as in emcee 3)', DeprecationWarning)
        self.nwalkers, self.ndim = int(nwalkers), int(ndim)
        self.log_prob_fn = log_prob_fn
        self.args = list(args or [])
        self.kwargs = dict(kwargs or {})
        self.a = float(a)
        self.vectorize = bool(vectorize)
        self.pool = pool
        # two independent streams of one seed: uniforms (stretch factors, partners, accept draws) and the
        # random split.  Each is consumed strictly row by row, so drawing m iterations in one call gives the
        # same numbers as m calls of one iteration -- the host loop (m = 1) and the device-resident loop
        # (m = chunk) walk the same chain
        ss = seed if isinstance(seed, np.random.SeedSequence) else np.random.SeedSequence(seed)
        self.rng, self._rng_split = (np.random.Generator(np.random.PCG64(c)) for c in ss.spawn(2))
        self.reset()

    # ---- bookkeeping ------------------------------------------------------------------------------
    def reset(self):
        self._chain = []
        self._logp = []
        self._accepted = np.zeros(self.nwalkers)
        self.iteration = 0
        self._last = None

    @property
    def chain(self):
        """(nwalkers, nsteps, ndim), the layout ``sampler.chain`` has in the reference (mft6.py:1527)."""
        if not self._chain:
            return np.empty((self.nwalkers, 0, self.ndim))
        return np.swapaxes(np.array(self._chain), 0, 1)

    def get_chain(self, flat=False, thin=1, discard=0):
        c = np.array(self._chain)[discard::thin]
        return c.reshape(-1, self.ndim) if flat else c

    def get_log_prob(self, flat=False, thin=1, discard=0):
        lp = np.array(self._logp)[discard::thin]
        return lp.reshape(-1) if flat else lp

    @property
    def acceptance_fraction(self):
        return self._accepted / max(self.iteration, 1)

    def get_last_sample(self):
        return self._last

    # ---- probability calls --------------------------------------------------------------------------
    def compute_log_prob(self, coords):
        coords = np.asarray(coords, dtype=float)
        if np.any(~np.isfinite(coords)):
            raise ValueError('At least one parameter value was infinite or NaN')
        if self.vectorize:
            lp = np.asarray(self.log_prob_fn(coords, *self.args, **self.kwargs), dtype=float)
        else:
            mapper = self.pool.map if self.pool is not None else map
            lp = np.array([float(v) for v in mapper(_Call(self.log_prob_fn, self.args, self.kwargs), coords)])
        if lp.shape != (len(coords),):
            raise ValueError('log_prob_fn returned the wrong shape')
        if np.any(np.isnan(lp)):
            raise ValueError('Probability function returned NaN')
        return lp

    # ---- the move -------------------------------------------------------------------------------------
    def _draw_steps(self, m):
        """All randomness of ``m`` iterations at once: per iteration the random split into two halves, then per
        half the stretch factors ``z = ((a-1)u+1)^2/a``, the partner index ``floor(u*n_half)`` and ``ln u`` of
        the accept draw.  Returns ``sidx, cidx, partner`` (int32) and ``zz, zfac, logu`` (float64), each of
        shape (m, 2, nwalkers/2); ``zfac = (ndim-1) ln z``.  Shared by the host loop and the device-resident
        loop; everything is vectorised over the m iterations (the host must stay ahead of a 23 us kernel)."""
        if self._draws is not None:
            out = self._draws(self._drawn, m)
            self._drawn += m
            return out
        return self._draw_split(m) + self._draw_moves(m)

    def _draw_split(self, m):
        """(sidx, cidx): the random split of each of m iterations (stream ``_rng_split``)."""
        nw, ns = self.nwalkers, self.nwalkers // 2
        # permuted() has its fast path for 8-byte items
        perm = self._rng_split.permuted(np.broadcast_to(np.arange(nw, dtype=np.int64), (m, nw)).copy(), axis=1)
        halves = perm.astype(np.int32).reshape(m, 2, ns)
        return halves, halves[:, ::-1]

    def _draw_moves(self, m):
        """(partner, zz, zfac, logu) of m iterations (stream ``rng``)."""
        ns = self.nwalkers // 2
        u = self.rng.random((m, 3, 2, ns))
        # contiguous operands: the elementwise loops (log in particular) then take the same code path for
        # every m, which the bit-identity of host and device chains relies on
        uz, up, ua = (np.ascontiguousarray(u[:, j]) for j in range(3))
        zz = ((self.a - 1.0) * uz + 1.0) ** 2 / self.a
        partner = np.minimum((up * ns).astype(np.int32), ns - 1)
        with np.errstate(divide='ignore'):
            logu = np.log(ua)
        zfac = (self.ndim - 1.0) * np.log(zz)
        return partner, zz, zfac, logu

    def _stretch_step(self, coords, logp):
        nw, nd = self.nwalkers, self.ndim
        accepted = np.zeros(nw, dtype=bool)
        sidx, cidx, part, zz_, zfac_, logu_ = self._draw_steps(1)
        for k in (0, 1):
            s_idx, c_idx, zz, partner_idx, logu, zfac = sidx[0, k], cidx[0, k], zz_[0, k], part[0, k], logu_[0, k], zfac_[0, k]
            s, c = coords[s_idx], coords[c_idx]
            partner = c[partner_idx]
            q = partner - (partner - s) * zz[:, None]
            new_lp = self.compute_log_prob(q)
            with np.errstate(invalid='ignore'):  # -inf - -inf = nan -> compares False -> rejected
                lnpdiff = zfac + new_lp - logp[s_idx]
            acc = logu < lnpdiff
            coords[s_idx[acc]] = q[acc]
            logp[s_idx[acc]] = new_lp[acc]
            accepted[s_idx[acc]] = True
        return accepted

    def sample(self, initial_state, iterations=1, store=True):
        if isinstance(initial_state, State):
            coords, logp = initial_state.coords.copy(), initial_state.log_prob.copy()
        else:
            coords, logp = np.array(initial_state, dtype=float), None
        if coords.shape != (self.nwalkers, self.ndim):
            raise ValueError('incompatible input dimensions')
        if logp is None or logp.shape != (self.nwalkers,):
            logp = self.compute_log_prob(coords)
        for _ in range(int(iterations)):
            acc = self._stretch_step(coords, logp)
            self._accepted += acc
            self.iteration += 1
            if store:
                self._chain.append(coords.copy())
                self._logp.append(logp.copy())
            self._last = State(coords, logp)
            yield self._last

    def run_mcmc(self, initial_state, nsteps, **kw):
        st = None
        for st in self.sample(initial_state, iterations=nsteps, **kw):
            pass
        return st

    # ---- diagnostics --------------------------------------------------------------------------------------
    def get_autocorr_time(self, quiet=False, c=5.0, tol=50.0, discard=0, thin=1):
        """Integrated autocorrelation time per dimension (Sokal window, like emcee's ``integrated_time``)."""
        x = self.get_chain(discard=discard, thin=thin)  # (nsteps, nwalkers, ndim)
        if x.shape[0] < 4:
            if quiet:
                return np.full(self.ndim, np.nan)
            raise ValueError('chain too short')
        n = x.shape[0]
        tau = np.empty(self.ndim)
        for d in range(self.ndim):
            f = np.zeros(n)
            for k in range(self.nwalkers):
                f += _autocorr_1d(x[:, k, d])
            f /= self.nwalkers
            taus = 2.0 * np.cumsum(f) - 1.0
            m = np.arange(len(taus)) < c * taus
            win = int(np.argmin(m)) if np.any(~m) else len(taus) - 1
            tau[d] = taus[win]
        tau *= thin
        if np.any(tol * tau > n * thin):
            msg = 'The chain is shorter than {} times the integrated autocorrelation time'.format(tol)
            if not quiet:
                raise RuntimeError(msg)
        return tau


class DeviceEnsembleSampler(EnsembleSampler):
    """The same sampler with the walker state resident in HBM: ``chunk`` iterations are queued on the GPU
    back to back (per half-step: proposal kernel, fused log-probability launch, accept kernel;
    ``msx_sampler_run``) and only the chain comes back.  Randomness is drawn on the host with exactly the
    calls of ``EnsembleSampler``, so for the same seed both samplers produce the same chain, bit for bit.

    ``engine`` is a staged ``mcmc_spec_amd.engine.Engine``; ``mode`` selects ``'logposterior'`` or
    ``'loglikelihood'`` as the target density."""

    def __init__(self, nwalkers, ndim, engine, mode='logposterior', a=2.0, seed=None, chunk=64, shard=None, rng='host', overlap=None):
        """``shard = (rank, world)`` runs the SHARDED form: every rank holds the whole ensemble on
        its GPU, evaluates block ``rank`` of each half-step's proposals, one RCCL all-gather of the new
        log-probabilities crosses xGMI and every rank applies the accept rule for all walkers on the device.
        Every rank must construct the sampler with the SAME seed and start from the same state; every rank then
        holds the same chain, bit-identical to the unsharded one.  ``world > 1`` needs the engine's RCCL
        communicator (``mcmc_spec_amd.dist.init_engine_comm``)."""
        from . import _lib
        # rng = 'host' (default): the randomness is drawn by this object's NumPy generators, exactly the calls of
        # EnsembleSampler -- same seed, same chain as the host loop.  rng = 'device': the library draws it on the GPU with a
        # counter-based generator keyed by `seed` (msx_sampler_enqueue_drawn: one launch per chunk, nothing uploaded, every
        # rank of a sharded run draws the same numbers); the chain is then the one the host loop walks when it is fed
        # `ctx.sampler_draw(seed, a, first_iteration, m, nwalkers, ndim)` (EnsembleSampler(draws=...)).  Up to 4096 walkers.
        # overlap = None: consecutive half-steps run concurrently when the library's rule allows (an unsharded run whose two
        # half-steps fit the chip together: include/msx.h, msx_sampler_policy); False: plain launches -- the choice for a
        # GPU shared with other work, where a waiting workgroup's producer may not get a CU (the chunk then fails with
        # MSX_W_HANDOVER after a bounded wait, a RuntimeError here, and the run has to be started again).
        self.overlap_policy = -1 if overlap is None or overlap else 0
        if rng not in ('host', 'device'):
            raise ValueError("rng must be 'host' or 'device'")
        self.rng_mode = rng
        self.device_seed = int(seed if seed is not None and not isinstance(seed, np.random.SeedSequence) else
                               np.random.SeedSequence(seed).generate_state(1, dtype=np.uint64)[0]) & 0xffffffffffffffff
        self.shard = None if shard is None else (int(shard[0]), int(shard[1]))
        self.overlapped = None   # set by the first chunk of a run: did its half-steps overlap (include/msx.h)?
        self.engine = engine
        self._mode = {'logposterior': _lib.MODE_LOGPOST, 'loglikelihood': _lib.MODE_LOGLIKE}[mode]
        fn = engine.logposterior if mode == 'logposterior' else engine.loglikelihood
        super().__init__(nwalkers, ndim, fn, a=a, vectorize=True, seed=seed)
        self.chunk = int(chunk)

    def sample(self, initial_state, iterations=1, store=True):
        from .engine import _raise_for_status
        if isinstance(initial_state, State):
            coords, logp = initial_state.coords.copy(), initial_state.log_prob.copy()
        else:
            coords, logp = np.array(initial_state, dtype=float), None
        if coords.shape != (self.nwalkers, self.ndim):
            raise ValueError('incompatible input dimensions')
        if logp is None or logp.shape != (self.nwalkers,):
            logp = self.compute_log_prob(coords)
        coords = np.ascontiguousarray(coords)
        logp = np.ascontiguousarray(logp)
        nd, ns = self.ndim, self.nwalkers // 2

        # Pipeline: the randomness of chunk i+1 is drawn (two streams, one thread each; NumPy's generators and
        # the ctypes calls release the GIL) and QUEUED on the GPU while chunk i runs; chunk i is collected only
        # after chunk i+1 has been queued, so the GPU never waits for the host between chunks.  Chunk sizes ramp
        # up from 8 so the first launch does not wait for a whole chunk of randomness.
        from collections import deque
        from concurrent.futures import ThreadPoolExecutor

        device_rng = self.rng_mode == 'device'

        def submit(pool, m):
            if device_rng:
                return None
            return (pool.submit(self._draw_split, m), pool.submit(self._draw_moves, m)) if m > 0 else None

        def next_size(prev, left):
            if self.rng_mode == 'device':   # (nothing to wait for on the host: whole chunks from the start)
                return min(left, self.chunk)
            return min(left, self.chunk, max(8, 2 * prev))

        ctx = self.engine.ctx
        left = int(iterations)
        if left <= 0:
            return
        base_acc = self._accepted.copy()
        ctx.sampler_policy(self.overlap_policy)
        ctx.sampler_begin(self._mode, coords, logp, self.chunk)
        if self.shard is not None:
            ctx.sampler_shard(*self.shard)
        try:
            with ThreadPoolExecutor(max_workers=2) as pool:
                queued = deque()
                m = next_size(4, left)
                fut = submit(pool, m)
                slot = 0
                while left > 0 or queued:
                    if left > 0:
                        arrays = None if device_rng else [x for f in fut for x in f.result()]
                        left -= m
                        m_next = next_size(m, left) if left > 0 else 0
                        fut = submit(pool, m_next)
                        if device_rng:
                            ctx.sampler_enqueue_drawn(slot, m, self.device_seed, self.a)
                        else:
                            ctx.sampler_enqueue(slot, *arrays)
                        self.overlapped = ctx.sampler_overlapped() == 1   # (half-steps on two streams: include/msx.h)
                        queued.append((slot, m))
                        slot ^= 1
                        m = m_next
                        if len(queued) < 2 and left > 0:
                            continue  # keep two chunks in flight
                    sl, mm = queued.popleft()
                    chain, lpc, nacc, worst = ctx.sampler_collect(sl, mm)
                    if worst:
                        _raise_for_status(np.array([worst]), chain[-1][:1])
                    self._accepted = base_acc + nacc
                    for i in range(mm):
                        self.iteration += 1
                        if store:
                            self._chain.append(chain[i])
                            self._logp.append(lpc[i])
                        self._last = State(chain[i], lpc[i])
                        yield self._last
        finally:
            ctx.sampler_end()

    @property
    def acceptance_fraction(self):
        return self._accepted / max(self.iteration, 1)


def _autocorr_1d(x):
    x = np.asarray(x, dtype=float)
    n = 1 << int(np.ceil(np.log2(max(len(x), 2))))
    f = np.fft.fft(x - np.mean(x), n=2 * n)
    acf = np.fft.ifft(f * np.conjugate(f))[: len(x)].real
    if acf[0] == 0:
        return np.ones(len(x))
    return acf / acf[0]


class _Call:
    def __init__(self, f, args, kwargs):
        self.f, self.args, self.kwargs = f, args, kwargs

    def __call__(self, x):
        return self.f(x, *self.args, **self.kwargs)


def run_reference_protocol(sampler, pos, nburn, nsteps, nthin=10, dirname=None, fname='run'):
    """The driver loop of ``run_emcee`` (mft6.py:1494-1529): burn-in, reset, production with the
    ``acl*50 < n`` / 10 % stability convergence test, thinned coordinate dumps and ``samples.txt``.
    Returns the flattened samples ``(nwalkers*nsteps_run, ndim)``."""
    import os
    for n, s in enumerate(sampler.sample(pos, iterations=nburn)):
        if dirname and n % nthin == 0:
            with open('{}/{}_{}_burnin.txt'.format(dirname, fname, n), 'ab') as f:
                f.write(b'\n')
                np.savetxt(f, s.coords)
    state = sampler.get_last_sample()
    sampler.reset()
    old_acl = np.inf
    for n, s in enumerate(sampler.sample(state, iterations=nsteps)):
        if n % nthin == 0:
            if dirname:
                with open('{}/{}_{}_results.txt'.format(dirname, fname, n), 'ab') as f:
                    f.write(b'\n')
                    np.savetxt(f, s.coords)
            acl = sampler.get_autocorr_time(quiet=True)
            macl = np.mean(acl)
            if dirname:
                with open('{}/{}_autocorr.txt'.format(dirname, fname), 'a') as f:
                    f.write(str(macl) + '\n')
            if not np.isnan(macl):
                converged = np.all(acl * 50 < n)
                converged &= np.all((np.abs(old_acl - acl) / acl) < 0.1)
                if converged:
                    break
            old_acl = acl
    samples = sampler.chain[:, :, :].reshape((-1, sampler.ndim))
    if dirname:
        np.savetxt(os.path.join(dirname, 'samples.txt'), samples)
    return samples
