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


class EnsembleSampler:
    def __init__(self, nwalkers, ndim, log_prob_fn, args=None, kwargs=None, a=2.0, vectorize=False, pool=None,
                 threads=None, seed=None):
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
        self.rng = np.random.default_rng(seed)
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
    def _draw_step(self):
        """All randomness of one iteration, in a fixed order: the split, then per half the stretch factors,
        the partner indices and the accept uniforms.  Shared by the host loop and the device-resident loop,
        which therefore walk the same chain for the same seed."""
        nw = self.nwalkers
        perm = self.rng.permutation(nw)
        halves = (perm[: nw // 2], perm[nw // 2:])
        draws = []
        for k in (0, 1):
            ns = len(halves[k])
            zz = ((self.a - 1.0) * self.rng.random(ns) + 1.0) ** 2 / self.a
            partner = self.rng.integers(len(halves[1 - k]), size=ns)
            logu = np.log(self.rng.random(ns))
            draws.append((halves[k], halves[1 - k], zz, partner, logu))
        return draws

    def _stretch_step(self, coords, logp):
        nw, nd = self.nwalkers, self.ndim
        accepted = np.zeros(nw, dtype=bool)
        for s_idx, c_idx, zz, partner_idx, logu in self._draw_step():
            s, c = coords[s_idx], coords[c_idx]
            partner = c[partner_idx]
            q = partner - (partner - s) * zz[:, None]
            new_lp = self.compute_log_prob(q)
            with np.errstate(invalid='ignore'):  # -inf - -inf = nan -> compares False -> rejected
                lnpdiff = (nd - 1.0) * np.log(zz) + new_lp - logp[s_idx]
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

    def __init__(self, nwalkers, ndim, engine, mode='logposterior', a=2.0, seed=None, chunk=64):
        from . import _lib
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

        def draw_chunk(m):
            sidx = np.empty((m, 2, ns), dtype=np.int32)
            cidx = np.empty((m, 2, ns), dtype=np.int32)
            partner = np.empty((m, 2, ns), dtype=np.int32)
            zz, zfac, logu = np.empty((m, 2, ns)), np.empty((m, 2, ns)), np.empty((m, 2, ns))
            for i in range(m):
                for k, (s_i, c_i, z, p_i, lu) in enumerate(self._draw_step()):
                    sidx[i, k], cidx[i, k], partner[i, k] = s_i, c_i, p_i
                    zz[i, k], zfac[i, k], logu[i, k] = z, (nd - 1.0) * np.log(z), lu
            return sidx, cidx, partner, zz, zfac, logu

        # the randomness of chunk i+1 is drawn (in stream order, by one worker thread) while the GPU runs
        # chunk i: the ctypes call releases the GIL for the whole chunk
        from concurrent.futures import ThreadPoolExecutor
        left = int(iterations)
        with ThreadPoolExecutor(max_workers=1) as pool:
            m = min(left, self.chunk)
            fut = pool.submit(draw_chunk, m) if m > 0 else None
            while left > 0:
                sidx, cidx, partner, zz, zfac, logu = fut.result()
                left -= m
                m_next = min(left, self.chunk)
                fut = pool.submit(draw_chunk, m_next) if m_next > 0 else None
                before = self._accepted.copy()
                chain, lpc, nacc, worst = self.engine.ctx.sampler_run(self._mode, coords, logp, sidx, cidx, partner, zz,
                                                                      zfac, logu)
                if worst:
                    _raise_for_status(np.array([worst]), coords[:1])
                self._accepted = before + nacc
                for i in range(m):
                    self.iteration += 1
                    if store:
                        self._chain.append(chain[i])
                        self._logp.append(lpc[i])
                    self._last = State(chain[i], lpc[i])
                    yield self._last
                m = m_next

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
