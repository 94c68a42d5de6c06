"""``Engine``: one GPU, one staged model grid, one staged dataset -- the object the drop-in functions
in ``mcmc_spec_amd.mft6`` cache behind the reference's call signatures.

All numerics run in ``libmsx.so`` (HIP, gfx950).  This class only moves arguments across the C ABI
and maps per-walker status codes onto the exceptions the reference would raise.
"""
from __future__ import annotations

import numpy as np

from . import _lib
from . import staging


def _raise_for_status(status, theta):
    bad = np.nonzero(status > _lib.W_REJECT)[0]
    if bad.size == 0:
        return
    i = int(bad[0])
    code = int(status[i])
    where = 'walker {} (theta = {})'.format(i, np.array2string(np.asarray(theta)[i], precision=6))
    if code == _lib.W_KEYERROR:
        raise KeyError('a model grid node needed by {} is not in specs'.format(where))  # mft6.py:489-500
    if code == _lib.W_INDEXERROR:
        raise IndexError('list index out of range: Teff/logg bracket past the last grid node for ' + where)
    if code == _lib.W_VALUEERROR:
        raise ValueError('A value in x_new is outside the interpolation range (isochrone Teff) for ' + where)
    if code == _lib.W_HANDOVER:
        raise RuntimeError('device fault: workgroups that wait for each other inside a launch (the linked form; the '
                           'overlapped half-steps of the device-resident sampler) did not meet within 20 ms for ' + where +
                           ' (stage the problem again before the next launch)')
    raise RuntimeError('unknown walker status {} for {}'.format(code, where))


class Engine:
    def __init__(self, device=0):
        self.ctx = _lib.Context(device)
        self.device = device
        self.grid = None
        self.tables = None

    # ---- A0 ---------------------------------------------------------------------------------------
    def stage_grid(self, wl, teff_nodes, logg_nodes, flux, present=None):
        self.ctx.stage_grid(wl, teff_nodes, logg_nodes, flux, present)
        self.grid = dict(wl=np.asarray(wl, dtype=float), teff=np.asarray(teff_nodes, dtype=float),
                         logg=np.asarray(logg_nodes, dtype=float))
        self.tables = None

    def stage_specs(self, specs):
        teff, logg, wl, flux, present = staging.parse_specs(specs)
        self.stage_grid(wl, teff, logg, flux, present)

    def broaden_grid_window(self, w_aa, resolution, placement='staging'):
        """Broaden every node over the data window ``[min(w), max(w)]`` [A] in place: the staging step
        of ``spec_interpolator`` (mft6.py:366-378).  ``placement='in_path'`` additionally keeps the raw window on the
        device: problems staged afterwards can be evaluated with the broadening applied per walker
        (``ctx.set_path(_lib.PATH_INPATH)``) beside the default forms."""
        self.ctx.set_broadening(placement)
        wl = self.grid['wl']
        idx = np.where((wl >= min(w_aa)) & (wl <= max(w_aa)))[0]
        self.ctx.broaden_grid(int(idx[0]), int(idx.size), resolution, 5.0)
        self.tables = None

    # ---- problem ------------------------------------------------------------------------------------
    def stage_problem(self, data, err, fr, r, ctm, ptm, tmi, tma, matrix, nspec=2, bands=None, av_table=None,
                      tmin=-np.inf, tmax=np.inf, prior=0, use_av=True, dist_fit=True, rad_prior=False, spectrum=True,
                      store='f64'):
        """``store='f32'`` keeps the per-node pixel table R in float32 (include/msx.h, msx_set_grid_storage): a SEPARATELY
        LABELLED storage precision -- ~1e-7 relative on the log-probability at S/N 100 instead of the float64 tables'
        1e-13, a quarter fewer bytes through the CU's L2 port; fused binaries only.  Default: float64."""
        if self.grid is None:
            raise RuntimeError('stage the model grid first')
        self.ctx.set_grid_storage(store)
        st = staging.build_problem(self.ctx, self.grid['wl'], data, err, fr, r, ctm, ptm, tmi, tma, matrix,
                                   nspec=nspec, bands=bands, av_table=av_table, tmin=tmin, tmax=tmax, prior=prior,
                                   use_av=use_av, dist_fit=dist_fit, rad_prior=rad_prior, spectrum=spectrum)
        self.ctx.stage_problem(st.prob)
        self.tables = st
        self.nspec = int(nspec)
        self.ndim = 2 * self.nspec + 2

    # ---- hot path -----------------------------------------------------------------------------------
    def _eval(self, theta, mode):
        theta = np.asarray(theta, dtype=float)
        single = theta.ndim == 1
        th = np.atleast_2d(theta)
        if th.shape[1] != self.ndim:
            raise ValueError("P0 doesn't match what I was expecting")  # mft6.py:1457
        logp, status = self.ctx.logprob_batch(th, mode)
        _raise_for_status(status, th)
        return float(logp[0]) if single else logp

    def logposterior(self, theta):
        return self._eval(theta, _lib.MODE_LOGPOST)

    def logprior(self, theta):
        return self._eval(theta, _lib.MODE_LOGPRIOR)

    def loglikelihood(self, theta, optimize=False):
        return self._eval(theta, _lib.MODE_CHISQ if optimize else _lib.MODE_LOGLIKE)

    def make_composite(self, teff, logg, rad, distance):
        """Returns (wl, spec, contrast list, phot_cwl, phot) like mft6.py:831."""
        st = self.tables
        use_d = not (type(distance) == bool)
        spec, con, ph, status = self.ctx.make_composite(teff, logg, rad, use_d, float(distance) if use_d else 0.0,
                                                        st.window[1], st.nc, st.nph)
        _raise_for_status(np.array([status]), np.array([list(teff) + list(logg)]))
        j0, n = st.window
        return (self.grid['wl'][j0:j0 + n].copy(), spec, [c for c in con], st.phot_cwl.copy(), ph)
