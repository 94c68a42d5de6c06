"""Drop-in replacements for the hot-path functions of ``mft6.py`` with the reference's own call
signatures: ``logposterior``, ``loglikelihood``, ``logprior``, ``make_composite``,
``broaden``.  Host code stays Python; every number comes out of the HIP library.

    import mcmc_spec_amd.mft6 as mft6          # instead of `import mft6`
    mft6.set_band_library(bands)               # replaces the module global `lib` (mft6.py:21)
    mft6.set_av_prior(edges_pc, mu, sigma)     # replaces the module global `bayestar` (mft6.py:23)
    sampler = EnsembleSampler(nwalkers, ndim, mft6.logposterior, args=[...], kwargs={...}, vectorize=True)
    sampler = mft6.device_sampler(nwalkers, ndim, args=[...], kwargs={...})   # the same chain, resident on the GPU

Differences from the reference, all at the boundary:
  * ``p0`` may also be a 2-D ``(n, ndim)`` array (emcee ``vectorize=True``): a length-n array comes back.
  * static arguments (``specs``, ``data``, ``err``, ``fr``, ``ctm``, ``ptm``, ``matrix`` ...) are staged to
    the GPU on first use and cached by object identity: treat them as immutable while you sample.
  * pyphot / dustmaps are not importable here, so the passbands and the A_V(distance) prior table are
    registered explicitly (functions above) instead of being created at import.
"""
from __future__ import annotations

import collections
import os

import numpy as np

from .engine import Engine

_BANDS = None
_AV_TABLE = None
_AV_OPTIONAL = False
_SPECTRUM = True
_DEVICE = int(os.environ.get('LOCAL_RANK', '0'))
_GRIDS = collections.OrderedDict()      # id(specs) -> (specs, Engine)
_MAX_GRIDS = 2
_LAST_DATASET = None                    # (specs, fr, data, err, r, ctm, ptm, tmi, tma) of the last staging call


def set_device(device):
    """GPU ordinal used by the drop-in functions (default: $LOCAL_RANK or 0)."""
    global _DEVICE
    _DEVICE = int(device)


def set_band_library(bands):
    """``{name: bands.Band}`` for the names pyphot's library uses (mft6.py:766-769)."""
    global _BANDS
    _BANDS = bands
    _invalidate_problems()


def set_av_prior(edges_pc, mu=None, sigma=None, optional=False):
    """Register the line-of-sight A_V prior: bin edges in parsec and per-bin mean / std of
    ``bayestar(...) * 3.1 * 0.884`` (mft6.py:1233-1236).  ``set_av_prior(None, optional=True)`` drops
    the Gaussian A_V term altogether (the reference cannot do that; use only for tests)."""
    global _AV_TABLE, _AV_OPTIONAL
    _AV_TABLE = None if edges_pc is None else (np.asarray(edges_pc, float), np.asarray(mu, float),
                                               np.asarray(sigma, float))
    _AV_OPTIONAL = bool(optional)
    _invalidate_problems()


def av_table_from_query(query, ra, dec, edges_pc):
    """Tabulate a Bayestar-like callable ``query(ra, dec, distance_pc) -> samples`` per distance bin
    (evaluated at the bin centre) exactly as the reference post-processes it (mft6.py:1234-1236)."""
    mu, sig = [], []
    for lo, hi in zip(edges_pc[:-1], edges_pc[1:]):
        s = np.asarray(query(ra, dec, 0.5 * (lo + hi))) * 3.1 * 0.884
        mu.append(np.mean(s))
        sig.append(np.std(s))
    return np.asarray(edges_pc, float), np.array(mu), np.array(sig)


def set_spectrum_term(on=True):
    """``False`` selects the ``mft6_nospec.py`` variant of the likelihood: contrast + photometry chi^2 only
    (mft6_nospec.py:1163-1196)."""
    global _SPECTRUM
    _SPECTRUM = bool(on)
    _invalidate_problems()


def _invalidate_problems():
    for _, (_, eng) in _GRIDS.items():
        eng._problem_key = None


def _engine_for(specs):
    key = id(specs)
    hit = _GRIDS.get(key)
    if hit is not None and hit[0] is specs:
        _GRIDS.move_to_end(key)
        return hit[1]
    eng = Engine(_DEVICE)
    eng.stage_specs(specs)
    eng._problem_key = None
    eng._problem_refs = None
    _GRIDS[key] = (specs, eng)  # the strong reference keeps id(specs) from being recycled
    while len(_GRIDS) > _MAX_GRIDS:
        _GRIDS.popitem(last=False)
    return eng


def _ids(*objs):
    out = []
    for o in objs:
        if isinstance(o, (list, tuple)):
            out.append(tuple(id(x) for x in o))
        else:
            out.append(id(o))
    return tuple(out)


def _prior_key(prior):
    if isinstance(prior, (int, float)) and prior == 0:
        return 0
    return tuple(float(x) for x in prior)


def _staged(specs, fr, nspec, data, err, r, ctm, ptm, tmi, tma, matrix, tmin=-np.inf, tmax=np.inf, prior=0, a=True,
            dist_fit=True, rad_prior=False, need_prior=False):
    eng = _engine_for(specs)
    key = (_ids(fr, data, err, ctm, ptm, matrix), tuple(float(x) for x in r), float(tmi), float(tma), int(nspec),
           float(tmin), float(tmax), _prior_key(prior), bool(a), bool(dist_fit), bool(rad_prior), bool(need_prior),
           _SPECTRUM)
    if eng._problem_key != key:
        av = None
        if need_prior and a:
            if _AV_TABLE is None and not _AV_OPTIONAL:
                raise RuntimeError('logposterior/logprior with a=True need the A_V(distance) prior table: call '
                                   'mcmc_spec_amd.mft6.set_av_prior(edges_pc, mu, sigma) first (the reference '
                                   'queries dustmaps Bayestar here, mft6.py:1233-1239)')
            av = _AV_TABLE
        eng.stage_problem(data, err, fr, r, ctm, ptm, tmi, tma, matrix, nspec=int(nspec), bands=_BANDS, av_table=av,
                          tmin=tmin, tmax=tmax, prior=prior, use_av=bool(a), dist_fit=dist_fit, rad_prior=rad_prior,
                          spectrum=_SPECTRUM)
        eng._problem_key = key
        eng._problem_refs = (fr, data, err, ctm, ptm, matrix)  # pin the ids in the key
    global _LAST_DATASET
    _LAST_DATASET = (specs, fr, data, err, r, ctm, ptm, tmi, tma)
    return eng


def _check_p0(p0, nspec):
    p = np.asarray(p0, dtype=float)
    if p.shape[-1] not in (6, 8) or p.shape[-1] != 2 * int(nspec) + 2:
        # the reference prints and returns None (mft6.py:1457) / raises UnboundLocalError (mft6.py:1161)
        raise ValueError("P0 doesn't match what I was expecting")
    return p


def logposterior(p0, fr, nspec, ndust, data, err, broadening, r, specs, ctm, ptm, tmi, tma, vs, tmin, tmax, matrix,
                 ra, dec, wu='aa', dust=False, norm=True, prior=0, a=True, models='btsettl', dist_fit=True,
                 rad_prior=False):
    """mft6.py:1459-1470: log prior + log likelihood, ``-inf`` outside the prior box.  emcee's ``log_prob_fn``."""
    p = _check_p0(p0, nspec)
    eng = _staged(specs, fr, nspec, data, err, r, ctm, ptm, tmi, tma, matrix, tmin, tmax, prior, a, dist_fit,
                  rad_prior, need_prior=True)
    return eng.logposterior(p)


def device_sampler(nwalkers, ndim, args, kwargs=None, a=2.0, seed=None, chunk=64):
    """The sampler line of ``run_emcee`` -- ``emcee.EnsembleSampler(nwalkers, ndim, logposterior, args=[...],
    kwargs={...})``, mft6.py:1490-1492 -- with the walker state RESIDENT on the GPU: ``args`` / ``kwargs`` are exactly
    what that line passes for ``logposterior`` (everything after ``p0``); the problem they describe is staged once and
    the stretch move runs on the device (``DeviceEnsembleSampler``: same interface -- ``sample``, ``run_mcmc``,
    ``get_chain``, ``get_log_prob``, ``acceptance_fraction``, ``reset`` -- and, for the same seed, the same chain bit
    for bit as ``EnsembleSampler(nwalkers, ndim, logposterior, args=args, kwargs=kwargs, vectorize=True)``)."""
    from .sampler import DeviceEnsembleSampler
    names = ('fr', 'nspec', 'ndust', 'data', 'err', 'broadening', 'r', 'specs', 'ctm', 'ptm', 'tmi', 'tma', 'vs', 'tmin',
             'tmax', 'matrix', 'ra', 'dec', 'wu', 'dust', 'norm', 'prior', 'a', 'models', 'dist_fit', 'rad_prior')
    given = dict(zip(names, args))
    if len(args) > len(names):
        raise TypeError('device_sampler: too many positional arguments for logposterior')
    for k, v in (kwargs or {}).items():
        if k not in names:
            raise TypeError("logposterior() got an unexpected keyword argument '{}'".format(k))
        if k in given:
            raise TypeError("logposterior() got multiple values for argument '{}'".format(k))
        given[k] = v
    missing = [k for k in names[:18] if k not in given]
    if missing:
        raise TypeError('device_sampler: logposterior arguments missing: ' + ', '.join(missing))
    g = given.get
    if 2 * int(g('nspec')) + 2 != int(ndim):
        raise ValueError("P0 doesn't match what I was expecting")
    eng = _staged(g('specs'), g('fr'), g('nspec'), g('data'), g('err'), g('r'), g('ctm'), g('ptm'), g('tmi'), g('tma'),
                  g('matrix'), g('tmin'), g('tmax'), g('prior', 0), g('a', True), g('dist_fit', True),
                  g('rad_prior', False), need_prior=True)
    return DeviceEnsembleSampler(nwalkers, ndim, eng, mode='logposterior', a=a, seed=seed, chunk=chunk)


def loglikelihood(p0, fr, nspec, ndust, data, err, broadening, r, specs, ctm, ptm, tmi, tma, vs, matrix, w='aa',
                  dust=False, norm=True, mode='spec', av=True, optimize=False, models='btsettl'):
    """mft6.py:1139-1205: ``-0.5 * total chi^2`` (or the chi^2 itself when ``optimize=True``)."""
    p = _check_p0(p0, nspec)
    eng = _staged(specs, fr, nspec, data, err, r, ctm, ptm, tmi, tma, matrix, a=av)
    return eng.loglikelihood(p, optimize=optimize)


def logprior(p0, nspec, ndust, tmin, tmax, matrix, ra, dec, prior=0, ext=True, dist_fit=True, rad_prior=False,
             specs=None):
    """mft6.py:1207-1272.  The value depends on ``p0`` and on the arguments above only; the reference signature
    carries neither grid nor dataset, but the device evaluates priors inside a staged problem, so one is needed:
    the dataset of the LAST ``logposterior`` / ``loglikelihood`` call (recorded explicitly, not "whichever engine
    was touched last"), or -- with ``specs=`` -- the last dataset staged on that grid."""
    p = _check_p0(p0, nspec)
    last = _LAST_DATASET
    if specs is not None and (last is None or last[0] is not specs):
        hit = _GRIDS.get(id(specs))
        if hit is None or hit[0] is not specs or hit[1]._problem_refs is None or hit[1].tables is None:
            raise RuntimeError('logprior(specs=...): nothing has been staged on this grid yet; call '
                               'logposterior/loglikelihood with it once first')
        fr, data, err, ctm, ptm, _ = hit[1]._problem_refs
        st = hit[1].tables
        last = (specs, fr, data, err, st.r, ctm, ptm, st.tmi, st.tma)
    if last is None:
        raise RuntimeError('logprior needs a staged dataset: call logposterior/loglikelihood once first')
    sp, fr, data, err, r, ctm, ptm, tmi, tma = last
    eng = _staged(sp, fr, nspec, data, err, r, ctm, ptm, tmi, tma, matrix, tmin, tmax, prior, ext, dist_fit, rad_prior,
                  need_prior=True)
    return eng.logprior(p)


def make_composite(teff, logg, rad, distance, contrast_filt, phot_filt, r, specs, ctm, ptm, tmi, tma, vs, nspec=2,
                   normalize=False, res=1000, npix=3, models='btsettl', plot=False):
    """mft6.py:651-831 (``plot=False``): ``(wl, spec, contrast, phot_cwl, phot)``."""
    if plot:
        raise NotImplementedError('make_composite(plot=True) is plotting support (out of scope, SURVEY.md §2)')
    eng = _engine_for(specs)
    wl4 = np.linspace(min(r), max(r), 4)
    key = ('composite', _ids(ctm, ptm), tuple(float(x) for x in r), float(tmi), float(tma), int(nspec),
           len(contrast_filt), len(phot_filt))
    if eng._problem_key != key:
        nc, nph = len(contrast_filt), len(phot_filt)
        fr = [np.zeros(nc), np.ones(nc), list(contrast_filt), np.zeros(nph), np.ones(nph), list(phot_filt)]
        eng.stage_problem([wl4, np.ones(4)], np.ones(4), fr, r, ctm, ptm, tmi, tma, _flat_matrix(), nspec=int(nspec),
                          bands=_BANDS)
        eng._problem_key = key
        eng._problem_refs = None
    return eng.make_composite(teff, logg, rad, distance)


def _flat_matrix():
    """A 2-row placeholder isochrone for problems that never look logg up (make_composite gets logg
    from its caller)."""
    m = np.zeros((2, 8))
    m[:, 1] = 9.0
    m[:, 4] = [1.0, 1.0e5]
    m[:, 5] = [4.5, 4.5]
    m[:, 6] = [1.0, 1.0]
    return m


def broaden(even_wl, modelspec_interp, res, vsini=0, limb=0, plot=False):
    """mft6.py:124-152 with ``vsini = limb = 0``: Gaussian instrumental broadening + the two edge patches."""
    if vsini != 0 and limb != 0:
        raise NotImplementedError('rotational broadening is dead code in the reference path (vsini = 0)')
    if _GRIDS:
        ctx = next(reversed(_GRIDS.values()))[1].ctx
    else:
        from ._lib import Context
        ctx = Context(_DEVICE)
    out = ctx.broaden(np.asarray(even_wl, float), np.asarray(modelspec_interp, float), res, 5.0)
    return np.array(even_wl), out


def clear_cache():
    global _LAST_DATASET
    _GRIDS.clear()
    _LAST_DATASET = None
