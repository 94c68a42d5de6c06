"""Grid loader + staging: the ``models='btsettl'`` branch of ``spec_interpolator``
(mft6.py:323-385) with the reference's signature.

Text parsing is host work by nature; every flux operation runs on the GPU: the per-node linear
resample onto the common 0.2 A grid (``msx_resample_linear``), the upload of the dense grid
(``msx_stage_grid``) and the Gaussian broadening of the data window of every node in place
(``msx_broaden_grid``).  The returned object is a ``dict`` with the reference's keys
(``'{T}, {g}'`` -> float64 array, ``'wl'``) -- so the reference's own code can keep indexing it --
that also remembers the engine holding the staged copy: ``mcmc_spec_amd.mft6`` reuses it instead
of uploading the grid again.
"""
from __future__ import annotations

import os
from glob import glob

import numpy as np

from .engine import Engine

GRID_DIR = 'BT-Settl_M-0.0a+0.0'  # mft6.py:324


class StagedSpecs(dict):
    """``specs`` dict + the Engine that already holds it in HBM."""
    engine = None


def _read_two_columns(path, lo, hi):
    """Samples of one BT-Settl text file with lo <= wavelength <= hi (mft6.py:353-357)."""
    xs, ys = [], []
    with open(path) as fh:
        for line in fh:
            li = line.split(' ')
            x = float(li[0])
            if lo <= x <= hi:
                xs.append(x)
                ys.append(float(li[1]))
    return np.array(xs), np.array(ys)


def _node_lists(files, trange, lgrange):
    t, l = [], []
    for f in files:  # mft6.py:330-340 (glob order; only membership matters downstream)
        base = os.path.basename(f)
        nu = int(float(base.split('-')[0].split('e')[1]) * 1e2)
        mu = float(base.split('-')[1])
        if nu not in t and min(trange) <= nu <= max(trange):
            t.append(nu)
        if mu not in l and min(lgrange) <= mu <= max(lgrange):
            l.append(mu)
    return t, l


def spec_interpolator(w, trange, lgrange, specrange, npix=3, resolution=10000, metal=0, write_file=True,
                      models='btsettl', grid_dir=GRID_DIR, device=None, cache=None):
    """Read, resample, stage and broaden the model grid.  ``w`` = data window [A] (``[spmin*1e4, spmax*1e4]``
    at mft6.py:3512), ``specrange`` = [specmin, specmax] [A].  Returns ``StagedSpecs``.

    ``cache``: optional ``.npz`` path; reused when the file list, mtimes, ranges and resolution match."""
    if models != 'btsettl':
        raise NotImplementedError("only models='btsettl' is on the hot path (SURVEY.md §2)")
    from . import mft6 as _api
    files = sorted(glob(os.path.join(grid_dir, 'lte*')))
    if not files:
        raise IndexError('list index out of range')  # what glob(...)[0] raises in find_model, mft6.py:251
    t, l = _node_lists(files, trange, lgrange)
    wl = np.arange(min(specrange), max(specrange), 0.2)  # mft6.py:343
    stamp = np.array([os.path.getmtime(f) for f in files] + [min(w), max(w), min(specrange), max(specrange),
                                                             float(resolution), len(t), len(l)])
    eng = Engine(_api._DEVICE if device is None else device)
    teff, logg = sorted(t), sorted(l)
    flux = np.zeros((len(teff), len(logg), len(wl)))
    cached = None
    if cache and os.path.exists(cache):
        z = np.load(cache)
        if z['stamp'].shape == stamp.shape and np.array_equal(z['stamp'], stamp):
            cached = z
    if cached is not None:
        flux = cached['flux']
        eng.stage_grid(wl, np.array(teff, float), np.array(logg, float), flux)
    else:
        for it, tt in enumerate(teff):
            for ig, ll in enumerate(logg):
                name = os.path.join(grid_dir, 'lte{}-{}-0.0a+0.0.BT-Settl.spec.7.txt'.format(
                    str(int(tt / 1e2)).zfill(3), str(ll)))  # find_model, mft6.py:246-251
                hits = glob(name)
                if not hits:
                    raise IndexError('list index out of range')
                xs, ys = _read_two_columns(hits[0], min(specrange) - 100, max(specrange) + 100)
                if np.any(np.diff(xs) < 0):  # interp1d sorts its abscissa (stable)
                    order = np.argsort(xs, kind='mergesort')
                    xs, ys = xs[order], ys[order]
                flux[it, ig] = eng.ctx.resample_linear(xs, ys, wl)  # mft6.py:369-371
        eng.stage_grid(wl, np.array(teff, float), np.array(logg, float), flux)
        eng.broaden_grid_window([min(w), max(w)], resolution)  # mft6.py:373-378
        for it in range(len(teff)):
            for ig in range(len(logg)):
                flux[it, ig] = eng.ctx.read_node(it, ig)
        if cache:
            np.savez(cache, stamp=stamp, flux=flux)
    specs = StagedSpecs()
    for it, tt in enumerate(teff):
        for ig, ll in enumerate(logg):
            specs['{}, {}'.format(tt, ll)] = flux[it, ig]
    specs['wl'] = wl  # the splice of mft6.py:381 re-assembles the same vector
    specs.engine = eng
    eng._problem_key = None
    eng._problem_refs = None
    _api._GRIDS[id(specs)] = (specs, eng)
    return specs
