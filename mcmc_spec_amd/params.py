"""Parameter-file reader and data preparation of the reference's ``main`` (mft6.py:3450-3597): the plumbing
that turns ``param_koi2298.txt`` + a three-column spectrum file into the arguments of ``logposterior``.

Only the parts that feed the hot path are here (the CLI itself is out of scope): the key/value
parser with its quirks, the telluric mask, the exclusive crop + median normalisation, the list / boolean
conventions, the KIC -> SDSS photometry conversion and the ``fr`` / ``prior`` assembly.  Every expression
follows the reference line it cites so that the prepared vectors are bit-identical to the reference's
(``tests/golden/make_golden_koi.py`` asserts that against the reference's own statements).
"""
from __future__ import annotations

import numpy as np


def read_param_file(parfile):
    """mft6.py:3458-3466.  ``key value`` separated by ONE space; the value ends at the first tab (a trailing
    newline is kept when there is no tab -- harmless for ``float()`` / substring tests, stripped by the list
    parsers).  Lines that start with ``#`` or are blank are skipped; a line without a space raises, as there."""
    pardict = {}
    with open(parfile) as fi:
        for line in fi:
            if not line.startswith('#') and not line.strip() == '':
                (key, val) = line.split(' ')[0:2]
                val = val.split('\t')[0]
                pardict[str(key)] = val
    return pardict


def truthy(val):
    """Booleans by substring: ``'t' in val.lower()`` (mft6.py:3480, 3496, 3517)."""
    return 't' in str(val).lower()


def float_list(val):
    """``[2.08,1.3]`` -> floats (mft6.py:3523-3524, 3539); the string ``np.nan`` -> NaN (mft6.py:3530-3536)."""
    toks = val.strip('[] \n').split(',')
    if toks == ['']:
        return []   # (an empty list; the reference itself cannot parse one)
    try:
        return [float(p) for p in toks]
    except ValueError:
        if 'np.nan' in val:
            return [np.nan if p.strip() == 'np.nan' else float(p) for p in toks]
        raise


def name_list(val):
    """``['lp600','Kp']`` -> array of names: split on quotes, drop the commas / empties (mft6.py:3525-3526)."""
    f = np.array([p.strip('\\') for p in val.strip('[] \n').split('\'')])
    return np.array([p for p in f if len(p) >= 1 and not p == ','])


def telluric_mask(data_wl, *cols):
    """The four kept wavelength intervals of mft6.py:3497-3499 (micron), applied to every column."""
    parts = (np.where(data_wl <= 0.6860), np.where((data_wl >= 0.6880) & (data_wl <= 0.7600)),
             np.where((data_wl >= 0.7660) & (data_wl <= 0.8210)), np.where(data_wl > 0.8240))
    return tuple(np.concatenate([c[p] for p in parts]) for c in (data_wl,) + cols)


def prepare_data(filename, spmin, spmax, mask=False):
    """mft6.py:3492-3507: read (wl [um], flux, err), optional telluric mask, crop ``spmin < wl < spmax``
    (exclusive), divide error then flux by the median flux.  Returns (data_wl, dsp, de)."""
    data_wl, dsp, de = np.genfromtxt(filename, unpack=True)
    if mask:
        data_wl, dsp, de = telluric_mask(data_wl, dsp, de)
    keep = np.where((data_wl > float(spmin)) & (data_wl < float(spmax)))
    data_wl, dsp, de = data_wl[keep], dsp[keep], de[keep]
    de /= np.median(dsp)
    dsp /= np.median(dsp)
    return data_wl, dsp, de


# m_KIC -> m_SDSS (mft6.py:3547-3549; the ab_to_vega table of :3546 is not applied, :3558)
_KIC_SLOPE = {'g': 0.0921, 'r': 0.0548, 'i': 0.0696, 'z': 0.1587}
_KIC_INT = {'g': -0.0985, 'r': -0.0383, 'i': -0.0583, 'z': -0.0597}
_KIC_COLOR = {'g': 'g-r', 'r': 'r-i', 'i': 'r-i', 'z': 'i-z'}


def convert_photometry(oldphot, phot_filt, synthetic=False):
    """mft6.py:3553-3562: SDSS bands are converted from the KIC system with a colour term taken from the same
    list; everything else passes through.  ``synthetic`` = the parameter file's name contains ``synth``."""
    oldphot = list(oldphot)
    phot_filt = np.asarray(phot_filt)
    if synthetic:
        return np.array(oldphot)
    phot = np.zeros(len(oldphot))
    for n, p in enumerate(phot_filt):
        if 'sdss' in p.lower():
            band = p.split(',')[1]
            a, b = _KIC_COLOR[band].split('-')
            color = oldphot[np.where('sdss,' + a == phot_filt)[0][0]] - oldphot[np.where('sdss,' + b == phot_filt)[0][0]]
            phot[n] = _KIC_INT[band] + _KIC_SLOPE[band] * color + oldphot[n]
        else:
            phot[n] = oldphot[n]
    return phot


class Run:
    """Everything ``main`` derives from the parameter file before it touches the model grid."""


def load_run(parfile, data_root=None):
    """Parse ``parfile`` and prepare the data the way ``main`` does (mft6.py:3469-3575).  ``data_root`` is
    prepended to a relative ``filename``.  Returns a ``Run`` with: ``pardict``, ``data = [wl, flux]``, ``err``,
    ``fr = [cmag, cerr, cfilt, pmag, perr, pfilt]``, ``r = [min wl, max wl]``, ``res``, ``tmin/tmax``,
    ``specrange``, ``plx/plx_err``, ``av/av_err``, ``ra/dec``, ``dist_fit``, ``rad_prior``, ``nspec``, ``ndust``,
    ``nwalk``, ``nstep``, ``nburn``, ``nsteps`` and ``prior(ndim)`` (the list handed to ``run_emcee``, :3689)."""
    import os
    pd = read_param_file(parfile)
    run = Run()
    run.pardict = pd
    run.models, run.res = pd['models'], int(pd['res'])
    run.mask = truthy(pd.get('mask', 'f'))          # mft6.py:3470-3473, 3496
    run.rad_prior = truthy(pd.get('rad_prior', 'f'))  # mft6.py:3475-3483
    fn = pd['filename'].strip()
    if data_root is not None and not os.path.isabs(fn):
        fn = os.path.join(data_root, fn)
    run.filename = fn
    run.spmin, run.spmax = float(pd['spmin']), float(pd['spmax'])
    wl, flux, err = prepare_data(fn, run.spmin, run.spmax, mask=run.mask)
    run.data, run.err = [wl, flux], err
    run.r = [min(wl), max(wl)]                      # mft6.py:3688
    run.tmin, run.tmax = int(pd['tmin']), int(pd['tmax'])
    run.specrange = [int(pd['specmin']), int(pd['specmax'])]
    run.logg_range = [4, 5.5]                       # hard-coded at mft6.py:3512 (lgmin / lgmax are ignored)
    run.plx, run.plx_err = float(pd['plx']), float(pd['plx_err'])
    run.dist_fit = truthy(pd['dist_fit'])
    mags, me = float_list(pd['cmag']), float_list(pd['cerr'])
    filts = name_list(pd['cfilt'])
    oldphot = float_list(pd['pmag'])
    phot_err = float_list(pd['perr'])
    phot_filt = name_list(pd['pfilt'])
    phot = convert_photometry(oldphot, phot_filt, synthetic='synth' in str(parfile))
    run.fr = [mags, me, filts, phot, phot_err, phot_filt]  # mft6.py:3575
    run.av, run.av_err = float(pd['av']), float(pd['av_err'])
    run.ra, run.dec = float(pd['ra']), float(pd['dec'])
    run.nwalk, run.nstep = int(pd['nwalk']), int(pd['nstep'])
    run.nspec, run.ndust = int(pd['nspec']), int(pd['ndust'])
    run.nburn, run.nsteps = int(pd['nburn']), int(pd['nsteps'])
    run.real_values = float_list(pd['real_values']) if 'real_values' in pd else []
    run.prior = lambda ndim: [*np.zeros(ndim * 2 - 2), run.plx, run.plx_err]  # mft6.py:3689
    return run
