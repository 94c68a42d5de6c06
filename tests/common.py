"""Shared builders for the test-suite: the synthetic problems of tests/golden/make_golden.py,
re-created from seeds (nothing here reads /root/reference)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from mcmc_spec_amd import synth  # noqa: E402
from oracle import mft6_oracle as orc  # noqa: E402

GOLDEN = os.path.join(ROOT, 'tests', 'golden', 'golden_reference.npz')

AV_EDGES, AV_MU, AV_SIG = synth.make_av_table()


def av_prior(dist_pc):
    """Same two-sample construction the golden generator fed to the reference's `bayestar` stub."""
    b = int(np.clip(np.searchsorted(AV_EDGES, dist_pc, side='right') - 1, 0, len(AV_MU) - 1))
    s = np.array([AV_MU[b] - AV_SIG[b], AV_MU[b] + AV_SIG[b]]) / (3.1 * 0.884) * 3.1 * 0.884
    return np.mean(s), np.std(s)


def av_table_exact():
    """(edges, mu, sigma) with mu/sigma evaluated exactly like `av_prior` (bit-identical inputs)."""
    mu, sig = [], []
    for b in range(len(AV_MU)):
        m, s = av_prior(0.5 * (AV_EDGES[b] + AV_EDGES[b + 1]))
        mu.append(m)
        sig.append(s)
    return AV_EDGES, np.array(mu), np.array(sig)


class Case:
    pass


_cache = {}


def golden_grid():
    """The grid tests/golden/make_golden.py used: 13 Teff x 4 logg x 95,000 samples."""
    if 'grid' not in _cache:
        teffs = np.arange(3000, 4300, 100)
        loggs = np.array([4.0, 4.5, 5.0, 5.5])
        wl = np.arange(5000, 24000, 0.2)
        flux = synth.make_grid(teffs, loggs, wl, nlines=1500, seed=11)
        _cache['grid'] = (teffs, loggs, wl, flux)
    return _cache['grid']


def tm_extrema(*tms):
    lo, hi = np.inf, 0
    for tm in tms:
        for w in tm[0]:
            lo, hi = min(lo, min(w)), max(hi, max(w))
    return lo, hi


def golden_case(which):
    """Dataset 'A' (reference synth file + real lp600/Kp tables, no photometry) or 'B' (700 px,
    non-monotonic wavelengths, synthetic filters + 6 photometric bands) of the golden file."""
    key = 'case' + which
    if key in _cache:
        return _cache[key]
    g = np.load(GOLDEN)
    teffs, loggs, wl, flux = golden_grid()
    c = Case()
    c.g = g
    c.teffs, c.loggs, c.wl, c.flux = teffs, loggs, wl, flux
    c.specs = synth.grid_to_specs(teffs, loggs, wl, flux)
    c.matrix = synth.make_isochrone_matrix()
    vw, vf = synth.synthetic_vega()
    c.tables = synth.synthetic_band_tables()
    c.vega = (vw, vf)
    c.bandlib = orc.make_band_library(c.tables, vw, vf)
    c.theta = g['theta']
    c.nspec = 2
    if which == 'A':
        c.data = [g['A_wl'], g['A_flux']]
        c.err = g['A_err']
        c.ctm = [[list(g['ctmA_w0']), list(g['ctmA_w1'])], [list(g['ctmA_t0']), list(g['ctmA_t1'])], [0, 0],
                 [np.mean(g['ctmA_w0']), np.mean(g['ctmA_w1'])]]
        c.ptm = [[], [], [], []]
        c.fr = [synth.EXAMPLE_CMAG, synth.EXAMPLE_CERR, np.array(['lp600', 'Kp']), np.zeros(0), [], np.array([])]
    else:
        c.data = [g['B_wl'], g['B_flux']]
        c.err = g['B_err']
        c.ctm = synth.synthetic_contrast_filters()
        c.ptm = synth.synthetic_phot_filters()
        c.fr = [synth.EXAMPLE_CMAG, synth.EXAMPLE_CERR, np.array(['lp600', 'Kp']), np.array(synth.EXAMPLE_PMAG),
                synth.EXAMPLE_PERR, np.array(['sdss,r', 'sdss,i', 'sdss,z', 'j', 'h', 'k'])]
    if which == 'C':  # triple system on dataset B: 4 contrast filters, first half secondary, second half tertiary
        c.nspec = 3
        c.theta = g['theta3']
        c.ctm = [x + x for x in c.ctm]
        c.fr = [[2.08, 1.3, 3.1, 2.2], [0.14, 0.02, 0.2, 0.05], np.array(['lp600', 'Kp', 'lp600', 'Kp']), c.fr[3], c.fr[4],
                c.fr[5]]
    c.tmi, c.tma = tm_extrema(c.ctm, c.ptm)
    c.r = [min(c.data[0]), max(c.data[0])]
    c.prior = [*np.zeros(10), 2.0732e-3, 0.0277e-3] if c.nspec == 2 else [*np.zeros(14), 2.0732e-3, 0.0277e-3]
    c.tmin, c.tmax = 3000.0, 4200.0
    _cache[key] = c
    return c


def oracle_loglike(c, theta, parts=None):
    return orc.loglikelihood(list(theta), c.fr, c.nspec, c.data, c.err, c.r, c.specs, c.ctm, c.ptm, c.tmi, c.tma,
                             c.matrix, bandlib=c.bandlib, parts=parts)


def oracle_logpost(c, theta, rad_prior=False):
    return orc.logposterior(list(theta), c.fr, c.nspec, c.data, c.err, c.r, c.specs, c.ctm, c.ptm, c.tmi, c.tma,
                            c.tmin, c.tmax, c.matrix, av_prior, prior=c.prior, rad_prior=rad_prior,
                            bandlib=c.bandlib)


def rel_err(a, b):
    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    both_inf = np.isinf(a) & np.isinf(b) & (np.sign(a) == np.sign(b))
    d = np.abs(a - b) / np.maximum(np.abs(b), 1e-300)
    return np.where(both_inf, 0.0, d)
