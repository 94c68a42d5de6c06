#!/usr/bin/env python3
"""Generate the committed golden vectors by running the REFERENCE's own functions.

Runs only in the build container (needs ``/root/reference/mft6.py``); the resulting ``.npz`` files
are data (inputs + expected outputs) and are what travels.  Usage::

    python tests/golden/make_golden.py            # writes tests/golden/golden_*.npz

How the reference is executed: ``mft6.py`` imports 13 third-party modules that are
not installed; they are pre-seeded in ``sys.modules`` with ``MagicMock`` so the import succeeds, and
the handful of third-party *functions* the hot path calls are replaced by functional stand-ins:

  * ``extinction.ccm89/apply``      -> published CCM89 (oracle.mft6_oracle.ccm89)      [unpinned]
  * ``pyphot`` ``lib[...]``         -> photon-counting band flux per pyphot's algorithm [unpinned]
  * ``bayestar`` / ``SkyCoord``     -> a two-sample line-of-sight table                 [unpinned]

Everything else -- ``get_logg, get_radius, get_spec, interp_2_spec, find_nearest, make_composite,
chisq, norm_spec, loglikelihood, logprior, logposterior`` -- is the reference's code, unmodified,
driven on a synthetic grid (``mcmc_spec_amd.synth``) through a scratch directory of *empty* files
named like the BT-Settl grid (``get_spec`` only parses the names, mft6.py:423-436,457).
"""
import os
import sys
import tempfile
import types
import warnings
from unittest.mock import MagicMock

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True
warnings.filterwarnings('ignore')

from mcmc_spec_amd import synth  # noqa: E402
from oracle import mft6_oracle as orc  # noqa: E402

REF = '/root/reference'
MISSING = ['synphot', 'astropy', 'astropy.io', 'astropy.io.fits', 'astropy.units', 'astropy.table',
           'astropy.coordinates', 'PyAstronomy', 'PyAstronomy.pyasl', 'emcee', 'corner', 'extinction',
           'pyphot', 'dustmaps', 'dustmaps.bayestar']


def import_reference():
    for m in MISSING:
        sys.modules.setdefault(m, MagicMock())
    sys.path.insert(0, REF)
    import mft6
    return mft6


class _Q(float):
    """float with the ``.value`` attribute the reference reads off pyphot quantities (mft6.py:780)."""

    @property
    def value(self):
        return float(self)

    def __truediv__(self, other):
        return _Q(float(self) / float(other))


class _BandStub:
    def __init__(self, band):
        self._b = band
        self.Vega_zero_flux = _Q(band.Vega_zero_flux)
        self.AB_zero_flux = _Q(band.AB_zero_flux)

    def get_flux(self, slamb, sflux):
        return _Q(self._b.get_flux(np.asarray(slamb), np.asarray(sflux)))


AV_EDGES, AV_MU, AV_SIG = synth.make_av_table()


def av_samples(dist_pc):
    """Two 'samples' whose mean*3.1*0.884 = mu(bin) and population std = sigma(bin)."""
    b = int(np.clip(np.searchsorted(AV_EDGES, dist_pc, side='right') - 1, 0, len(AV_MU) - 1))
    return np.array([AV_MU[b] - AV_SIG[b], AV_MU[b] + AV_SIG[b]]) / (3.1 * 0.884)


def av_prior(dist_pc):
    s = av_samples(dist_pc) * 3.1 * 0.884
    return np.mean(s), np.std(s)


class _NumpyCompat:
    """``mft6.fit_spec`` stacks ragged rows with ``np.vstack((sp, gi))`` (mft6.py:1066), which old NumPy
    turned into an object array and NumPy >= 1.24 refuses.  This proxy restores exactly that behaviour
    (object rows) and forwards everything else to the installed NumPy untouched -- no arithmetic is
    involved, ``sp`` is only bookkeeping that fit_spec writes to ``params{n}.txt``."""

    def __getattr__(self, name):
        return getattr(np, name)

    @staticmethod
    def vstack(tup):
        try:
            return np.vstack(tup)
        except ValueError:
            rows = []
            for a in tup:
                if isinstance(a, np.ndarray) and a.dtype == object and a.ndim == 2:
                    rows += [list(r) for r in a]
                else:
                    rows.append(list(a))
            out = np.empty((len(rows), len(rows[0])), dtype=object)
            for i, r in enumerate(rows):
                for j, x in enumerate(r):
                    out[i, j] = x
            return out


def patch_third_party(mft6, bandlib):
    mft6.extinction = types.SimpleNamespace(
        ccm89=lambda wl, av, rv: orc.ccm89(wl, av, rv),
        apply=lambda ext, flux: np.asarray(flux) * 10.0 ** (-0.4 * np.asarray(ext)))
    mft6.pyphot = types.SimpleNamespace(unit=lambda s: 1.0)
    mft6.lib = {k: _BandStub(v) for k, v in bandlib.items()}
    mft6.u = types.SimpleNamespace(deg=1.0, pc=1.0)
    mft6.SkyCoord = lambda ra, dec, distance=None: distance
    mft6.bayestar = lambda pos, mode='samples': av_samples(pos)


def scratch_grid_dir(teffs, loggs):
    d = tempfile.mkdtemp(prefix='msx_golden_')
    os.mkdir(os.path.join(d, 'BT-Settl_M-0.0a+0.0'))
    for t in teffs:
        for g in loggs:
            name = 'lte{}-{}-0.0a+0.0.BT-Settl.spec.7.txt'.format(str(int(t / 1e2)).zfill(3), str(float(g)))
            open(os.path.join(d, 'BT-Settl_M-0.0a+0.0', name), 'w').close()
    return d


def write_btsettl_files(root, seed=21):
    """Small synthetic BT-Settl-format text files (two space-separated columns, irregular wavelength
    steps) for Teff 3000..3300 x logg 4.5, 5.0 -- the loader test regenerates them from the same seed."""
    from mcmc_spec_amd import synth as sy
    return sy.write_btsettl_text_grid(os.path.join(root, 'BT-Settl_M-0.0a+0.0'), seed=seed)


def load_reference_filter_tables():
    """Real contrast filter tables as the reference reads them (mft6.py:598-600, 631-634)."""
    lp = np.genfromtxt(os.path.join(REF, 'bps/lp600.csv'), delimiter=',')
    kp = np.genfromtxt(os.path.join(REF, 'bps/keck_kp.txt'))
    wls = [list(lp[:, 0] * 10), list(kp[:, 0] * 1e4)]
    tras = [list(lp[:, 1]), list(kp[:, 1])]
    res = 1700
    nres = [(max(w) - min(w)) / (np.mean(w) / res) for w in wls]
    cwl = [np.mean(w) for w in wls]
    return [wls, tras, nres, cwl]


def edge_case_walkers(rng, n_random):
    base = synth.TRUTH_THETA
    th = [base.copy()]
    th.append(np.array([3800.0, 3100.0, 0.106, 0.4994, 0.31, 2.0732e-3]))      # both Teff on nodes
    th.append(np.array([3850.0, 3000.0, 0.0, 0.4994, 0.31, 2.0732e-3]))        # A_V = 0 -> no reddening
    th.append(np.array([3849.999, 3050.0, 0.2, 0.6, 0.5, 2.5e-3]))             # Teff half-way: tie in find_nearest
    th.append(np.array([4150.0, 3950.0, 0.05, 0.7, 0.9, 1.0e-3]))
    th.append(np.array([3050.0, 3049.0, 0.5, 0.3, 1.2, 5.0e-3]))
    for _ in range(n_random):
        t = base + rng.normal(size=6) * synth.WALKER_SIGMA * np.array([4, 4, 2, 3, 3, 5])
        t[0:2] = np.clip(t[0:2], 3001.0, 4199.0)
        t[2] = abs(t[2])
        t[3] = np.clip(t[3], 0.06, 1.4)
        t[4] = np.clip(t[4], 0.06, None)
        t[5] = np.clip(t[5], 1 / 2900.0, 1 / 5.0)
        th.append(t)
    return np.array(th)


def main():
    mft6 = import_reference()
    rng = np.random.default_rng(7)

    # ---------------------------------------------------------------- synthetic model grid
    teffs = np.arange(3000, 4300, 100)
    loggs = np.array([4.0, 4.5, 5.0, 5.5])
    wl = np.arange(5000, 24000, 0.2)
    flux = synth.make_grid(teffs, loggs, wl, nlines=1500, seed=11)
    specs = synth.grid_to_specs(teffs, loggs, wl, flux)
    matrix = synth.make_isochrone_matrix()
    vega_w, vega_f = synth.synthetic_vega()
    bandlib = orc.make_band_library(synth.synthetic_band_tables(), vega_w, vega_f)
    patch_third_party(mft6, bandlib)

    cwd = os.getcwd()
    os.chdir(scratch_grid_dir(teffs, loggs))
    out = {}
    try:
        # ------------------------------------------------------------ A1: get_logg / get_radius
        tq = np.array([3000.0, 3025.0, 3850.0, 4199.5, 5000.0, 2900.0, 6500.0])
        out['a1_teff'] = tq
        out['a1_logg'] = np.array([float(mft6.get_logg(t, matrix)) for t in tq])
        out['a1_radius'] = np.array([float(mft6.get_radius(t, matrix)) for t in tq])
        assert np.array_equal(out['a1_logg'], [float(orc.get_logg(t, matrix)) for t in tq])
        assert np.array_equal(out['a1_radius'], [float(orc.get_radius(t, matrix)) for t in tq])

        # ------------------------------------------------------------ A2: get_spec
        cases = [(3850.0, 4.76), (3800.0, 4.76), (3850.0, 4.5), (3800.0, 4.5), (3849.999, 4.2499),
                 (3050.0, 5.16), (4199.0, 4.01), (3950.0, 4.75)]
        reg = [0.55, 0.90]
        sub = slice(0, None, 97)
        a2 = []
        for (t, g) in cases:
            w_ref, s_ref = mft6.get_spec(t, g, reg, specs)
            w_o, s_o = orc.get_spec(t, g, reg, specs)
            assert np.array_equal(w_ref, w_o) and np.array_equal(s_ref, s_o), (t, g)
            a2.append(s_ref[sub])
        out['a2_cases'] = np.array(cases)
        out['a2_flux_sub'] = np.array(a2)
        out['a2_npts'] = np.array([len(w_ref)])

        # ------------------------------------------------------------ datasets
        ctm_real = load_reference_filter_tables()
        ctm_syn = synth.synthetic_contrast_filters()
        ptm_empty = [[], [], [], []]
        ptm6 = synth.synthetic_phot_filters()

        def tm_extrema(*tms):
            lo, hi = np.inf, 0
            for tm in tms:
                for w in tm[0]:
                    lo, hi = min(lo, min(w)), max(hi, max(w))
            return lo, hi

        # dataset A: the reference's own synthetic spectrum file, cropped/normalised like mft6.py:3502-3507
        dw, ds, de = np.genfromtxt(os.path.join(REF, 'Data/synth_spec_3850_3025.txt'), unpack=True)
        keep = np.where((dw > 0.55) & (dw < 0.90))
        dw, ds, de = dw[keep], ds[keep], de[keep]
        de = de / np.median(ds)
        ds = ds / np.median(ds)
        out['A_wl'], out['A_flux'], out['A_err'] = dw, ds, de

        # dataset B: 700 px, KOI-like two overlapping arms (non-monotonic wavelengths), sigma errors
        wb = np.concatenate((np.linspace(0.5601, 0.7103, 380), np.linspace(0.6952, 0.8897, 320)))
        out['B_wl'] = wb

        theta = edge_case_walkers(rng, 26)
        out['theta'] = theta

        def run_ll(name, data, err, fr, ctm, ptm, store_model_for=(0, 2, 3)):
            tmi, tma = tm_extrema(ctm, ptm)
            r = [min(data[0]), max(data[0])]
            ll_ref, ll_orc, parts_list = [], [], []
            for i, p in enumerate(theta):
                v = mft6.loglikelihood(list(p), fr, 2, 0, data, err, 1700, r, specs, ctm, ptm, tmi, tma, None,
                                       matrix)
                parts = {}
                o = orc.loglikelihood(list(p), fr, 2, data, err, r, specs, ctm, ptm, tmi, tma, matrix,
                                      bandlib=bandlib, parts=parts)
                ll_ref.append(v)
                ll_orc.append(o)
                parts_list.append(parts)
            ll_ref, ll_orc = np.array(ll_ref), np.array(ll_orc)
            rel = np.max(np.abs(ll_ref - ll_orc) / np.abs(ll_ref))
            print('{}: max rel |oracle - reference| on loglikelihood = {:.3e}'.format(name, rel))
            assert rel < 1e-13, rel
            out[name + '_loglike'] = ll_ref
            out[name + '_contrast'] = np.array([q['contrast'] for q in parts_list])
            out[name + '_phot'] = np.array([q['phot'] for q in parts_list])
            out[name + '_iic'] = np.array([q['iic'] for q in parts_list])
            out[name + '_model'] = np.array([parts_list[i]['model'] for i in store_model_for])
            out[name + '_model_idx'] = np.array(store_model_for)
            # make_composite pieces straight from the reference for one walker
            p = theta[0]
            lg = [mft6.get_logg(t, matrix) for t in p[:2]]
            w1, c1, con, pcw, ph = mft6.make_composite(p[:2], lg, p[3:5], p[5], fr[2], fr[5], r, specs, ctm, ptm,
                                                       tmi, tma, None, nspec=2)
            out[name + '_mc_wl_ends'] = np.array([w1[0], w1[-1], len(w1)])
            out[name + '_mc_spec_sub'] = c1[::211]
            out[name + '_mc_contrast'] = np.array(con)
            out[name + '_mc_phot'] = np.array(ph, dtype=float)
            return tmi, tma, r

        # config-1-like: real data file, real lp600/Kp tables, no photometry
        frA = [synth.EXAMPLE_CMAG, synth.EXAMPLE_CERR, np.array(['lp600', 'Kp']), np.zeros(0), [], np.array([])]
        out['ctmA_w0'], out['ctmA_t0'] = np.array(ctm_real[0][0]), np.array(ctm_real[1][0])
        out['ctmA_w1'], out['ctmA_t1'] = np.array(ctm_real[0][1]), np.array(ctm_real[1][1])
        tmiA, tmaA, rA = run_ll('A', [dw, ds], de, frA, ctm_real, ptm_empty)

        # dataset B: generated from the reference's make_composite like mft6.py:3632-3642, then 1 % noise
        frB = [synth.EXAMPLE_CMAG, synth.EXAMPLE_CERR, np.array(['lp600', 'Kp']), np.array(synth.EXAMPLE_PMAG),
               synth.EXAMPLE_PERR, np.array(['sdss,r', 'sdss,i', 'sdss,z', 'j', 'h', 'k'])]
        tmiB, tmaB = tm_extrema(ctm_syn, ptm6)
        p = synth.TRUTH_THETA
        lg = [mft6.get_logg(t, matrix) for t in p[:2]]
        w1, c1, _, _, _ = mft6.make_composite(p[:2], lg, p[3:5], p[5], frB[2], frB[5], [min(wb), max(wb)], specs,
                                              ctm_syn, ptm6, tmiB, tmaB, None, nspec=2)
        c1 = mft6.extinct(w1, c1, p[2])
        fb = mft6.interp1d(w1, c1)(wb * 1e4)
        noise = rng.normal(0, 0.01 * fb)
        sb, eb = (fb + noise) / np.median(fb + noise), 0.01 * fb / np.median(fb + noise)
        out['B_flux'], out['B_err'] = sb, eb
        run_ll('B', [wb, sb], eb, frB, ctm_syn, ptm6)

        # ------------------------------------------------------------ prior + posterior (dataset A)
        prior = [*np.zeros(10), 2.0732e-3, 0.0277e-3]
        extra = np.array([[2999.0, 3025.0, 0.1, 0.5, 0.3, 2e-3], [3850.0, 4201.0, 0.1, 0.5, 0.3, 2e-3],
                          [3850.0, 3025.0, -0.01, 0.5, 0.3, 2e-3], [3850.0, 3025.0, 0.1, 0.049, 0.3, 2e-3],
                          [3850.0, 3025.0, 0.1, 1.51, 0.3, 2e-3], [3850.0, 3025.0, 0.1, 0.5, 0.04, 2e-3],
                          [3850.0, 3025.0, 0.1, 0.5, 0.3, 1 / 3001.0], [3850.0, 3025.0, 0.1, 0.5, 0.3, 0.2501]])
        th2 = np.vstack([theta, extra])
        out['theta_post'] = th2
        for rp in (False, True):
            lp_ref, po_ref = [], []
            for pq in th2:
                lp = mft6.logprior(list(pq), 2, 0, 3000.0, 4200.0, matrix, 10.0, 20.0, prior=prior, ext=True,
                                   dist_fit=True, rad_prior=rp)
                lo = orc.logprior(list(pq), 2, 3000.0, 4200.0, matrix, av_prior, prior=prior, rad_prior=rp)
                assert (lp == lo) or abs(lp - lo) <= 1e-13 * abs(lp), (lp, lo)
                po = mft6.logposterior(list(pq), frA, 2, 0, [dw, ds], de, 1700, rA, specs, ctm_real, ptm_empty,
                                       tmiA, tmaA, None, 3000.0, 4200.0, matrix, 10.0, 20.0, prior=prior,
                                       rad_prior=rp)
                oo = orc.logposterior(list(pq), frA, 2, [dw, ds], de, rA, specs, ctm_real, ptm_empty, tmiA, tmaA,
                                      3000.0, 4200.0, matrix, av_prior, prior=prior, rad_prior=rp,
                                      bandlib=bandlib)
                assert (po == oo) or abs(po - oo) <= 1e-13 * abs(po), (po, oo)
                lp_ref.append(lp)
                po_ref.append(po)
            tag = 'radprior' if rp else 'noradprior'
            out['A_logprior_' + tag] = np.array(lp_ref)
            out['A_logpost_' + tag] = np.array(po_ref)

        # ------------------------------------------------------------ the mft6_nospec.py variant (dataset B)
        import importlib.util
        nsp = importlib.util.spec_from_file_location('mft6_nospec', os.path.join(REF, 'mft6_nospec.py'))
        mns = importlib.util.module_from_spec(nsp)
        nsp.loader.exec_module(mns)
        patch_third_party(mns, bandlib)
        ns_ref = []
        for pq in theta[:12]:
            v = mns.loglikelihood(list(pq), frB, 2, 0, [wb, sb], eb, 1700, [min(wb), max(wb)], specs, ctm_syn, ptm6, tmiB,
                                  tmaB, None, matrix)
            o = orc.loglikelihood(list(pq), frB, 2, [wb, sb], eb, [min(wb), max(wb)], specs, ctm_syn, ptm6, tmiB, tmaB,
                                  matrix, bandlib=bandlib, spectrum=False)
            assert v == o, (v, o)
            ns_ref.append(v)
        out['B_nospec_loglike'] = np.array(ns_ref)

        # ------------------------------------------------------------ dist_fit=False branch (mft6.py:1275-1327)
        prior_nd = [3800.0, 0.0, 60.0, 1.0, 0.12, 0.03, 0.5, 0.0, 0.05, 1.0, 2.0732e-3, 0.0277e-3]
        th_nd = np.vstack([theta[:10], [[3850.0, 3025.0, 0.1, 1.7, 0.3, 2e-3], [3850.0, 3025.0, 0.1, 0.5, 0.3, 0.3],
                                        [3850.0, 3025.0, 0.1, 0.5, 0.049, 2e-3], [3850.0, 3025.0, -0.1, 0.5, 0.3, 2e-3],
                                        [2999.0, 3025.0, 0.1, 0.5, 0.3, 2e-3]]])
        out['theta_nodist'] = th_nd
        for rp in (False, True):
            lp_ref, po_ref = [], []
            for pq in th_nd:
                lp = mft6.logprior(list(pq), 2, 0, 3000.0, 4200.0, matrix, 10.0, 20.0, prior=prior_nd, ext=True,
                                   dist_fit=False, rad_prior=rp)
                lo = orc.logprior(list(pq), 2, 3000.0, 4200.0, matrix, av_prior, prior=prior_nd, dist_fit=False,
                                  rad_prior=rp)
                assert (lp == lo) or abs(lp - lo) <= 1e-13 * abs(lp), (lp, lo)
                po = mft6.logposterior(list(pq), frA, 2, 0, [dw, ds], de, 1700, rA, specs, ctm_real, ptm_empty, tmiA, tmaA,
                                       None, 3000.0, 4200.0, matrix, 10.0, 20.0, prior=prior_nd, dist_fit=False,
                                       rad_prior=rp)
                lp_ref.append(lp)
                po_ref.append(po)
            tag = 'radprior' if rp else 'noradprior'
            out['A_nodist_logprior_' + tag] = np.array(lp_ref)
            out['A_nodist_logpost_' + tag] = np.array(po_ref)
        out['prior_nodist'] = np.array(prior_nd)

        # ------------------------------------------------------------ triple system (ndim 8), dataset B
        # contrast list follows the reference's own triple example ['880','Kp','880','Kp'] (mft6.py:3632):
        # first half of the filters = secondary - primary, second half = tertiary - primary (mft6.py:747-749)
        ctm4 = [ctm_syn[0] + ctm_syn[0], ctm_syn[1] + ctm_syn[1], ctm_syn[2] + ctm_syn[2], ctm_syn[3] + ctm_syn[3]]
        frC = [[2.08, 1.3, 3.1, 2.2], [0.14, 0.02, 0.2, 0.05], np.array(['lp600', 'Kp', 'lp600', 'Kp']),
               np.array(synth.EXAMPLE_PMAG), synth.EXAMPLE_PERR, np.array(['sdss,r', 'sdss,i', 'sdss,z', 'j', 'h', 'k'])]
        th3 = []
        for t in theta[:14]:
            th3.append([t[0], t[1], max(3001.0, t[1] - 150.0 + 40.0 * len(th3)) if t[1] > 3200 else 3001.0 + 13.0 * len(th3),
                        t[2], t[3], t[4], 0.6 * t[4], t[5]])
        th3 = np.array(th3)
        th3_bad = np.array([[3850.0, 3400.0, 2999.0, 0.1, 0.5, 0.4, 0.3, 2e-3], [3850.0, 3400.0, 3100.0, 0.1, 0.5, 0.4, 0.04, 2e-3],
                            [3850.0, 3400.0, 3100.0, 0.1, 0.5, 0.4, 0.3, 1 / 1001.0], [3850.0, 3400.0, 3100.0, -0.1, 0.5, 0.4, 0.3, 2e-3]])
        out['theta3'] = np.vstack([th3, th3_bad])
        prior3 = [*np.zeros(14), 2.0732e-3, 0.0277e-3]
        rB = [min(wb), max(wb)]
        ll3, lp3, po3 = [], [], []
        for pq in out['theta3']:
            lp = mft6.logprior(list(pq), 3, 0, 3000.0, 4200.0, matrix, 10.0, 20.0, prior=prior3, ext=True,
                               dist_fit=True, rad_prior=True)
            lo = orc.logprior(list(pq), 3, 3000.0, 4200.0, matrix, av_prior, prior=prior3, rad_prior=True)
            assert (lp == lo) or abs(lp - lo) <= 1e-13 * abs(lp), (lp, lo)
            lp3.append(lp)
            po = mft6.logposterior(list(pq), frC, 3, 0, [wb, sb], eb, 1700, rB, specs, ctm4, ptm6, tmiB, tmaB, None,
                                   3000.0, 4200.0, matrix, 10.0, 20.0, prior=prior3, rad_prior=True)
            oo = orc.logposterior(list(pq), frC, 3, [wb, sb], eb, rB, specs, ctm4, ptm6, tmiB, tmaB, 3000.0, 4200.0,
                                  matrix, av_prior, prior=prior3, rad_prior=True, bandlib=bandlib)
            assert (po == oo) or abs(po - oo) <= 1e-13 * abs(po), (po, oo)
            po3.append(po)
            if np.isfinite(lp):
                v = mft6.loglikelihood(list(pq), frC, 3, 0, [wb, sb], eb, 1700, rB, specs, ctm4, ptm6, tmiB, tmaB, None,
                                       matrix)
                o = orc.loglikelihood(list(pq), frC, 3, [wb, sb], eb, rB, specs, ctm4, ptm6, tmiB, tmaB, matrix,
                                      bandlib=bandlib)
                assert v == o or abs(v - o) <= 1e-13 * abs(v), (v, o)
                ll3.append(v)
            else:
                ll3.append(np.nan)
        out['C_loglike'], out['C_logprior'], out['C_logpost'] = np.array(ll3), np.array(lp3), np.array(po3)
        print('triple: {} walkers, {} inside the box'.format(len(po3), int(np.isfinite(po3).sum())))

        # ------------------------------------------------------------ triple system, dist_fit=False (mft6.py:1397-1455)
        # Gates (:1411): Teff box, the two RATIOS >= 0.05 (R1 is not tested), plx >= 0, A_V >= 0; the Gaussian list is
        # shorter (:1425-1442: Teff x 3, A_V, R1, ratio 2 -- no ratio 3, no parallax term).  Extra walkers: R1 = 0.03 and
        # plx = 0.3 (both pass here, fail with dist_fit), ratio 2 / ratio 3 below 0.05, plx < 0, A_V < 0, Teff outside.
        prior3_nd = [3800.0, 3400.0, 0.0, 60.0, 80.0, 1.0, 0.12, 0.03, 0.5, 0.4, 0.3, 0.05, 0.04, 0.03, 2.0732e-3, 0.0277e-3]
        th3_nd = np.vstack([th3[:10], [[3850.0, 3400.0, 3100.0, 0.1, 0.03, 0.4, 0.3, 2e-3], [3850.0, 3400.0, 3100.0, 0.1, 0.5, 0.4, 0.3, 0.3],
                                       [3850.0, 3400.0, 3100.0, 0.1, 0.5, 0.049, 0.3, 2e-3], [3850.0, 3400.0, 3100.0, 0.1, 0.5, 0.4, 0.049, 2e-3],
                                       [3850.0, 3400.0, 3100.0, 0.1, 0.5, 0.4, 0.3, -1e-3], [3850.0, 3400.0, 3100.0, -0.1, 0.5, 0.4, 0.3, 2e-3],
                                       [3850.0, 3400.0, 4201.0, 0.1, 0.5, 0.4, 0.3, 2e-3]]])
        out['theta3_nodist'] = th3_nd
        out['prior3_nodist'] = np.array(prior3_nd)
        for rp in (False, True):
            lp_ref, po_ref = [], []
            for pq in th3_nd:
                lp = mft6.logprior(list(pq), 3, 0, 3000.0, 4200.0, matrix, 10.0, 20.0, prior=prior3_nd, ext=True,
                                   dist_fit=False, rad_prior=rp)
                lo = orc.logprior(list(pq), 3, 3000.0, 4200.0, matrix, av_prior, prior=prior3_nd, dist_fit=False,
                                  rad_prior=rp)
                assert (lp == lo) or abs(lp - lo) <= 1e-13 * abs(lp), (lp, lo)
                po = mft6.logposterior(list(pq), frC, 3, 0, [wb, sb], eb, 1700, rB, specs, ctm4, ptm6, tmiB, tmaB, None,
                                       3000.0, 4200.0, matrix, 10.0, 20.0, prior=prior3_nd, dist_fit=False, rad_prior=rp)
                oo = orc.logposterior(list(pq), frC, 3, [wb, sb], eb, rB, specs, ctm4, ptm6, tmiB, tmaB, 3000.0, 4200.0,
                                      matrix, av_prior, prior=prior3_nd, dist_fit=False, rad_prior=rp, bandlib=bandlib)
                assert (po == oo) or abs(po - oo) <= 1e-13 * abs(po), (po, oo)
                lp_ref.append(lp)
                po_ref.append(po)
            tag = 'radprior' if rp else 'noradprior'
            out['C_nodist_logprior_' + tag] = np.array(lp_ref)
            out['C_nodist_logpost_' + tag] = np.array(po_ref)
        print('triple, dist_fit=False: {} walkers, {} inside the box'.format(len(th3_nd), int(np.isfinite(po_ref).sum())))

        # ------------------------------------------------------------ f4: fit_spec (pre-optimiser), dataset B
        # The reference draws its proposals from the unseeded global RNG (make_varied_param, mft6.py:211-228);
        # to get a reproducible trajectory the draw is redirected to a seeded Generator with the same call
        # pattern.  Everything else -- bounds, counters, repair loops, accept rule, chi^2 -- is fit_spec itself.
        prng = np.random.default_rng(123)
        mft6.make_varied_param = lambda init, sig: [prng.normal(init[n], sig[n]) for n in range(len(init))]
        outdir = tempfile.mkdtemp(prefix='msx_golden_fit_')
        mft6.np = _NumpyCompat()
        steps = 24
        start = dict(t=[3700.0, 3200.0], av=0.2, rad=[0.6, 0.45], plx=2.2e-3)
        best, cs = mft6.fit_spec(0, outdir, wb.copy(), sb.copy(), eb, [min(wb), max(wb)], list(start['t']),
                                 [start['av'], 0.106, 0.01], list(start['rad']), frB, specs, [3000.0, 4200.0],
                                 [start['plx'], 2.0732e-3, 0.0277e-3], ctm_syn, ptm6, tmiB, tmaB, None, matrix, 10.0, 20.0,
                                 nspec=2, steps=steps, dist_fit=True, rad_prior=True)
        out['D_start'] = np.array(start['t'] + [start['av']] + start['rad'] + [start['plx']])
        out['D_steps'] = np.array([steps])
        out['D_best'] = np.array([float(x) for x in best.split()])
        out['D_best_chi'] = np.array([cs])
        out['D_params'] = np.atleast_2d(np.genfromtxt(os.path.join(outdir, 'params0.txt')))
        # chisq0.txt holds `savechi[n] savetest[n]`; savetest starts with four non-chi^2 entries
        # ([t_guess, lg_guess, extinct_guess, rad_guess], mft6.py:939), so column 2 is the test chi^2 of
        # proposal n-3 from line 4 on and a list / scalar guess before that (kept as NaN here).
        rows = []
        for line in open(os.path.join(outdir, 'chisq0.txt')):
            tok = line.split()
            try:
                second = float(' '.join(tok[1:]))
            except ValueError:
                second = np.nan
            rows.append([float(tok[0]), second])
        out['D_chisq'] = np.array(rows)
        # oracle restatement of the two chi^2 kernels of fit_spec on the first proposals
        chi0, flux_n = orc.fit_spec_init(wb * 1e4, sb, eb, [min(wb), max(wb)], start['t'], start['rad'], start['plx'],
                                         frB, specs, ctm_syn, ptm6, tmiB, tmaB, matrix, bandlib=bandlib)
        out['D_init_like'] = np.array([chi0])
        out['D_flux_norm_sub'] = flux_n[::7]
        print('fit_spec: {} evaluated proposals, best chi^2 {:.6g} (initial likelihood chi^2 {:.6g})'.format(
            len(out['D_chisq']), cs, chi0))

        mft6.np = np

        # ------------------------------------------------------------ f3: the grid loader on text files
        os.chdir(tempfile.mkdtemp(prefix='msx_golden_loader_'))  # a directory holding ONLY the text files
        gdir = write_btsettl_files(os.getcwd(), seed=21)
        mft6.pyasl = types.SimpleNamespace(instrBroadGaussFast=lambda wl_, f_, res_, maxsig=5:
                                           orc.instr_broad_gauss_fast(wl_, f_, res_, maxsig=maxsig))
        import io
        import contextlib
        with contextlib.redirect_stdout(io.StringIO()):
            sp_ref = mft6.spec_interpolator([6000.0, 8000.0], [3000, 3200], [4, 5.5], [5000, 9000], resolution=1700)
        sp_orc = orc.spec_interpolator([6000.0, 8000.0], [3000, 3200], [4, 5.5], [5000, 9000], resolution=1700,
                                       grid_dir=gdir)
        assert sorted(sp_ref.keys()) == sorted(sp_orc.keys())
        for k in sp_ref:
            assert np.array_equal(sp_ref[k], sp_orc[k]), k
        out['L_keys'] = np.array(sorted(k for k in sp_ref if k != 'wl'))
        out['L_wl_ends'] = np.array([sp_ref['wl'][0], sp_ref['wl'][-1], len(sp_ref['wl'])])
        out['L_sub'] = np.array([sp_ref[k][::53] for k in out['L_keys']])
        print('loader: {} nodes x {} samples'.format(len(out['L_keys']), len(sp_ref['wl'])))

        # ------------------------------------------------------------ small helpers
        xm, xd, xv = rng.uniform(1, 2, 50), rng.uniform(1, 2, 50), rng.uniform(0.01, 0.02, 50)
        out['chisq_in'] = np.array([xm, xd, xv])
        out['chisq_out'] = mft6.chisq(xm, xd, xv)
        xw = np.linspace(0.55, 0.9, 50)
        out['norm_spec_out'] = mft6.norm_spec(xw, xm, xd)
        assert np.array_equal(out['norm_spec_out'], orc.norm_spec(xw, xm, xd))
    finally:
        os.chdir(cwd)

    np.savez_compressed(os.path.join(HERE, 'golden_reference.npz'), **out)
    sz = os.path.getsize(os.path.join(HERE, 'golden_reference.npz'))
    print('wrote golden_reference.npz ({} arrays, {:.0f} KiB)'.format(len(out), sz / 1024))


if __name__ == '__main__':
    main()
