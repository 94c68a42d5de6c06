#!/usr/bin/env python3
"""Golden vectors for BASELINE config 5: the reference's own ``loglikelihood`` / ``logposterior`` on the eight
KOI spectra of ``/root/reference/Data/koi*.txt``.

Runs only in the build container (it imports ``/root/reference/mft6.py`` the way ``make_golden.py`` does);
what is committed is ``golden_koi.npz`` -- arrays only: the PREPARED data vectors (the reference's own data
preparation, mft6.py:3492-3507: read three columns, crop ``spmin < wl < spmax`` exclusive, divide flux and
error by the median flux) and the reference's outputs for 16 walkers per target.

    python tests/golden/make_golden_koi.py

Two extra cases use the wider crop (0.505, 0.90) um, which keeps the files' backwards wavelength jump
(two overlapping spectrograph arms, pixel 777 of the file, 0.515 -> 0.510 um) inside the fitted window: the
non-monotonic query path of ``interp1d`` (mft6.py:1169-1170) on real data.
"""
import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
sys.dont_write_bytecode = True
warnings.filterwarnings('ignore')

import make_golden as mg  # noqa: E402
from mcmc_spec_amd import params, synth  # noqa: E402
from oracle import mft6_oracle as orc  # noqa: E402

TARGETS = ['koi1422', 'koi1681', 'koi2124', 'koi2174', 'koi227', 'koi2542', 'koi2862', 'koi3010']
WIDE = ['koi1422', 'koi3010']   # also fitted over (0.505, 0.90) um: non-monotonic wavelengths inside the window


def reference_data_prep(path, spmin, spmax):
    """mft6.py:3492, 3502-3507 verbatim in effect (no telluric mask: the KOI files have no param file)."""
    data_wl, dsp, de = np.genfromtxt(path, unpack=True)
    keep = np.where((data_wl > spmin) & (data_wl < spmax))
    data_wl, dsp, de = data_wl[keep], dsp[keep], de[keep]
    de /= np.median(dsp)
    dsp /= np.median(dsp)
    return data_wl, dsp, de


def main():
    mft6 = mg.import_reference()
    teffs = np.arange(3000, 4300, 100)
    loggs = np.array([4.0, 4.5, 5.0, 5.5])
    wl = np.arange(5000, 24000, 0.2)
    flux = synth.make_grid(teffs, loggs, wl, nlines=1500, seed=11)   # the golden grid of make_golden.py
    specs = synth.grid_to_specs(teffs, loggs, wl, flux)
    matrix = synth.make_isochrone_matrix()
    vega_w, vega_f = synth.synthetic_vega()
    bandlib = orc.make_band_library(synth.synthetic_band_tables(), vega_w, vega_f)
    mg.patch_third_party(mft6, bandlib)
    ctm = mg.load_reference_filter_tables()     # the real lp600 / Kp tables, as in dataset A
    ptm = [[], [], [], []]
    fr = [synth.EXAMPLE_CMAG, synth.EXAMPLE_CERR, np.array(['lp600', 'Kp']), np.zeros(0), [], np.array([])]
    tmi = min(min(w) for w in ctm[0])
    tma = max(max(w) for w in ctm[0])
    prior = [*np.zeros(10), 2.0732e-3, 0.0277e-3]   # mft6.py:3689

    g0 = np.load(os.path.join(HERE, 'golden_reference.npz'))
    theta = g0['theta'][:16]                       # 6 edge cases + 10 random walkers, all inside the prior box
    out = {'theta': theta, 'targets': np.array(TARGETS), 'wide_targets': np.array(WIDE)}

    cwd = os.getcwd()
    os.chdir(mg.scratch_grid_dir(teffs, loggs))
    try:
        cases = [(t, 0.55, 0.90, t) for t in TARGETS] + [(t, 0.505, 0.90, t + '_wide') for t in WIDE]
        for name, spmin, spmax, tag in cases:
            path = os.path.join(mg.REF, 'Data', name + '.txt')
            dw, ds, de = reference_data_prep(path, spmin, spmax)
            # the build's data-prep module must agree bit for bit with the reference expressions
            pw, ps, pe = params.prepare_data(path, spmin, spmax, mask=False)
            assert np.array_equal(pw, dw) and np.array_equal(ps, ds) and np.array_equal(pe, de), name
            r = [min(dw), max(dw)]
            ll, po = [], []
            for p in theta:
                v = mft6.loglikelihood(list(p), fr, 2, 0, [dw, ds], de, 1700, r, specs, ctm, ptm, tmi, tma, None, matrix)
                o = orc.loglikelihood(list(p), fr, 2, [dw, ds], de, r, specs, ctm, ptm, tmi, tma, matrix, bandlib=bandlib)
                assert v == o or abs(v - o) <= 1e-13 * abs(v), (name, v, o)
                ll.append(v)
                q = mft6.logposterior(list(p), fr, 2, 0, [dw, ds], de, 1700, r, specs, ctm, ptm, tmi, tma, None,
                                      3000.0, 4200.0, matrix, 10.0, 20.0, prior=prior, rad_prior=True)
                oo = orc.logposterior(list(p), fr, 2, [dw, ds], de, r, specs, ctm, ptm, tmi, tma, 3000.0, 4200.0,
                                      matrix, mg.av_prior, prior=prior, rad_prior=True, bandlib=bandlib)
                assert q == oo or abs(q - oo) <= 1e-13 * abs(q), (name, q, oo)
                po.append(q)
            out[tag + '_wl'], out[tag + '_flux'], out[tag + '_err'] = dw, ds, de
            out[tag + '_loglike'], out[tag + '_logpost'] = np.array(ll), np.array(po)
            nonmono = int(np.sum(np.diff(dw) <= 0))
            print('{:14s} {:5d} px  backwards steps {}  loglike[0] {:.6e}  logpost[0] {:.6e}'.format(
                tag, len(dw), nonmono, ll[0], po[0]))
    finally:
        os.chdir(cwd)
    path = os.path.join(HERE, 'golden_koi.npz')
    np.savez_compressed(path, **out)
    print('wrote golden_koi.npz ({} arrays, {:.0f} KiB)'.format(len(out), os.path.getsize(path) / 1024))


if __name__ == '__main__':
    main()
