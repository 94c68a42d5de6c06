"""CPU: host-side staging logic (index/weight tables) against NumPy/SciPy and the oracle."""
import warnings

import numpy as np
import pytest
from scipy.interpolate import interp1d

import common
from common import golden_case
from mcmc_spec_amd import bands, staging, synth
from oracle import mft6_oracle as orc

warnings.filterwarnings('ignore')


def test_resample_tables_reproduce_interp1d():
    rng = np.random.default_rng(0)
    wave = np.arange(5000, 6000, 0.2)
    y = rng.uniform(1, 2, len(wave))
    x = np.concatenate([rng.uniform(wave[0], wave[-1], 500), [wave[0], wave[-1], wave[17], wave[-2]]])
    lo, t = staging.resample_tables(wave, x)
    got = y[lo] + (y[lo + 1] - y[lo]) * t
    want = interp1d(wave, y)(x)
    assert np.max(np.abs(got - want) / want) < 1e-14
    assert got[-4] == y[0] and got[-3] == y[-1] and got[-2] == y[17]
    with pytest.raises(ValueError):
        staging.resample_tables(wave, np.array([4999.9]))
    with pytest.raises(ValueError):
        staging.resample_tables(wave, np.array([6000.0]))


def test_window_slice_matches_get_spec_crop():
    c = golden_case('B')
    reg = orc.composite_window(c.r, c.tmi, c.tma, c.ctm, c.ptm)
    assert reg == staging.composite_window_um(c.r, c.tmi, c.tma, c.ctm, c.ptm)
    j0, n = staging.window_slice(c.wl, reg)
    w, _ = orc.get_spec(3800.0, 4.5, reg, c.specs)
    assert n == len(w) and np.array_equal(c.wl[j0:j0 + n], w)


def test_contrast_weights_equal_trapz_incl_unsorted_filter_tables():
    c = golden_case('A')  # real lp600 / keck_kp tables: unsorted, repeated abscissa
    reg = orc.composite_window(c.r, c.tmi, c.tma, c.ctm, c.ptm)
    j0, n = staging.window_slice(c.wl, reg)
    wave = c.wl[j0:j0 + n]
    s = c.flux[5, 2, j0:j0 + n]
    for f in range(2):
        ran, tm = c.ctm[0][f], c.ctm[1][f]
        i0, w = staging.contrast_weights(wave, ran, tm)
        inband = np.where((wave <= max(ran)) & (wave >= min(ran)))
        want = np.trapz(s[inband] * interp1d(ran, tm)(wave[inband]), wave[inband])
        got = np.sum(w * s[i0:i0 + len(w)])
        assert abs(got - want) / want < 1e-13


def test_band_library_matches_oracle_band_flux_and_zero_points():
    c = golden_case('B')
    bl = bands.make_bands(c.tables, *c.vega)
    reg = orc.composite_window(c.r, c.tmi, c.tma, c.ctm, c.ptm)
    j0, n = staging.window_slice(c.wl, reg)
    wave, s = c.wl[j0:j0 + n], c.flux[3, 1, j0:j0 + n]
    for name in bands.BAND_NAMES_6:
        ob = c.bandlib[name]
        i0, w = bl[name].weights_on(wave)
        got = np.sum(w * s[i0:i0 + len(w)])
        assert abs(got - ob.get_flux(wave, s)) / got < 1e-13
        zero = ob.Vega_zero_flux if '2MASS' in name else ob.AB_zero_flux
        assert abs(bl[name].zero_flux - zero) / zero < 1e-13


def test_prior_vectors_and_isochrone_order():
    prior = [*np.zeros(10), 2.0732e-3, 0.0277e-3]
    mean, sig, has = staging.prior_vectors(prior, 2)
    assert has == 1 and mean[5] == 2.0732e-3 and sig[5] == 0.0277e-3 and not mean[:5].any()
    assert staging.prior_vectors(0, 2)[2] == 0
    full = [3850, 3025, 50, 60, 0.1, 0.02, 0.5, 0.3, 0.05, 0.06, 2e-3, 1e-5]
    mean, sig, _ = staging.prior_vectors(full, 2)
    assert list(mean[:6]) == [3850, 3025, 0.1, 0.5, 0.3, 2e-3] and list(sig[:6]) == [50, 60, 0.02, 0.05, 0.06, 1e-5]
    mean, sig, _ = staging.prior_vectors(full, 2, dist_fit=False)   # mft6.py:1303-1312: one radius entry, no parallax
    assert list(mean[:6]) == [3850, 3025, 0.1, 0.5, 0, 0] and list(sig[:4]) == [50, 60, 0.02, 0.05]
    m = synth.make_isochrone_matrix()
    x, g, l = staging.sorted_isochrone(m)
    assert len(x) == 220 and np.all(np.diff(x) > 0) and x[0] == 2900.0 and x[-1] == 6500.0
    assert staging.isochrone_logg(3850.0, m) == float(orc.get_logg(3850.0, m))


def test_parse_specs_round_trip_and_missing_nodes():
    c = golden_case('B')
    specs = dict(c.specs)
    del specs['3100, 5.5']
    teff, logg, wl, flux, present = staging.parse_specs(specs)
    assert list(teff) == list(c.teffs) and list(logg) == list(c.loggs)
    assert present.sum() == present.size - 1 and present[1, 3] == 0
    assert np.array_equal(flux[4, 2], c.flux[4, 2])


def test_synthetic_generators_are_deterministic():
    wl = np.arange(6000, 6100, 0.2)
    a = synth.make_grid([3000, 3100], [4.5, 5.0], wl, nlines=50, seed=3)
    b = synth.make_grid([3000, 3100], [4.5, 5.0], wl, nlines=50, seed=3)
    assert np.array_equal(a, b) and np.all(a > 0)
    assert np.array_equal(synth.draw_walkers(8), synth.draw_walkers(8))
    th = synth.draw_walkers(512)
    assert th[:, 0].min() >= 3000 and th[:, 3].min() >= 0.05 and th[:, 5].max() <= 0.25


def test_from_pyphot_adapts_a_library_object():
    """bands.from_pyphot against a stand-in with pyphot's attribute shapes: quantities carrying `.magnitude`
    (pint-style) or `.value` (astropy-style), plain arrays, Vega zero points for 2MASS and AB for SDSS
    (mft6.py:778-782).  pyphot itself is not installed here: the adapter's own logic is what is pinned."""
    from mcmc_spec_amd import bands

    class Q:
        def __init__(self, v, attr):
            setattr(self, attr, v)

    class F:
        def __init__(self, name, k):
            self.name = name
            w = np.linspace(5000.0 + 100 * k, 6000.0 + 100 * k, 11)
            self.wavelength = Q(w, 'magnitude') if k % 2 else w
            self.transmit = list(np.exp(-0.5 * ((w - w.mean()) / 300.0) ** 2))
            self.Vega_zero_flux = Q(1e-9 * (k + 1), 'magnitude') if k % 2 else Q(1e-9 * (k + 1), 'value')
            self.AB_zero_flux = 2e-9 * (k + 1)

    lib = {n: F(n, k) for k, n in enumerate(bands.BAND_NAMES_6)}
    out = bands.from_pyphot(lib)
    assert list(out) == bands.BAND_NAMES_6
    for k, n in enumerate(bands.BAND_NAMES_6):
        b = out[n]
        assert isinstance(b, bands.Band) and b.wavelength.dtype == float and b.wavelength.shape == (11,)
        assert b.transmit.shape == (11,) and b.transmit.max() == 1.0
        want = 1e-9 * (k + 1) if '2MASS' in n else 2e-9 * (k + 1)
        assert b.zero_flux == want
    three = bands.from_pyphot(lib, bands.BAND_NAMES_3)
    assert list(three) == bands.BAND_NAMES_3
    # the adapted band integrates like any other: a flat spectrum has mean flux 1
    wave = np.arange(4000.0, 8000.0, 1.0)
    assert abs(out['SDSS_r'].mean_flux(wave, np.ones_like(wave)) - 1.0) < 1e-12
