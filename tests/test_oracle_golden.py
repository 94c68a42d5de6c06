"""CPU: the oracle (oracle/mft6_oracle.py) against the golden vectors produced by the reference's own
functions (tests/golden/make_golden.py ran them in the build container).  These pin the oracle."""
import warnings

import numpy as np
import pytest

import common
from common import golden_case, oracle_loglike, oracle_logpost, rel_err
from oracle import mft6_oracle as orc

warnings.filterwarnings('ignore')


def test_a1_isochrone_lookups_bitwise():
    c = golden_case('A')
    g = c.g
    assert np.array_equal([float(orc.get_logg(t, c.matrix)) for t in g['a1_teff']], g['a1_logg'])
    assert np.array_equal([float(orc.get_radius(t, c.matrix)) for t in g['a1_teff']], g['a1_radius'])
    with pytest.raises(ValueError):
        orc.get_logg(2899.0, c.matrix)


def test_a2_get_spec_bitwise_incl_on_node_and_tie_cases():
    c = golden_case('A')
    g = c.g
    for (t, lg), want in zip(g['a2_cases'], g['a2_flux_sub']):
        w, s = orc.get_spec(float(t), float(lg), [0.55, 0.90], c.specs)
        assert len(w) == int(g['a2_npts'][0])
        assert np.array_equal(s[::97], want), (t, lg)


@pytest.mark.parametrize('which', ['A', 'B'])
def test_loglikelihood_bitwise(which):
    c = golden_case(which)
    g = c.g
    idx = list(g[which + '_model_idx'])
    for i, th in enumerate(c.theta[:12]):
        parts = {}
        v = oracle_loglike(c, th, parts)
        assert v == g[which + '_loglike'][i]
        assert np.array_equal(parts['contrast'], g[which + '_contrast'][i])
        assert np.array_equal(parts['phot'], g[which + '_phot'][i])
        assert parts['iic'] == g[which + '_iic'][i]
        if i in idx:
            assert np.array_equal(parts['model'], g[which + '_model'][idx.index(i)])


@pytest.mark.parametrize('rad_prior', [False, True])
def test_logprior_logposterior(rad_prior):
    c = golden_case('A')
    tag = 'radprior' if rad_prior else 'noradprior'
    th = c.g['theta_post']
    lp = np.array([orc.logprior(list(t), 2, c.tmin, c.tmax, c.matrix, common.av_prior, prior=c.prior,
                                rad_prior=rad_prior) for t in th])
    want = c.g['A_logprior_' + tag]
    assert np.array_equal(np.isinf(lp), np.isinf(want))
    assert rel_err(lp, want).max() < 1e-13
    # every rejection case of the box (mft6.py:1227-1230) is in the last 8 rows
    assert np.all(np.isinf(want[-8:]))
    sel = [0, 3, len(th) - 1, len(th) - 5]
    po = np.array([oracle_logpost(c, th[i], rad_prior) for i in sel])
    assert rel_err(po, c.g['A_logpost_' + tag][sel]).max() < 1e-13


def test_chisq_and_norm_spec_bitwise():
    g = golden_case('A').g
    m, d, v = g['chisq_in']
    assert np.array_equal(orc.chisq(m, d, v), g['chisq_out'])
    assert np.array_equal(orc.norm_spec(np.linspace(0.55, 0.9, 50), m, d), g['norm_spec_out'])


def test_make_composite_reference_pieces():
    c = golden_case('B')
    p = c.theta[0]
    lg = [orc.get_logg(t, c.matrix) for t in p[:2]]
    w, s, con, pcw, ph, _ = orc.make_composite(p[:2], lg, p[3:5], p[5], c.fr[2], c.fr[5], c.r, c.specs, c.ctm, c.ptm,
                                               c.tmi, c.tma, bandlib=c.bandlib)
    g = c.g
    assert [w[0], w[-1], len(w)] == list(g['B_mc_wl_ends'])
    assert np.array_equal(s[::211], g['B_mc_spec_sub'])
    assert np.array_equal(con, g['B_mc_contrast'])
    assert np.array_equal(ph, g['B_mc_phot'])


def test_broadening_kernel_properties():
    """A3 is restated from PyAstronomy's published algorithm (parity unpinned): check the properties
    the algorithm guarantees -- unit area, symmetry for odd length, flat spectrum preserved inside."""
    wl = np.arange(6450.0, 8400.0, 0.2)
    sigma = np.mean(wl) / 1700 / (2 * np.sqrt(2 * np.log(2)))
    e = orc.broad_gauss_kernel(0.2, sigma, 5)
    assert len(e) == int(sigma * 5 / 0.2 * 2) + 1 and abs(e.sum() - 1) < 1e-15
    if len(e) % 2:
        assert np.allclose(e, e[::-1], rtol=0, atol=1e-18)
    _, b = orc.broaden(wl, np.ones_like(wl), 1700)
    assert np.allclose(b[200:-200], 1.0, atol=1e-14)
    assert np.all(b[:5] == b[5]) and np.all(b[-10:] == b[-11])  # mft6.py:129-130
    with pytest.raises(ValueError):
        orc.instr_broad_gauss_fast(wl**1.01, np.ones_like(wl), 1700)


def test_ccm89_known_values():
    """A_V normalisation: a(x)+b(x)/R_V = 1 at x = 1.82 (V band) by construction of CCM89; IR power law."""
    assert abs(orc.ccm89(np.array([1e4 / 1.82]), 1.0, 3.1)[0] - 1.0) < 1e-12
    k = orc.ccm89(np.array([20000.0]), 1.0, 3.1)[0]
    assert abs(k - (0.574 - 0.527 / 3.1) * 0.5**1.61) < 1e-15
    f = orc.extinct(np.array([5500.0, 8000.0]), np.array([1.0, 1.0]), 0.5)
    assert np.all(f < 1) and f[0] < f[1]


# Cardelli, Clayton & Mathis 1989, Table 3: the standard bands' x (1 / micron), a(x), b(x) and A(lambda) / A(V) at
# R_V = 3.1 -- the one published set of numbers for the law `extinction.ccm89` evaluates (mft6.py:62-63).  The B row of
# the table is inconsistent with the paper's own polynomial (1.337 against 1.3226: known); H, K, L are the table's
# rounding of the IR power law.  tests/test_gpu_parity.py holds the device's ccm89_kernel to the same rows.
CCM89_TABLE3 = {  # band: (x, a, b, A/A_V)
    'U': (2.78, 0.9530, 1.9090, 1.569), 'B': (2.27, 0.9982, 1.0495, 1.337), 'V': (1.82, 1.0000, 0.0000, 1.000),
    'R': (1.43, 0.8686, -0.3660, 0.751), 'I': (1.11, 0.6800, -0.6239, 0.479), 'J': (0.80, 0.4008, -0.3679, 0.282),
    'H': (0.63, 0.2693, -0.2473, 0.190), 'K': (0.46, 0.1615, -0.1483, 0.114), 'L': (0.29, 0.0800, -0.0734, 0.056),
}


def check_ccm89_against_table3(k_of_wl_rv):
    """`k_of_wl_rv(wl_angstrom, r_v)` -> a + b / r_v.  A mistyped polynomial coefficient fails here: the optical
    polynomials are sampled at y = x - 1.82 = 0.96, 0.45, 0, -0.39, -0.71 and the IR power law at three more points."""
    for band, (x, a, b, k31) in CCM89_TABLE3.items():
        wl = np.array([1e4 / x])
        k1, k2 = float(k_of_wl_rv(wl, 3.1)[0]), float(k_of_wl_rv(wl, 5.0)[0])
        tol = 2e-2 if band == 'B' else 3e-3
        assert abs(k1 - k31) < tol, (band, k1, k31)
        bb = (k1 - k2) / (1 / 3.1 - 1 / 5.0)      # a + b / R_V at two R_V pins a and b separately
        aa = k1 - bb / 3.1
        if band in 'UVRIJ':                        # (rows whose a, b columns the paper's formulas reproduce to the table's digits)
            assert abs(aa - a) < 1.5e-3 and abs(bb - b) < 1.5e-3, (band, aa, a, bb, b)
        else:
            assert abs(aa - a) < (5e-3 if band != 'B' else 2e-2) and abs(bb - b) < (5e-3 if band != 'B' else 6e-2), (band, aa, a, bb, b)


def test_ccm89_matches_the_papers_table3():
    check_ccm89_against_table3(lambda wl, rv: orc.ccm89(wl, 1.0, rv))
    # continuity of the law where its branches meet (x = 1.1: IR power law | optical polynomial)
    lo, hi = orc.ccm89(np.array([1e4 / (1.1 - 1e-9)]), 1.0, 3.1)[0], orc.ccm89(np.array([1e4 / (1.1 + 1e-9)]), 1.0, 3.1)[0]
    assert abs(lo - hi) < 2e-3


def gaussian_line_case(resolution=1700.0, s=1.4, depth=0.55):
    """A Gaussian absorption line of width s on a flat continuum, and what a Gaussian instrumental profile of
    sigma = mean(wl) / R / (2 sqrt(2 ln 2)) must turn it into: the same line with width sqrt(s^2 + sigma^2) and its
    equivalent width preserved (depth x s / sqrt(s^2 + sigma^2)) -- the analytic convolution, exact up to the kernel's
    truncation at 5 sigma (erfc(5 / sqrt 2) = 5.7e-7) and the 0.2 A sampling (s / dx = 7: negligible)."""
    wl = np.arange(6400.0, 8400.0, 0.2)
    l0 = 7400.0
    flux = 1.0 - depth * np.exp(-0.5 * ((wl - l0) / s) ** 2)
    sigma = np.mean(wl) / resolution / (2 * np.sqrt(2 * np.log(2)))
    w = np.hypot(s, sigma)
    want = 1.0 - depth * (s / w) * np.exp(-0.5 * ((wl - l0) / w) ** 2)
    return wl, flux, want, sigma


def check_broadened_gaussian_line(broaden_fn):
    wl, flux, want, sigma = gaussian_line_case()
    got = np.asarray(broaden_fn(wl, flux, 1700.0))
    mid = slice(500, -500)                                  # away from the zero-padded, patched edges (mft6.py:129-130)
    assert np.max(np.abs(got[mid] - want[mid])) < 5e-6      # of a line 0.33 deep after broadening
    # second moment of the broadened line = s^2 + sigma^2 (what "Gaussian of width sigma" means), to 1e-4
    d = 1.0 - got[mid]
    x = wl[mid] - 7400.0
    assert abs(np.sum(d * x * x) / np.sum(d) - (1.4 ** 2 + sigma ** 2)) < 1e-4 * (1.4 ** 2 + sigma ** 2)
    # ... and the equivalent width is the unbroadened line's
    assert abs(np.sum(d) - np.sum(1.0 - flux[mid])) < 1e-5 * np.sum(d)


def test_broadening_of_a_gaussian_line_is_the_analytic_convolution():
    """pyasl.instrBroadGaussFast restated (mft6.py:124-152): pinned to the one case with a closed form."""
    check_broadened_gaussian_line(lambda wl, f, r: orc.broaden(wl, f, r)[1])


def test_triple_system_ndim8():
    c = golden_case('C')
    g = c.g
    for i, th in enumerate(c.theta):
        lp = orc.logprior(list(th), 3, c.tmin, c.tmax, c.matrix, common.av_prior, prior=c.prior, rad_prior=True)
        assert lp == g['C_logprior'][i] or abs(lp - g['C_logprior'][i]) < 1e-13 * abs(lp)
    for i in (0, 5, 13, 15):
        po = oracle_logpost(c, c.theta[i], rad_prior=True)
        want = g['C_logpost'][i]
        assert (po == want) or abs(po - want) < 1e-13 * abs(want)
    assert np.all(np.isinf(g['C_logpost'][-4:]))


def test_fit_spec_pieces_against_reference_trajectory():
    """f4: the oracle's restatement of fit_spec's two chi^2 computations reproduces the reference's own
    numbers: initial chi^2 (first savechi minus the opt_prior terms is checked through the total) and the
    test chi^2 of the first accepted proposal."""
    c = golden_case('B')
    g = c.g
    st = g['D_start']
    chi0, flux_n = orc.fit_spec_init(c.data[0] * 1e4, c.data[1], c.err, c.r, st[:2], st[3:5], st[5], c.fr, c.specs,
                                     c.ctm, c.ptm, c.tmi, c.tma, c.matrix, bandlib=c.bandlib)
    assert chi0 == g['D_init_like'][0] and np.array_equal(flux_n[::7], g['D_flux_norm_sub'])
    # proposal 1 of the reference run = params row 0 when accepted; rebuild its total with the prior terms
    row = g['D_params'][2]  # first row that differs from the start = an accepted proposal
    like = orc.fit_spec_proposal(c.data[0] * 1e4, flux_n, c.err, c.r, row[:2], row[2], row[3:5], row[5], c.fr, c.specs,
                                 c.ctm, c.ptm, c.tmi, c.tma, c.matrix, bandlib=c.bandlib)
    mu, sg = common.av_prior(1.0 / row[5])
    mr = [float(orc.get_radius(t, c.matrix)) for t in row[:2]]
    si_rad = [0.1 * r for r in st[3:5]]  # coarse radius steps are the sigma of the radius prior (mft6.py:1042)
    total = like + ((row[2] - mu) / sg) ** 2 + ((row[5] - 2.0732e-3) / 0.0277e-3) ** 2 \
        + ((row[3] - mr[0]) / si_rad[0]) ** 2 + ((row[4] - mr[1] / mr[0]) / si_rad[1]) ** 2
    assert abs(total - g['D_chisq'][2, 0]) < 1e-9 * total


@pytest.mark.parametrize('rad_prior', [False, True])
def test_dist_fit_false_prior(rad_prior):
    c = golden_case('A')
    g = c.g
    tag = 'radprior' if rad_prior else 'noradprior'
    lp = np.array([orc.logprior(list(t), 2, c.tmin, c.tmax, c.matrix, common.av_prior, prior=list(g['prior_nodist']),
                                dist_fit=False, rad_prior=rad_prior) for t in g['theta_nodist']])
    want = g['A_nodist_logprior_' + tag]
    assert np.array_equal(np.isinf(lp), np.isinf(want)) and rel_err(lp, want).max() < 1e-13


@pytest.mark.parametrize('rad_prior', [False, True])
def test_triple_dist_fit_false_prior_and_posterior(rad_prior):
    """ndim 8, dist_fit=False (mft6.py:1397-1455): the oracle against the reference's logprior on every walker and
    its logposterior on a few (a posterior costs a full spectrum synthesis on the CPU)."""
    c = golden_case('C')
    g = c.g
    tag = 'radprior' if rad_prior else 'noradprior'
    prior = list(g['prior3_nodist'])
    th = g['theta3_nodist']
    lp = np.array([orc.logprior(list(t), 3, c.tmin, c.tmax, c.matrix, common.av_prior, prior=prior, dist_fit=False,
                                rad_prior=rad_prior) for t in th])
    want = g['C_nodist_logprior_' + tag]
    assert np.array_equal(np.isinf(lp), np.isinf(want)) and rel_err(lp, want).max() < 1e-13
    assert np.isfinite(want[10]) and np.isfinite(want[11]) and np.all(np.isinf(want[12:]))
    for i in (0, 10, 11, 13):
        po = orc.logposterior(list(th[i]), c.fr, 3, c.data, c.err, c.r, c.specs, c.ctm, c.ptm, c.tmi, c.tma, c.tmin,
                              c.tmax, c.matrix, common.av_prior, prior=prior, dist_fit=False, rad_prior=rad_prior,
                              bandlib=c.bandlib)
        w = g['C_nodist_logpost_' + tag][i]
        assert (po == w) or abs(po - w) < 1e-13 * abs(w)


def test_nospec_variant():
    c = golden_case('B')
    got = [orc.loglikelihood(list(t), c.fr, 2, c.data, c.err, c.r, c.specs, c.ctm, c.ptm, c.tmi, c.tma, c.matrix,
                             bandlib=c.bandlib, spectrum=False) for t in c.theta[:12]]
    assert np.array_equal(got, c.g['B_nospec_loglike'])
