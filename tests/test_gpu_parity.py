"""GPU parity tests proper: the HIP path (through the C ABI) against the reference-generated golden
vectors and against the CPU oracle on identical inputs.  Tolerance on log-probabilities is the
north-star's 1e-6 relative; what is actually observed is ~1e-12, and the tighter bound 1e-9 is
asserted so regressions in summation order or table staging are caught early."""
import numpy as np
import pytest

import common
from common import golden_case, oracle_loglike, oracle_logpost, rel_err

pytestmark = pytest.mark.gpu

TOL = 1e-6      # north-star bar
TIGHT = 1e-9    # what float64 with re-ordered sums should easily meet


def make_engine(c, rad_prior=False, with_prior=True):
    from mcmc_spec_amd.engine import Engine
    from mcmc_spec_amd import bands
    eng = Engine(0)
    eng.stage_specs(c.specs)
    bl = bands.make_bands(c.tables, *c.vega)
    eng.stage_problem(c.data, c.err, c.fr, c.r, c.ctm, c.ptm, c.tmi, c.tma, c.matrix, nspec=2, bands=bl,
                      av_table=common.av_table_exact() if with_prior else None, tmin=c.tmin, tmax=c.tmax,
                      prior=c.prior if with_prior else 0, rad_prior=rad_prior)
    return eng


@pytest.fixture(scope='module')
def engA():
    return make_engine(golden_case('A'))


@pytest.fixture(scope='module')
def engB():
    return make_engine(golden_case('B'))


@pytest.mark.parametrize('which', ['A', 'B'])
def test_loglikelihood_matches_reference_golden(which, engA, engB):
    c = golden_case(which)
    eng = engA if which == 'A' else engB
    got = eng.loglikelihood(c.theta)
    want = c.g[which + '_loglike']
    e = rel_err(got, want)
    print('case', which, 'max rel err vs reference', e.max())
    assert e.max() < TIGHT
    # walker-at-a-time entry returns a Python float like the reference
    one = eng.loglikelihood(c.theta[0])
    assert isinstance(one, float) and one == got[0]


def test_chisq_mode_matches(engB):
    c = golden_case('B')
    got = engB.loglikelihood(c.theta, optimize=True)
    assert rel_err(got, -2.0 * c.g['B_loglike']).max() < TIGHT


@pytest.mark.parametrize('rad_prior', [False, True])
def test_logposterior_matches_reference_golden(rad_prior):
    c = golden_case('A')
    eng = make_engine(c, rad_prior=rad_prior)
    th = c.g['theta_post']
    got = eng.logposterior(th)
    want = c.g['A_logpost_' + ('radprior' if rad_prior else 'noradprior')]
    assert np.array_equal(np.isinf(got), np.isinf(want))
    assert rel_err(got, want).max() < TIGHT


def test_vs_oracle_random_walkers(engB):
    c = golden_case('B')
    rng = np.random.default_rng(99)
    th = c.theta[0] + rng.normal(size=(64, 6)) * np.array([120, 120, 0.05, 0.05, 0.05, 1e-4])
    th[:, 0:2] = np.clip(th[:, 0:2], 3000.0, 4200.0)
    th[:, 2] = np.abs(th[:, 2])
    th[:, 3:5] = np.clip(th[:, 3:5], 0.05, 1.4)
    got = engB.loglikelihood(th)
    want = np.array([oracle_loglike(c, t) for t in th])
    assert rel_err(got, want).max() < TIGHT


def test_run_to_run_bitwise_deterministic(engB):
    c = golden_case('B')
    a = engB.logposterior(c.theta)
    b = engB.logposterior(c.theta)
    assert np.array_equal(a, b)
    # batch composition must not matter either: same walkers in another order / batch size
    perm = np.random.default_rng(1).permutation(len(c.theta))
    d = engB.logposterior(c.theta[perm])
    assert np.array_equal(d, a[perm])


def test_make_composite_matches_reference_golden(engB):
    from oracle import mft6_oracle as orc
    c = golden_case('B')
    p = c.theta[0]
    lg = [float(orc.get_logg(t, c.matrix)) for t in p[:2]]
    wl, spec, con, pcw, ph = engB.make_composite(p[:2], lg, p[3:5], p[5])
    g = c.g
    assert wl[0] == g['B_mc_wl_ends'][0] and wl[-1] == g['B_mc_wl_ends'][1] and len(wl) == g['B_mc_wl_ends'][2]
    assert rel_err(spec[::211], g['B_mc_spec_sub']).max() < 1e-13
    assert rel_err(con, g['B_mc_contrast']).max() < 1e-11
    assert rel_err(ph, g['B_mc_phot']).max() < 1e-11


def test_error_conventions():
    c = golden_case('B')
    eng = make_engine(c)
    # Teff outside the isochrone table -> interp1d ValueError in the reference (mft6.py:95)
    bad = c.theta[0].copy()
    bad[1] = 2800.0
    with pytest.raises(ValueError):
        eng.loglikelihood(bad)
    # ... but the posterior rejects it in the prior box first (mft6.py:1227) -> -inf, no exception
    assert eng.logposterior(bad) == -np.inf
    # wrong length
    with pytest.raises(ValueError):
        eng.loglikelihood(np.zeros(5))
    # missing node -> KeyError (mft6.py:489-500)
    specs = dict(c.specs)
    del specs['3800, 5.0']
    from mcmc_spec_amd.engine import Engine
    e2 = Engine(0)
    e2.stage_specs(specs)
    e2.stage_problem(c.data, c.err, c.fr, c.r, c.ctm, c.ptm, c.tmi, c.tma, c.matrix, nspec=2,
                     bands=__import__('mcmc_spec_amd.bands', fromlist=['x']).make_bands(c.tables, *c.vega))
    with pytest.raises(KeyError):
        e2.loglikelihood(c.theta[0])
    # non-finite coordinates never produce NaN
    nf = c.theta[0].copy()
    nf[2] = np.nan
    assert eng.logposterior(nf) == -np.inf


def test_broaden_and_ccm89_vs_oracle():
    from oracle import mft6_oracle as orc
    from mcmc_spec_amd._lib import Context
    ctx = Context(0)
    wl = np.arange(6450.0, 8400.0, 0.2)
    rng = np.random.default_rng(5)
    f = 1.0 + 0.3 * np.sin(wl / 7.0) + 0.05 * rng.normal(size=len(wl))
    for R in (1700, 5000, 400):
        got = ctx.broaden(wl, f, R)
        _, want = orc.broaden(wl, f, R)
        assert rel_err(got, want).max() < 1e-12, R
    w = np.concatenate([np.linspace(3050, 30000, 4001), [1e4 / 1.1, 1e4 / 3.3]])
    k = ctx.ccm89_k(w, 3.1)
    assert rel_err(k, orc.ccm89(w, 1.0, 3.1)).max() < 1e-13


def test_median_edge_cases():
    """Odd pixel counts, tiny spectra and heavy duplication exercise every branch of the select."""
    c = golden_case('B')
    from mcmc_spec_amd.engine import Engine
    from mcmc_spec_amd import bands
    bl = bands.make_bands(c.tables, *c.vega)
    eng = Engine(0)
    eng.stage_specs(c.specs)
    for npix in (699, 64, 5):
        sel = np.arange(npix)
        data = [c.data[0][sel], c.data[1][sel]]
        err = c.err[sel]
        r = [min(data[0]), max(data[0])]
        eng.stage_problem(data, err, c.fr, r, c.ctm, c.ptm, c.tmi, c.tma, c.matrix, nspec=2, bands=bl)
        got = eng.loglikelihood(c.theta[:8])
        from oracle import mft6_oracle as orc
        want = np.array([orc.loglikelihood(list(t), c.fr, 2, data, err, r, c.specs, c.ctm, c.ptm, c.tmi, c.tma,
                                           c.matrix, bandlib=c.bandlib) for t in c.theta[:8]])
        assert rel_err(got, want).max() < TIGHT, npix
