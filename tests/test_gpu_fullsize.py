"""GPU: BASELINE.json's full sizes (config 2: 4096 px + 2 contrasts; config 4: 16384 px + 6-band photometry)
on the full 26x4x135,000 synthetic grid.  A handful of walkers are checked against the oracle (which also
re-does the per-node broadening on the CPU); the rest of the ensemble is covered by size-independent
properties of the path."""
import os
import sys

import numpy as np
import pytest

import common
from common import rel_err

sys.path.insert(0, common.ROOT)
pytestmark = pytest.mark.gpu

TOL = 1e-6
TIGHT = 1e-9


@pytest.fixture(scope='module', params=[(4096, False), (16384, True)], ids=['config2', 'config4'])
def work(request):
    from bench import build_workload
    from mcmc_spec_amd.engine import Engine
    npix, phot = request.param
    eng = Engine(0)
    W = build_workload(eng, npix, phot, keep_host_grid=True)
    return eng, W, npix, phot


def test_full_size_matches_oracle_and_cpu_broadening(work):
    from mcmc_spec_amd import synth
    from oracle import mft6_oracle as orc
    eng, W, npix, phot = work
    # 64 walkers of the benched ensemble against the oracle (4 ms each), among them the corners of the path:
    th = synth.draw_walkers(64, seed=11, tmin=W['tmin'], tmax=W['tmax'])
    th[1, 2] = 1e-12                    # A_V -> 0+ : reddened with a factor that rounds to 1 (mft6.py:1161 takes the branch)
    th[2, 2] = 0.0                      # A_V = 0 exactly: the unreddened branch
    th[3, 0] = 3800.0                   # Teff on a grid node: both Teff brackets are that node (mft6.py:441-443)
    th[4, 1] = 3000.0                   # ... the FIRST node, where the prior's box ends
    th[5, 0], th[5, 1] = 3900.0, 3100.0  # both stars on nodes
    th[6, 0] = 3849.999999999           # a hair below the midpoint between two nodes / (7) above it: nearest node flips
    th[7, 0] = 3850.000000001
    th[8, 3] = 0.04                     # a walker the prior box rejects (R1 < 0.05, mft6.py:1227): -inf, its neighbours untouched
    th[9, 5] = 1.0                      # ... and one by the parallax bound
    got = eng.logposterior(th)
    specs = synth.grid_to_specs(W['teffs'], W['loggs'], W['wl'], W['flux'])
    specs = orc.broaden_specs_window(specs, W['win'], W['resolution'])
    edges, mu, sig = synth.make_av_table()

    def avp(d):
        b = int(np.clip(np.searchsorted(edges, d, side='right') - 1, 0, len(mu) - 1))
        return mu[b], sig[b]

    bl = orc.make_band_library(W['tabs'], *W['vega'])
    want = np.array([orc.logposterior(list(t), W['fr'], 2, W['data'], W['err'], W['r'], specs, W['ctm'], W['ptm'],
                                      W['tmi'], W['tma'], W['tmin'], W['tmax'], W['matrix'], avp, prior=W['prior'],
                                      bandlib=bl) for t in th])
    assert np.isneginf(want[8]) and np.isneginf(want[9]) and np.array_equal(np.isneginf(got), np.isneginf(want))
    fin = np.isfinite(want)
    assert fin.sum() == 62 and rel_err(got[fin], want[fin]).max() < TIGHT
    # the likelihood alone (no prior terms to dominate the sum) holds the same bar
    ll = eng.loglikelihood(th[fin][:16])
    want_ll = np.array([orc.loglikelihood(list(t), W['fr'], 2, W['data'], W['err'], W['r'], specs, W['ctm'], W['ptm'], W['tmi'],
                                          W['tma'], W['matrix'], bandlib=bl) for t in th[fin][:16]])
    assert rel_err(ll, want_ll).max() < TIGHT
    # the device-broadened node equals the CPU restatement of mft6.py:366-378 (incl. edge patches)
    node = eng.ctx.read_node(8, 2)
    assert rel_err(node, specs['{}, {}'.format(int(W['teffs'][8]), float(W['loggs'][2]))]).max() < 1e-12


def test_size_independent_properties(work):
    from mcmc_spec_amd import synth
    eng, W, npix, phot = work
    n = 2048 if npix == 4096 else 1024  # BASELINE configs 3 / 4 ensemble sizes
    th = synth.draw_walkers(n, seed=5, tmin=W['tmin'], tmax=W['tmax'])
    post = eng.logposterior(th)
    assert np.all(np.isfinite(post))
    # (1) posterior = prior + likelihood, walker by walker
    assert rel_err(post, eng.logprior(th) + eng.loglikelihood(th)).max() < 1e-12
    # (2) chi^2 mode is -2 x log-likelihood
    assert rel_err(eng.loglikelihood(th, optimize=True), -2.0 * eng.loglikelihood(th)).max() < 1e-15
    # (3) a walker's value does not depend on its batch: permutation + sub-batch give identical bits
    perm = np.random.default_rng(0).permutation(n)
    assert np.array_equal(eng.logposterior(th[perm]), post[perm])
    assert np.array_equal(eng.logposterior(th[:100]), post[:100])
    # (4) outside the prior box -> -inf, inside -> finite, regardless of batch neighbours
    mixed = th[:8].copy()
    mixed[3, 0] = W['tmax'] + 1.0
    mixed[5, 2] = -1e-9
    out = eng.logposterior(mixed)
    assert np.isneginf(out[3]) and np.isneginf(out[5]) and np.array_equal(np.delete(out, [3, 5]), np.delete(post[:8], [3, 5]))
    # (5) flux-scale invariance: median scaling + continuum fit make chi^2 invariant to data,err -> c*data,c*err
    c = 3.7
    ll0 = eng.loglikelihood(th[:64])
    kw = dict(nspec=2, bands=__import__('mcmc_spec_amd.bands', fromlist=['x']).make_bands(W['tabs'], *W['vega']))
    eng.stage_problem([W['data'][0], c * W['data'][1]], c * W['err'], W['fr'], W['r'], W['ctm'], W['ptm'], W['tmi'],
                      W['tma'], W['matrix'], **kw)
    ll1 = eng.loglikelihood(th[:64])
    assert rel_err(ll1, ll0).max() < 1e-10
    # restore for the other tests of this module
    eng.stage_problem(W['data'], W['err'], W['fr'], W['r'], W['ctm'], W['ptm'], W['tmi'], W['tma'], W['matrix'],
                      av_table=synth.make_av_table(), tmin=W['tmin'], tmax=W['tmax'], prior=W['prior'], **kw)


def test_broadening_is_linear_and_preserves_flat_spectra():
    from mcmc_spec_amd._lib import Context
    ctx = Context(0)
    wl = np.arange(5500.0, 19303.0, 0.2)  # config-4 window: 69,015 samples, 155 taps
    rng = np.random.default_rng(1)
    f, g = rng.uniform(1, 2, len(wl)), rng.uniform(1, 2, len(wl))
    bf, bg, bfg = ctx.broaden(wl, f, 1700), ctx.broaden(wl, g, 1700), ctx.broaden(wl, 2.5 * f - 0.5 * g, 1700)
    assert np.max(np.abs(bfg - (2.5 * bf - 0.5 * bg))) < 1e-12
    flat = ctx.broaden(wl, np.ones_like(wl), 1700)
    assert np.max(np.abs(flat[500:-500] - 1.0)) < 1e-13
    assert np.all(flat[:5] == flat[5]) and np.all(flat[-10:] == flat[-11])  # mft6.py:129-130


def test_device_ccm89_matches_the_papers_table3():
    """ccm89_kernel (A7: the staged extinction curve) against Cardelli, Clayton & Mathis 1989, Table 3 -- the same rows
    tests/test_oracle_golden.py holds the oracle to."""
    from mcmc_spec_amd._lib import Context
    from test_oracle_golden import check_ccm89_against_table3
    ctx = Context(0)
    check_ccm89_against_table3(lambda wl, rv: ctx.ccm89_k(wl, rv))


def test_device_broadening_of_a_gaussian_line_is_the_analytic_convolution():
    """broaden_conv_kernel + broaden_patch_kernel (A3) on the one case with a closed form: a Gaussian line of width s
    comes out with width sqrt(s^2 + sigma^2) and its equivalent width (tests/test_oracle_golden.py: the oracle likewise)."""
    from mcmc_spec_amd._lib import Context
    from test_oracle_golden import check_broadened_gaussian_line
    ctx = Context(0)
    check_broadened_gaussian_line(lambda wl, f, r: ctx.broaden(wl, f, r))
