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
    th = synth.draw_walkers(4, seed=11, tmin=W['tmin'], tmax=W['tmax'])
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
    assert rel_err(got, want).max() < TIGHT
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
