"""Overlapped half-steps of the device-resident sampler (include/msx.h: msx_sampler_overlapped; logprob_kernel.h,
walker_done; msx.hip, chunk_half_eval).

Consecutive half-steps run concurrently on two streams; what orders them is inside the kernel: a walker's workgroup
waits until the two walkers its move reads hold the versions the move is defined on, coordinates are double-buffered by
version parity, every finished walker publishes its version.  None of that may change a single bit of the chain -- the
host-driven loop over the same randomness is the reference -- and a walker that never publishes must end in a loud
MSX_W_HANDOVER, not in a hang or a value.
"""
import os

import numpy as np
import pytest

import common
from common import golden_case
from test_gpu_parity import make_engine

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(os.environ.get('MSX_SMP_OVERLAP', '') == '0', reason='overlap switched off in the environment')]


def _config2():
    import bench
    from mcmc_spec_amd.engine import Engine
    if 'config2' not in common._cache:
        eng = Engine(0)
        common._cache['config2'] = (eng, bench.build_workload(eng, 4096, False))
    return common._cache['config2']


def _chains(eng, p0, nsteps, seed, chunk, mode='logposterior'):
    from mcmc_spec_amd.sampler import DeviceEnsembleSampler, EnsembleSampler
    fn = eng.logposterior if mode == 'logposterior' else eng.loglikelihood
    host = EnsembleSampler(p0.shape[0], p0.shape[1], fn, vectorize=True, seed=seed)
    hs = host.run_mcmc(p0, nsteps)
    dev = DeviceEnsembleSampler(p0.shape[0], p0.shape[1], eng, mode=mode, seed=seed, chunk=chunk)
    ds = dev.run_mcmc(p0, nsteps)
    return host, hs, dev, ds


@pytest.mark.parametrize('nw,nsteps,chunk', [(256, 40, 16), (64, 27, 8), (12, 9, 4)])
def test_overlapped_half_steps_walk_the_host_chain(nw, nsteps, chunk, monkeypatch):
    """BASELINE config 2's ensemble (256 walkers: two half-steps of 128 fill the chip), a small one over several chunk
    boundaries with an odd number of iterations (the final state sits in the second coordinate buffer), and six
    walkers per half-step -- against the host loop, and against the same run with the overlap switched off."""
    from mcmc_spec_amd import synth
    eng, W = _config2()
    p0 = synth.draw_walkers(nw, seed=5 + nw, tmin=W['tmin'], tmax=W['tmax'])
    host, hs, dev, ds = _chains(eng, p0, nsteps, 17, chunk)
    assert dev.overlapped is True
    assert np.array_equal(dev.get_chain(), host.get_chain())
    assert np.array_equal(dev.get_log_prob(), host.get_log_prob())
    assert np.array_equal(ds.coords, hs.coords) and np.array_equal(ds.log_prob, hs.log_prob)
    assert np.array_equal(dev.acceptance_fraction, host.acceptance_fraction)
    assert 0.1 < dev.acceptance_fraction.mean() < 0.9
    # a second run on the same context starts from version 0 again
    dev2 = _chains(eng, ds.coords, 5, 3, chunk)
    assert np.array_equal(dev2[2].get_chain(), dev2[0].get_chain())
    monkeypatch.setenv('MSX_SMP_OVERLAP', '0')
    host3, hs3, dev3, ds3 = _chains(eng, p0, nsteps, 17, chunk)
    assert dev3.overlapped is False
    assert np.array_equal(dev3.get_chain(), dev.get_chain()) and np.array_equal(ds3.log_prob, ds.log_prob)
    monkeypatch.delenv('MSX_SMP_OVERLAP')
    # ... and by the caller's say (msx_sampler_policy: a device shared with other work): plain launches, same chain
    from mcmc_spec_amd.sampler import DeviceEnsembleSampler
    dev4 = DeviceEnsembleSampler(nw, 6, eng, seed=17, chunk=chunk, overlap=False)
    dev4.run_mcmc(p0, nsteps)
    assert dev4.overlapped is False and np.array_equal(dev4.get_chain(), dev.get_chain())
    dev5 = DeviceEnsembleSampler(nw, 6, eng, seed=17, chunk=chunk)      # (the policy is per run: back to the rule)
    dev5.run_mcmc(p0, 3)
    assert dev5.overlapped is True


def test_overlap_is_not_taken_where_it_could_deadlock_or_does_not_apply():
    """More walkers per half-step than half the CUs: the next half-step's waiting workgroups could keep the previous
    one's off the chip -- plain launches.  A sharded run has a collective between its half-steps -- plain launches."""
    from mcmc_spec_amd import synth
    from mcmc_spec_amd.sampler import DeviceEnsembleSampler
    eng, W = _config2()
    cus = eng.ctx.device_info()['cus']
    nw = 2 * (cus // 2 + 2)
    p0 = synth.draw_walkers(nw, seed=2, tmin=W['tmin'], tmax=W['tmax'])
    s = DeviceEnsembleSampler(nw, 6, eng, seed=1, chunk=4)
    s.run_mcmc(p0, 4)
    assert s.overlapped is False
    s = DeviceEnsembleSampler(64, 6, eng, seed=1, chunk=4, shard=(0, 1))
    s.run_mcmc(p0[:64], 4)
    assert s.overlapped is False


def test_a_walker_error_in_an_overlapped_run_is_reported():
    """Likelihood mode has no prior box: a proposal below the isochrone table is an error status, not a value.  The
    failing walker still publishes its version (nobody waits for it for ever) and the chunk reports the status."""
    from mcmc_spec_amd.sampler import DeviceEnsembleSampler
    c = golden_case('B')
    eng = make_engine(c, rad_prior=False)
    nw = 48
    rng = np.random.default_rng(9)
    p0 = c.theta[0] + rng.normal(size=(nw, 6)) * np.array([30, 30, 0.02, 0.02, 0.02, 2e-5])
    p0[:, 1] = 2905.0 + np.abs(np.random.default_rng(2).normal(size=nw)) * 3   # stretch moves step below 2900 K
    s = DeviceEnsembleSampler(nw, 6, eng, mode='loglikelihood', seed=2, chunk=16)
    with pytest.raises(ValueError):
        s.run_mcmc(p0, 60)
    assert s.overlapped is True


def test_a_version_that_is_never_published_ends_in_a_loud_failure():
    """The fault hook keeps every walker from publishing its new version: the next half-step's workgroups wait 20 ms of
    wall clock, give up, and the chunk's worst status is MSX_W_HANDOVER -- Python raises; no hang, no chain.  The next
    run on the context (hook off) is healthy: the versions belong to the run, not to the context."""
    import time
    from mcmc_spec_amd import _lib, synth
    from mcmc_spec_amd.sampler import DeviceEnsembleSampler
    eng, W = _config2()
    p0 = synth.draw_walkers(32, seed=4, tmin=W['tmin'], tmax=W['tmax'])
    eng.ctx.test_hook(_lib.HOOK_LINKED_FAULT, 1)
    try:
        s = DeviceEnsembleSampler(32, 6, eng, seed=1, chunk=2)
        t0 = time.time()
        with pytest.raises(RuntimeError, match='did not meet'):
            s.run_mcmc(p0, 2)
        assert time.time() - t0 < 10.0
    finally:
        eng.ctx.test_hook(_lib.HOOK_LINKED_FAULT, 0)
    host, hs, dev, ds = _chains(eng, p0, 6, 8, 4)
    assert dev.overlapped is True and np.array_equal(dev.get_chain(), host.get_chain())


def test_half_steps_that_do_not_partition_the_ensemble_are_refused():
    """The version protocol rests on every walker moving exactly once per iteration; an enqueue whose two half-steps
    name a walker twice is refused on the host instead of failing after a 20 ms wait on the device."""
    from mcmc_spec_amd import _lib, synth
    from mcmc_spec_amd.sampler import EnsembleSampler
    eng, W = _config2()
    nw = 32
    p0 = synth.draw_walkers(nw, seed=6, tmin=W['tmin'], tmax=W['tmax'])
    lp0 = eng.logposterior(p0)
    c = eng.ctx
    draw = EnsembleSampler(nw, 6, lambda x: x, seed=1)
    c.sampler_begin(_lib.MODE_LOGPOST, p0.copy(), lp0.copy(), 4)
    try:
        arrays = [np.array(a) for a in draw._draw_steps(2)]
        c.sampler_enqueue(0, *arrays)                      # a proper split: taken (and the run overlaps)
        assert c.sampler_overlapped() == 1
        c.sampler_collect(0, 2)
        bad = [np.array(a) for a in draw._draw_steps(2)]
        bad[0] = bad[0].copy()
        bad[0][1, 1, 0] = bad[0][1, 0, 0]                  # second iteration: one walker in both half-steps
        with pytest.raises(_lib.MsxError, match='appears twice'):
            c.sampler_enqueue(1, *bad)
    finally:
        c.sampler_end()


def test_overlapped_half_steps_of_a_triple_system():
    """ndim 8 (three recipe waves wait for the walkers' versions, the fourth wave fetches the accept inputs): against
    the host loop, bit for bit."""
    c = golden_case('C')
    eng = make_engine(c, rad_prior=True)
    good = c.theta[np.isfinite(eng.logposterior(c.theta))]
    rng = np.random.default_rng(12)
    p0 = good[0] + rng.normal(size=(24, 8)) * np.array([10, 10, 10, 0.01, 0.01, 0.01, 0.01, 1e-5])
    host, hs, dev, ds = _chains(eng, p0, 21, 5, 8)
    assert dev.overlapped is True
    assert np.array_equal(dev.get_chain(), host.get_chain()) and np.array_equal(dev.get_log_prob(), host.get_log_prob(), equal_nan=True)
    assert np.array_equal(ds.coords, hs.coords)


# ---- the stretch move's randomness drawn on the device (msx_sampler_enqueue_drawn) --------------------
def test_device_generator_is_the_counter_stream_its_numpy_restatement_describes():
    """sampler_draw_kernel against mcmc_spec_amd.sampler.counter_draws: the split (walkers sorted by a 64-bit key), the
    partners and z bit for bit; the two logarithms to the last place or two (device log vs NumPy's).  Every iteration is a
    permutation of the ensemble, a chunk started at iteration k equals rows k.. of a chunk started at 0, and odd sizes
    (18 walkers: the sort pads to 32) work like powers of two."""
    from mcmc_spec_amd.sampler import counter_draws
    eng, W = _config2()
    for nw, ndim, seed in ((256, 6, 17), (18, 6, 2**63 + 5), (2048, 8, 3)):
        got = eng.ctx.sampler_draw(seed, 2.0, 0, 7, nw, ndim)
        want = counter_draws(seed, 2.0, ndim, 0, 7, nw)
        for g, w in zip(got[:4], want[:4]):           # sidx, cidx, partner, zz
            assert np.array_equal(g, w)
        for g, w in zip(got[4:], want[4:]):           # zfac = (ndim - 1) ln z, logu = ln u
            assert np.allclose(g, w, rtol=4e-16, atol=1e-300)
        s = np.concatenate([got[0][:, 0], got[0][:, 1]], axis=1)
        assert np.array_equal(np.sort(s, axis=1), np.broadcast_to(np.arange(nw), s.shape))
        assert np.array_equal(got[1], got[0][:, ::-1])
        later = eng.ctx.sampler_draw(seed, 2.0, 3, 4, nw, ndim)
        assert all(np.array_equal(a, b[3:]) for a, b in zip(later, got))
        assert got[2].min() >= 0 and got[2].max() < nw // 2 and np.all(got[3] >= 0.5) and np.all(got[3] <= 2.0) and np.all(got[5] <= 0)
    # the uniforms behind z are uniform: mean 1/2, variance 1/12 over 7 x 2048 draws (a = 2: z = (u + 1)^2 / 2)
    u = np.sqrt(2.0 * got[3]) - 1.0
    assert abs(u.mean() - 0.5) < 0.01 and abs(u.var() - 1 / 12) < 0.005


@pytest.mark.parametrize('nw,nsteps,chunk', [(256, 40, 16), (18, 11, 4)])
def test_device_drawn_chain_is_the_host_loop_fed_with_the_device_stream(nw, nsteps, chunk):
    """rng='device': nothing is drawn or uploaded by the host.  The chain must be, bit for bit, the one the HOST loop walks
    over the same posterior when it is fed the generator's own numbers (msx_sampler_draw) -- overlapped half-steps and
    chunk boundaries included; a second sampler with the same seed repeats it, another seed does not."""
    from mcmc_spec_amd import synth
    from mcmc_spec_amd.sampler import DeviceEnsembleSampler, EnsembleSampler
    eng, W = _config2()
    p0 = synth.draw_walkers(nw, seed=9 + nw, tmin=W['tmin'], tmax=W['tmax'])
    seed = 2024
    dev = DeviceEnsembleSampler(nw, 6, eng, seed=seed, chunk=chunk, rng='device')
    ds = dev.run_mcmc(p0, nsteps)
    host = EnsembleSampler(nw, 6, eng.logposterior, vectorize=True,
                           draws=lambda i, m: eng.ctx.sampler_draw(seed, 2.0, i, m, nw, 6))
    hs = host.run_mcmc(p0, nsteps)
    assert dev.overlapped is True
    assert np.array_equal(dev.get_chain(), host.get_chain()) and np.array_equal(dev.get_log_prob(), host.get_log_prob())
    assert np.array_equal(ds.coords, hs.coords) and np.array_equal(dev.acceptance_fraction, host.acceptance_fraction)
    assert 0.1 < dev.acceptance_fraction.mean() < 0.9
    again = DeviceEnsembleSampler(nw, 6, eng, seed=seed, chunk=chunk + 3, rng='device')   # (the chunking does not matter)
    again.run_mcmc(p0, nsteps)
    assert np.array_equal(again.get_chain(), dev.get_chain())
    other = DeviceEnsembleSampler(nw, 6, eng, seed=seed + 1, chunk=chunk, rng='device')
    other.run_mcmc(p0, nsteps)
    assert not np.array_equal(other.get_chain(), dev.get_chain())
