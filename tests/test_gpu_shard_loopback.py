"""The rank >= 1 paths of the sharded device-resident sampler (msx_sampler_shard) on ONE GPU.

A LOOPBACK group (include/msx.h: msx_comm_init_loopback / msx_sampler_enqueue_group; SURVEY §4 (4)'s "fake
collective") makes `world` contexts of this process the ranks of one job: the all-gather becomes device copies
between the ranks' gathered vectors, placed exactly where RCCL's in-place all-gather puts them.  Everything that
depends on the rank -- block offsets into the half-step's arrays, ragged and empty shards, the status that travels in
a NaN's payload, the apply kernel over the gathered vector -- runs for rank 1, 2, ... and every rank's chain must
equal the fully fused one-GPU chain bit for bit.
"""
import numpy as np
import pytest

from common import golden_case
from test_gpu_parity import make_engine

pytestmark = pytest.mark.gpu


def run_group(engs, p0, nsteps, seed, mode='logposterior', chunk=8):
    """Drive a loopback group like DeviceEnsembleSampler drives one context: same randomness (the host sampler's own
    draws, one seed) for every rank, chunks queued in lock-step.  Returns per rank (chain, logp chain, naccept,
    worst status, final coords, final logp)."""
    from mcmc_spec_amd import _lib
    from mcmc_spec_amd.sampler import EnsembleSampler
    world = len(engs)
    nw, nd = p0.shape
    md = {'logposterior': _lib.MODE_LOGPOST, 'loglikelihood': _lib.MODE_LOGLIKE}[mode]
    draw = EnsembleSampler(nw, nd, lambda x: x, seed=seed)
    fn = engs[0].logposterior if mode == 'logposterior' else engs[0].loglikelihood
    lp0 = fn(p0)
    ctxs = [e.ctx for e in engs]
    for r, c in enumerate(ctxs):
        c.sampler_begin(md, p0.copy(), lp0.copy(), chunk)
        c.sampler_shard(r, world)
    outs = [dict(chain=[], lp=[], nacc=None, worst=0) for _ in engs]
    try:
        left = nsteps
        while left > 0:
            m = min(chunk, left)
            left -= m
            arrays = draw._draw_steps(m)
            _lib.Context.sampler_enqueue_group(ctxs, 0, *arrays)
            for r, c in enumerate(ctxs):
                chain, lpc, nacc, worst = c.sampler_collect(0, m)
                outs[r]['chain'].append(chain)
                outs[r]['lp'].append(lpc)
                outs[r]['nacc'] = nacc
                outs[r]['worst'] = max(outs[r]['worst'], worst)
    finally:
        finals = [c.sampler_end(want_state=True) for c in ctxs]
    return [(np.concatenate(o['chain']), np.concatenate(o['lp']), o['nacc'], o['worst'], f[0], f[1])
            for o, f in zip(outs, finals)]


def fused_chain(eng, p0, nsteps, seed, mode='logposterior', chunk=8):
    from mcmc_spec_amd.sampler import DeviceEnsembleSampler
    s = DeviceEnsembleSampler(p0.shape[0], p0.shape[1], eng, mode=mode, seed=seed, chunk=chunk)
    st = s.run_mcmc(p0, nsteps)
    return s.get_chain(), s.get_log_prob(), st


def _p0(c, nw, seed):
    rng = np.random.default_rng(seed)
    return c.theta[0] + rng.normal(size=(nw, 6)) * np.array([30, 30, 0.02, 0.02, 0.02, 2e-5])


@pytest.mark.parametrize('world,nw', [(2, 48), (2, 50), (3, 52), (4, 12)])
def test_every_rank_of_a_loopback_group_walks_the_fused_chain(world, nw):
    """world 2 with an even split (24 proposals: 12 + 12), a ragged one (25: 13 + 12), world 3 (26: 9 + 9 + 8) and
    world 4 with 6 proposals per half-step (2 + 2 + 2 + 0: rank 3's shard is EMPTY -- it launches nothing, takes the
    others' blocks and still applies every walker)."""
    from mcmc_spec_amd import _lib
    c = golden_case('B')
    engs = [make_engine(c, rad_prior=False) for _ in range(world)]   # one context per rank, all on device 0
    _lib.Context.comm_init_loopback([e.ctx for e in engs])
    p0 = _p0(c, nw, 6 + nw)
    want_chain, want_lp, want_state = fused_chain(engs[0], p0, 24, seed=13)
    got = run_group(engs, p0, 24, seed=13)
    for r, (chain, lp, nacc, worst, coords, logp) in enumerate(got):
        assert worst == 0
        assert np.array_equal(chain, want_chain), 'rank {}'.format(r)
        assert np.array_equal(lp, want_lp), 'rank {}'.format(r)
        assert np.array_equal(coords, want_state.coords) and np.array_equal(logp, want_state.log_prob)
    acc = got[0][2] / 24.0
    assert 0.05 < acc.mean() < 0.95 and all(np.array_equal(g[2], got[0][2]) for g in got)


def test_a_walker_error_on_one_rank_reaches_every_rank():
    """Likelihood mode has no prior box: a proposal below the isochrone table is an error status, not a value.  It
    is produced by whichever rank evaluates that proposal and travels to the others inside the NaN it yields."""
    from mcmc_spec_amd import _lib
    c = golden_case('B')
    engs = [make_engine(c, rad_prior=False) for _ in range(2)]
    _lib.Context.comm_init_loopback([e.ctx for e in engs])
    nw = 48
    p0 = _p0(c, nw, 9)
    p0[:, 1] = 2905.0 + np.abs(np.random.default_rng(2).normal(size=nw)) * 3   # stretch moves step below 2900 K
    got = run_group(engs, p0, 60, seed=2, mode='loglikelihood')
    assert got[0][3] == _lib.W_VALUEERROR and got[1][3] == _lib.W_VALUEERROR
    assert np.array_equal(got[0][0], got[1][0])     # and both ranks still hold the same chain


def test_loopback_group_call_order_is_checked():
    from mcmc_spec_amd import _lib
    c = golden_case('B')
    engs = [make_engine(c, rad_prior=False) for _ in range(2)]
    ctxs = [e.ctx for e in engs]
    p0 = _p0(c, 16, 1)
    lp0 = engs[0].logposterior(p0)
    ctxs[0].sampler_begin(_lib.MODE_LOGPOST, p0, lp0, 4)
    with pytest.raises(_lib.MsxError):          # no communicator of any kind yet
        ctxs[0].sampler_shard(0, 2)
    ctxs[0].sampler_end()
    _lib.Context.comm_init_loopback(ctxs)
    with pytest.raises(_lib.MsxError):          # a context joins one group only
        _lib.Context.comm_init_loopback(ctxs)
    from mcmc_spec_amd.sampler import EnsembleSampler
    arrays = EnsembleSampler(16, 6, lambda x: x, seed=1)._draw_steps(2)
    ctxs[0].sampler_begin(_lib.MODE_LOGPOST, p0, lp0, 4)
    ctxs[0].sampler_shard(0, 2)
    with pytest.raises(_lib.MsxError):          # rank 1 has not begun its run
        _lib.Context.sampler_enqueue_group(ctxs, 0, *arrays)
    with pytest.raises(_lib.MsxError):          # a rank of a group does not advance alone
        ctxs[0].sampler_enqueue(0, *arrays)
    ctxs[0].sampler_end()
