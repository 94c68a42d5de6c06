"""CPU: the N > 1 walker-sharding path with a world_size-2 gloo process group.  The per-rank evaluator
is the ORACLE here (test infrastructure standing in for the GPU engine): what is under test is the
partition / pad / all-gather logic of mcmc_spec_amd.dist, which must reproduce the 1-rank vector
bit for bit for every ragged size."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

import common
from mcmc_spec_amd.dist import shard_bounds


def test_shard_bounds_cover_exactly_once():
    for n in (0, 1, 5, 16, 17, 255, 256):
        for w in (1, 2, 3, 8):
            seen = []
            for r in range(w):
                lo, hi, m = shard_bounds(n, w, r)
                assert 0 <= lo <= hi <= n and hi - lo <= m
                seen += list(range(lo, hi))
            assert seen == list(range(n))


def _worker(rank, world, port, sizes, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, common.ROOT)
    sys.path.insert(0, os.path.join(common.ROOT, 'tests'))
    import warnings
    warnings.filterwarnings('ignore')
    import torch.distributed as dist
    from mcmc_spec_amd.dist import ShardedLogProb
    dist.init_process_group('gloo', rank=rank, world_size=world)
    c = common.golden_case('B')
    calls = []

    def local_eval(block):
        calls.append(len(block))
        return np.array([common.oracle_logpost(c, t) for t in block])

    f = ShardedLogProb(local_eval, device='cpu')
    out = {}
    for n in sizes:
        out[n] = f(c.g['theta_post'][:n])
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, out, calls))


@pytest.mark.timeout(300)
def test_world2_gather_equals_single_rank_for_ragged_sizes():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    sizes = [1, 2, 5, 8]
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, sizes, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=280) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    c = common.golden_case('B')
    import warnings
    warnings.filterwarnings('ignore')
    for n in sizes:
        want = np.array([common.oracle_logpost(c, t) for t in c.g['theta_post'][:n]])
        for rank, out, calls in res:
            assert np.array_equal(out[n], want), (n, rank)
    # each rank evaluated only its own (padded) block: ceil(n/2) walkers per call
    for rank, out, calls in res:
        assert calls == [-(-n // 2) for n in sizes]


def _worker_errors(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, common.ROOT)
    import torch
    import torch.distributed as dist
    from mcmc_spec_amd.benchutil import capture_agreed
    from mcmc_spec_amd.dist import ShardedLogProb
    dist.init_process_group('gloo', rank=rank, world_size=world)
    log = []

    # (1) a data-dependent exception on ONE rank (a bad walker in its shard) must surface on EVERY rank, after the
    # collectives, instead of leaving the other rank blocked in all_gather
    def local_eval(block):
        if np.any(block[:, 0] < 0):
            raise KeyError('a model grid node needed by walker ... is not in specs')
        return block.sum(axis=1)

    f = ShardedLogProb(local_eval, device='cpu')
    ncoll = [0]
    real_gather = dist.all_gather_into_tensor

    def counting_gather(*a, **k):
        ncoll[0] += 1
        return real_gather(*a, **k)

    f.dist = type('D', (), {'all_gather_into_tensor': staticmethod(counting_gather)})()
    coords = np.arange(24, dtype=float).reshape(8, 3)
    log.append(('ok', f(coords).tolist()))
    bad = coords.copy()
    bad[6, 0] = -1.0                       # lives in rank 1's shard only
    try:
        f(bad)
        log.append(('no exception',))
    except KeyError as e:
        log.append(('KeyError', 'rank 1' in str(e) or rank == 1))
    log.append(('after', f(coords).tolist()))      # the group is still usable: nobody is stuck in a collective
    log.append(('collectives', ncoll[0]))          # ONE per evaluation, failing or not: the error class rides in the NaN

    # (1b) an evaluator that RETURNS the device path's tagged NaN (payload 2 = IndexError in ShardedLogProb's own numbering) without raising, on rank 0:
    # rank 0 too must raise IndexError (it used to `raise None`)
    from mcmc_spec_amd.dist import _nan_with_code

    def tagging_eval(block):
        out = block.sum(axis=1)
        if rank == 0:
            out[0] = _nan_with_code(2)
        return out

    g = ShardedLogProb(tagging_eval, device='cpu')
    try:
        g(coords)
        log.append(('tagged', 'no exception'))
    except Exception as e:  # noqa: BLE001 - the class is the assertion
        log.append(('tagged', type(e).__name__))

    # (2) bench.py's graph capture: rank 1 fails to capture; nobody may replay (a replay holds collectives)
    replays = []

    def all_min(flag):
        t = torch.tensor([flag])
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return int(t.item())

    def capture():
        if rank == 1:
            raise RuntimeError('capture failure on this rank')
        return 'graph'

    def replay(g):
        replays.append(g)
        t = torch.ones(1)
        dist.all_reduce(t)                 # what a real replay contains

    log.append(('capture', capture_agreed(capture, replay, all_min, rank), len(replays)))
    # ... and when every rank captures, every rank replays once
    log.append(('capture2', capture_agreed(lambda: 'graph', replay, all_min, rank), len(replays)))
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, log))


@pytest.mark.timeout(120)
def test_world2_one_rank_failing_does_not_deadlock_the_other():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_errors, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=100) for _ in procs)
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    want = np.arange(24, dtype=float).reshape(8, 3).sum(axis=1).tolist()
    for rank in (0, 1):
        log = res[rank]
        assert log[0] == ('ok', want)
        assert log[1] == ('KeyError', True)            # same exception class on both ranks
        assert log[2] == ('after', want)
        assert log[3] == ('collectives', 3)
        assert log[4] == ('tagged', 'IndexError')      # on the rank that produced the NaN and on the other
        assert log[5] == ('capture', None, 0)          # nobody replayed
        assert log[6] == ('capture2', 'graph', 1)


def _worker_chain(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, common.ROOT)
    import torch.distributed as dist
    from mcmc_spec_amd.dist import ShardedLogProb
    from mcmc_spec_amd.sampler import EnsembleSampler
    dist.init_process_group('gloo', rank=rank, world_size=world)

    def lnp(x):
        return -0.5 * np.sum((x / np.array([0.5, 1.0, 2.0])) ** 2, axis=1) + 0.1 * np.sin(x[:, 0])

    f = ShardedLogProb(lnp, device='cpu')
    nw = 18                                # half-steps of 9 walkers: ragged over 2 ranks (5 + 4)
    p0 = np.random.default_rng(8).normal(size=(nw, 3))
    s = EnsembleSampler(nw, 3, f, vectorize=True, seed=4)   # replicated sampler, shared seed
    s.run_mcmc(p0, 25)
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, s.get_chain(), s.get_log_prob(), s.acceptance_fraction))


@pytest.mark.timeout(120)
def test_world2_sharded_chain_equals_the_one_rank_chain():
    """The N > 1 sampler protocol on CPU: every rank runs the same sampler from the same seed, each half-step's
    proposals are evaluated in rank blocks and all-gathered; both ranks must hold the chain a single rank produces,
    bit for bit (ragged half-steps included)."""
    from mcmc_spec_amd.sampler import EnsembleSampler
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_chain, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=100) for _ in procs]
    for p in procs:
        p.join(30)
        assert p.exitcode == 0

    def lnp(x):
        return -0.5 * np.sum((x / np.array([0.5, 1.0, 2.0])) ** 2, axis=1) + 0.1 * np.sin(x[:, 0])

    p0 = np.random.default_rng(8).normal(size=(18, 3))
    one = EnsembleSampler(18, 3, lnp, vectorize=True, seed=4)
    one.run_mcmc(p0, 25)
    for rank, chain, lp, acc in res:
        assert np.array_equal(chain, one.get_chain()) and np.array_equal(lp, one.get_log_prob())
        assert np.array_equal(acc, one.acceptance_fraction)
