"""Walker sharding across the GPUs of one node.

Walkers are independent (no cross-walker term anywhere in mft6.py:1139-1205), so rank r evaluates the
contiguous block ``[r*m, (r+1)*m)`` of the ``(n, ndim)`` coordinate array with ``m = ceil(n / G)`` and the
log-probabilities -- and any rank-local failure, as a NaN payload -- are combined by ONE collective:
``all_gather_into_tensor`` of ``m`` float64 per rank
(backend ``nccl`` = RCCL over xGMI on the GPU box; ``gloo`` in the CPU tests).  The ragged tail is padded
with copies of the last walker so no sentinel reaches the kernel; the pad is dropped after the gather.
The staged grid/tables are replicated on every GPU.  Every rank must call with the same ``coords``
(replicated sampler with a shared seed) and gets the same full vector back, bit-identical to a 1-GPU
evaluation of the same walkers because each walker's value does not depend on its batch.

One process per GPU, launched with ``python -m torch.distributed.run``.
"""
from __future__ import annotations

import numpy as np


_ERR_CODES = [KeyError, IndexError, ValueError]   # the reference's conventions, in status order


_QNAN = 0x7ff8000000000000


def _nan_with_code(code):
    """A quiet NaN whose low mantissa byte carries ``code`` (1..255), like the kernel's nan_with_status."""
    return np.array([_QNAN | (int(code) & 0xff)], dtype=np.uint64).view(np.float64)[0]


def _codes_of(v):
    """Per element: the code a `_nan_with_code` NaN carries, 0 for anything else (numbers, -inf, a plain NaN)."""
    b = np.ascontiguousarray(v, dtype=np.float64).view(np.uint64)
    tagged = (b & np.uint64(0xfff8000000000000)) == np.uint64(_QNAN)
    return np.where(tagged, (b & np.uint64(0xff)).astype(np.int64), 0)


def shard_bounds(n, world, rank):
    """Contiguous block partition with equal block length m = ceil(n / world).  Returns (lo, hi, m)."""
    m = -(-n // world)
    lo = min(rank * m, n)
    hi = min(lo + m, n)
    return lo, hi, m


class ShardedLogProb:
    """Wrap a local batch evaluator into an emcee-style vectorised ``log_prob_fn`` over a process group.

    ``local_eval(coords_block) -> float64[len(block)]`` is the per-rank evaluator
    (``Engine.logposterior`` on this rank's GPU).  ``device`` is the torch device that holds the
    gather buffers (``cuda:LOCAL_RANK`` for nccl, ``cpu`` for gloo)."""

    def __init__(self, local_eval, group=None, device='cpu'):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.local_eval = local_eval
        self.group = group
        self.device = torch.device(device)
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0

    def __call__(self, coords, *args, **kwargs):
        torch = self.torch
        coords = np.atleast_2d(np.asarray(coords, dtype=float))
        n = len(coords)
        if self.world == 1:
            return np.asarray(self.local_eval(coords, *args, **kwargs), dtype=float)
        lo, hi, m = shard_bounds(n, self.world, self.rank)
        block = coords[lo:hi]
        if len(block) < m:  # ragged tail (or an empty shard): pad with the last walker, dropped below
            pad = np.repeat(coords[-1:], m - len(block), axis=0)
            block = np.concatenate([block, pad], axis=0)
        # The local evaluator follows the reference's error conventions: it may RAISE (KeyError for a missing grid
        # node, IndexError, ValueError -- data-dependent, so possibly on one rank only).  A rank that raised before
        # the collective would leave the others blocked in it, so the exception is held back and its class travels
        # INSIDE the one collective there is, as the payload of the quiet NaN that fills the failing rank's block --
        # the convention of the device path (csrc/logprob_kernel.h, nan_with_status) -- and every rank then raises
        # the same exception.  (A log-probability is never NaN by itself: the engine returns -inf.)
        err, local = None, None
        try:
            local = np.asarray(self.local_eval(block, *args, **kwargs), dtype=float)
            if local.shape != (m,):
                raise ValueError('local_eval returned shape {} for a block of {} walkers'.format(local.shape, m))
        except Exception as exc:  # noqa: BLE001 - re-raised below, on every rank
            err = exc
            code = _ERR_CODES.index(type(err)) + 1 if type(err) in _ERR_CODES else len(_ERR_CODES) + 1
            local = np.full(m, _nan_with_code(code))
        send = torch.from_numpy(np.ascontiguousarray(local)).to(self.device)
        recv = torch.empty(m * self.world, dtype=torch.float64, device=self.device)
        self.dist.all_gather_into_tensor(recv, send, group=self.group)   # the ONE collective of the evaluation
        out = recv.cpu().numpy()
        codes = _codes_of(out).reshape(self.world, m).max(axis=1)
        if codes.any():
            first = int(np.nonzero(codes)[0][0])           # the lowest failing rank decides, on every rank
            if first == self.rank and err is not None:
                raise err
            # (a local evaluator may also RETURN a tagged NaN without raising -- the device path's own values do: then this
            # rank builds the class from the code like every other rank)
            c = int(codes[first])
            cls = _ERR_CODES[c - 1] if c <= len(_ERR_CODES) else RuntimeError
            raise cls('walker evaluation failed on rank {} ({}); raised on every rank'.format(first, cls.__name__))
        # rank r's valid entries are the first (hi_r - lo_r) of its block
        parts = []
        for r in range(self.world):
            l, h, _ = shard_bounds(n, self.world, r)
            parts.append(out[r * m: r * m + (h - l)])
        return np.concatenate(parts)


def init_engine_comm(engine, group=None):
    """Give ``engine`` (one per rank, on this rank's GPU) the library's own RCCL communicator over the ranks of
    ``group``: rank 0 creates the id, torch.distributed broadcasts it, every rank joins (``msx_comm_init``).  Needed
    by the sharded device-resident sampler, whose per-half-step all-gather is enqueued from C on the compute stream.
    Returns (rank, world)."""
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    dev = torch.device('cuda', engine.device) if dist.get_backend(group) == 'nccl' else torch.device('cpu')
    idt = torch.zeros(128, dtype=torch.uint8, device=dev)
    if rank == 0:
        idt.copy_(torch.frombuffer(bytearray(engine.ctx.comm_unique_id()), dtype=torch.uint8))
    dist.broadcast(idt, src=0, group=group)
    engine.ctx.comm_init(bytes(idt.cpu().numpy().tobytes()), rank, world)
    return rank, world
