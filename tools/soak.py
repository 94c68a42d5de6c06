#!/usr/bin/env python3
"""Soak check for the hot kernel's intra-workgroup hand-offs (histogram filled in phase A, per-wave scans, gather,
barrier-free rank): many random ensembles, every workgroup size, repeated launches -- all results must agree to
the bit with each other (parity with the oracle is the GPU test suite's job).

    python tools/soak.py --rounds 200
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--rounds', type=int, default=100)
    ap.add_argument('--walkers', type=int, default=1024)
    ap.add_argument('--npix', type=int, default=4096)
    args = ap.parse_args()
    import torch
    from bench import build_workload
    from mcmc_spec_amd import _lib, synth
    from mcmc_spec_amd.engine import Engine
    eng = Engine(0)
    W = build_workload(eng, args.npix, False, keep_host_grid=True)
    dev = torch.device('cuda', 0)
    s = torch.cuda.current_stream(dev).cuda_stream
    n = args.walkers
    rng = np.random.default_rng(123)
    t0 = time.time()
    bad = 0
    checked = 0
    for r in range(args.rounds):
        th = synth.draw_walkers(n, seed=1000 + r, tmin=W['tmin'], tmax=W['tmax'])
        # sprinkle edge cases: zero extinction, on-node temperatures, out-of-box and non-finite coordinates
        k = rng.integers(0, n, size=24)
        th[k[:6], 2] = 0.0
        th[k[6:12], 0] = 3800.0
        th[k[12:16], 1] = 3000.0
        th[k[16:20], 3] = -1.0
        th[k[20:22], 4] = np.nan
        th[k[22:24], 5] = np.inf
        tht = torch.from_numpy(np.ascontiguousarray(th)).to(dev)
        ref = None
        for mode in (_lib.MODE_LOGPOST, _lib.MODE_CHISQ):
            outs = []
            for block in (0, 256, 512, 1024, 0):
                lp = torch.empty(n, dtype=torch.float64, device=dev)
                st = torch.empty(n, dtype=torch.int32, device=dev)
                eng.ctx.logprob_batch_dev(tht.data_ptr(), n, 6, lp.data_ptr(), st.data_ptr(), s, mode, block)
                outs.append((lp, st))
            torch.cuda.synchronize()
            a = outs[0][0].cpu().numpy()
            for lp, st in outs[1:]:
                if not np.array_equal(lp.cpu().numpy(), a, equal_nan=True):
                    bad += 1
            # shards of the batch evaluated alone
            for lo, m in ((0, min(100, n)), (n // 2, min(256, n - n // 2)), (max(n - 37, 0), min(37, n))):
                lp = torch.empty(m, dtype=torch.float64, device=dev)
                st = torch.empty(m, dtype=torch.int32, device=dev)
                eng.ctx.logprob_batch_dev(tht[lo:lo + m].contiguous().data_ptr(), m, 6, lp.data_ptr(), st.data_ptr(), s, mode, 0)
                torch.cuda.synchronize()
                if not np.array_equal(lp.cpu().numpy(), a[lo:lo + m], equal_nan=True):
                    bad += 1
            checked += 1
            if mode == _lib.MODE_LOGPOST:
                ref = a
        if r % 25 == 0:
            print('round {:4d}: mismatching launches so far {}  ({:.0f} s)'.format(r, bad, time.time() - t0), flush=True)
    print('SOAK {}: {} rounds x 2 modes, {} mismatching launches'.format('OK' if bad == 0 else 'FAILED', args.rounds, bad))
    sys.exit(1 if bad else 0)


if __name__ == '__main__':
    main()
