#!/usr/bin/env python3
"""Batch-size x workgroup-size x path (fused / linked) sweep of the hot path: device time per batch from HIP events
on the launch stream.  `--graph` replays each batch from a hipGraph (no host launch gaps)."""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--npix', type=int, default=4096)
    ap.add_argument('--phot', action='store_true')
    ap.add_argument('--walkers', default='128,256,512,1024,2048,4096,16384')
    ap.add_argument('--blocks', default='256,512,1024')
    ap.add_argument('--iters', type=int, default=50)
    ap.add_argument('--paths', default='fused')
    ap.add_argument('--graph', action='store_true')
    ap.add_argument('--sort-cells', action='store_true', help='order the batch by Teff cell: neighbours mostly share their grid rows (what the pair form\'s planner arranges)')
    ap.add_argument('--spread', action='store_true', help='walkers uniform over the whole Teff range: every walker its own grid rows')
    args = ap.parse_args()
    import torch
    from bench import build_workload
    from mcmc_spec_amd import _lib, synth
    from mcmc_spec_amd.engine import Engine
    dev = torch.device('cuda', 0)
    eng = Engine(0)
    W = build_workload(eng, args.npix, args.phot, broaden='in_path' if 'inpath' in args.paths.split(',') else 'staging')
    b_alg = 2 * 4 * W['nwin'] * 8 + 56
    stream = torch.cuda.current_stream(dev)
    rows = []
    for n in [int(x) for x in args.walkers.split(',')]:
        thn = synth.draw_walkers(n, seed=3, tmin=W['tmin'], tmax=W['tmax'])
        if args.spread:
            thn[:, 0:2] = np.random.default_rng(1).uniform(W['tmin'] + 1, W['tmax'] - 1, size=(n, 2))
        if args.sort_cells:
            thn = thn[np.lexsort((thn[:, 1] // 100, thn[:, 0] // 100))]
        th = torch.from_numpy(np.ascontiguousarray(thn)).to(dev)
        lp = torch.empty(n, dtype=torch.float64, device=dev)
        st = torch.empty(n, dtype=torch.int32, device=dev)
        for path in args.paths.split(','):
          eng.ctx.set_path({'auto': _lib.PATH_AUTO, 'fused': _lib.PATH_FUSED, 'linked': _lib.PATH_LINKED, 'pair': _lib.PATH_PAIR,
                             'inpath': _lib.PATH_INPATH}[path])   # (inpath: the broadening applied per walker, include/msx.h)
          for B in [int(x) for x in args.blocks.split(',')]:
            def go(sp):
                eng.ctx.logprob_batch_dev(th.data_ptr(), n, 6, lp.data_ptr(), st.data_ptr(), sp, _lib.MODE_LOGPOST, B)
            iters = max(3, min(args.iters, int(2e6 / n)))
            if args.graph:
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=torch.cuda.Stream(dev)):
                    for _ in range(iters):
                        go(torch.cuda.current_stream(dev).cuda_stream)
                run = g.replay
                reps = 3
            else:
                def run():
                    for _ in range(iters):
                        go(stream.cuda_stream)
                reps = 1
            run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(reps):
                run()
            e1.record(stream)
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / (iters * reps)
            extra = {}
            if path == 'pair':
                extra['pairs_singles'] = eng.ctx.pair_stats()
            row = dict(extra, walkers=n, npix=args.npix, path=path, block=B, batch_us=ms * 1e3, evals_per_s=n / (ms * 1e-3),
                       alg_GBps=n * b_alg / (ms * 1e-3) / 1e9, graph=bool(args.graph))
            rows.append(row)
            print(json.dumps(row), flush=True)


if __name__ == '__main__':
    main()
