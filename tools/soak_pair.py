#!/usr/bin/env python3
"""Soak test of the pair form (planner + pair kernel): batches of random size and make-up -- a tight ensemble, one spread
over the whole grid, mixtures, walkers outside the prior box and outside the isochrone, Teff exactly on nodes, A_V = 0 --
evaluated through the pair form back to back; every value must equal the fused kernel's for the same walker, bit for bit,
every walker must be placed exactly once (the planner's counts), and the working counters must be back at zero.

    python3 tools/soak_pair.py --batches 300
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batches', type=int, default=300)
    ap.add_argument('--npix', type=int, default=4096)
    ap.add_argument('--phot', action='store_true')
    ap.add_argument('--seed', type=int, default=1)
    args = ap.parse_args()
    from bench import build_workload
    from mcmc_spec_amd import _lib, synth
    from mcmc_spec_amd.engine import Engine
    eng = Engine(0)
    W = build_workload(eng, args.npix, args.phot)
    rng = np.random.default_rng(args.seed)
    pool_n = 40000
    tight = synth.draw_walkers(pool_n, seed=5, tmin=W['tmin'], tmax=W['tmax'])
    spread = tight.copy()
    spread[:, 0:2] = rng.uniform(W['tmin'] - 30, W['tmax'] + 30, size=(pool_n, 2))   # some outside the prior box
    odd = tight.copy()
    odd[:, 0] = np.round(odd[:, 0] / 100.0) * 100.0         # on a Teff node
    odd[::3, 2] = 0.0                                      # A_V = 0
    odd[::7, 5] = -odd[::7, 5]                             # negative parallax: rejected
    odd[::11, 1] = 2000.0                                  # below the isochrone: ValueError status
    pools = [tight, spread, odd]
    eng.ctx.set_path(_lib.PATH_FUSED)
    want = [eng.ctx.logprob_batch(p, _lib.MODE_LOGPOST) for p in pools]   # (values, statuses) of the fused kernel
    eng.ctx.set_path(_lib.PATH_PAIR)
    bad = bad_status = bad_count = total = 0
    for b in range(args.batches):
        n = int(rng.choice([1, 2, 3, 63, 64, 65, 255, 256, 257, 1000, 2048, 4096, 9000, 16384, 20000])) if b % 3 == 0 else int(rng.integers(1, 20001))
        mix = rng.random(3) ** 2
        mix /= mix.sum()
        which = rng.choice(3, size=n, p=mix)
        idx = rng.integers(0, pool_n, size=n)
        th = np.empty((n, 6))
        exp = np.empty(n)
        exs = np.empty(n, dtype=np.int32)
        for k in range(3):
            m = which == k
            th[m] = pools[k][idx[m]]
            exp[m] = want[k][0][idx[m]]
            exs[m] = want[k][1][idx[m]]
        g, st = eng.ctx.logprob_batch(th, _lib.MODE_LOGPOST)
        total += n
        bad += int((~((g == exp) | (np.isnan(g) & np.isnan(exp)))).sum())
        bad_status += int((st != exs).sum())
        if n <= 16384:   # (larger batches run as sub-batches: the counts are the last sub-batch's)
            p, s = eng.ctx.pair_stats()
            live = int((exs == 0).sum())
            bad_count += int(2 * p + s != live)
        if (b + 1) % 50 == 0:
            print('  {} batches, {} walkers: {} values differ, {} statuses differ, {} batches with a wrong count'.format(b + 1, total, bad, bad_status, bad_count), flush=True)
    print('pair form, {} px: {} batches, {} walkers: {} values differ from the fused kernel, {} statuses differ, {} batches whose '
          'planner counts do not add up'.format(args.npix, args.batches, total, bad, bad_status, bad_count))
    sys.exit(1 if (bad or bad_status or bad_count) else 0)


if __name__ == '__main__':
    main()
