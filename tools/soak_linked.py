#!/usr/bin/env python3
"""Soak test of the linked form: many launches over rotating batches of walkers; every launch's values must equal the fused
kernel's for the same batch, bit for bit (a race in the meetings would show as a differing value or a status).

    python3 tools/soak_linked.py --launches 50000
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--launches', type=int, default=50000)
    ap.add_argument('--walkers', type=int, default=128)
    args = ap.parse_args()
    import torch
    from bench import build_workload
    from mcmc_spec_amd import _lib, synth
    from mcmc_spec_amd.engine import Engine
    eng = Engine(0)
    W = build_workload(eng, 16384, True)
    dev = torch.device('cuda', 0)
    n, nb = args.walkers, 16
    th = [torch.from_numpy(synth.draw_walkers(n, seed=100 + b, tmin=W['tmin'], tmax=W['tmax'])).to(dev) for b in range(nb)]
    s = torch.cuda.current_stream(dev).cuda_stream
    want = []
    eng.ctx.set_path(_lib.PATH_FUSED)
    for b in range(nb):
        lp = torch.empty(n, dtype=torch.float64, device=dev)
        st = torch.empty(n, dtype=torch.int32, device=dev)
        eng.ctx.logprob_batch_dev(th[b].data_ptr(), n, 6, lp.data_ptr(), st.data_ptr(), s, _lib.MODE_LOGPOST, 0)
        torch.cuda.synchronize()
        want.append(lp.clone())
    eng.ctx.set_path(_lib.PATH_LINKED)
    ring = 64   # launches between checks: their outputs are kept
    lps = [torch.empty(n, dtype=torch.float64, device=dev) for _ in range(ring)]
    sts = [torch.empty(n, dtype=torch.int32, device=dev) for _ in range(ring)]
    bad = 0
    done = 0
    while done < args.launches:
        for k in range(ring):
            b = (done + k) % nb
            eng.ctx.logprob_batch_dev(th[b].data_ptr(), n, 6, lps[k].data_ptr(), sts[k].data_ptr(), s, _lib.MODE_LOGPOST, 0)
        torch.cuda.synchronize()
        for k in range(ring):
            b = (done + k) % nb
            if not torch.equal(lps[k], want[b]) or int(sts[k].max()) > 1:
                bad += 1
        done += ring
    print('linked form, {} walkers x 16,384 px: {} launches over {} rotating batches, {} differ from the fused kernel'.format(n, done, nb, bad))
    sys.exit(1 if bad else 0)


if __name__ == '__main__':
    main()
