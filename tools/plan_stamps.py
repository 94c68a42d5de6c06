#!/usr/bin/env python3
"""Diagnostic: where pair_plan_kernel's time goes (a -DMSX_STAMPS build; thread 0 of every planner workgroup stamps the phases).

    MSX_LIB=build/libmsx_stamps.so python tools/plan_stamps.py --walkers 2048
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--walkers', type=int, default=2048)
    ap.add_argument('--npix', type=int, default=4096)
    args = ap.parse_args()
    import torch
    from bench import build_workload
    from mcmc_spec_amd import _lib, synth
    from mcmc_spec_amd.engine import Engine
    eng = Engine(0)
    W = build_workload(eng, args.npix, False)
    dev = torch.device('cuda', 0)
    n = args.walkers
    th = torch.from_numpy(synth.draw_walkers(n, seed=3, tmin=W['tmin'], tmax=W['tmax'])).to(dev)
    lp = torch.empty(n, dtype=torch.float64, device=dev)
    st = torch.empty(n, dtype=torch.int32, device=dev)
    s = torch.cuda.current_stream(dev).cuda_stream
    eng.ctx.set_path(_lib.PATH_PAIR)
    for _ in range(20):
        eng.ctx.logprob_batch_dev(th.data_ptr(), n, 6, lp.data_ptr(), st.data_ptr(), s, _lib.MODE_LOGPOST, 0)
    torch.cuda.synchronize()
    nb = (n + 255) // 256
    out = np.zeros((nb, 16), dtype=np.uint64)
    fn = eng.ctx.lib.msx_diag_read_stamps
    fn.argtypes = [C.c_void_p, C.c_int64, C.c_void_p]
    fn.restype = C.c_int
    assert fn(eng.ctx.h, nb, out.ctypes.data) == 0
    o = out.astype(np.int64)
    names = ['tables + theta (barrier)', 'recipe_scalar2 (+ band rows requested)', 'pairs in the wave + barrier', 'card rounds',
             'global adds issued', 'prior terms', 'band terms', 'adds back + barrier + where + records stored', 'ticket']
    print('planner, {} walkers, {} workgroups; stamps are shader cycles (2.3-2.4 GHz)'.format(n, nb))
    print('  entry -> end: median {} max {}'.format(int(np.median(o[:, 9] - o[:, 0])), int((o[:, 9] - o[:, 0]).max())))
    pass
    for k, nm in enumerate(names):
        v = o[:, k + 1] - o[:, k]
        print('  {:32s} median {:6d}  max {:6d}'.format(nm, int(np.median(v)), int(v.max())))


if __name__ == '__main__':
    main()
