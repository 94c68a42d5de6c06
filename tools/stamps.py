#!/usr/bin/env python3
"""Diagnostic: per-phase shader-clock shares of the hot kernel (build with -DMSX_STAMPS, never shipped).

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -mllvm -amdgpu-kernarg-preload-count=8 -DMSX_STAMPS -o build/libmsx_stamps.so mcmc_spec_amd/csrc/msx.hip
    MSX_LIB=build/libmsx_stamps.so python tools/stamps.py --walkers 256 --block 1024
    MSX_LIB=build/libmsx_stamps.so python tools/stamps.py --walkers 128 --npix 16384 --path linked     # each walker's last segment
(-DMSX_STAMPS=2 -o build/libmsx_stamps2.so: the workgroup of the walker's FIRST segment writes the stamps instead)
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--walkers', type=int, default=256)
    ap.add_argument('--block', type=int, default=0)
    ap.add_argument('--npix', type=int, default=4096)
    ap.add_argument('--mode', default='logpost')
    ap.add_argument('--av0', action='store_true', help='all walkers at A_V = 0: no reddening, the blend loads R only')
    ap.add_argument('--path', default='fused', help='fused | linked')
    args = ap.parse_args()
    import torch
    from bench import build_workload
    from mcmc_spec_amd import _lib, synth
    from mcmc_spec_amd.engine import Engine
    eng = Engine(0)
    W = build_workload(eng, args.npix, args.npix >= 16384)
    dev = torch.device('cuda', 0)
    n = args.walkers
    thn = synth.draw_walkers(n, seed=3, tmin=W['tmin'], tmax=W['tmax'])
    if args.av0:
        thn[:, 2] = 0.0
    th = torch.from_numpy(thn).to(dev)
    lp = torch.empty(n, dtype=torch.float64, device=dev)
    st = torch.empty(n, dtype=torch.int32, device=dev)
    s = torch.cuda.current_stream(dev).cuda_stream
    eng.ctx.set_path({'linked': _lib.PATH_LINKED}.get(args.path, _lib.PATH_FUSED))
    for _ in range(20):
        eng.ctx.logprob_batch_dev(th.data_ptr(), n, 6, lp.data_ptr(), st.data_ptr(), s, {'logpost': _lib.MODE_LOGPOST, 'loglike': _lib.MODE_LOGLIKE}[args.mode], args.block)
    torch.cuda.synchronize()
    out = np.zeros((n, 16), dtype=np.uint64)
    fn = eng.ctx.lib.msx_diag_read_stamps
    fn.argtypes = [C.c_void_p, C.c_int64, C.c_void_p]
    fn.restype = C.c_int
    assert fn(eng.ctx.h, n, out.ctypes.data) == 0
    if args.path == 'linked':  # the stamps of each walker's LAST segment's workgroup (-DMSX_STAMPS=2: of its first)
        o = out.astype(np.int64)
        order = [0, 1, 2, 3, 4, 5, 6]
        names = ['phase0 recipe', 'phaseA blend (one segment)', 'fit sums + range + partials stored', 'first meeting (release, wait, acquire)',
                 'totals + running totals', 'locate + chi2 / candidates pass + stored']
        print('linked form, walkers {}: median entry -> pass stored {} cycles'.format(n, int(np.median(o[:, 6] - o[:, 0]))))
        for i, nm in enumerate(names):
            v = o[:, order[i + 1]] - o[:, order[i]]
            print('  {:40s} median {:8d}   max {:8d} cycles'.format(nm, int(np.median(v)), int(v.max())))
        fin = o[:, 15] - o[:, 7]
        print('  {:40s} median {:8d} cycles'.format('finisher: gather + rank + last lines', int(np.median(fin))))
        mine = o[:, 7] > o[:, 6]   # (this workgroup was the finisher of the last launch -- or of an earlier one)
        if mine.any():
            print('  {:40s} median {:8d} cycles  ({} of {} walkers)'.format('second meeting (when the finisher)', int(np.median((o[:, 7] - o[:, 6])[mine])), int(mine.sum()), n))
        print('  first start -> last end: {} cycles'.format(int(o[:, 15].max() - o[:, 0].min())))
        return
    d = np.diff(out[:, :8].astype(np.int64), axis=1)
    names = ['phase0 recipe', 'phaseA blend', 'fit sums + range (barriers)', 'median + chi2 pass', '-', 'tail', 'final reduce']
    tot = (out[:, 7] - out[:, 0]).astype(np.int64)
    print('walkers {} block {}: median total {} cycles'.format(n, args.block or 'auto', int(np.median(tot))))
    for i, nm in enumerate(names):
        print('  {:18s} median {:8d} cycles  ({:5.1f} %)'.format(nm, int(np.median(d[:, i])), 100 * np.median(d[:, i]) / np.median(tot)))
    # wave 0's chain through phase 0, in time order (recipe.h: stamps 9, 12, 13, 14, 10, 11)
    e = out[:, [0, 8, 9, 12, 13, 14, 10, 11, 1]].astype(np.int64)
    de = np.diff(e, axis=1)
    for i, nm in enumerate(['(entry)', 'theta load', 'iso interp', 'two brackets', 'presence mask', 'weights', 'gates + LDS store',
                            'barrier (other waves)']):
        print('    phase0/{:22s} median {:8d} cycles'.format(nm, int(np.median(de[:, i]))))
    med = np.zeros((n, 8), dtype=np.uint64)
    fm = eng.ctx.lib.msx_diag_read_med_stamps
    fm.argtypes = [C.c_void_p, C.c_int64, C.c_void_p]
    fm.restype = C.c_int
    assert fm(eng.ctx.h, n, med.ctypes.data) == 0
    dm = np.diff(med[:, :5].astype(np.int64), axis=1)
    for i, nm in enumerate(['(entry)', 'bin scan', 'chi2 + gather pass + barrier', 'rank']):
        v = int(np.median(dm[:, i]))
        if nm == 'rank' and v < 0:   # <= 64 candidates: ranked by four waves, picked behind the closing barrier (no stamp of its own)
            print('    median/rank: split over the idle waves (part of "closing barrier + combine + store" below)')
            continue
        print('    median/{:24s} median {:8d} cycles'.format(nm, v))
    if med[:, 6].max() > 0:
        print('    median/candidates: median {} max {}'.format(int(np.median(med[:, 6])), int(med[:, 6].max())))
    fin = (out[:, 15] - out[:, 7]).astype(np.int64)
    print('  closing barrier + combine + store: median {} cycles  (walker total {})'.format(int(np.median(fin)), int(np.median(fin + tot))))
    # the clock the CUs ran at: shader cycles per tick of the 100 MHz wall clock, first stamp -> last stamp of each walker
    wall = (med[:, 7] - med[:, 5]).astype(np.int64)
    ok = wall > 0
    if ok.any():
        mhz = (out[ok, 15] - out[ok, 0]).astype(np.float64) / wall[ok] * 100.0
        print('  shader clock during the walker chain: median {:.0f} MHz (min {:.0f}, max {:.0f}); walker chain {:.2f} us median, {:.2f} us max'.format(
            float(np.median(mhz)), float(mhz.min()), float(mhz.max()), float(np.median(wall[ok])) / 100.0, float(wall[ok].max()) / 100.0))
        w0, w1 = med[ok, 5].astype(np.int64), med[ok, 7].astype(np.int64)
        print('  wall clock, first walker start -> last walker end: {:.2f} us; starts spread over {:.2f} us, ends over {:.2f} us'.format(
            (w1.max() - w0.min()) / 100.0, (w0.max() - w0.min()) / 100.0, (w1.max() - w1.min()) / 100.0))
    span = int(out[:, 15].max() - out[:, 0].min())
    print('  first start -> last end: {} cycles'.format(span))


if __name__ == '__main__':
    main()
