#!/usr/bin/env python3
"""How much of the big-batch rate is recovered at 256 walkers/launch when independent launches are kept in
flight on several streams (e.g. several independent ensembles / targets per GPU)."""
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    from bench import build_workload
    from mcmc_spec_amd import _lib, synth
    from mcmc_spec_amd.engine import Engine
    dev = torch.device('cuda', 0)
    eng = Engine(0)
    W = build_workload(eng, 4096, False)
    n, steps = 256, 600
    for nstreams in (1, 2, 3, 4):
        for block in (256, 512):
            streams = [torch.cuda.Stream(dev) for _ in range(nstreams)]
            th = [torch.from_numpy(synth.draw_walkers(n, seed=3 + k, tmin=W['tmin'], tmax=W['tmax'])).to(dev)
                  for k in range(nstreams)]
            lp = [torch.empty(n, dtype=torch.float64, device=dev) for _ in range(nstreams)]
            st = [torch.empty(n, dtype=torch.int32, device=dev) for _ in range(nstreams)]
            fn = eng.ctx.lib.msx_logprob_batch_dev
            calls = [(eng.ctx.h, _lib.MODE_LOGPOST, C.c_void_p(th[k].data_ptr()), n, 6, C.c_void_p(lp[k].data_ptr()),
                      C.c_void_p(st[k].data_ptr()), C.c_void_p(streams[k].cuda_stream), block) for k in range(nstreams)]
            for i in range(50):
                fn(*calls[i % nstreams])
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(steps):
                fn(*calls[i % nstreams])
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            print(json.dumps(dict(streams=nstreams, block=block, us_per_step=dt / steps * 1e6, evals_per_s=n * steps / dt)),
                  flush=True)


if __name__ == '__main__':
    main()
