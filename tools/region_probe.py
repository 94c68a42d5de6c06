#!/usr/bin/env python3
"""Where do the microseconds of a SHORT timed region go?  (VERDICT r3 #2a: the driver's 20-step shape reads 16-16.5 us
per "kernel" where a 200-step run of the same build on the same box reads 14.6.)

Runs bench.py's step loop -- `head` plain launches, then one hipGraph of the rest -- between synchronize() calls, like
bench.py's timed region, and reports per repetition:
    wall_us        host clock, synchronize -> synchronize (what bench.py reports / steps)
    issue_us       host clock until the last launch call returned
    gpu_span_us    HIP events: before the first launch -> after the last kernel (stream timeline)
    head_us        ... the plain launches alone;   graph_us  the graph's kernels alone (what bench.py divides by `chunk`)
    start_lat_us   wall - gpu_span: host-side latency outside the stream's busy time (first dispatch + completion wake-up)
Environment knobs are the process's own (e.g. HSA_ENABLE_INTERRUPT=0 python tools/region_probe.py ...).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--heads', default='0,2,4,8,20')
    ap.add_argument('--reps', type=int, default=7)
    ap.add_argument('--walkers', type=int, default=256)
    ap.add_argument('--ramp', type=int, default=2000)
    ap.add_argument('--tag', default='')
    args = ap.parse_args()
    import ctypes as C
    import torch
    from bench import build_workload
    from mcmc_spec_amd import _lib, synth
    from mcmc_spec_amd.engine import Engine
    dev = torch.device('cuda', 0)
    eng = Engine(0)
    W = build_workload(eng, 4096, False)
    n = args.walkers
    thetas = [torch.from_numpy(synth.draw_walkers(n, seed=3 + b, tmin=W['tmin'], tmax=W['tmax'])).to(dev) for b in range(4)]
    logp = [torch.empty(n, dtype=torch.float64, device=dev) for _ in range(2)]
    st = [torch.zeros(n, dtype=torch.int32, device=dev) for _ in range(2)]
    stream = torch.cuda.current_stream(dev)
    fn = eng.ctx.lib.msx_logprob_batch_dev

    def calls(sp):
        return [[(eng.ctx.h, _lib.MODE_LOGPOST, C.c_void_p(t.data_ptr()), n, 6, C.c_void_p(logp[b].data_ptr()),
                  C.c_void_p(st[b].data_ptr()), C.c_void_p(sp), 0) for b in range(2)] for t in thetas]

    tab0 = calls(stream.cuda_stream)

    def launch(i, tab=tab0):
        assert fn(*tab[i % 4][i & 1]) == 0

    for head in [int(x) for x in args.heads.split(',')]:
        head = min(head, args.steps)
        chunk = args.steps - head
        g = None
        if chunk > 0:
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=torch.cuda.Stream(dev)):
                tab = calls(torch.cuda.current_stream(dev).cuda_stream)
                for i in range(chunk):
                    launch(i, tab)
            g.replay()
            torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        rows = []
        for rep in range(args.reps + 1):
            for i in range(args.ramp):
                launch(i)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ev[0].record(stream)
            for i in range(head):
                launch(i)
            ev[1].record(stream)
            if g is not None:
                g.replay()
            ev[2].record(stream)
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            if rep == 0:
                continue
            rows.append(dict(wall_us=(t2 - t0) * 1e6, issue_us=(t1 - t0) * 1e6, gpu_span_us=ev[0].elapsed_time(ev[2]) * 1e3,
                             head_us=ev[0].elapsed_time(ev[1]) * 1e3, graph_us=ev[1].elapsed_time(ev[2]) * 1e3))
        med = {k: float(np.median([r[k] for r in rows])) for k in rows[0]}
        med['start_lat_us'] = med['wall_us'] - med['gpu_span_us']
        out = dict(tag=args.tag, steps=args.steps, head=head, graph_steps=chunk, us_per_step=med['wall_us'] / args.steps,
                   graph_us_per_step=(med['graph_us'] / chunk) if chunk else None,
                   head_us_per_step=(med['head_us'] / head) if head else None, **med,
                   env={k: os.environ.get(k) for k in ('HSA_ENABLE_INTERRUPT', 'HIP_FORCE_DEV_KERNARG', 'GPU_MAX_HW_QUEUES')})
        print(json.dumps(out), flush=True)


if __name__ == '__main__':
    main()
