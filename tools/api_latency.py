#!/usr/bin/env python3
"""Host-visible latency of one drop-in call (PCIe-inclusive; what a host-driven sampler pays per half-step)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from bench import build_workload
    from mcmc_spec_amd import synth
    from mcmc_spec_amd.engine import Engine
    eng = Engine(0)
    W = build_workload(eng, 4096, False)
    for n in (128, 256, 2048):
        th = synth.draw_walkers(n, seed=1, tmin=W['tmin'], tmax=W['tmax'])
        eng.logposterior(th)
        t0 = time.perf_counter()
        for _ in range(1000):
            eng.logposterior(th)
        print('Engine.logposterior, {:5d} walkers x 4096 px: {:.1f} us per call (host pointers in and out)'.format(
            n, (time.perf_counter() - t0) / 1000 * 1e6), flush=True)


if __name__ == '__main__':
    main()
