#!/usr/bin/env python3
"""A single DEPENDENT chain (what emcee actually runs): host-driven stretch move over the drop-in
log-posterior vs the device-resident loop (msx_sampler_run), BASELINE config 2 (256 walkers x 4096 px)."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument('--walkers', default='256,2048')
    ap.add_argument('--device-steps', type=int, default=4000)
    ap.add_argument('--chunk', type=int, default=100)
    ap.add_argument('--no-host', action='store_true', help='skip the host-driven loop (profiling runs)')
    ap.add_argument('--rng', default='host', help="host | device: who draws the move's randomness (DeviceEnsembleSampler)")
    args = ap.parse_args()
    from bench import build_workload
    from mcmc_spec_amd import synth
    from mcmc_spec_amd.engine import Engine
    from mcmc_spec_amd.sampler import DeviceEnsembleSampler, EnsembleSampler
    eng = Engine(0)
    W = build_workload(eng, 4096, False)
    for nw in [int(x) for x in args.walkers.split(',')]:
        p0 = synth.draw_walkers(nw, seed=9, tmin=W['tmin'], tmax=W['tmax'])
        steps = 200
        th = float('nan')
        if not args.no_host:
            host = EnsembleSampler(nw, 6, eng.logposterior, vectorize=True, seed=1)
            host.run_mcmc(p0, 5)
            t0 = time.perf_counter()
            host.run_mcmc(p0, steps)
            th = time.perf_counter() - t0
        dev = DeviceEnsembleSampler(nw, 6, eng, seed=1, chunk=args.chunk, rng=args.rng)
        dev.run_mcmc(p0, 5)
        dsteps = args.device_steps
        t0 = time.perf_counter()
        dev.run_mcmc(p0, dsteps, store=False)
        td = time.perf_counter() - t0
        print(json.dumps(dict(walkers=nw, steps=steps, host_loop_us_per_step=th / steps * 1e6,
                              device_us_per_step=td / dsteps * 1e6, device_steps=dsteps, randomness=args.rng, overlapped=bool(dev.overlapped), host_evals_per_s=nw * steps / th,
                              device_evals_per_s=nw * dsteps / td,
                              acceptance=float(dev.acceptance_fraction.mean()))), flush=True)


if __name__ == '__main__':
    main()
