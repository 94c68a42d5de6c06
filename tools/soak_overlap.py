#!/usr/bin/env python3
"""Soak test of the overlapped half-steps: a long chain with the overlap on and the same chain with plain launches must end
in the same state, bit for bit (any race in the version protocol would show as a diverging chain or a status).

    python3 tools/soak_overlap.py --walkers 256 --steps 200000
"""
import argparse
import os
import subprocess
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(nw, steps, seed, rng='host'):
    from bench import build_workload
    from mcmc_spec_amd import synth
    from mcmc_spec_amd.engine import Engine
    from mcmc_spec_amd.sampler import DeviceEnsembleSampler
    eng = Engine(0)
    W = build_workload(eng, 4096, False)
    p0 = synth.draw_walkers(nw, seed=9, tmin=W['tmin'], tmax=W['tmax'])
    s = DeviceEnsembleSampler(nw, 6, eng, seed=seed, chunk=200, rng=rng)
    t0 = time.perf_counter()
    st = s.run_mcmc(p0, steps, store=False)
    dt = time.perf_counter() - t0
    return st.coords, st.log_prob, s.acceptance_fraction, bool(s.overlapped), dt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--walkers', type=int, default=256)
    ap.add_argument('--steps', type=int, default=200000)
    ap.add_argument('--child', default='')
    ap.add_argument('--rng', default='host', help='host | device: who draws the randomness')
    args = ap.parse_args()
    if args.child:
        c, lp, acc, ov, dt = run(args.walkers, args.steps, 5, args.rng)
        np.savez(args.child, coords=c, lp=lp, acc=acc, ov=ov, dt=dt)
        return
    outs = []
    for ov in ('1', '0'):   # (a process each: the choice is read from the environment by the library)
        f = '/tmp/soak_{}.npz'.format(ov)
        env = dict(os.environ, MSX_SMP_OVERLAP=ov)
        subprocess.check_call([sys.executable, os.path.abspath(__file__), '--walkers', str(args.walkers), '--steps', str(args.steps),
                               '--child', f, '--rng', args.rng], env=env)
        outs.append(np.load(f))
    a, b = outs
    same = np.array_equal(a['coords'], b['coords']) and np.array_equal(a['lp'], b['lp']) and np.array_equal(a['acc'], b['acc'])
    print('randomness ' + args.rng + ', walkers {} iterations {}: overlapped {} ({:.1f} us per iteration) against plain {} ({:.1f} us): final state {}; acceptance {:.3f}'.format(
        args.walkers, args.steps, bool(a['ov']), float(a['dt']) / args.steps * 1e6, bool(b['ov']), float(b['dt']) / args.steps * 1e6,
        'IDENTICAL' if same else 'DIFFERENT', float(a['acc'].mean())))
    sys.exit(0 if same and bool(a['ov']) and not bool(b['ov']) else 1)


if __name__ == '__main__':
    main()
