#!/usr/bin/env python3
"""Longer device-resident vs host-driven chains (several ensemble sizes and chunk sizes, incl. 512 walkers whose
half-steps take the two-workgroups-per-CU kernel variant): chains, log-probabilities and acceptance must be identical."""
import sys, numpy as np
sys.path.insert(0, '.')
from bench import build_workload
from mcmc_spec_amd import synth
from mcmc_spec_amd.engine import Engine
from mcmc_spec_amd.sampler import DeviceEnsembleSampler, EnsembleSampler
eng = Engine(0); W = build_workload(eng, 4096, False)
for nw, steps, chunk in ((64, 1500, 37), (256, 600, 100), (512, 300, 64)):
    p0 = synth.draw_walkers(nw, seed=9, tmin=W['tmin'], tmax=W['tmax'])
    h = EnsembleSampler(nw, 6, eng.logposterior, vectorize=True, seed=4); h.run_mcmc(p0, steps)
    d = DeviceEnsembleSampler(nw, 6, eng, seed=4, chunk=chunk); d.run_mcmc(p0, steps)
    print(nw, steps, chunk, 'chain equal', np.array_equal(h.chain, d.chain), 'logp equal', np.array_equal(h.get_log_prob(), d.get_log_prob()),
          'acc equal', np.array_equal(h.acceptance_fraction, d.acceptance_fraction), float(d.acceptance_fraction.mean()))
