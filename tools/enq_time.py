import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
from bench import build_workload
from mcmc_spec_amd import synth, _lib
from mcmc_spec_amd.engine import Engine
from mcmc_spec_amd.sampler import EnsembleSampler
import torch
eng = Engine(0)
W = build_workload(eng, 4096, False)
nw = 256
p0 = synth.draw_walkers(nw, seed=9, tmin=W['tmin'], tmax=W['tmax'])
lp0 = eng.logposterior(p0)
c = eng.ctx
draw = EnsembleSampler(nw, 6, lambda x: x, seed=1)
m = 100
c.sampler_begin(_lib.MODE_LOGPOST, p0.copy(), lp0.copy(), m)
for rep in range(4):
    arrays = draw._draw_steps(m)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    c.sampler_enqueue(0, *arrays)
    t1 = time.perf_counter()
    c.sampler_collect(0, m)
    t2 = time.perf_counter()
    print('enqueue of %d steps: host %.1f us (%.2f us per launch); until collected %.1f us (%.2f us per step)' % (m, (t1 - t0) * 1e6, (t1 - t0) * 1e6 / (2 * m), (t2 - t0) * 1e6, (t2 - t0) * 1e6 / m))
c.sampler_end()
