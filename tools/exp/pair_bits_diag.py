#!/usr/bin/env python3
"""Diagnostic: where the pair form's bits leave the fused kernel's (golden case B / A), per mode."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'tests'))
from common import golden_case
from test_gpu_parity import make_engine
from mcmc_spec_amd import _lib
c = golden_case('B')
eng = make_engine(c)
def forms(fn, *a, **k):
    eng.ctx.set_path(_lib.PATH_FUSED); f = fn(*a, **k)
    eng.ctx.set_path(_lib.PATH_PAIR); p = fn(*a, **k)
    return f, p
for name, fn, th in (('loglike B', eng.loglikelihood, c.theta), ('logpost A-theta', eng.logposterior, golden_case('A').g['theta_post'])):
    f, p = forms(fn, th)
    fin = np.isfinite(f)
    d = np.abs(f[fin] - p[fin]) / np.abs(f[fin])
    print(name, 'n', len(f), 'differ', int((f[fin] != p[fin]).sum()), 'max rel', d.max() if len(d) else 0, 'nan/inf equal', np.array_equal(np.isfinite(f), np.isfinite(p)))
    bad = np.where(fin)[0][f[fin] != p[fin]][:5]
    for i in bad: print('   walker', i, th[i], f[i], p[i])
