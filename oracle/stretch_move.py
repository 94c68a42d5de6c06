"""TEST INFRASTRUCTURE -- CPU restatement of the ensemble move the reference's sampler block runs
(``emcee.EnsembleSampler`` with its default ``StretchMove``, mft6.py:1490-1529; emcee itself is not installed
here and its source is not under /root/reference, so this follows the published algorithm: Goodman & Weare
2010, eq. 7-10, as emcee 3's red-blue scheme applies it).  Parity unpinned against emcee itself.

Only ``tests/`` import this module.  It is written walker by walker, in plain Python loops, on purpose: it
shares no code with ``mcmc_spec_amd.sampler`` (whose loops are vectorised) nor with the fused device kernel.

One iteration = two half-steps.  In half-step k the walkers ``sidx[k]`` move, each against ONE walker drawn
from the complementary half ``cidx[k]``:

    q      = c - (c - s) * z                      z ~ g(z) ∝ 1/sqrt(z) on [1/a, a]:  z = ((a-1) u + 1)^2 / a
    accept   iff  ln u' < (ndim - 1) ln z + ln p(q) - ln p(s)

The random numbers are INPUTS (the caller draws them), so that a chain produced by any implementation from
the same numbers can be compared with this one value by value.
"""
import math

import numpy as np


def stretch_factor(u, a=2.0):
    """z from a uniform u in [0, 1): inverse CDF of g(z) ∝ z^(-1/2) on [1/a, a]."""
    return ((a - 1.0) * u + 1.0) ** 2 / a


def half_step(coords, logp, s_idx, c_idx, partner, zz, logu, log_prob_fn, naccept=None):
    """Advance the walkers ``s_idx`` in place.  ``partner[i]`` indexes ``c_idx``; ``zz[i]`` and ``logu[i]``
    are walker i's stretch factor and ln of its accept draw.  Proposals are all built from the ensemble as it
    stands BEFORE the half-step (the two halves are disjoint, so this equals emcee's batch proposal), then
    evaluated with ``log_prob_fn(q[n, ndim]) -> [n]`` and accepted one walker at a time."""
    ndim = coords.shape[1]
    props = []
    for i in range(len(s_idx)):
        s = coords[s_idx[i]]
        c = coords[c_idx[partner[i]]]
        q = np.empty(ndim)
        for d in range(ndim):
            q[d] = c[d] - (c[d] - s[d]) * zz[i]
        props.append(q)
    new_lp = np.asarray(log_prob_fn(np.array(props)), dtype=float)
    for i in range(len(s_idx)):
        j = s_idx[i]
        lnpdiff = (ndim - 1.0) * math.log(zz[i]) + new_lp[i] - logp[j]   # nan (-inf - -inf) compares False
        if logu[i] < lnpdiff:
            coords[j] = props[i]
            logp[j] = new_lp[i]
            if naccept is not None:
                naccept[j] += 1


def run_chain(coords0, logp0, draws, log_prob_fn):
    """``draws`` = (sidx, cidx, partner, zz, logu), each of shape (nsteps, 2, nwalkers/2).  Returns
    (chain [nsteps][nw][ndim], logp chain [nsteps][nw], naccept [nw])."""
    sidx, cidx, partner, zz, logu = draws
    coords = np.array(coords0, dtype=float)
    logp = np.array(logp0, dtype=float)
    nacc = np.zeros(len(coords), dtype=np.int64)
    chain, lpc = [], []
    for t in range(len(zz)):
        for k in (0, 1):
            half_step(coords, logp, sidx[t, k], cidx[t, k], partner[t, k], zz[t, k], logu[t, k], log_prob_fn, nacc)
        chain.append(coords.copy())
        lpc.append(logp.copy())
    return np.array(chain), np.array(lpc), nacc
