"""Pre-optimiser: ``fit_spec`` / ``optimize_fit`` (mft6.py:856-1137, :1686-1765) with every
start point advancing in lock-step so that each proposal round is ONE batched GPU launch instead of
``mp.Pool(15).apply_async`` per start point.

The chi^2 of a proposal (composite, reddening, resample, median scale against the chain's own
normalised data, contrast + photometry terms, spectrum weight 3) comes from the HIP kernel
(``msx_opt_init`` / ``msx_opt_step``).  What stays on the host is the reference's per-chain state
machine -- step-size schedule, bounds test, repair-loop counters, accept rule -- and the three scalar
``opt_prior`` terms, all restated below line by line.
"""
from __future__ import annotations

import os

import numpy as np

from . import staging


def opt_prior_sum(vals, pval, psig):
    """``opt_prior`` for list arguments (mft6.py:845-853): sum of ((v-p)/s)^2 over entries with p != 0."""
    tot = 0.0
    for v, p, s in zip(vals, pval, psig):
        if p != 0:
            tot = tot + ((float(v) - p) / s) ** 2
    return tot


def opt_prior_one(val, pval, psig):
    """``opt_prior`` with one-element lists (mft6.py:839-843)."""
    return ((float(val) - float(pval)) / float(psig)) ** 2


class _Chain:
    pass


def _av_lookup(av_table, dist_pc):
    edges, mu, sig = av_table
    b = int(np.clip(np.searchsorted(edges, dist_pc, side='right') - 1, 0, len(mu) - 1))
    s = sig[b]
    return mu[b], (0.05 if s == 0 else s)  # mft6.py:927-928, 994-995


def _step_sizes(nspec, rad0, dist0, fine):
    if not fine:  # mft6.py:952-955
        return [[250.0] * nspec, [0.05], [0.1 * r for r in rad0], [(0.02 if nspec == 2 else 0.05) * dist0]]
    return [[20.0] * nspec, [0.01], [0.05 * r for r in rad0], [(0.005 if nspec == 2 else 0.01) * dist0]]  # :970-973


def _in_bounds(var, tlim):
    """mft6.py:981-982."""
    return (all(min(tlim) < v < max(tlim) for v in var[0]) and 0 <= var[1] and 0.05 <= var[2][0] <= 1.5
            and 0.05 < var[2][1] < 1 and 1 / 10 > var[3] > 1 / 3000 and var[1] >= 0)


def _repair_count(var, tlim, total_n, cap):
    """The out-of-bounds branch (mft6.py:1071-1103).  The repaired proposal is thrown away by the
    reference (the next trip draws afresh from ``gi``); only ``total_n`` survives, so only it is kept."""
    total_n += 1
    T, av, rad, plx = [np.array(v, dtype=float).copy() for v in var]
    while any(v < min(tlim) for v in T) and total_n < cap:
        total_n += 1
        T[np.where(T < min(tlim))] += 100
    while any(v > max(tlim) for v in T) and total_n < cap:
        total_n += 1
        T[np.where(T > max(tlim))] -= 100
    while T[0] < T[1] and total_n < cap:
        total_n += 1
        T[1] -= 100
    while av < 0 and total_n < cap:
        total_n += 1
        av += 0.1
    while any(v < 0.05 for v in rad) and total_n < cap:
        total_n += 1
        rad[np.where(rad < 0.05)] += 0.01
    while plx > 1 / 100 and total_n < cap:
        total_n += 1
        plx -= 0.01 * np.abs(plx)
    while plx < 1 / 3000 and total_n < cap:
        total_n += 1
        plx += 0.01 * np.abs(plx)
    return total_n


def default_propose(gi, si, rng):
    """``make_varied_param`` (mft6.py:211-228): one normal draw per parameter group."""
    return [rng.normal(gi[n], si[n]) for n in range(len(gi))]


def fit_spec_batch(engine, starts, tlim, dist_prior, matrix, av_table, nspec=2, steps=200, dist_fit=True,
                   rad_prior=False, rngs=None, propose=default_propose, dirname=None, first_index=0):
    """Advance every start point of ``starts`` (rows ``[T.., A_V, rad.., plx]``) through ``fit_spec``'s
    random-walk chi^2 descent in lock-step.  The dataset must already be staged on ``engine``.

    Returns a list of ``(best_line, best_chi2, chain)`` in start order; ``best_line`` is the text
    ``fit_spec`` returns (mft6.py:1118,1137)."""
    starts = np.atleast_2d(np.asarray(starts, dtype=float))
    nch = len(starts)
    ndim = 2 * nspec + 2
    pprior, psig = dist_prior
    if rngs is None:
        rngs = [np.random.default_rng() for _ in range(nch)]
    like0, status = engine.ctx.opt_init(starts)
    from .engine import _raise_for_status
    _raise_for_status(status, starts)
    chains = []
    for c in range(nch):
        ch = _Chain()
        T, av, rad, plx = list(starts[c, :nspec]), float(starts[c, nspec]), list(starts[c, nspec + 1:2 * nspec + 1]), \
            float(starts[c, 2 * nspec + 1])
        cs = like0[c]
        if dist_fit:
            cs = cs + opt_prior_one(plx, pprior, psig)  # mft6.py:909-910
        if rad_prior:  # mft6.py:912-922
            mr = staging.isochrone_radius(np.array(T), matrix)
            cs = cs + opt_prior_sum(rad, [mr[0]] + [m / mr[0] for m in mr[1:]], [0.05 * r for r in rad])
        mu, sg = _av_lookup(av_table, 1.0 / plx)
        cs = cs + opt_prior_one(av, mu, sg)  # mft6.py:924-929
        ch.chi = cs
        ch.rad0, ch.dist0 = list(rad), plx
        ch.gi = [np.array(T), av, np.array(rad), plx]
        ch.n, ch.total_n = 0, 0
        ch.savechi = [cs]
        ch.savetest = []
        ch.sp = [[np.array(T), av, np.array(rad), plx]]
        ch.rng = rngs[c]
        ch.pending = None
        ch.done = False
        chains.append(ch)
    cap = 50 * steps
    while True:
        batch, owners = [], []
        for ci, ch in enumerate(chains):
            if ch.done:
                continue
            # draw until this chain has an in-bounds proposal to evaluate or runs out of budget
            while ch.n < steps and ch.total_n < cap:
                ch.si = _step_sizes(nspec, ch.rad0, ch.dist0, ch.n > steps / 2)
                var = propose(ch.gi, ch.si, ch.rng)
                if _in_bounds(var, tlim):
                    if nspec == 3:
                        while var[2][2] >= var[2][1] or var[2][2] < 0:  # mft6.py:984-985
                            var[2][2] = var[2][1] * 0.9
                    ch.total_n += 1
                    ch.n += 1
                    ch.pending = var
                    break
                ch.total_n = _repair_count(var, tlim, ch.total_n, cap)
            if ch.pending is None:
                ch.done = True
                continue
            v = ch.pending
            batch.append(np.concatenate([np.ravel(v[0]), np.ravel(v[1]), np.ravel(v[2]), np.ravel(v[3])]))
            owners.append(ci)
        if not batch:
            break
        like, status = engine.ctx.opt_step(np.array(batch), np.array(owners, dtype=np.int32))
        _raise_for_status(status, np.array(batch))
        for k, ci in enumerate(owners):
            ch = chains[ci]
            var = ch.pending
            ch.pending = None
            plx = float(np.ravel(var[3])[0])
            av = float(np.ravel(var[1])[0])
            mu, sg = _av_lookup(av_table, 1.0 / plx)  # mft6.py:991-995
            test = like[k] + opt_prior_one(av, mu, sg)  # mft6.py:1030
            if dist_fit:
                test = test + opt_prior_one(plx, pprior, psig)  # mft6.py:1034-1035
            if rad_prior:  # mft6.py:1037-1050: sigma = the current radius step sizes
                mr = staging.isochrone_radius(np.asarray(var[0], dtype=float), matrix)
                test = test + opt_prior_sum(var[2], [mr[0]] + [m / mr[0] for m in mr[1:]], np.array(ch.si[2]))
            if test < ch.chi:  # mft6.py:1053-1063
                ch.gi = var
                ch.chi = test
                ch.n = steps / 2 + 1 if ch.n > steps / 2 else 0
            ch.sp.append(ch.gi)
            ch.savechi.append(ch.chi)
            ch.savetest.append(test)
    out = []
    for ci, ch in enumerate(chains):
        g = ch.gi
        vals = list(np.ravel(g[0])) + [float(np.ravel(g[1])[0])] + list(np.ravel(g[2])) + [float(np.ravel(g[3])[0])]
        line = ' '.join(str(v) for v in vals[:nspec]) + ' ' + str(float(vals[nspec])) + ' ' + \
            ' '.join(str(v) for v in vals[nspec + 1:2 * nspec + 1]) + ' ' + str(float(vals[ndim - 1])) + '\n'
        if dirname:
            with open(os.path.join(dirname, 'params{}.txt'.format(first_index + ci)), 'a') as f:
                for row in ch.sp[1:]:
                    rv = list(np.ravel(row[0])) + [float(np.ravel(row[1])[0])] + list(np.ravel(row[2])) + \
                        [float(np.ravel(row[3])[0])]
                    f.write(' '.join(str(x) for x in rv) + '\n')
            with open(os.path.join(dirname, 'chisq{}.txt'.format(first_index + ci)), 'a') as f:
                for n in range(1, len(ch.savechi)):
                    f.write('{} {}\n'.format(ch.savechi[n], ch.savetest[n - 1]))
        out.append((line, ch.savechi[-1], ch))
    return out


def optimize_fit(dirname, data, err, specs, nwalk, fr, dist_arr, av, res, ctm, ptm, tmi, tma, vs, matrix, ra, dec,
                 cutoff=2, nspec=2, nstep=200, nburn=20, con=True, models='btsettl', err2=0, dist_fit=True,
                 rad_prior=False, seed=None, av_table=None, bands=None):
    """``optimize_fit`` (mft6.py:1686-1765): random start points, ``fit_spec`` on each, results appended to
    ``optimize_res.txt`` / ``optimize_cs.txt``.  All ``nwalk`` chains run as one batch on the GPU."""
    from . import mft6 as api
    rng = np.random.default_rng(seed)
    t = [float(k.split(', ')[0]) for k in specs.keys() if k != 'wl']
    tmin, tmax = min(t), max(t)
    rmin, rmax = 0.05, 1
    t1 = rng.uniform(tmin, tmax, nwalk)  # mft6.py:1712-1722
    t2 = np.array([rng.uniform(tmin, tt) for tt in t1])
    cols = [t1, t2]
    if nspec == 3:
        cols.append(np.array([rng.uniform(tmin, tt) for tt in t2]))
    e1 = rng.uniform(0.1, 0.5, nwalk)  # mft6.py:1724
    rg1 = rng.uniform(rmin, rmax, nwalk)  # mft6.py:1727-1736
    rg2 = np.array([rng.uniform(rmin, r) / r for r in rg1])
    rcols = [rg1, rg2]
    if nspec == 3:
        rcols.append(np.array([rng.uniform(rmin, r) / r for r in rg2]))
    dist = np.abs(rng.normal(dist_arr[0], dist_arr[1], nwalk))  # mft6.py:1741-1743
    starts = np.column_stack(cols + [e1] + rcols + [dist])
    reg = [min(data[0]), max(data[0])]
    eng = api._engine_for(specs)
    eng.stage_problem(data, err, fr, reg, ctm, ptm, tmi, tma, matrix, nspec=nspec, bands=bands or api._BANDS)
    eng._problem_key = None
    table = av_table if av_table is not None else api._AV_TABLE
    if table is None:
        raise RuntimeError('optimize_fit needs the A_V(distance) table (mcmc_spec_amd.mft6.set_av_prior)')
    rngs = [np.random.default_rng(rng.integers(1 << 62)) for _ in range(nwalk)]
    res_ = fit_spec_batch(eng, starts, [tmin, tmax], (dist_arr[0], dist_arr[1]), matrix, table, nspec=nspec,
                          steps=nstep, dist_fit=dist_fit, rad_prior=rad_prior, rngs=rngs, dirname=dirname)
    with open(os.path.join(dirname, 'optimize_res.txt'), 'a') as f:  # mft6.py:1757-1763
        for line, cs, _ in res_:
            f.write(line)
    with open(os.path.join(dirname, 'optimize_cs.txt'), 'a') as f:
        for line, cs, _ in res_:
            f.write(str(cs) + '\n')
    return res_
