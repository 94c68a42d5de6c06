"""The PAIR form of the hot path (mcmc_spec_amd/csrc/pair_kernel.h): two walkers of one grid cell per workgroup, one
set of row loads, model values in registers -- against the fused kernel.

The contract is the same as for every other form: a walker's value has the SAME BITS whichever form evaluates it --
whoever its partner is, whether it was paired at all, whatever the workgroup size.
"""
import numpy as np
import pytest

import common
from common import golden_case, rel_err
from test_gpu_parity import make_engine

pytestmark = pytest.mark.gpu
TIGHT = 1e-9


def forms(eng, fn, *a, **k):
    """fused, pair"""
    from mcmc_spec_amd import _lib
    eng.ctx.set_path(_lib.PATH_FUSED)
    out = [fn(*a, **k)]
    eng.ctx.set_path(_lib.PATH_PAIR)
    out.append(fn(*a, **k))
    eng.ctx.set_path(_lib.PATH_AUTO)
    return out


def same_bits(outs):
    return all(np.array_equal(outs[0], o, equal_nan=True) for o in outs[1:])


def test_pair_matches_reference_golden_and_fused_bits():
    c = golden_case('B')
    eng = make_engine(c)
    outs = forms(eng, eng.loglikelihood, c.theta)
    assert same_bits(outs)
    assert rel_err(outs[1], c.g['B_loglike']).max() < TIGHT
    assert same_bits(forms(eng, eng.loglikelihood, c.theta, optimize=True))
    th = golden_case('A').g['theta_post']        # includes walkers outside the prior box: -inf from the recipe alone
    outs = forms(eng, eng.logposterior, th)
    assert same_bits(outs) and np.isinf(outs[0]).sum() == 8


@pytest.mark.parametrize('nphot', [1, 3, 5])
def test_pair_band_terms_with_odd_band_counts(nphot):
    """The planner reads the band table two bands to a 16-byte load when a row holds an even number of bands and band by
    band when it does not, four bands to a trip (recipe.h, band_terms_scalar2): 2 contrast filters + 1 / 3 / 5 photometric
    bands = rows of 3 / 5 / 7 -- the odd path, one trip and two -- against the fused kernel (bits) and the oracle."""
    import copy
    from mcmc_spec_amd.engine import Engine
    from mcmc_spec_amd import bands
    c = copy.copy(golden_case('B'))
    c.ptm = [x[:nphot] for x in c.ptm]
    c.fr = [c.fr[0], c.fr[1], c.fr[2], c.fr[3][:nphot], c.fr[4][:nphot], c.fr[5][:nphot]]
    c.tmi, c.tma = common.tm_extrema(c.ctm, c.ptm)
    eng = Engine(0)
    eng.stage_specs(c.specs)
    eng.stage_problem(c.data, c.err, c.fr, c.r, c.ctm, c.ptm, c.tmi, c.tma, c.matrix, nspec=2,
                      bands=bands.make_bands(c.tables, *c.vega), av_table=common.av_table_exact(), tmin=c.tmin, tmax=c.tmax,
                      prior=c.prior)
    th = np.repeat(c.theta, 2, axis=0)
    for fn in (eng.loglikelihood, eng.logposterior):
        outs = forms(eng, fn, th)
        assert same_bits(outs) and np.isfinite(outs[0]).sum() >= len(c.theta)
    want = np.array([common.orc.loglikelihood(list(t), c.fr, 2, c.data, c.err, c.r, c.specs, c.ctm, c.ptm, c.tmi, c.tma,
                                              c.matrix, bandlib=c.bandlib) for t in c.theta[:4]])
    eng.ctx.set_path(__import__('mcmc_spec_amd._lib', fromlist=['x']).PATH_PAIR)
    assert rel_err(eng.loglikelihood(th)[:8:2], want).max() < TIGHT


def test_pair_problem_staged_without_extinction():
    """`a=False` (mft6.py:1161): no walker is reddened; the pair kernel's only variant loads the extinction terms and
    must still return the unreddened values, bit for bit."""
    from mcmc_spec_amd.engine import Engine
    from mcmc_spec_amd import bands
    c = golden_case('B')
    eng = Engine(0)
    eng.stage_specs(c.specs)
    eng.stage_problem(c.data, c.err, c.fr, c.r, c.ctm, c.ptm, c.tmi, c.tma, c.matrix, nspec=2,
                      bands=bands.make_bands(c.tables, *c.vega), use_av=False)
    th = np.repeat(c.theta, 3, axis=0)
    outs = forms(eng, eng.loglikelihood, th)
    assert same_bits(outs) and np.all(np.isfinite(outs[0]))
    want = np.array([common.orc.loglikelihood(list(t), c.fr, 2, c.data, c.err, c.r, c.specs, c.ctm, c.ptm, c.tmi, c.tma,
                                              c.matrix, bandlib=c.bandlib, av=False) for t in c.theta[:4]])
    assert rel_err(outs[1][:12:3], want).max() < TIGHT


def test_pair_values_do_not_depend_on_the_partner():
    """Many walkers in few grid cells, then the same walkers shuffled, duplicated, with an odd count, in tiny batches:
    identical bits every time.  Includes a Teff exactly on a node (duplicated corners with weight 0), A_V = 0 (the
    unreddened value from a kernel that loads the extinction terms) and a walker outside the prior box."""
    c = golden_case('B')
    rng = np.random.default_rng(17)
    th = c.theta[0] + rng.normal(size=(701, 6)) * np.array([120, 120, 0.05, 0.05, 0.05, 1e-4])
    th[:, 0:2] = np.clip(th[:, 0:2], 3000.0, 4200.0)
    th[:, 2] = np.abs(th[:, 2])
    th[:, 3:5] = np.clip(th[:, 3:5], 0.05, 1.4)
    th[5, 0] = 3800.0
    th[6, 2] = 0.0
    th[7, 1] = 2999.0
    th[8] = th[9]                        # identical neighbours: certainly one cell
    th[10, :2] = th[11, :2]              # same temperatures, other radii / extinction / parallax
    eng = make_engine(c)
    outs = forms(eng, eng.logposterior, th)
    assert same_bits(outs) and np.isinf(outs[0][7]) and np.isfinite(outs[0]).sum() >= 690
    f = outs[0]
    from mcmc_spec_amd import _lib
    eng.ctx.set_path(_lib.PATH_PAIR)
    perm = rng.permutation(len(th))
    assert np.array_equal(eng.logposterior(th[perm]), f[perm])
    order = np.lexsort((th[:, 1] // 100, th[:, 0] // 100))     # neighbours in the batch mostly share their cell
    assert np.array_equal(eng.logposterior(th[order]), f[order])
    npairs, nsingles = eng.ctx.pair_stats()
    assert 2 * npairs + nsingles == np.isfinite(f).sum() and npairs > 300   # every live walker exactly once
    for n in (1, 2, 3, 8, 9):
        assert np.array_equal(eng.logposterior(th[:n]), f[:n]), n
    dup = np.repeat(th[:40], 2, axis=0)                       # every pair: the same walker twice
    assert np.array_equal(eng.logposterior(dup), np.repeat(f[:40], 2))
    spread = th.copy()                                        # spread over the whole grid: 144 cell combinations
    spread[:, 0] = rng.uniform(3001.0, 4199.0, len(th))
    spread[:, 1] = rng.uniform(3001.0, 4199.0, len(th))
    want = eng.logposterior(spread)
    few = eng.ctx.pair_stats()
    one_each = spread[:140].copy()                            # every walker a cell of its own: nothing to pair
    one_each[:, 0] = 3005.0 + 100.0 * (np.arange(140) % 12)
    one_each[:, 1] = 3005.0 + 100.0 * (np.arange(140) // 12)
    want1 = eng.logposterior(one_each)
    assert eng.ctx.pair_stats() == (0, int(np.isfinite(want1).sum()))
    eng.ctx.set_path(_lib.PATH_FUSED)
    assert np.array_equal(eng.logposterior(spread), want) and few[0] < npairs
    assert np.array_equal(eng.logposterior(one_each), want1)


def test_pair_error_statuses_and_problems_without_a_pair_form():
    from mcmc_spec_amd import _lib, bands
    from mcmc_spec_amd.engine import Engine
    c = golden_case('B')
    eng = make_engine(c)
    eng.ctx.set_path(_lib.PATH_PAIR)
    bad = c.theta[:12].copy()
    bad[3, 1] = 2800.0                   # outside the isochrone table: ValueError in likelihood mode (mft6.py:95)
    with pytest.raises(ValueError):
        eng.loglikelihood(bad)
    lp = eng.logposterior(bad)           # the prior box rejects it first: a value; its partner is evaluated alone
    assert lp[3] == -np.inf and np.all(np.isfinite(np.delete(lp, 3)))
    with pytest.raises(_lib.MsxError):   # logprior alone has no spectrum pass, hence no pair form
        eng.logprior(c.theta)
    specs = dict(c.specs)
    del specs['3800, 5.0']
    e2 = Engine(0)
    e2.stage_specs(specs)
    e2.stage_problem(c.data, c.err, c.fr, c.r, c.ctm, c.ptm, c.tmi, c.tma, c.matrix, nspec=2,
                     bands=bands.make_bands(c.tables, *c.vega))
    e2.ctx.set_path(_lib.PATH_PAIR)
    with pytest.raises(KeyError):
        e2.loglikelihood(c.theta[:9])
    e3 = make_engine(golden_case('C'), rad_prior=True)    # a triple: twelve rows per walker, no pair form
    e3.ctx.set_path(_lib.PATH_PAIR)
    with pytest.raises(_lib.MsxError):
        e3.logposterior(golden_case('C').theta)
    e4 = make_engine(golden_case('A'))                    # 4154 pixels: beyond the registers
    e4.ctx.set_path(_lib.PATH_PAIR)
    with pytest.raises(_lib.MsxError):
        e4.loglikelihood(c.theta[:4])


@pytest.mark.parametrize('npix,n', [(4096, 2305), (1194, 1024)], ids=['config2', 'config5_length'])
def test_pair_full_size_bits(npix, n):
    """BASELINE config 2 / 3's spectrum (8 element trips per lane with 256 threads) and one of config 5's length."""
    import bench
    from mcmc_spec_amd import synth
    from mcmc_spec_amd.engine import Engine
    eng = Engine(0)
    W = bench.build_workload(eng, npix, False)
    th = synth.draw_walkers(n, seed=9, tmin=W['tmin'], tmax=W['tmax'])
    outs = forms(eng, eng.logposterior, th)
    assert same_bits(outs) and np.all(np.isfinite(outs[0]))
    order = np.lexsort((th[:, 1] // 100, th[:, 0] // 100))
    outs2 = forms(eng, eng.logposterior, th[order])
    assert same_bits(outs2) and np.array_equal(outs2[0], outs[0][order])


@pytest.mark.parametrize('shape', ['two_clusters', 'wide_range', 'plateau_at_median', 'flat', 'steps'])
def test_pair_median_exits(shape):
    """Every exit of the median with the model values in registers: the upper middle value in a later bin, a vector
    spanning more than 8 binades (spilled to the scratch row, min/max-binned select), 65..256 equal candidates (ranked
    through LDS), a constant vector, thousands of duplicates (radix select on the spilled row)."""
    from mcmc_spec_amd.engine import Engine
    from mcmc_spec_amd import synth
    from oracle import mft6_oracle as orc
    teffs = np.arange(3000, 3500, 100)
    loggs = np.array([4.5, 5.0, 5.5])
    wl = np.arange(5400, 9100, 0.2)
    x = (wl - 5600.0) / (8800.0 - 5600.0)
    if shape == 'two_clusters':
        base = np.where(x < 0.5, 1.0e5 * (1 + 1e-3 * x), 2.0e5 * (1 + 1e-3 * x))
    elif shape == 'wide_range':
        base = 1.0e5 * 10.0 ** (4.0 * np.clip(x, 0, 1))
    elif shape == 'plateau_at_median':
        base = 1.0e5 * (1 + 0.5 * np.where(np.abs(x - 0.5) < 0.035, 0.5, x))
    elif shape == 'steps':
        base = np.where(wl < 7000.0, 1.0e5, 3.0e5) + np.where((wl > 7500) & (wl < 7600), 1.0e5 * np.sin(wl), 0.0)
    else:
        base = np.full_like(wl, 1.0e5)
    flux = np.empty((len(teffs), len(loggs), len(wl)))
    for i in range(len(teffs)):
        for j in range(len(loggs)):
            flux[i, j] = base * (1 + 0.01 * i + 0.02 * j)
    specs = synth.grid_to_specs(teffs, loggs, wl, flux)
    matrix = synth.make_isochrone_matrix()
    ctm = [[list(np.linspace(6000, 8800, 40))], [list(np.ones(40))], [0], [7400.0]]
    ptm = [[], [], [], []]
    fr = [[1.0], [0.1], ['x'], [], [], []]
    th = np.array([[3250.0, 3120.0, 0.0, 0.5, 0.4, 2e-3], [3260.0, 3149.0, 0.0, 0.7, 0.9, 3e-3],     # one cell
                   [3250.0, 3120.0, 0.3, 0.5, 0.4, 2e-3], [3300.0, 3049.0, 0.2, 0.7, 0.9, 3e-3],     # two cells
                   [3210.0, 3110.0, 0.1, 0.6, 0.5, 2e-3]])                                            # alone
    for npix in (2048, 2047, 300):
        wl_um = np.linspace(0.56, 0.88, npix)
        rng = np.random.default_rng(4)
        data = [wl_um, 1 + 0.05 * rng.normal(size=npix)]
        err = np.full(npix, 0.05)
        r = [wl_um.min(), wl_um.max()]
        eng = Engine(0)
        eng.stage_specs(specs)
        eng.stage_problem(data, err, fr, r, ctm, ptm, 6000.0, 8800.0, matrix, nspec=2)
        outs = forms(eng, eng.loglikelihood, th)
        assert same_bits(outs), (shape, npix)
        want = np.array([orc.loglikelihood(list(t), fr, 2, data, err, r, specs, ctm, ptm, 6000.0, 8800.0, matrix)
                         for t in th[:3]])
        assert rel_err(outs[1][:3], want).max() < TIGHT, (shape, npix)
        if shape == 'steps' and npix == 2048:
            # every walker of a large batch on the spill path: more workgroups than scratch rows, which are leased --
            # workgroup b and b + 1024 take turns on one row
            many = np.repeat(th, 560, axis=0)
            many[:, 3] *= 1.0 + 1e-4 * np.arange(len(many))
            outs = forms(eng, eng.loglikelihood, many)
            assert same_bits(outs) and np.all(np.isfinite(outs[0]))
            # ... and a lease nobody gives back (a launch torn down mid-spill): the wait is bounded -- the spilling walkers
            # end with MSX_W_HANDOVER (a RuntimeError at this level), not with a hung GPU; the synchronous entry point
            # that saw the status clears the leases, and the next launch is whole again
            from mcmc_spec_amd import _lib
            eng.ctx.set_path(_lib.PATH_PAIR)
            eng.ctx.test_hook(_lib.HOOK_PAIR_LEASES, 1)
            few = many[:64]
            _, st = eng.ctx.logprob_batch(few, _lib.MODE_LOGLIKE)
            assert np.all(st == _lib.W_HANDOVER)
            lp2, st2 = eng.ctx.logprob_batch(few, _lib.MODE_LOGLIKE)
            assert np.all(st2 == _lib.W_OK) and np.array_equal(lp2, outs[0][:64])
            eng.ctx.set_path(_lib.PATH_AUTO)


def test_auto_choice_never_changes_values():
    """MSX_PATH_AUTO takes the pair form from 8 walkers per CU on (spectra of <= 3,072 px: 12) while the planner's last count says pairing pays, the
    fused kernel while it says the ensemble is spread over the grid, and looks again every 32nd launch: whatever it takes,
    in whatever order the two kinds of batches arrive, the values are the fused kernel's."""
    import bench
    from mcmc_spec_amd import _lib, synth
    from mcmc_spec_amd.engine import Engine
    eng = Engine(0)
    W = bench.build_workload(eng, 1194, False)
    n = 4200
    near = synth.draw_walkers(n, seed=21, tmin=W['tmin'], tmax=W['tmax'])
    far = near.copy()
    far[:, 0:2] = np.random.default_rng(5).uniform(W['tmin'] + 1, W['tmax'] - 1, size=(n, 2))
    eng.ctx.set_path(_lib.PATH_FUSED)
    want_near, want_far = eng.logposterior(near), eng.logposterior(far)
    eng.ctx.set_path(_lib.PATH_AUTO)
    assert np.array_equal(eng.logposterior(near), want_near)
    p, s = eng.ctx.pair_stats()
    assert 2 * p + s == n and 4 * s < 3 * p                   # the pair form ran, and it paid
    assert np.array_equal(eng.logposterior(far), want_far)    # still the pair form (the count came from `near`) ...
    p, s = eng.ctx.pair_stats()
    assert 2 * p + s == np.isfinite(want_far).sum() and 4 * s >= 3 * p   # ... which now says: spread
    for _ in range(40):                                       # the fused kernel, with a look through the pair form in between
        assert np.array_equal(eng.logposterior(far), want_far)
    for _ in range(40):                                       # back to an ensemble in a few cells: found at the next look
        assert np.array_equal(eng.logposterior(near), want_near)
    p, s = eng.ctx.pair_stats()
    assert 4 * s < 3 * p
