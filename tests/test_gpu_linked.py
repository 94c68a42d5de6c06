"""The LINKED form of the hot path (several workgroups per walker in one launch, hand-over inside the kernel:
mcmc_spec_amd/csrc/logprob_kernel.h, LK) against the fused kernel, the oracle and its own failure mode.

The contract is stronger than the parity bar: both forms call the same per-pixel function and sum in the same
canonical order (long spectra segment by segment in either form), so a walker's value must have the SAME BITS
whichever form evaluates it.  And a hand-over that fails must never turn into a silent value -- not in the launch
that fails, and not in any later one.
"""
import os

import numpy as np
import pytest

import common
from common import golden_case, rel_err
from test_gpu_parity import make_engine

pytestmark = pytest.mark.gpu
TIGHT = 1e-9


def both(eng, fn, *a, **k):
    from mcmc_spec_amd import _lib
    out = []
    for path in (_lib.PATH_FUSED, _lib.PATH_LINKED):
        eng.ctx.set_path(path)
        out.append(fn(*a, **k))
    eng.ctx.set_path(_lib.PATH_AUTO)
    return out


@pytest.mark.parametrize('n', [1, 5, 128, 300])
def test_linked_config4_bits(n):
    """BASELINE config 4's spectrum (16,384 px + photometry): the linked form (one workgroup per walker and 8192-pixel
    segment) gives the fused kernel's bits for any walker count, also beyond the sizes the automatic choice would
    take it for; the fused kernel itself sums such a spectrum segment by segment."""
    import bench
    from mcmc_spec_amd import synth
    from mcmc_spec_amd.engine import Engine
    key = 'wide16k'
    if key not in common._cache:
        eng = Engine(0)
        common._cache[key] = (eng, bench.build_workload(eng, 16384, True))
    eng, W = common._cache[key]
    th = synth.draw_walkers(n, seed=40 + n, tmin=W['tmin'], tmax=W['tmax'])
    if n >= 5:
        th[1, 1] = 2999.0            # rejected by the prior box: segment 0's workgroup alone finishes it
        th[3, 2] = 0.0               # no reddening
    f, l = both(eng, eng.logposterior, th)
    assert np.array_equal(f, l)
    assert np.isfinite(f).sum() >= n - 1
    # the automatic choice (the linked form while walkers x segments <= #CUs), and again: the hand-over flags are back at zero
    for _ in range(2):
        assert np.array_equal(f, eng.logposterior(th))
    f, l = both(eng, eng.loglikelihood, th[:1], optimize=True)
    assert np.array_equal(f, l)


def test_linked_three_segments_against_the_oracle():
    """17,000 pixels = two full segments and a short third, unsorted wavelengths: against the oracle and the fused
    kernel; an error status (Teff outside the isochrone, likelihood mode) comes back through segment 0."""
    from mcmc_spec_amd import _lib, bands
    from mcmc_spec_amd.engine import Engine
    c = golden_case('B')
    rng = np.random.default_rng(5)
    wl = rng.uniform(0.56, 0.89, 17000)
    data = [wl, 1.0 + 0.05 * rng.normal(size=wl.size)]
    err = np.full(wl.size, 0.05)
    eng = Engine(0)
    eng.stage_specs(c.specs)
    eng.stage_problem(data, err, c.fr, [wl.min(), wl.max()], c.ctm, c.ptm, c.tmi, c.tma, c.matrix, nspec=2,
                      bands=bands.make_bands(c.tables, *c.vega))
    f, l = both(eng, eng.loglikelihood, c.theta[:12])
    assert np.array_equal(f, l) and np.all(np.isfinite(f))
    one = common.orc.loglikelihood(list(c.theta[0]), c.fr, 2, data, err, [wl.min(), wl.max()], c.specs, c.ctm, c.ptm,
                                   c.tmi, c.tma, c.matrix, bandlib=c.bandlib)
    assert rel_err(l[0], one) < TIGHT
    bad = c.theta[:6].copy()
    bad[2, 0] = 2800.0
    eng.ctx.set_path(_lib.PATH_LINKED)
    with pytest.raises(ValueError):
        eng.loglikelihood(bad)
    # a one-segment spectrum has no linked form (and allocates no scratch for one)
    e1 = make_engine(c)
    e1.ctx.set_path(_lib.PATH_LINKED)
    with pytest.raises(_lib.MsxError):
        e1.loglikelihood(c.theta[:4])
    with pytest.raises(_lib.MsxError):   # the wide form of rounds 1-2 is gone (and 2, once the split form, is now the pair form)
        e1.ctx.set_path(3)


def _config4_engine():
    import bench
    from mcmc_spec_amd.engine import Engine
    eng = Engine(0)
    return eng, bench.build_workload(eng, 16384, True)


def test_linked_hand_over_that_never_comes_fails_loudly_and_poisons_the_context():
    """The fault hook makes the workgroups skip their signal: every one of them gives up after 20 ms of wall clock, the
    walker reports MSX_W_HANDOVER and Python raises -- no hang, no value.  From then on, WITHOUT staging again and
    with the fault gone, the context must never hand out a value computed from the flags the failed launch left
    behind: the automatic choice takes the fused form (bit-equal), an explicit PATH_LINKED is refused; staging again
    brings the linked form back."""
    import time
    from mcmc_spec_amd import _lib, synth
    eng, W = _config4_engine()
    th = synth.draw_walkers(40, seed=77, tmin=W['tmin'], tmax=W['tmax'])
    eng.ctx.set_path(_lib.PATH_FUSED)
    want = eng.logposterior(th)
    eng.ctx.set_path(_lib.PATH_LINKED)
    assert np.array_equal(eng.logposterior(th), want)
    eng.ctx.test_hook(_lib.HOOK_LINKED_FAULT, 1)
    t0 = time.time()
    with pytest.raises(RuntimeError, match='did not meet'):
        eng.logposterior(th)
    assert time.time() - t0 < 5.0
    eng.ctx.test_hook(_lib.HOOK_LINKED_FAULT, 0)          # the fault is gone; the problem is NOT staged again
    with pytest.raises(_lib.MsxError, match='timed out'):
        eng.logposterior(th)                               # explicit linked form: refused
    eng.ctx.set_path(_lib.PATH_AUTO)                       # 40 walkers x 2 segments <= #CUs / 2: would be linked
    assert np.array_equal(eng.logposterior(th), want)      # ... takes the fused form instead
    W = __import__('bench').build_workload(eng, 16384, True)   # staged afresh: flags and poison cleared
    eng.ctx.set_path(_lib.PATH_LINKED)
    assert np.array_equal(eng.logposterior(th), want) and np.all(np.isfinite(want))


def test_linked_poison_reaches_callers_who_never_read_a_status():
    """The device-pointer entry point does not synchronise, so its caller may launch again before -- or without ever --
    looking at the statuses of a launch whose hand-over failed.  The poison word lives on the device: every later
    linked launch on the context reports MSX_W_HANDOVER for ALL its walkers (NaN with the status in its payload),
    not a value read through stale flags."""
    import torch
    from mcmc_spec_amd import _lib, synth
    eng, W = _config4_engine()
    n = 24
    thn = synth.draw_walkers(n, seed=78, tmin=W['tmin'], tmax=W['tmax'])
    dev = torch.device('cuda', 0)
    th = torch.from_numpy(thn).to(dev)
    lp = torch.zeros(n, dtype=torch.float64, device=dev)
    st = torch.zeros(n, dtype=torch.int32, device=dev)
    sp = torch.cuda.current_stream(dev).cuda_stream
    eng.ctx.set_path(_lib.PATH_LINKED)

    def launch():
        eng.ctx.logprob_batch_dev(th.data_ptr(), n, 6, lp.data_ptr(), st.data_ptr(), sp, _lib.MODE_LOGPOST, 0)

    launch()
    torch.cuda.synchronize()
    good = lp.cpu().numpy().copy()
    assert np.all(st.cpu().numpy() == _lib.W_OK) and np.all(np.isfinite(good))
    eng.ctx.test_hook(_lib.HOOK_LINKED_FAULT, 1)
    launch()                                               # every workgroup times out; nobody reads the statuses
    eng.ctx.test_hook(_lib.HOOK_LINKED_FAULT, 0)
    for _ in range(2):                                     # healthy launches, same context, not staged again
        lp.zero_()
        st.zero_()
        launch()
        torch.cuda.synchronize()
        assert np.all(st.cpu().numpy() == _lib.W_HANDOVER)
        assert np.all(np.isnan(lp.cpu().numpy()))
    eng.ctx.set_path(_lib.PATH_FUSED)                      # the fused form does not depend on any of it
    launch()
    torch.cuda.synchronize()
    assert np.array_equal(lp.cpu().numpy(), good)


@pytest.mark.parametrize('npix', [8193, 9001, 16383])
def test_segment_forms_with_an_odd_pixel_count(npix):
    """Odd pixel counts (the scratch rows are then only 8-byte aligned: the segment copies take their scalar paths), a
    second segment of ONE pixel, and one pixel short of two full segments: linked against fused, and the oracle."""
    from mcmc_spec_amd import bands
    from mcmc_spec_amd.engine import Engine
    c = golden_case('B')
    rng = np.random.default_rng(npix)
    wl = rng.uniform(0.56, 0.89, npix)
    data = [wl, 1.0 + 0.05 * rng.normal(size=wl.size)]
    err = np.full(wl.size, 0.05)
    eng = Engine(0)
    eng.stage_specs(c.specs)
    eng.stage_problem(data, err, c.fr, [wl.min(), wl.max()], c.ctm, c.ptm, c.tmi, c.tma, c.matrix, nspec=2,
                      bands=bands.make_bands(c.tables, *c.vega))
    f, l = both(eng, eng.loglikelihood, c.theta[:9])
    assert np.array_equal(f, l) and np.all(np.isfinite(f))
    one = common.orc.loglikelihood(list(c.theta[0]), c.fr, 2, data, err, [wl.min(), wl.max()], c.specs, c.ctm, c.ptm,
                                   c.tmi, c.tma, c.matrix, bandlib=c.bandlib)
    assert rel_err(l[0], one) < TIGHT


@pytest.mark.parametrize('shape', ['two_clusters', 'wide_range', 'plateau_at_median', 'flat', 'steps', 'split_plateau'])
def test_linked_median_exits(shape):
    """Every exit of the median with the model vector spread over several workgroups: the upper middle value in a later
    bin (and in ANOTHER segment), candidates gathered from all segments, 65..256 equal candidates, and the vectors the
    early histogram cannot handle -- more than 8 binades, a constant vector, thousands of duplicates -- for which
    every segment goes to the scratch row and the last arrival runs the general select.  Two segments, a short third
    one, odd and even pixel counts; against the fused kernel bit for bit, and the oracle."""
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
        base = 1.0e5 * (1 + 0.5 * np.where(np.abs(x - 0.5) < 0.004, 0.5, x))
    elif shape == 'split_plateau':   # the median's bin holds values of the first AND the last segment
        base = 1.0e5 * (1 + 0.5 * np.where((np.abs(x - 0.1) < 0.002) | (np.abs(x - 0.9) < 0.002), 0.5, x))
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
    th = np.array([[3250.0, 3120.0, 0.0, 0.5, 0.4, 2e-3], [3260.0, 3149.0, 0.0, 0.7, 0.9, 3e-3],
                   [3250.0, 3120.0, 0.3, 0.5, 0.4, 2e-3], [3300.0, 3049.0, 0.2, 0.7, 0.9, 3e-3],
                   [3210.0, 3110.0, 0.1, 0.6, 0.5, 2e-3]])
    for npix in (16384, 9001, 20000):
        wl_um = np.linspace(0.56, 0.88, npix)
        rng = np.random.default_rng(4)
        data = [wl_um, 1 + 0.05 * rng.normal(size=npix)]
        err = np.full(npix, 0.05)
        r = [wl_um.min(), wl_um.max()]
        eng = Engine(0)
        eng.stage_specs(specs)
        eng.stage_problem(data, err, fr, r, ctm, ptm, 6000.0, 8800.0, matrix, nspec=2)
        f, l = both(eng, eng.loglikelihood, th)
        assert np.array_equal(f, l, equal_nan=True), (shape, npix)
        assert np.all(np.isfinite(l)), (shape, npix)
        want = np.array([orc.loglikelihood(list(t), fr, 2, data, err, r, specs, ctm, ptm, 6000.0, 8800.0, matrix)
                         for t in th[:2]])
        assert rel_err(l[:2], want).max() < TIGHT, (shape, npix)
        if shape == 'steps' and npix == 16384:
            # more walkers than CUs, all of them through the scratch rows, and again (the counters went on counting)
            many = np.repeat(th, 80, axis=0)
            many[:, 3] *= 1.0 + 1e-4 * np.arange(len(many))
            for _ in range(2):
                f, l = both(eng, eng.loglikelihood, many)
                assert np.array_equal(f, l) and np.all(np.isfinite(l))


def test_device_resident_sampler_through_the_linked_form():
    """The stretch move resident on the GPU over config 4's spectrum: a half-step of 24 walkers x 2 segments takes the
    linked form by itself (MSX_PATH_AUTO) -- proposal, both meetings, accept step by whichever workgroup finishes the
    walker -- and must walk the chain the fused kernel walks, bit for bit; the same through a sharded loopback group
    (two ranks: 12 proposals each, log p(q) only, the accept step after the gather)."""
    from mcmc_spec_amd import _lib, synth
    from mcmc_spec_amd.sampler import DeviceEnsembleSampler
    from test_gpu_shard_loopback import run_group
    eng, W = _config4_engine()
    nw = 48
    p0 = synth.draw_walkers(nw, seed=21, tmin=W['tmin'], tmax=W['tmax'])
    chains = []
    for path in (_lib.PATH_FUSED, _lib.PATH_AUTO, _lib.PATH_LINKED):
        eng.ctx.set_path(path)
        s = DeviceEnsembleSampler(nw, 6, eng, seed=3, chunk=8)
        st = s.run_mcmc(p0, 20)
        chains.append((s.get_chain(), s.get_log_prob(), st.coords, st.log_prob))
    eng.ctx.set_path(_lib.PATH_AUTO)
    for c in chains[1:]:
        assert all(np.array_equal(a, b) for a, b in zip(chains[0], c))
    acc = np.mean(np.any(np.diff(chains[0][0], axis=0) != 0, axis=2))
    assert 0.05 < acc < 0.95
    eng2, _ = _config4_engine()
    _lib.Context.comm_init_loopback([eng.ctx, eng2.ctx])
    got = run_group([eng, eng2], p0, 20, seed=3)
    for chain, lp, nacc, worst, coords, logp in got:
        assert worst == 0 and np.array_equal(chain, chains[0][0]) and np.array_equal(lp, chains[0][1])


@pytest.mark.parametrize('npix', [16384, 12001])
def test_linked_triple_system(npix):
    """nspec = 3 (ndim 8) over a two-segment spectrum: the triple's linked variant (twelve corners, the third star's
    recipe wave) against the fused kernel bit for bit and against the oracle, posterior mode (the triple prior box with
    rad_prior), and through the device-resident sampler."""
    from mcmc_spec_amd import _lib, bands
    from mcmc_spec_amd.engine import Engine
    from mcmc_spec_amd.sampler import DeviceEnsembleSampler
    c = golden_case('C')
    rng = np.random.default_rng(npix)
    wl = np.sort(rng.uniform(0.56, 0.89, npix))
    data = [wl, 1.0 + 0.05 * rng.normal(size=wl.size)]
    err = np.full(wl.size, 0.05)
    eng = Engine(0)
    eng.stage_specs(c.specs)
    eng.stage_problem(data, err, c.fr, [wl.min(), wl.max()], c.ctm, c.ptm, c.tmi, c.tma, c.matrix, nspec=3,
                      bands=bands.make_bands(c.tables, *c.vega), av_table=common.av_table_exact(), tmin=c.tmin, tmax=c.tmax,
                      prior=c.prior, rad_prior=True)
    f, l = both(eng, eng.logposterior, c.theta)
    assert np.array_equal(f, l, equal_nan=True) and np.isfinite(l).sum() >= 3
    ok = np.isfinite(l)
    fl, ll = both(eng, eng.loglikelihood, c.theta[ok][:6])
    assert np.array_equal(fl, ll)
    one = common.orc.loglikelihood(list(c.theta[ok][0]), c.fr, 3, data, err, [wl.min(), wl.max()], c.specs, c.ctm, c.ptm,
                                   c.tmi, c.tma, c.matrix, bandlib=c.bandlib)
    assert rel_err(ll[0], one) < TIGHT
    good = c.theta[ok]
    nw = 16
    p0 = good[0] + rng.normal(size=(nw, 8)) * np.array([10, 10, 10, 0.01, 0.01, 0.01, 0.01, 1e-5])
    chains = []
    for path in (_lib.PATH_FUSED, _lib.PATH_LINKED):
        eng.ctx.set_path(path)
        s = DeviceEnsembleSampler(nw, 8, eng, seed=3, chunk=4)
        s.run_mcmc(p0, 6)
        chains.append((s.get_chain(), s.get_log_prob()))
    eng.ctx.set_path(_lib.PATH_AUTO)
    assert np.array_equal(chains[0][0], chains[1][0]) and np.array_equal(chains[0][1], chains[1][1], equal_nan=True)
