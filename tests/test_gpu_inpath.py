"""In-path broadening (include/msx.h, MSX_PATH_INPATH; SURVEY A3 placement (ii)): the instrumental broadening applied per
walker to the unreddened composite inside the data window, instead of once per grid node at staging.  Broadening is
linear, so the form must agree with the default one to the order of the sums -- and with the oracle's restatement of the
same placement (oracle/mft6_oracle.py, loglikelihood(inpath=...)) to the usual 1e-9."""
import sys

import numpy as np
import pytest

import common
from common import rel_err

sys.path.insert(0, common.ROOT)
pytestmark = pytest.mark.gpu
TIGHT = 1e-9


@pytest.fixture(scope='module')
def work():
    from bench import build_workload
    from mcmc_spec_amd.engine import Engine
    eng = Engine(0)
    W = build_workload(eng, 4096, False, keep_host_grid=True, broaden='in_path')
    return eng, W


def test_inpath_matches_the_staging_placement_and_the_oracle(work):
    from mcmc_spec_amd import _lib, synth
    from oracle import mft6_oracle as orc
    eng, W = work
    th = synth.draw_walkers(300, seed=11, tmin=W['tmin'], tmax=W['tmax'])
    th[1, 2] = 0.0          # unreddened
    th[2, 0] = 3800.0       # on a Teff node
    th[3, 3] = 0.04         # rejected by the prior box
    th[4, 1] = 2000.0       # outside the isochrone: ValueError status (never reaches the in-path kernels' output)
    eng.ctx.set_path(_lib.PATH_FUSED)
    ref, st_ref = eng.ctx.logprob_batch(th, _lib.MODE_LOGPOST)
    ll_ref, _ = eng.ctx.logprob_batch(th, _lib.MODE_LOGLIKE)
    eng.ctx.set_path(_lib.PATH_INPATH)
    got, st = eng.ctx.logprob_batch(th, _lib.MODE_LOGPOST)
    ll, _ = eng.ctx.logprob_batch(th, _lib.MODE_LOGLIKE)
    assert eng.ctx.last_form() == _lib.FORM_INPATH and 'GIVEN' in eng.ctx.launch_info(300)['kernel']
    assert np.array_equal(st, st_ref) and np.isneginf(got[3]) and st[4] != 0
    fin = np.isfinite(ref)
    assert np.array_equal(np.isfinite(got), fin) and fin.sum() == 298
    e = rel_err(got[fin], ref[fin]).max()
    e_ll = rel_err(ll[fin], ll_ref[fin]).max()
    print('in-path against the staging placement: max relative difference', e, '(log-likelihood alone', e_ll, ')')
    assert e < 1e-11 and e_ll < 1e-11
    # the oracle's restatement of the same placement: composite of the RAW nodes, broadened per evaluation
    raw = synth.grid_to_specs(W['teffs'], W['loggs'], W['wl'], W['flux'])
    specs = orc.broaden_specs_window(raw, W['win'], W['resolution'])
    bl = orc.make_band_library(W['tabs'], *W['vega'])
    pick = [0, 1, 2, 5, 6, 7]
    want = np.array([orc.loglikelihood(list(th[k]), W['fr'], 2, W['data'], W['err'], W['r'], specs, W['ctm'], W['ptm'], W['tmi'],
                                       W['tma'], W['matrix'], bandlib=bl,
                                       inpath=dict(specs_raw=raw, w=W['win'], resolution=W['resolution'])) for k in pick])
    assert rel_err(ll[pick], want).max() < TIGHT
    # sub-batches (the form's scratch holds ~1,300 walkers of this problem), permutations: a walker's value is its own
    big = synth.draw_walkers(3000, seed=5, tmin=W['tmin'], tmax=W['tmax'])
    a = eng.logposterior(big)
    perm = np.random.default_rng(0).permutation(len(big))
    assert np.array_equal(eng.logposterior(big[perm]), a[perm]) and np.array_equal(eng.logposterior(big[:7]), a[:7])
    eng.ctx.set_path(_lib.PATH_AUTO)
    assert rel_err(a, eng.logposterior(big)).max() < 1e-11
    assert eng.ctx.last_form() != _lib.FORM_INPATH            # AUTO never takes it


def test_inpath_on_config_4s_share():
    """16,384 px + 6 photometric bands: a window of 69,015 samples and 155 taps (sub-batches of ~390 walkers), the band terms
    through the fused kernel's own recipe."""
    from bench import build_workload
    from mcmc_spec_amd import _lib, synth
    from mcmc_spec_amd.engine import Engine
    eng = Engine(0)
    W = build_workload(eng, 16384, True, broaden='in_path')
    th = synth.draw_walkers(500, seed=4, tmin=W['tmin'], tmax=W['tmax'])
    th[7, 2] = 0.0
    eng.ctx.set_path(_lib.PATH_FUSED)
    ref = eng.logposterior(th)
    eng.ctx.set_path(_lib.PATH_INPATH)
    got = eng.logposterior(th)
    assert eng.ctx.last_form() == _lib.FORM_INPATH
    e = rel_err(got, ref).max()
    print('config 4 share, in-path against the staging placement:', e)
    assert np.all(np.isfinite(ref)) and e < 1e-11


def test_inpath_is_refused_without_the_raw_window():
    from bench import build_workload
    from mcmc_spec_amd import _lib, synth
    from mcmc_spec_amd.engine import Engine
    eng = Engine(0)
    W = build_workload(eng, 1194, False)                      # staging placement: nothing kept
    th = synth.draw_walkers(8, seed=1, tmin=W['tmin'], tmax=W['tmax'])
    eng.ctx.set_path(_lib.PATH_INPATH)
    with pytest.raises(_lib.MsxError):
        eng.logposterior(th)
    eng.ctx.set_path(_lib.PATH_AUTO)
    assert np.all(np.isfinite(eng.logposterior(th)))
    # ... for a triple (the form is the binaries'), and for float32-stored tables, even with the raw window kept
    from common import golden_case
    from mcmc_spec_amd import bands
    c = golden_case('C')
    e3 = Engine(0)
    e3.stage_specs(c.specs)
    w = [float(min(c.data[0])) * 1e4 - 2.0, float(max(c.data[0])) * 1e4 + 2.0]
    e3.broaden_grid_window(w, 1700.0, 'in_path')
    e3.stage_problem(c.data, c.err, c.fr, c.r, c.ctm, c.ptm, c.tmi, c.tma, c.matrix, nspec=3, bands=bands.make_bands(c.tables, *c.vega))
    e3.ctx.set_path(_lib.PATH_INPATH)
    with pytest.raises(_lib.MsxError):
        e3.loglikelihood(c.theta)
    b = golden_case('B')
    e2 = Engine(0)
    e2.stage_specs(b.specs)
    e2.broaden_grid_window(w, 1700.0, 'in_path')
    kw = dict(nspec=2, bands=bands.make_bands(b.tables, *b.vega))
    e2.stage_problem(b.data, b.err, b.fr, b.r, b.ctm, b.ptm, b.tmi, b.tma, b.matrix, store='f32', **kw)
    e2.ctx.set_path(_lib.PATH_INPATH)
    with pytest.raises(_lib.MsxError):
        e2.loglikelihood(b.theta)
    e2.stage_problem(b.data, b.err, b.fr, b.r, b.ctm, b.ptm, b.tmi, b.tma, b.matrix, **kw)   # float64 again: the form is there
    got = e2.loglikelihood(b.theta)
    e2.ctx.set_path(_lib.PATH_FUSED)
    assert e2.ctx.last_form() == _lib.FORM_INPATH and rel_err(got, e2.loglikelihood(b.theta)).max() < 1e-11
