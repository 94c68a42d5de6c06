"""float32 STORAGE of the staged grid table R (include/msx.h, msx_set_grid_storage; SURVEY 8b's `store_dtype`): a separately
labelled precision.  The arithmetic stays float64; the grid values carry 2^-24.  Held to BASELINE's own tolerance
(1e-6 relative on the log-probability against the reference's goldens), never to the 1e-9 of the float64 tables -- and
never mixed into a float64 problem: the forms that have no float32 variant are refused."""
import numpy as np
import pytest

import common
from common import golden_case, rel_err

pytestmark = pytest.mark.gpu
BASELINE_TOL = 1e-6   # BASELINE.json: "chi^2 matching CPU reference to <= 1e-6 rel"


def _engine(c, store, **kw):
    from mcmc_spec_amd import bands
    from mcmc_spec_amd.engine import Engine
    eng = Engine(0)
    eng.stage_specs(c.specs)
    eng.stage_problem(c.data, c.err, c.fr, c.r, c.ctm, c.ptm, c.tmi, c.tma, c.matrix, nspec=c.nspec,
                      bands=bands.make_bands(c.tables, *c.vega), av_table=common.av_table_exact(), tmin=c.tmin, tmax=c.tmax,
                      prior=c.prior, store=store, **kw)
    return eng


@pytest.mark.parametrize('which', ['A', 'B'])
def test_f32_stored_tables_hold_the_baseline_tolerance_against_the_reference(which):
    from mcmc_spec_amd import _lib
    c = golden_case(which)
    e32, e64 = _engine(c, 'f32'), _engine(c, 'f64')
    cases = [(lambda e: e.loglikelihood(c.theta), which + '_loglike')]
    if which == 'A':
        cases.append((lambda e: e.logposterior(c.g['theta_post']), 'A_logpost_noradprior'))
    for fn, key in cases:
        want = c.g[key]
        g32, g64 = fn(e32), fn(e64)
        fin = np.isfinite(want)
        assert np.array_equal(np.isfinite(g32), fin)
        err32, err64 = rel_err(g32[fin], want[fin]).max(), rel_err(g64[fin], want[fin]).max()
        print(which, key, 'max rel err vs reference: f32-stored', err32, 'f64', err64)
        assert err32 < BASELINE_TOL and err64 < 1e-9
        assert not np.array_equal(g32[fin], g64[fin])          # it IS another precision ...
        assert rel_err(g32[fin], g64[fin]).max() < BASELINE_TOL  # ... inside the tolerance
    # what runs says so, and asks for fewer bytes
    i32, i64 = e32.ctx.launch_info(len(c.theta)), e64.ctx.launch_info(len(c.theta))
    assert 'R32' in i32['kernel'] and 'float32' in i32['kernel'] and 'R32' not in i64['kernel']
    assert i32['requested_bytes_per_eval'] < i64['requested_bytes_per_eval']
    # one staged problem, one precision: no pair / linked form beside it
    e32.ctx.set_path(_lib.PATH_PAIR)
    with pytest.raises(_lib.MsxError):
        e32.loglikelihood(c.theta)
    e32.ctx.set_path(_lib.PATH_AUTO)
    big = np.tile(c.theta, (300, 1))                             # a batch AUTO would hand to the pair form: stays fused
    assert np.array_equal(e32.loglikelihood(big)[:len(c.theta)], e32.loglikelihood(c.theta))
    assert e32.ctx.last_form() == _lib.FORM_FUSED
    # every workgroup size of the fused kernel reads the same float32 table: same bits
    th = c.theta
    import torch
    dev = torch.device('cuda', 0)
    t = torch.from_numpy(np.ascontiguousarray(th)).to(dev)
    outs = []
    for blk in (0, 256, 512, _lib.BLOCK_512_SHARED):
        lp, st = torch.empty(len(th), dtype=torch.float64, device=dev), torch.empty(len(th), dtype=torch.int32, device=dev)
        e32.ctx.logprob_batch_dev(t.data_ptr(), len(th), th.shape[1], lp.data_ptr(), st.data_ptr(), torch.cuda.current_stream(dev).cuda_stream,
                                  _lib.MODE_LOGLIKE, blk)
        torch.cuda.synchronize()
        outs.append(lp.cpu().numpy())
    assert all(np.array_equal(outs[0], o, equal_nan=True) for o in outs[1:])
    # back to float64 on the same context: the float64 bits again
    e32.stage_problem(c.data, c.err, c.fr, c.r, c.ctm, c.ptm, c.tmi, c.tma, c.matrix, nspec=c.nspec,
                      bands=__import__('mcmc_spec_amd.bands', fromlist=['x']).make_bands(c.tables, *c.vega),
                      av_table=common.av_table_exact(), tmin=c.tmin, tmax=c.tmax, prior=c.prior)
    assert np.array_equal(e32.loglikelihood(c.theta), e64.loglikelihood(c.theta))


def test_f32_storage_is_refused_where_no_variant_exists():
    from mcmc_spec_amd import _lib
    c = golden_case('C')   # a triple
    with pytest.raises(_lib.MsxError):
        _engine(c, 'f32', rad_prior=True)
