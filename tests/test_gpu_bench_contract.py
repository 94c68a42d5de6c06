"""bench.py's contract with the driver (the task's ④): one JSON line on stdout with the fixed keys, `roofline` and
`cpu_baseline` objects of the prescribed shape, values that are consistent with each other."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*extra, env=None):
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '8', '--warmup', '2',
                          '--no-extras', '--cpu-budget', '2'] + list(extra), capture_output=True, text=True, timeout=600, cwd=ROOT,
                         env=dict(os.environ, **(env or {})))
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines          # exactly ONE line on stdout
    return json.loads(lines[0])


def test_bench_prints_one_json_line_with_the_contract_keys():
    j = _run()
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
              'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert k in j, k
    assert j['n_gpus'] == 1 and j['steps'] == 8 and j['warmup'] == 2 and j['higher_is_better'] is True
    assert j['scaling'] == 'weak' and j['vs_baseline'] is None and j['dtype'] == 'f64' and j['data'] == 'synthetic'
    assert 'workload' in j['config'] and 'model' not in j['config'] and j['config']['walkers_total'] == 256
    # value = walkers x steps / timed region
    assert abs(j['value'] - 256 / (j['ms_per_step'] * 1e-3)) < 1e-6 * j['value']
    r = j['roofline']
    for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'):
        assert k in r, k
    assert abs(r['frac'] - r['achieved'] / r['peak']) < 1e-9 and 0 < r['frac'] <= 1 and r['kernel_ms'] <= j['ms_per_step'] * 1.02
    c = j['cpu_baseline']
    for k in ('value', 'unit', 'cores', 'kind', 'sample'):
        assert k in c, k
    assert c['kind'] in ('port', 'reference') and c['cores'] >= 1 and c['value'] > 0
    assert c['max_rel_err_gpu_vs_oracle'] < 1e-9            # the baseline leg doubles as a parity check
    assert j['walker_error_statuses'] == 0


def test_bench_config4_takes_the_linked_form():
    j = _run('--config', '4', '--no-cpu-baseline')
    assert j['config']['baseline_config'] == 4 and j['config']['npix'] == 16384
    assert 'linked' in j['roofline']['kernel'] and j['cpu_baseline'] is None


def test_bench_launches_its_own_ranks_and_passes_the_line_through():
    """The N > 1 entry rehearsed on one GPU: bench.py starts `python -m torch.distributed.run` as a child, the rank
    builds a (one-rank) RCCL communicator, all-gathers every step, and its line arrives on the parent's stdout."""
    j = _run('--no-cpu-baseline', env={'MSX_BENCH_SELF_LAUNCH': '1', 'MSX_BENCH_FORCE_GATHER': '1'})
    assert j['n_gpus'] == 1 and j['n_ranks_seen'] == 1 and j['gather_verified'] is True
    assert j['walker_error_statuses'] == 0 and j['value'] > 0
    assert j['config']['collective'] != 'none' and 'multi_gpu_diag' in j


def test_the_rooflines_kernel_name_comes_from_the_library():
    """bench.py prints what msx_launch_info says -- the launcher's own variant table -- and mirrors nothing: no kernel
    name is spelt out in bench.py, and the line's name / resources equal the library's answer for the same staged problem."""
    src = open(os.path.join(ROOT, 'bench.py')).read()
    assert 'logprob_kernel<' not in src and 'pick_block' not in src
    j = _run('--no-cpu-baseline')
    sys.path.insert(0, ROOT)
    import bench
    from mcmc_spec_amd import _lib
    from mcmc_spec_amd.engine import Engine
    eng = Engine(0)
    bench.build_workload(eng, 4096, False)
    info = eng.ctx.launch_info(256, _lib.MODE_LOGPOST, 0)
    r = j['roofline']
    assert r['kernel'] == info['kernel'] and r['kernel'].startswith("logprob_kernel<NS=2, 512 threads, PF")
    assert r['requested_bytes_per_eval'] == info['requested_bytes_per_eval'] == eng.ctx.bytes_per_eval(256)
    assert r['kernel_resources']['vgprs'] == info['vgprs'] and r['kernel_resources']['form'] == 'fused'
    assert j['storage'] == {k: j['storage'][k] for k in j['storage']} and j['storage']['R'] == 'f64' and j['storage']['H'] == 'f32' and j['storage']['dk'] == 'f32'
    assert j['clock_probe']['shader_mhz'] > 500 and 5 < j['clock_probe']['walker_us_median'] < j['roofline']['kernel_ms'] * 1e3
    # the forms: what a launch WOULD take (launch_info) is what it TOOK (last_form), for the batch sizes of the sweep
    import torch
    from mcmc_spec_amd import synth
    dev = torch.device('cuda', 0)
    for m in (128, 1024, 2048, 4096):
        th = torch.from_numpy(synth.draw_walkers(m, seed=3, tmin=3000.0, tmax=5500.0)).to(dev)
        lp, st = torch.empty(m, dtype=torch.float64, device=dev), torch.empty(m, dtype=torch.int32, device=dev)
        would = eng.ctx.launch_info(m)
        for _ in range(3):   # (MSX_PATH_AUTO's evidence is the planner's count of an earlier launch)
            eng.ctx.logprob_batch_dev(th.data_ptr(), m, 6, lp.data_ptr(), st.data_ptr(), torch.cuda.current_stream(dev).cuda_stream)
        torch.cuda.synchronize()
        assert eng.ctx.last_form() == would['form_id'], (m, would)
    assert eng.ctx.launch_info(2048)['form'].startswith('pair') and eng.ctx.launch_info(1024)['form'] == 'fused'   # (8 walkers per CU)
