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
