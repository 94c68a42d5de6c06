"""`python bench.py --gpus N` outside a launcher starts its own ranks: the child command is the task
statement's torch.distributed.run form, rank 0's JSON line is passed through, the child's return code is ours -- and
all of it happens before bench.py imports torch or touches a GPU (the parent never execs: it starts a child)."""
import json
import os
import subprocess
import sys
import types

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _fake_run(rc, stdout, seen):
    def run(cmd, **kw):
        seen['cmd'] = list(cmd)
        seen['kw'] = kw
        return types.SimpleNamespace(returncode=rc, stdout=stdout)
    return run


@pytest.fixture
def bench_mod(monkeypatch):
    import bench
    monkeypatch.delenv('WORLD_SIZE', raising=False)
    monkeypatch.delenv('RANK', raising=False)
    return bench


def test_gpus_n_without_world_size_launches_the_ranks_as_a_child(bench_mod, monkeypatch, capsys):
    seen = {}
    line = json.dumps({'metric': 'walker log-likelihood evals/sec (whole node)', 'value': 1.0, 'n_gpus': 8})
    monkeypatch.setattr(subprocess, 'run', _fake_run(0, 'NCCL banner on stdout\n' + line + '\n', seen))
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--gpus', '8', '--steps', '20', '--warmup', '5'])
    imported_before = 'torch' in sys.modules
    with pytest.raises(SystemExit) as ei:
        bench_mod.main()
    assert ei.value.code == 0
    cmd = seen['cmd']
    assert cmd[0] == sys.executable and cmd[1:3] == ['-m', 'torch.distributed.run']
    assert '--nnodes=1' in cmd and cmd[cmd.index('--nproc-per-node') + 1] == '8'
    assert cmd[cmd.index('--master-addr') + 1] == '127.0.0.1' and int(cmd[cmd.index('--master-port') + 1]) > 0
    script = cmd.index(os.path.join(ROOT, 'bench.py'))
    assert cmd[script + 1:] == ['--gpus', '8', '--steps', '20', '--warmup', '5']   # the same arguments, untouched
    assert seen['kw']['env']['MASTER_ADDR'] == '127.0.0.1' and seen['kw']['env']['HSA_ENABLE_IPC_MODE_LEGACY'] == '0'
    out = capsys.readouterr()
    assert out.out == line + '\n'                      # exactly ONE line on stdout: rank 0's
    assert 'NCCL banner' in out.err                    # what else the ranks wrote there goes to stderr
    if not imported_before:
        assert 'torch' not in sys.modules              # the parent did not import torch on the way


def test_the_childs_return_code_is_ours(bench_mod, monkeypatch, capsys):
    monkeypatch.setattr(subprocess, 'run', _fake_run(3, '{"value": null, "invalid": "walker error statuses"}\n', {}))
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--gpus', '2'])
    with pytest.raises(SystemExit) as ei:
        bench_mod.main()
    assert ei.value.code == 3
    assert json.loads(capsys.readouterr().out)['value'] is None   # the line still comes through


def test_no_result_line_is_a_failure(bench_mod, monkeypatch):
    monkeypatch.setattr(subprocess, 'run', _fake_run(0, '', {}))
    assert bench_mod.self_launch(4, ['--gpus', '4']) == 1


def test_inside_a_launcher_nothing_is_spawned():
    """WORLD_SIZE set (the driver's torch.distributed.run form): bench.py is a rank, it must not launch again.  (A process
    of its own: a rank redirects its stdout for good.)  It goes on into the GPU set-up, which this box does not have."""
    env = dict(os.environ, WORLD_SIZE='2', RANK='0', LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT='29999')
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '4'], env=env,
                         capture_output=True, text=True, timeout=300)
    assert '[bench] --gpus' not in out.stderr and 'torch.distributed.run' not in out.stderr
    assert out.stdout.strip() == ''
