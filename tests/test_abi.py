"""CPU: the C-ABI library loads and exports exactly what include/msx.h declares; the ctypes mirror of
struct msx_problem has the C layout.  No compute calls (no GPU here)."""
import ctypes
import os
import re
import subprocess
import tempfile

import common  # noqa: F401
from mcmc_spec_amd import _lib

ROOT = common.ROOT
HDR = os.path.join(ROOT, 'include', 'msx.h')


def declared_functions():
    txt = open(HDR).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    return sorted(set(re.findall(r'\b(msx_[a-z0-9_]+)\s*\(', txt)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as ge
    ge.build()
    lib = _lib.load()
    names = declared_functions()
    assert len(names) >= 14
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(_lib.EXPORTED) == names


def test_struct_layout_matches_c():
    src = '#include <stdio.h>\n#include <stddef.h>\n#include "msx.h"\nint main(){printf("%zu %zu %zu %zu %zu\\n",' \
          'sizeof(msx_problem), offsetof(msx_problem, fit_minv), offsetof(msx_problem, win_j0),' \
          'offsetof(msx_problem, tmin), offsetof(msx_problem, has_prior_list));return 0;}\n'
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, 't.c'), 'w').write(src)
        subprocess.check_call(['gcc', '-I', os.path.join(ROOT, 'include'), '-o', os.path.join(d, 't'),
                               os.path.join(d, 't.c')])
        out = subprocess.check_output([os.path.join(d, 't')]).decode().split()
    P = _lib.MsxProblem
    assert [int(x) for x in out] == [ctypes.sizeof(P), P.fit_minv.offset, P.win_j0.offset, P.tmin.offset,
                                     P.has_prior_list.offset]


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, 'LIB_PATH', '/nonexistent/libmsx.so')
    try:
        _lib.load()
    except ImportError as e:
        assert 'no CPU fallback' in str(e)
    else:
        raise AssertionError('load() must raise when the HIP library is missing')


def test_hot_kernel_variants_do_not_spill_to_scratch():
    """Every variant of the hot kernel must keep its working set in registers.  (The problem struct is a by-value
    kernel argument: a helper that stops being inlined makes the compiler copy all 1.2 KB of it into per-lane
    scratch, which doubled the kernel time once; every such helper is force-inlined.)  The variants capped at
    168 / 128 VGPRs by their occupancy target (256 threads; 512 threads sharing a CU) may park a few dwords; the others
    may at most carry a small frame slot that NO instruction touches (the register allocator leaves one behind when a
    variant at the 256-VGPR / 102-SGPR limits spills scalars into vector lanes): not one scratch instruction."""
    src = os.path.join(ROOT, 'mcmc_spec_amd', 'csrc', 'msx.hip')
    with tempfile.TemporaryDirectory() as d:
        asm = os.path.join(d, 't.s')
        out = subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-S', '--cuda-device-only',
                              '-mllvm', '-amdgpu-kernarg-preload-count=8',
                              '-Rpass-analysis=kernel-resource-usage', '-o', asm, src],
                             capture_output=True, text=True)
        assert out.returncode == 0, out.stderr[-2000:]
        text = open(asm).read()
    lines = out.stderr.splitlines()
    seen = 0
    for i, ln in enumerate(lines):
        if 'Function Name' in ln and ('logprob_kernel' in ln or 'logprob_pair_kernel' in ln or 'pair_plan_kernel' in ln):
            block = '\n'.join(lines[i:i + 14])
            m = re.search(r'ScratchSize \[bytes/lane\]: (\d+)', block)
            capped = 'Li256E' in ln or 'Li512ELb0ELb1E' in ln   # 256 threads, or 512 with SH (shared CU)
            assert m and int(m.group(1)) <= 64, block          # (the struct in scratch is > 1 KB)
            if not capped:
                name = re.search(r'Function Name: (\S+)', ln).group(1)
                body = text[text.index('\n' + name + ':'):]
                body = body[:body.index('.Lfunc_end')]
                assert 'scratch_' not in body, name            # no instruction reads or writes scratch
            seen += 1
    assert seen >= 15   # binary + triple; 256 / 512 threads; global-model, shared-CU, LDS-staged variants; linked; pair + planner


def test_python_constants_mirror_the_header():
    """Every status / mode / path / limit constant `_lib.py` restates has the value `include/msx.h` defines."""
    from mcmc_spec_amd import _lib
    text = open(os.path.join(ROOT, 'include', 'msx.h')).read()
    defs = {m.group(1): int(m.group(2).strip('()')) for m in re.finditer(r'#define (MSX_[A-Z0-9_]+) (\(?-?\d+\)?)', text)}
    pairs = {'MSX_OK': _lib.MSX_OK, 'MSX_ERR_INVALID': _lib.MSX_ERR_INVALID, 'MSX_ERR_HIP': _lib.MSX_ERR_HIP,
             'MSX_ERR_STATE': _lib.MSX_ERR_STATE, 'MSX_ERR_RANGE': _lib.MSX_ERR_RANGE,
             'MSX_W_OK': _lib.W_OK, 'MSX_W_REJECT': _lib.W_REJECT, 'MSX_W_KEYERROR': _lib.W_KEYERROR,
             'MSX_W_INDEXERROR': _lib.W_INDEXERROR, 'MSX_W_VALUEERROR': _lib.W_VALUEERROR, 'MSX_W_HANDOVER': _lib.W_HANDOVER,
             'MSX_MODE_LOGLIKE': _lib.MODE_LOGLIKE, 'MSX_MODE_LOGPOST': _lib.MODE_LOGPOST, 'MSX_MODE_CHISQ': _lib.MODE_CHISQ,
             'MSX_MODE_LOGPRIOR': _lib.MODE_LOGPRIOR, 'MSX_BLOCK_512_SHARED': _lib.BLOCK_512_SHARED,
             'MSX_PATH_AUTO': _lib.PATH_AUTO, 'MSX_PATH_FUSED': _lib.PATH_FUSED, 'MSX_PATH_PAIR': _lib.PATH_PAIR,
             'MSX_PATH_LINKED': _lib.PATH_LINKED, 'MSX_PATH_INPATH': _lib.PATH_INPATH,
             'MSX_BROADEN_STAGING': _lib.BROADEN_STAGING, 'MSX_BROADEN_IN_PATH': _lib.BROADEN_IN_PATH,
             'MSX_FORM_FUSED': _lib.FORM_FUSED, 'MSX_FORM_PAIR': _lib.FORM_PAIR, 'MSX_FORM_LINKED': _lib.FORM_LINKED, 'MSX_FORM_INPATH': _lib.FORM_INPATH,
             'MSX_HOOK_LINKED_FAULT': _lib.HOOK_LINKED_FAULT,
             'MSX_MAX_SPEC': _lib.MAX_SPEC, 'MSX_MAX_BANDS': _lib.MAX_BANDS, 'MSX_MAX_DIM': _lib.MAX_DIM}
    for name, val in pairs.items():
        assert defs[name] == val, name
    # and nothing of those families is defined in the header without a mirror
    for name in defs:
        if name.startswith(('MSX_W_', 'MSX_PATH_', 'MSX_ERR_')):
            assert name in pairs, name
