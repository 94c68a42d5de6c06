#!/usr/bin/env python3
"""Diagnostic: innermost loops of one kernel in an ISA dump (hipcc -S --cuda-device-only) with their instruction mix.

    python tools/isa_loops.py build/isa/msx.s 'logprob_kernelILi2ELi2ELi256ELb0ELb0ELb0ELi0E' [min VALU per iteration: 100]
(innermost loops only)
"""
import re
import sys
from collections import Counter


def classify(op):
    if op.startswith(('v_cmp', 'v_cmpx')): return 'valu_cmp'
    if op.startswith('v_') and ('_f64' in op): return 'valu_f64'
    if op.startswith('v_') and ('_f32' in op or '_f16' in op): return 'valu_f32'
    if op.startswith(('v_readlane', 'v_readfirstlane', 'v_writelane')): return 'valu_lane'
    if op.startswith('v_'): return 'valu_int'
    if op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')): return 'vmem'
    if op.startswith('ds_'): return 'lds'
    if op.startswith('s_waitcnt'): return 'wait'
    if op.startswith(('s_cbranch', 's_branch')): return 'branch'
    if op.startswith('s_'): return 'salu'
    return 'other'


def main():
    path, pat = sys.argv[1], sys.argv[2]
    min_valu = int(sys.argv[3]) if len(sys.argv) > 3 else 100
    lines = open(path).read().split('\n')
    start = next(i for i, l in enumerate(lines) if l.startswith('_Z') and pat in l and l.rstrip().split(';')[0].strip().endswith(':'))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith('.Lfunc_end'))
    body = lines[start:end]
    labels = {}
    ins = []  # (index, op, text)
    for l in body:
        s = l.strip()
        m = re.match(r'^(\.LBB\d+_\d+):', s)
        if m:
            labels[m.group(1)] = len(ins)
            continue
        if not s or s.startswith((';', '.', '//')) or s.endswith(':'):
            continue
        op = s.split()[0]
        ins.append((op, s))
    print('kernel at line {}: {} instructions'.format(start + 1, len(ins)))
    tot = Counter(classify(op) for op, _ in ins)
    print('  static mix:', dict(tot))
    loops = []
    for i, (op, s) in enumerate(ins):
        if op.startswith(('s_cbranch', 's_branch')):
            tgt = s.split()[-1]
            if tgt in labels and labels[tgt] <= i:
                loops.append((labels[tgt], i, tgt))
    for a, b, t in sorted(loops):
        inner = [x for x in loops if a <= x[0] and x[1] <= b and (x[0], x[1]) != (a, b)]
        c = Counter(classify(op) for op, _ in ins[a:b + 1])
        valu = sum(v for k, v in c.items() if k.startswith('valu'))
        if valu < min_valu or inner:
            continue
        print('  loop {:12s} [{:6d}..{:6d}] {:5d} instr, VALU {:5d}  {}{}'.format(t, a, b, b - a + 1, valu, dict(c), '  (contains {} loops)'.format(len(inner)) if inner else ''))


if __name__ == '__main__':
    main()
