#!/usr/bin/env python3
"""The measured tables of DESIGN.md, generated from the tracked files under profiles/ (no number is
typed twice).

    python tools/design_tables.py            # prints the block
    python tools/design_tables.py --write    # rewrites the block between the markers in DESIGN.md

tests/test_design_tables.py checks that DESIGN.md holds exactly what this prints.
"""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROF = os.path.join(ROOT, 'profiles')
TAG = 'r4'
BEGIN, END = '<!-- BEGIN GENERATED TABLES (tools/design_tables.py) -->', '<!-- END GENERATED TABLES -->'


def jload(name):
    p = os.path.join(PROF, name)
    if not os.path.exists(p):
        return None
    with open(p) as f:
        return json.load(f)


def jlines(name):
    p = os.path.join(PROF, name)
    if not os.path.exists(p):
        return []
    out = []
    for ln in open(p):
        ln = ln.strip()
        if ln.startswith('{'):
            out.append(json.loads(ln))
    return out


def f(x, nd=1):
    return '–' if x is None else ('{:,.%df}' % nd).format(x)


def kernel_stats():
    p = os.path.join(PROF, TAG + '_logprob_kernel_stats.csv')
    if not os.path.exists(p):
        return None
    for row in csv.DictReader(open(p)):
        if 'logprob_kernel' in row['Name']:
            return dict(calls=int(row['Calls']), avg_us=float(row['AverageNs']) / 1e3, min_us=float(row['MinNs']) / 1e3,
                        max_us=float(row['MaxNs']) / 1e3)
    return None


def bench_row(name):
    j = jload(name)
    if not j:
        return None
    cp = j.get('clock_probe') or {}
    return dict(file=name, steps=j['steps'], us_step=j['ms_per_step'] * 1e3, mevals=(j['value'] or 0) / 1e6, kern=j['roofline']['kernel_ms'] * 1e3,
                frac=j['roofline']['frac'], unr=(j.get('unramped') or {}).get('ms_per_step'), mhz=cp.get('shader_mhz'),
                walker=cp.get('walker_us_median'), span=cp.get('span_us'))


def build():
    o = []
    ks = kernel_stats()
    o.append('**Headline kernel, config 2 (256 walkers × 4096 px per launch).**  Files: `profiles/%s_logprob_kernel_stats.csv`, '
             '`%s_bench_*.json`, `%s_logprob_traffic.json`.' % (TAG, TAG, TAG))
    o.append('')
    o.append('| measurement | value |')
    o.append('|---|---|')
    if ks:
        o.append('| rocprofv3 `--kernel-trace --stats`, `logprob_kernel` average over %s dispatches | **%s µs** (min %s, max %s) |' %
                 (f(ks['calls'], 0), f(ks['avg_us'], 2), f(ks['min_us'], 2), f(ks['max_us'], 2)))
    tr = jload(TAG + '_logprob_traffic.json')
    if tr and tr.get('hbm_bytes_per_launch'):
        o.append('| HBM traffic per launch (rocprofv3 `--pmc`, FETCH_SIZE × 2 + WRITE_SIZE, separate passes) | %s MB |' % f(tr['hbm_bytes_per_launch'] / 1e6, 2))
    for name in sorted(glob.glob(os.path.join(PROF, TAG + '_bench_*.json'))):
        b = bench_row(os.path.basename(name))
        if not b or 'under_rocprof' in name:
            continue
        j = jload(os.path.basename(name))
        cfg = j['config'].get('baseline_config')
        lab = '' if (j.get('storage') or {}).get('R', 'f64') == 'f64' else ' — **float32-STORED grid table: a separately labelled precision, not the headline**'
        o.append(('| `%s`: config %s, %d steps' + lab + ' | %s µs/step = **%s M evals/s**; kernel %s µs by HIP events; `roofline.frac` %s; un-ramped %s µs/step; '
                  'clock probe %s MHz, walker %s µs, first start → last end %s µs |') %
                 (b['file'], cfg, b['steps'], f(b['us_step'], 2), f(b['mevals'], 2), f(b['kern'], 2), f(b['frac'], 3),
                  f(b['unr'] * 1e3 if b['unr'] else None, 2), f(b['mhz'], 0), f(b['walker'], 2), f(b['span'], 2)))
    for name in sorted(glob.glob(os.path.join(PROF, TAG + '_rehearsal_*.json'))):
        j = jload(os.path.basename(name))
        bt = j['config'].get('block_tuned_next_to_the_collective') or {}
        cands = '; '.join('%s: %s' % (k, f(v, 1)) for k, v in (bt.get('candidates_us_per_step') or {}).items())
        o.append('| `%s`: ONE-GPU rehearsal of the N > 1 loop (self-launch through `torch.distributed.run`, one-rank RCCL communicator, all-gather '
                 'every step), config %s, %d steps | %s µs/step; `%s`; `gather_verified` %s; set-up candidates, µs per step (%s): %s → *%s* |' %
                 (os.path.basename(name), j['config'].get('baseline_config'), j['steps'], f(j['ms_per_step'] * 1e3, 2), j['config']['collective'],
                  j.get('gather_verified'), bt.get('timed_as'), cands, bt.get('taken')))
    o.append('')
    # ---- batch sweep
    rows = jlines(TAG + '_sweep_4096px.jsonl')
    if rows:
        o.append('**Batch sweep, 4096 px** (`profiles/%s_sweep_4096px.jsonl`: `tools/sweep.py`, device time per batch from HIP events; '
                 '`auto` = the form `MSX_PATH_AUTO` takes).' % TAG)
        o.append('')
        o.append('| walkers | path | µs per batch | M evals/s |')
        o.append('|---|---|---|---|')
        for r in rows:
            o.append('| %s | %s | %s | %s |' % (f(r['walkers'], 0), r['path'], f(r['batch_us'], 1), f(r['evals_per_s'] / 1e6, 1)))
        o.append('')
    rows = jlines(TAG + '_linked_sweep_16384px.jsonl')
    if rows:
        o.append('**16,384 px + 6 bands, fused against linked** (`profiles/%s_linked_sweep_16384px.jsonl`).' % TAG)
        o.append('')
        ws = sorted({r['walkers'] for r in rows})
        o.append('| walkers | ' + ' | '.join(str(w) for w in ws) + ' |')
        o.append('|---|' + '---|' * len(ws))
        for path in ('fused', 'linked'):
            o.append('| %s, µs | ' % path + ' | '.join(f(next((r['batch_us'] for r in rows if r['walkers'] == w and r['path'] == path), None), 1) for w in ws) + ' |')
        o.append('')
    # ---- instruction counts
    v = jload(TAG + '_valu.json')
    if v:
        o.append('**Instructions per evaluation** (`profiles/%s_valu.json`: rocprofv3 `--pmc`, SQ counters only).' % TAG)
        o.append('')
        o.append('| px | walkers | form | kernel | VALU wave-instructions | vector-memory reads | LDS instructions | SQ_WAIT_ANY / SQ_WAVE_CYCLES |')
        o.append('|---|---|---|---|---|---|---|---|')
        for pt in v['points']:
            o.append('| %s | %s | %s | `%s` | %s | %s | %s | %s |' % (f(pt['npix'], 0), f(pt['walkers'], 0), pt['path'], pt['kernel'][:48],
                                                                   f(pt['valu_insts_per_eval'], 0), f(pt['vmem_read_insts_per_eval'], 0),
                                                                   f(pt['lds_insts_per_eval'], 0), f(pt['wave_cycle_shares'].get('SQ_WAIT_ANY'), 2)))
        o.append('')
    # ---- LDS conflicts
    sq = []
    for name in sorted(glob.glob(os.path.join(PROF, TAG + '_sq_*.json'))):
        j = jload(os.path.basename(name))
        for k, m in j.items():
            if 'pair_plan' in k or 'SQ_LDS_BANK_CONFLICT' not in m:
                continue
            sq.append((os.path.basename(name), k, m['SQ_LDS_BANK_CONFLICT'], m['SQ_LDS_IDX_ACTIVE']))
    if sq:
        o.append('**LDS bank conflicts** (`profiles/%s_sq_*.json`; what they are: `profiles/r4_lds_conflicts.txt`).' % TAG)
        o.append('')
        o.append('| file | kernel | SQ_LDS_BANK_CONFLICT | SQ_LDS_IDX_ACTIVE | ratio |')
        o.append('|---|---|---|---|---|')
        for name, k, a, b in sq:
            o.append('| `%s` | `%s` | %s | %s | %s %% |' % (name, k[:44], f(a, 0), f(b, 0), f(100.0 * a / b, 1)))
        o.append('')
    # ---- chain
    ch = jlines(TAG + '_chain_bench.jsonl')
    if ch:
        o.append('**One dependent chain** (`profiles/%s_chain_bench.jsonl`: `tools/chain_bench.py`, wall time of `run_mcmc`).' % TAG)
        o.append('')
        o.append('| walkers | randomness | µs per iteration | M evals/s |')
        o.append('|---|---|---|---|')
        import collections
        acc = collections.OrderedDict()
        for r in ch:
            acc.setdefault((r['walkers'], r.get('randomness', 'host')), []).append(r['device_us_per_step'])
        for (w, rng), vs in acc.items():
            vs = sorted(vs)
            med = vs[len(vs) // 2]
            o.append('| %s | %s | %s%s | %s |' % (f(w, 0), rng, f(med, 2), '' if len(vs) == 1 else ' (median of %d: %s … %s)' % (len(vs), f(vs[0], 1), f(vs[-1], 1)),
                                                 f(w / med, 2)))
        o.append('')
    return '\n'.join(o).rstrip() + '\n'


def main():
    block = build()
    if '--write' in sys.argv:
        p = os.path.join(ROOT, 'DESIGN.md')
        s = open(p).read()
        a, b = s.index(BEGIN) + len(BEGIN), s.index(END)
        open(p, 'w').write(s[:a] + '\n' + block + s[b:])
    else:
        sys.stdout.write(block)


if __name__ == '__main__':
    main()
