#!/bin/bash
# Extra PMC passes for the hot kernel (each pass its own run, counters only): where the waves' cycles go, LDS
# conflicts, L2 hit rate.   gpurun --timeout 900 -- 'tools/pmc_extra.sh r1'
set -e
tag=${1:-r1}
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/pmcx_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES --output-format csv -d $out/sq -o sq -- python3 $root/bench.py --steps 50 --warmup 5 --no-cpu-baseline > /dev/null 2> $out/sq.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d $out/sq2 -o sq2 -- python3 $root/bench.py --steps 50 --warmup 5 --no-cpu-baseline > /dev/null 2> $out/sq2.err
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum --output-format csv -d $out/tcc -o tcc -- python3 $root/bench.py --steps 50 --warmup 5 --no-cpu-baseline > /dev/null 2> $out/tcc.err
cd $root
python3 - <<PY
import csv, glob, json, collections
out = {}
for d in ('sq', 'sq2', 'tcc'):
    f = glob.glob('$out/%s/**/*_counter_collection.csv' % d, recursive=True)
    acc = collections.defaultdict(list)
    for row in csv.DictReader(open(f[0])):
        if 'logprob_kernel' in row.get('Kernel_Name', ''):
            acc[row['Counter_Name']].append(float(row['Counter_Value']))
    for k, v in acc.items():
        out[k] = {'mean_per_dispatch': sum(v) / len(v), 'dispatches': len(v)}
g = lambda k: out[k]['mean_per_dispatch']
der = {}
if 'SQ_WAVE_CYCLES' in out:
    wc = g('SQ_WAVE_CYCLES')
    der['wave_cycles_share'] = {k: g(k) / wc for k in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_ACTIVE_INST_VALU', 'SQ_ACTIVE_INST_LDS', 'SQ_ACTIVE_INST_VMEM') if k in out}
if 'TCC_HIT_sum' in out:
    der['l2_hit_rate'] = g('TCC_HIT_sum') / (g('TCC_HIT_sum') + g('TCC_MISS_sum'))
if 'SQ_LDS_BANK_CONFLICT' in out and 'SQ_LDS_IDX_ACTIVE' in out:
    der['lds_bank_conflict_share_of_lds_cycles'] = g('SQ_LDS_BANK_CONFLICT') / g('SQ_LDS_IDX_ACTIVE')
json.dump({'kernel': 'logprob_kernel<2,2,512,PF>', 'config': '256 walkers x 4096 px (bench default)', 'note': 'rocprofv3 --pmc, three separate passes over python3 bench.py --steps 50 --warmup 5; SQ_* cycle counters count quad-cycles summed over waves', 'counters': out, 'derived': der}, open('gpurun_out/${tag}_logprob_pmc_extra.json', 'w'), indent=1)
print(json.dumps(der, indent=1))
PY
