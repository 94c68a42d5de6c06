#!/bin/bash
# SQ counter pass (counters only, its own run) over one sweep point; prints per-kernel means.
#   gpurun -- 'tools/pmc_kernels.sh tag 4096 3072 fused'
tag=${1:-x}; npix=${2:-4096}; n=${3:-3072}; path=${4:-fused}
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/pmck_${tag}_${npix}_${n}_$path
mkdir -p $out
extra=""; [ "$npix" = "16384" ] && extra="--phot"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES --output-format csv -d $out/sq -o sq -- python3 $root/tools/sweep.py --npix $npix $extra --walkers $n --blocks 0 --paths $path --iters 6 > $out/sq.json 2> $out/sq.err
rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_INSTS_SMEM --output-format csv -d $out/sq2 -o sq2 -- python3 $root/tools/sweep.py --npix $npix $extra --walkers $n --blocks 0 --paths $path --iters 6 > $out/sq2.json 2> $out/sq2.err
python3 - $out $n <<'PY'
import csv, glob, sys, collections, json
out, n = sys.argv[1], int(sys.argv[2])
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ('sq', 'sq2'):
    for f in glob.glob(out + '/%s/**/*_counter_collection.csv' % d, recursive=True):
        for row in csv.DictReader(open(f)):
            k = row.get('Kernel_Name', '')
            if not any(s in k for s in ('logprob_kernel', 'logprob_pair_kernel', 'pair_plan_kernel')):
                continue
            short = k.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
            acc[short][row['Counter_Name']].append(float(row['Counter_Value']))
res = {}
for k, cs in acc.items():
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    res[k] = m
    print(k)
    for c in sorted(m):
        print('    {:24s} {:16.0f}   per walker {:12.1f}'.format(c, m[c], m[c] / n))
    if 'SQ_WAVE_CYCLES' in m:
        wc = m['SQ_WAVE_CYCLES']
        print('    shares of wave cycles: wait_any {:.2f} wait_inst {:.2f} active {:.2f} valu {:.2f}; VALU busy of SIMD time (4*active_valu/busy/4 SIMDs...) cycles/inst {:.2f}'.format(
            m.get('SQ_WAIT_ANY', 0) / wc, m.get('SQ_WAIT_INST_ANY', 0) / wc, m.get('SQ_ACTIVE_INST_ANY', 0) / wc, m.get('SQ_ACTIVE_INST_VALU', 0) / wc,
            4 * m.get('SQ_ACTIVE_INST_VALU', 0) / max(m.get('SQ_INSTS_VALU', 1), 1)))
json.dump(res, open(out + '/summary.json', 'w'), indent=1)
PY
