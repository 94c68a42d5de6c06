#!/bin/bash
# L2 hit rate and HBM fetch of one sweep point, fused against linked (counters only, one pass per counter set).
#   gpurun -- 'tools/pmc_l2.sh r2 16384 128'
tag=${1:-x}; npix=${2:-16384}; n=${3:-128}
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/pmcl2_${tag}_${npix}_${n}
mkdir -p $out
extra=""; [ "$npix" = "16384" ] && extra="--phot"
cd /tmp && export TMPDIR=/tmp
for path in fused linked; do
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $out/tcc_$path -o t -- python3 $root/tools/sweep.py --npix $npix $extra --walkers $n --blocks 0 --paths $path --iters 6 > /dev/null 2> $out/tcc_$path.err
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch_$path -o f -- python3 $root/tools/sweep.py --npix $npix $extra --walkers $n --blocks 0 --paths $path --iters 6 > /dev/null 2> $out/fetch_$path.err
done
python3 - $out $npix $n $root/gpurun_out/${tag}_l2_${npix}px_${n}walkers.json <<'PY'
import csv, glob, sys, collections, json
out, npix, n, dst = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
res = {'source': 'rocprofv3 --pmc over tools/sweep.py, one pass per counter set; means per dispatch of logprob_kernel', 'npix': npix, 'walkers': n, 'paths': {}}
for path in ('fused', 'linked'):
    acc = collections.defaultdict(list)
    for d in ('tcc', 'fetch'):
        for f in glob.glob('%s/%s_%s/**/*_counter_collection.csv' % (out, d, path), recursive=True):
            for row in csv.DictReader(open(f)):
                if 'logprob_kernel' in row.get('Kernel_Name', ''):
                    acc[row['Counter_Name']].append(float(row['Counter_Value']))
    m = {k: sum(v) / len(v) for k, v in acc.items()}
    if 'TCC_HIT_sum' in m:
        m['l2_hit_rate'] = m['TCC_HIT_sum'] / (m['TCC_HIT_sum'] + m['TCC_MISS_sum'])
    if 'FETCH_SIZE' in m:
        m['hbm_read_bytes_x2_corrected'] = m['FETCH_SIZE'] * 1024 * 2   # guide: FETCH_SIZE in KiB, gfx950 counts half
    res['paths'][path] = m
json.dump(res, open(dst, 'w'), indent=1)
print(json.dumps(res, indent=1))
PY
