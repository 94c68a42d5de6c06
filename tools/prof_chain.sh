#!/bin/bash
# kernel durations inside the device-resident sampler loop (rocprofv3 --kernel-trace --stats over tools/chain_bench.py)
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/prof_chain
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -o kt -- python3 $root/tools/chain_bench.py > $out/chain.json 2> $out/chain.err
cat $out/chain.json
python3 - $out/kt/kt_kernel_stats.csv <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'logprob_kernel' in r['Name'] or 'sampler' in r['Name']:
        print('{:90s} calls {:>7s} avg {:>8.2f} us min {:>8.2f} max {:>8.2f}'.format(r['Name'].replace('(anonymous namespace)::', '')[:90], r['Calls'], float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3, float(r['MaxNs']) / 1e3))
PY
