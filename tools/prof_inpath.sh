#!/bin/bash
# The in-path broadening form (MSX_PATH_INPATH) beside the default forms: device time per batch and, under rocprofv3, per kernel.
#   gpurun -- 'tools/prof_inpath.sh r4'
tag=${1:-r4}
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/prof_inpath_$tag
mkdir -p $out
cd $root
python3 tools/sweep.py --blocks 0 --paths fused,inpath --walkers 64,256,1024,2048 > $out/sweep.jsonl 2> $out/err.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -o kt -- python3 $root/tools/sweep.py --blocks 0 --paths inpath --walkers 256 > $out/kt.jsonl 2> $out/kt.err
cd $root
{
  echo "in-path broadening (MSX_PATH_INPATH) against the fused kernel, 4096 px, resolution 1700 (window 17,259 samples, 91 taps): us per batch (tools/sweep.py)"
  python3 - $out/sweep.jsonl <<'PY'
import json, sys
for l in open(sys.argv[1]):
    j = json.loads(l); print('  walkers %5d  %-6s %8.1f us  %6.2f M evals/s' % (j['walkers'], j['path'], j['batch_us'], j['evals_per_s'] / 1e6))
PY
  echo "per kernel, 256 walkers per launch (rocprofv3 --kernel-trace --stats):"
  python3 - $out/kt/kt_kernel_stats.csv <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Name']
    if any(k in n for k in ('inpath_', 'logprob_kernel')):
        import re
        short = re.search(r'(inpath_\w+|logprob_kernel<[^>]*>)', n).group(1)
        print('  %-72s calls %6s  avg %8.2f us' % (short, r['Calls'], float(r['AverageNs']) / 1e3))
PY
} > $root/gpurun_out/${tag}_inpath.txt
cat $root/gpurun_out/${tag}_inpath.txt
