#!/bin/bash
# Per-kernel device time of the forms of the path (rocprofv3 --kernel-trace --stats) at a few batch sizes.
#   gpurun --timeout 600 -- 'PATHS="fused linked" tools/prof_forms.sh tag "4096:256,4096:4096,4096:16384,16384:128"'
tag=${1:-x}
cases=${2:-4096:4096}
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/prof_forms_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for cs in ${cases//,/ }; do
  npix=${cs%%:*}; n=${cs##*:}
  extra=""; [ "$npix" = "16384" ] && extra="--phot"
  for path in ${PATHS:-fused}; do
    rocprofv3 --kernel-trace --stats --output-format csv -d $out/${npix}_${n}_$path -o kt -- python3 $root/tools/sweep.py --npix $npix $extra --walkers $n --blocks 0 --paths $path --iters 20 ${SWEEP_EXTRA:-} > $out/${npix}_${n}_$path.json 2> $out/${npix}_${n}_$path.err
    f=$(find $out/${npix}_${n}_$path -name '*kernel_stats.csv' | head -1)
    echo "== npix $npix walkers $n path $path"; cat $out/${npix}_${n}_$path.json
    [ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    name = r['Name']
    if any(k in name for k in ('logprob_kernel', 'logprob_pair_kernel', 'pair_plan_kernel')):
        print('   {:70s} calls {:>6s} avg {:>10.2f} us  total {:>10.1f} us'.format(name[:70], r['Calls'], float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e3))
PY
  done
done
