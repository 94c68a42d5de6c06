#!/bin/bash
# Same-box A/B of two builds on the batch sweep: tools/ab_sweep.sh <out dir> <libA> <libB> <walkers csv> [paths] [reps]
out=$1; A=$2; B=$3; W=$4; paths=${5:-auto}; reps=${6:-2}
mkdir -p $out
for r in $(seq 1 $reps); do
  for tag in A B; do
    lib=$A; [ $tag = B ] && lib=$B
    MSX_LIB=$lib python tools/sweep.py --blocks 0 --paths $paths --walkers $W > $out/${tag}_$r.jsonl 2>> $out/err.log
  done
done
python - $out <<'PY'
import glob, json, sys, collections
out = sys.argv[1]
for tag in 'AB':
    acc = collections.defaultdict(list)
    for f in sorted(glob.glob('%s/%s_*.jsonl' % (out, tag))):
        for l in open(f):
            j = json.loads(l)
            acc[(j['walkers'], j['path'])].append(j['batch_us'])
    print(tag, ' '.join('%d/%s: %s' % (k[0], k[1], '/'.join('%.1f' % v for v in vs)) for k, vs in sorted(acc.items())))
PY
