#!/bin/bash
# Same-box A/B of two builds of the library on bench.py's own shapes: tools/ab_bench.sh <out dir> <libA> <libB> [reps]
# (alternating runs, so that the box's drift hits both alike); prints us/step and the kernel's own us per shape.
out=$1; A=$2; B=$3; reps=${4:-3}
mkdir -p $out
for r in $(seq 1 $reps); do
  for tag in A B; do
    lib=$A; [ $tag = B ] && lib=$B
    MSX_LIB=$lib python bench.py --steps 200 --warmup 20 --no-extras --no-cpu-baseline > $out/${tag}_200_$r.json 2>> $out/err.log
    MSX_LIB=$lib python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $out/${tag}_20_$r.json 2>> $out/err.log
  done
done
python - $out <<'PY'
import glob, json, sys
import numpy as np
out = sys.argv[1]
for shape in ('200', '20'):
    for tag in 'AB':
        js = [json.load(open(f)) for f in sorted(glob.glob('%s/%s_%s_*.json' % (out, tag, shape)))]
        print(shape, tag, 'us/step', ' '.join('%.2f' % (j['ms_per_step'] * 1e3) for j in js), '| kernel us', ' '.join('%.2f' % (j['roofline']['kernel_ms'] * 1e3) for j in js),
              '| unramped', ' '.join('%.2f' % (j['unramped']['ms_per_step'] * 1e3) for j in js if j.get('unramped')))
PY
