#!/bin/bash
# Same-box A/B of the FULL (no-clamp) variants of the fused kernel against the general ones: one library, MSX_NO_FULL=1 for A.
# tools/ab_full.sh <out dir> [reps]
out=$1; reps=${2:-3}
mkdir -p $out
for r in $(seq 1 $reps); do
  for tag in A B; do
    nf=0; [ $tag = A ] && nf=1
    MSX_NO_FULL=$nf python bench.py --steps 200 --warmup 20 --no-extras --no-cpu-baseline > $out/${tag}_200_$r.json 2>> $out/err.log
    MSX_NO_FULL=$nf python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $out/${tag}_20_$r.json 2>> $out/err.log
    MSX_NO_FULL=$nf python tools/sweep.py --blocks 0 --paths fused --walkers 256,512,768,1024,1536,2048,2304 > $out/${tag}_sweep_$r.jsonl 2>> $out/err.log
  done
done
python - $out <<'PY'
import glob, json, sys, collections
out = sys.argv[1]
for shape in ('200', '20'):
    for tag in 'AB':
        js = [json.load(open(f)) for f in sorted(glob.glob('%s/%s_%s_*.json' % (out, tag, shape)))]
        print(shape, tag, 'us/step', ' '.join('%.2f' % (j['ms_per_step'] * 1e3) for j in js), '| kernel us', ' '.join('%.2f' % (j['roofline']['kernel_ms'] * 1e3) for j in js),
              '|', js[0]['roofline']['kernel'][:70] if js else '')
for tag in 'AB':
    acc = collections.defaultdict(list)
    for f in sorted(glob.glob('%s/%s_sweep_*.jsonl' % (out, tag))):
        for l in open(f):
            j = json.loads(l)
            acc[j['walkers']].append(j['batch_us'])
    print(tag, ' '.join('%d: %s' % (k, '/'.join('%.1f' % v for v in vs)) for k, vs in sorted(acc.items())))
PY
