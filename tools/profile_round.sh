#!/bin/bash
# The round's profile set (run on the GPU box through gpurun); condensed summaries land in gpurun_out/profiles_<tag>/
# and are copied into profiles/ (tracked) afterwards.
#   gpurun --timeout 1100 -- 'tools/profile_round.sh r4'
# 1. kernel trace + stats of the default bench (config 2: 256 walkers x 4096 px), then two separate PMC passes
#    (FETCH_SIZE, WRITE_SIZE -- never combined with trace domains) -> <tag>_logprob_kernel_stats.csv, _traffic.json
# 2. SQ counter passes (counters only) at 256 / 3072 walkers (fused), 16,384 walkers (fused and pair) and config 4's
#    share (fused and linked) -> <tag>_valu.json, <tag>_sq_*.json
# 3. per-kernel times of the forms: fused against pair (planner + pair kernel) at 4,096 / 16,384 walkers, fused against
#    linked at 16,384 px
# 4. sweeps (device time per batch), the dependent chain
tag=${1:-r4}
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/prof_$tag
dst=$root/gpurun_out/profiles_$tag
mkdir -p $out $dst
cd /tmp && export TMPDIR=/tmp
B="--no-cpu-baseline --no-extras"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -o kt -- python3 $root/bench.py --steps 400 --warmup 40 $B > $out/bench_kt.json 2> $out/kt.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -o f -- python3 $root/bench.py --steps 50 --warmup 5 $B > /dev/null 2> $out/f.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -o w -- python3 $root/bench.py --steps 50 --warmup 5 $B > /dev/null 2> $out/w.err
cd $root
python3 tools/pmc_summary.py --kt $out/kt --fetch $out/pmc_fetch --write $out/pmc_write --kernel logprob_kernel \
  --out $dst/${tag}_logprob \
  --note "rocprofv3 on: python3 bench.py --steps 400 --warmup 40 --no-cpu-baseline --no-extras (kernel-trace/stats) and --steps 50 --warmup 5 (two separate --pmc passes: FETCH_SIZE, WRITE_SIZE); config 2: 256 walkers, 4096 px, block auto (512 threads, one workgroup per CU: the PF quad-trip variant named in the kernel column)" > /dev/null
grep '^{' $out/bench_kt.json > $dst/${tag}_bench_under_rocprof.json
echo "[profile_round] 1 done" >&2
# ---- SQ counters
for cs in 4096:256:fused 4096:3072:fused 4096:16384:fused 4096:16384:pair 16384:128:fused 16384:128:linked; do
  IFS=: read npix n path <<< "$cs"
  tools/pmc_kernels.sh $tag $npix $n $path > $out/sq_${npix}_${n}_$path.txt 2>&1
  cp $root/gpurun_out/pmck_${tag}_${npix}_${n}_$path/summary.json $dst/${tag}_sq_${npix}px_${n}walkers_$path.json
done
python3 - $dst $tag <<'PY'
import json, sys
dst, tag = sys.argv[1], sys.argv[2]
out = {'source': 'rocprofv3 --pmc (SQ counters only, two passes per point; tools/pmc_kernels.sh) over tools/sweep.py',
       'note': 'SQ_INSTS_VALU = vector ALU wave-instructions per launch; SQ_ACTIVE_INST_VALU counts quad-cycles; VALU busy = '
               '4 x SQ_ACTIVE_INST_VALU / (kernel time x clock x 1024 SIMDs) is derived in DESIGN.md from the kernel times of the same points',
       'points': []}
for npix, n, path in ((4096, 256, 'fused'), (4096, 3072, 'fused'), (4096, 16384, 'fused'), (4096, 16384, 'pair'), (16384, 128, 'fused'), (16384, 128, 'linked')):
    j = json.load(open('%s/%s_sq_%dpx_%dwalkers_%s.json' % (dst, tag, npix, n, path)))
    for k, m in j.items():
        if 'pair_plan' in k:
            continue
        out['points'].append({'npix': npix, 'walkers': n, 'path': path, 'kernel': k, 'valu_insts_per_eval': m['SQ_INSTS_VALU'] / n,
                              'valu_insts_per_pixel_lane': m['SQ_INSTS_VALU'] / n / (npix / 64.0),
                              'vmem_read_insts_per_eval': m.get('SQ_INSTS_VMEM_RD', 0) / n, 'lds_insts_per_eval': m.get('SQ_INSTS_LDS', 0) / n,
                              'wave_cycle_shares': {c: m[c] / m['SQ_WAVE_CYCLES'] for c in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_ACTIVE_INST_VALU') if c in m},
                              'active_valu_quadcycles_per_eval': m['SQ_ACTIVE_INST_VALU'] / n})
json.dump(out, open('%s/%s_valu.json' % (dst, tag), 'w'), indent=1)
print(json.dumps(out['points'], indent=1))
PY
echo "[profile_round] 2 done" >&2
# ---- the forms of the path, per kernel
PATHS="fused pair" tools/prof_forms.sh $tag "4096:4096,4096:16384" > $dst/${tag}_forms_4096px.txt 2>&1
PATHS="fused linked" tools/prof_forms.sh $tag "16384:32,16384:128" > $dst/${tag}_forms_16384px.txt 2>&1
echo "[profile_round] 3 done" >&2
python3 tools/sweep.py --blocks 0 --paths auto --walkers 26,128,256,512,1024,2048,4096,8192,16384 > $dst/${tag}_sweep_4096px.jsonl 2>/dev/null
python3 tools/sweep.py --blocks 0 --paths fused --walkers 4096,8192,16384 >> $dst/${tag}_sweep_4096px.jsonl 2>/dev/null
python3 tools/sweep.py --blocks 0 --paths auto --npix 16384 --phot --walkers 32,128,512 > $dst/${tag}_sweep_16384px.jsonl 2>/dev/null
python3 tools/sweep.py --blocks 0 --paths fused,linked --npix 16384 --phot --iters 200 --walkers 8,16,32,48,64,96,128 > $dst/${tag}_linked_sweep_16384px.jsonl 2>/dev/null
tools/r3_pair_sweep.sh 2048,2304,3072,4096,8192,16384 > /dev/null 2>&1; cp $root/gpurun_out/r3_pair_sweep.txt $dst/${tag}_pair_sweep.txt
python3 tools/chain_bench.py > $dst/${tag}_chain_bench.jsonl 2>/dev/null
for r in 1 2 3 4 5; do python3 tools/chain_bench.py --walkers 256 --no-host --rng device >> $dst/${tag}_chain_bench.jsonl 2>/dev/null; python3 tools/chain_bench.py --walkers 256 --no-host >> $dst/${tag}_chain_bench.jsonl 2>/dev/null; done
echo "[profile_round] 4 done" >&2
ls -la $dst
