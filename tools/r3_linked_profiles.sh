#!/bin/bash
# The linked form's share of the round's profile set (run on the GPU box through gpurun; summaries land in
# gpurun_out/profiles_<tag>/ and are copied into profiles/ afterwards).
#   gpurun --timeout 900 -- 'tools/r3_linked_profiles.sh r3'
tag=${1:-r3}
root=${GRAFT_REPO_ROOT:-$PWD}
dst=$root/gpurun_out/profiles_$tag
mkdir -p $dst
cd $root
python3 tools/sweep.py --blocks 0 --paths fused,linked --npix 16384 --phot --iters 200 --walkers 8,16,32,48,64,96,128 > $dst/${tag}_linked_sweep_16384px.jsonl 2>/dev/null
python3 tools/sweep.py --blocks 0 --paths auto --npix 16384 --phot --walkers 32,128,512 > $dst/${tag}_sweep_16384px.jsonl 2>/dev/null
echo "[linked_profiles] sweeps done" >&2
PATHS="fused linked" tools/prof_forms.sh $tag "16384:32,16384:128" > $dst/${tag}_forms_16384px.txt 2>&1
echo "[linked_profiles] forms done" >&2
tools/pmc_kernels.sh $tag 16384 128 linked > $root/gpurun_out/sq_16384_128_linked.txt 2>&1
cp $root/gpurun_out/pmck_${tag}_16384_128_linked/summary.json $dst/${tag}_sq_16384px_128walkers_linked.json
echo "[linked_profiles] counters done" >&2
if [ -f build/libmsx_stamps.so ]; then
  for n in 8 64 128; do MSX_LIB=$root/build/libmsx_stamps.so python3 tools/stamps.py --walkers $n --npix 16384 --path linked; done > $dst/${tag}_linked_stamps.txt 2>/dev/null
fi
python3 bench.py --config 4 > $dst/${tag}_bench_c4.json 2> $root/gpurun_out/bench_c4.err
python3 bench.py > $dst/${tag}_bench_default.json 2> $root/gpurun_out/bench_default.err
ls -la $dst
