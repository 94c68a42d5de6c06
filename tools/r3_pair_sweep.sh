#!/bin/bash
# fused against the pair form (256 / 512 threads), batch ordered by grid cell (neighbours = partners)
out=${GRAFT_REPO_ROOT:-$PWD}/gpurun_out/r3_pair_sweep.txt
: > $out
W=${1:-512,1024,2048,4096,16384}
echo "== fused" >> $out
python3 tools/sweep.py --blocks 0 --walkers $W --paths fused --iters 30 >> $out 2>/dev/null
for t in 256 512; do
  echo "== pair $t threads" >> $out
  MSX_PAIR_THREADS=$t python3 tools/sweep.py --blocks 0 --walkers $W --paths pair --iters 30 >> $out 2>/dev/null
done
cut -c1-130 $out
