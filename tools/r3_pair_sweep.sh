#!/bin/bash
# fused against the pair form (planner + pair kernel), config 2's ensemble and one spread over the whole grid
out=${GRAFT_REPO_ROOT:-$PWD}/gpurun_out/r3_pair_sweep.txt
: > $out
W=${1:-512,1024,2048,4096,16384}
echo "== fused" >> $out
python3 tools/sweep.py --blocks 0 --walkers $W --paths fused --iters 30 >> $out 2>/dev/null
echo "== pair" >> $out
python3 tools/sweep.py --blocks 0 --walkers $W --paths pair --iters 30 >> $out 2>/dev/null
echo "== pair, walkers spread over the whole grid (nothing to share)" >> $out
python3 tools/sweep.py --blocks 0 --walkers $W --paths fused,pair --iters 30 --spread >> $out 2>/dev/null
cut -c1-130 $out
