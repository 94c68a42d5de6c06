#!/bin/bash
# where a pair workgroup's time goes: measurement builds (-DMSX_PAIR_EXP=mask, wrong values) at one batch size
out=${GRAFT_REPO_ROOT:-$PWD}/gpurun_out/r3_pair_exp.txt
: > $out
W=${1:-16384}
for t in 512 256; do
for e in 0 1 2 3 4 8 15; do
  lib=build/libmsx_pexp$e.so; [ $e = 0 ] && lib=mcmc_spec_amd/libmsx.so
  echo "== threads $t exp $e" >> $out
  MSX_LIB=$lib MSX_PAIR_THREADS=$t python3 tools/sweep.py --blocks 0 --walkers $W --paths pair --iters 30 --sort-cells 2>/dev/null | cut -c1-110 >> $out
done; done
cat $out
