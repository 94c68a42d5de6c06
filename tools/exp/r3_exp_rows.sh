#!/bin/bash
# Round-3 sensitivity experiments for the walker-pairing decision (measurement builds, wrong values where noted):
#   base      the 256-thread fused kernel, three workgroups per CU
#   pad       the same, two workgroups per CU (MSX_PAD_LDS pads the dynamic LDS)
#   halfrows  the second star re-uses the first star's row loads (-DMSX_EXP_HALFROWS: half the row requests)
#   both
out=${GRAFT_REPO_ROOT:-$PWD}/gpurun_out/r3_exp_rows.txt
: > $out
W=${1:-2048,16384}
for cfg in base pad halfrows both; do
  case $cfg in
    base) env_=""; ;;
    pad) env_="MSX_PAD_LDS=28672"; ;;
    halfrows) env_="MSX_LIB=build/libmsx_halfrows.so"; ;;
    both) env_="MSX_LIB=build/libmsx_halfrows.so MSX_PAD_LDS=28672"; ;;
  esac
  echo "== $cfg ($env_)" >> $out
  env $env_ python3 tools/sweep.py --blocks 256 --walkers $W --paths fused --iters 30 >> $out 2>/dev/null
done
cat $out
