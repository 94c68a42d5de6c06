#!/bin/bash
# The round's profile set (run on the GPU box through gpurun): kernel trace + stats of the default bench, then two
# separate PMC passes (FETCH_SIZE, WRITE_SIZE -- never combined with trace domains), condensed into profiles/.
#   gpurun --timeout 900 -- 'tools/profile_round.sh r1'
set -e
tag=${1:-r1}
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -o kt -- python3 $root/bench.py --steps 400 --warmup 40 --no-cpu-baseline > $out/bench_kt.json 2> $out/kt.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -o f -- python3 $root/bench.py --steps 50 --warmup 5 --no-cpu-baseline > /dev/null 2> $out/f.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -o w -- python3 $root/bench.py --steps 50 --warmup 5 --no-cpu-baseline > /dev/null 2> $out/w.err
cd $root
python3 tools/pmc_summary.py --kt $out/kt --fetch $out/pmc_fetch --write $out/pmc_write --kernel logprob_kernel \
  --out gpurun_out/profiles_$tag/${tag}_logprob \
  --note "rocprofv3 on: python3 bench.py --steps 400 --warmup 40 --no-cpu-baseline (kernel-trace/stats) and --steps 50 --warmup 5 (two separate --pmc passes: FETCH_SIZE, WRITE_SIZE); config 2: 256 walkers, 4096 px, block auto (512)"
grep '^{' $out/bench_kt.json > gpurun_out/profiles_$tag/${tag}_bench_under_rocprof.json
ls -la gpurun_out/profiles_$tag
