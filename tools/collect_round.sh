#!/bin/bash
# The round's bench lines and small probes for profiles/ (one gpurun call): tools/collect_round.sh r4
tag=${1:-r4}
root=${GRAFT_REPO_ROOT:-$PWD}
dst=$root/gpurun_out/collect_$tag
mkdir -p $dst
cd $root
python -m pytest tests -m gpu -x -q > $dst/${tag}_gpu_tests.log 2>&1; tail -2 $dst/${tag}_gpu_tests.log
python bench.py --steps 200 --warmup 20 > $dst/${tag}_bench_default.json 2> $dst/err.log
for r in 1 2 3; do python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $dst/${tag}_bench_driver_shape_$r.json 2>> $dst/err.log; done
python bench.py --config 4 --steps 200 --warmup 20 --no-cpu-baseline > $dst/${tag}_bench_c4.json 2>> $dst/err.log
python bench.py --config 5 --steps 200 --warmup 20 --no-cpu-baseline > $dst/${tag}_bench_c5.json 2>> $dst/err.log
python bench.py --store f32 --steps 200 --warmup 20 --no-extras > $dst/${tag}_bench_f32_labelled.json 2>> $dst/err.log
MSX_BENCH_SELF_LAUNCH=1 MSX_BENCH_FORCE_GATHER=1 python bench.py --gpus 1 --steps 200 --warmup 20 --no-extras --no-cpu-baseline > $dst/${tag}_rehearsal_one_rank_gather.json 2>> $dst/err.log
MSX_BENCH_SELF_LAUNCH=1 MSX_BENCH_FORCE_GATHER=1 python bench.py --gpus 1 --config 4 --steps 200 --warmup 20 --no-extras --no-cpu-baseline > $dst/${tag}_rehearsal_one_rank_gather_c4.json 2>> $dst/err.log
python tools/region_probe.py --tag default > $dst/${tag}_region_probe.jsonl 2>> $dst/err.log
MSX_LIB=$root/build/libmsx_stamps.so python tools/stamps.py --walkers 256 > $dst/${tag}_fused_stamps.txt 2>&1
MSX_LIB=$root/build/libmsx_stamps.so python tools/stamps.py --walkers 128 >> $dst/${tag}_fused_stamps.txt 2>&1
ls -la $dst
