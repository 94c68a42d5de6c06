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
# the planner's phases (stamps build), the forms' crossovers in one process, the pair form's soak
MSX_LIB=$root/build/libmsx_stamps.so python tools/plan_stamps.py --walkers 2048 > $dst/${tag}_plan_stamps.txt 2>&1
MSX_LIB=$root/build/libmsx_stamps.so python tools/plan_stamps.py --walkers 16384 >> $dst/${tag}_plan_stamps.txt 2>&1
python tools/sweep.py --blocks 0 --paths fused,pair --walkers 1024,1536,2048,2304,3072,4096 > $dst/${tag}_crossover_4096px.jsonl 2>> $dst/err.log
python tools/sweep.py --npix 1194 --blocks 0 --paths fused,pair --walkers 2304,3072,4096,6144 > $dst/${tag}_crossover_1194px.jsonl 2>> $dst/err.log
python tools/soak_pair.py --batches 300 > $dst/${tag}_soak_pair.txt 2>&1
python tools/soak_pair.py --batches 150 --npix 1194 --seed 2 2>&1 | tail -1 >> $dst/${tag}_soak_pair.txt
python tools/soak_pair.py --batches 100 --phot --seed 3 2>&1 | tail -1 >> $dst/${tag}_soak_pair.txt
ls -la $dst
# the in-path broadening form beside the fused kernel
tools/prof_inpath.sh $tag > /dev/null 2>&1; cp $root/gpurun_out/${tag}_inpath.txt $dst/${tag}_inpath.txt
ls -la $dst
