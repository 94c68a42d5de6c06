#!/bin/bash
# One-GPU rehearsal of the N > 1 step loop: world-size-1 RCCL all-gather forced on, eager loop vs hipGraph replay.
export MSX_BENCH_FORCE_GATHER=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533
for g in 0 1; do
  MSX_BENCH_GRAPH=$g python bench.py --steps ${1:-400} --warmup 20 --no-cpu-baseline 2>gpurun_out/gather_ab_$g.err | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(round(d['value']), round(d['ms_per_step']*1e3, 2), 'us/step; kernel', round(d['roofline']['kernel_ms']*1e3, 2), 'us;', d['config']['step_loop'])
"
  tail -3 gpurun_out/gather_ab_$g.err
done
