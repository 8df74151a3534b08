#!/bin/bash
# what makes the first ~20 steps of a process 2 % slower?  the driver's short form as is / with 30 extra untimed steps ahead of the
# warm-up / with the GC pass after the warm-up (idle device right before the timed steps)
F="--gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-stage-rooflines --no-renderer-only"
show='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d["ms_per_step"], d["step_ms"]["p50"], d["step_ms"]["max"], d["allocator"])'
for i in 1 2 3; do
python bench.py $F 2>/dev/null | tail -1 | python -c "$show" short
GSPL_BENCH_PREHEAT=30 python bench.py $F 2>/dev/null | tail -1 | python -c "$show" short+30
GSPL_BENCH_PREHEAT=100 python bench.py $F 2>/dev/null | tail -1 | python -c "$show" short+100
done
rocm-smi --showclocks 2>/dev/null | head -20
