#!/bin/bash
# the driver's short form: where do its extra 3 % come from?  per-step spans + hipMalloc calls inside the timed region
F="--gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-stage-rooflines --no-workload-stats --no-renderer-only"
for i in 1 2 3; do
GSPL_BENCH_DUMP_STEPS=1 python bench.py $F 2>&1 | grep -v amdgpu | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('step spans'): print(l.strip())
    elif l.startswith('{'):
        d = json.loads(l); print('short', d['ms_per_step'], d['step_ms']['p50'], d['allocator'])
"
done
GSPL_BENCH_DUMP_STEPS=1 python bench.py --no-cpu-baseline --no-stage-rooflines --no-workload-stats --no-renderer-only --steps 48 --warmup 5 2>&1 | grep -v amdgpu | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('step spans'): print(l.strip())
    elif l.startswith('{'):
        d = json.loads(l); print('48 steps after 5', d['ms_per_step'], d['step_ms']['p50'], d['allocator'])
"
