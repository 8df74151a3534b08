#!/bin/bash
# is the short form's 3 % a warm-up effect?  20 timed steps after 5 / 21 / 53 warm-up steps (same cameras in the timed region)
F="--gpus 1 --steps 20 --no-cpu-baseline --no-stage-rooflines --no-workload-stats --no-renderer-only"
for w in 5 21 53 5 21 53; do
python bench.py $F --warmup $w 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('warmup', d['warmup'], d['ms_per_step'], d['step_ms']['p50'], d['allocator'])
"
done
