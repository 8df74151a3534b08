#!/bin/bash
# the driver's exact command (20 steps, 5 warm-up): gc inside the warm-up vs after it
F="--gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-stage-rooflines --no-workload-stats --no-renderer-only"
for i in 1 2 3; do
python bench.py $F 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('gc-early', d['ms_per_step'], d['step_ms'])"
GSPL_BENCH_GC_LATE=1 python bench.py $F 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('gc-late ', d['ms_per_step'], d['step_ms'])"
done
python bench.py $F --no-overlap-sh-update 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('gc-early plain', d['ms_per_step'], d['step_ms'])"
python bench.py $F --no-overlap-sh-update 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('gc-early plain', d['ms_per_step'], d['step_ms'])"
