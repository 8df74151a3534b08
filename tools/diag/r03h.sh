#!/bin/bash
python -m pytest tests/test_adam.py tests/test_abi.py -x -q -m gpu 2>&1 | tail -2
python -m pytest tests/test_training_loop.py -x -q -m gpu -k "deferred" 2>&1 | tail -2
F="--no-cpu-baseline --no-stage-rooflines --no-workload-stats --no-renderer-only"
for i in 1 2; do
python bench.py $F 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('hooked', d['value'], d['ms_per_step'], d['step_ms']['p50'], d['step_ms']['p99'])"
python bench.py $F --no-overlap-sh-update 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('plain ', d['value'], d['ms_per_step'], d['step_ms']['p50'], d['step_ms']['p99'])"
done
python bench.py --gpus 1 --steps 20 --warmup 5 $F 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('hooked 20 steps', d['value'], d['ms_per_step'], d['step_ms'])"
