#!/bin/bash
python -m pytest tests/test_hip_parity.py tests/test_renderers_gpu.py tests/test_distributed_renderer.py tests/test_scores.py -x -q -m gpu 2>&1 | tail -4
for a in "--api gsplat" "--parallelism sharded" ""; do
python bench.py $a --no-cpu-baseline --no-stage-rooflines --no-workload-stats --no-renderer-only 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$a', d['value'], d['ms_per_step'], d['step_ms']['p50'], d['speculation'])"
done
