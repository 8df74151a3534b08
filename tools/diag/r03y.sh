#!/bin/bash
# do single-step stalls show up with the deferred update on the default-priority colour stream?  default flags, right after a test run
mkdir -p gpurun_out/r03y
python -m pytest tests/test_adam.py tests/test_loss.py -x -q -m gpu 2>&1 | tail -1
for i in 1 2 3 4 5 6; do
python bench.py 2>/dev/null | tail -1 > gpurun_out/r03y/d$i.json
python -c "
import json
d=json.loads(open('gpurun_out/r03y/d$i.json').read()); print('default$i', d['value'], d['ms_per_step'], d['step_ms']['p50'], d['step_ms']['p99'], d['step_ms']['slowest'])"
done
python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/r03y/s20.json
python -c "
import json
d=json.loads(open('gpurun_out/r03y/s20.json').read()); print('steps20', d['value'], d['ms_per_step'], d['step_ms'])"
python bench.py --no-overlap-sh-update 2>/dev/null | tail -1 > gpurun_out/r03y/plain.json
python -c "
import json
d=json.loads(open('gpurun_out/r03y/plain.json').read()); print('plain', d['value'], d['ms_per_step'], d['step_ms'])"
