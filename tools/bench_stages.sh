#!/bin/bash
# usage (on the GPU box): tools/bench_stages.sh [bench args]   -> one line: images/s, ms/step, per-stage ms
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --stage-times "$@" 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print(d['value'], d['ms_per_step'], d['stages_ms'], 'frac', d['roofline']['frac'])"
