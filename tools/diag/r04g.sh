#!/bin/bash
# sharded W = 1: where does the long tail (p99 2.1 ms) of the default line come from?  exchange auto / padded x workload stats on / off
show='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d["ms_per_step"], d["step_ms"], d["allocator"])'
F="--parallelism sharded --no-cpu-baseline --no-stage-rooflines"
python bench.py $F --exchange auto 2>/dev/null | tail -1 | python -c "$show" auto+stats
python bench.py $F --exchange padded 2>/dev/null | tail -1 | python -c "$show" padded+stats
python bench.py $F --exchange auto --no-workload-stats 2>/dev/null | tail -1 | python -c "$show" auto
python bench.py $F --exchange padded --no-workload-stats 2>/dev/null | tail -1 | python -c "$show" padded
python bench.py $F --exchange counted 2>/dev/null | tail -1 | python -c "$show" counted+stats
