#!/bin/bash
mkdir -p gpurun_out/r03w
run() { name=$1; shift; env "$@" python bench.py --no-stage-rooflines --no-workload-stats $EXTRA 2>/dev/null | tail -1 > gpurun_out/r03w/$name.json; python -c "
import json,sys
d=json.loads(open('gpurun_out/r03w/$name.json').read()); print('$name', d['value'], d['ms_per_step'], d['step_ms'])"; }
EXTRA="" run default1 A=1
EXTRA="" run default2 A=1
EXTRA="" run highprio GSPL_SIDE_LOW_PRIORITY=0
EXTRA="--no-overlap-sh-update" run plain A=1
EXTRA="--no-cpu-baseline" run nocpu A=1
EXTRA="--no-renderer-only" run noro A=1
