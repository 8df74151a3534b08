#!/bin/bash
mkdir -p gpurun_out/r03x
run() { name=$1; shift; env "$@" python bench.py --no-stage-rooflines --no-workload-stats --no-cpu-baseline --no-renderer-only $EXTRA 2>/dev/null | tail -1 > gpurun_out/r03x/$name.json; python -c "
import json,sys
d=json.loads(open('gpurun_out/r03x/$name.json').read()); print('$name', d['value'], d['ms_per_step'], d['step_ms']['p50'], d['step_ms']['p99'], d['step_ms']['slowest'][:2])"; }
for rep in 1 2; do
EXTRA="--no-overlap-sh-update" run plain_low_$rep A=1
EXTRA="--no-overlap-sh-update" run plain_norm_$rep GSPL_SIDE_LOW_PRIORITY=0
EXTRA="" run def_low_512_$rep A=1
EXTRA="" run def_norm_512_$rep GSPL_SIDE_LOW_PRIORITY=0
EXTRA="" run def_norm_1024_$rep GSPL_SIDE_LOW_PRIORITY=0 GSPL_ADAM_DEFERRED_BLOCKS=1024
EXTRA="" run def_norm_2048_$rep GSPL_SIDE_LOW_PRIORITY=0 GSPL_ADAM_DEFERRED_BLOCKS=2048
EXTRA="" run def_norm_all_$rep GSPL_SIDE_LOW_PRIORITY=0 GSPL_ADAM_DEFERRED_BLOCKS=16384
done
