#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out
run() { python bench.py --steps 200 --warmup 20 --no-cpu-baseline --loop none --no-stage-rooflines --no-workload-stats --no-renderer-only 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 ms/step', d['ms_per_step'], 'p50', d['step_ms']['p50'], 'p99', d['step_ms']['p99'])"; }
for rep in 1 2 3 4; do
  GSPL_EARLY_LENGTH=0 run "early_length=0"
  GSPL_EARLY_LENGTH=1 run "early_length=1"
done
GSPL_EARLY_LENGTH=0 python tools/micro/host_step.py 300 0 2>&1 | tail -2
GSPL_EARLY_LENGTH=1 python tools/micro/host_step.py 300 0 2>&1 | tail -2
for v in 0 1; do
export GSPL_EARLY_LENGTH=$v TMPDIR=/tmp
rm -rf /tmp/prof
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-renderer-only --loop none --no-stage-rooflines --no-workload-stats > /tmp/log.txt 2>&1)
f=$(find /tmp/prof -name "*kernel_trace.csv" | head -1); python tools/prof_summary.py seq $f composite_fwd $O/r06v_seq_early$v.txt > /dev/null; echo "== early_length=$v"; cat $O/r06v_seq_early$v.txt | awk '$3 > 2.0 || /span/'
done
