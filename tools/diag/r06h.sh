#!/bin/bash
# call h: the colour kernel behind the count half of the binning (GSPL_COLOUR_LATE=1) against behind the geometry kernel
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out
export TMPDIR=/tmp
run() { python bench.py --steps 100 --warmup 10 --no-cpu-baseline --loop none --no-stage-rooflines --no-workload-stats --no-renderer-only 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 ms/step', d['ms_per_step'])"; }
for rep in 1 2 3; do
  GSPL_COLOUR_LATE=0 run "late=0"
  GSPL_COLOUR_LATE=1 run "late=1"
done
for v in 0 1; do
export GSPL_COLOUR_LATE=$v
rm -rf /tmp/prof
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-renderer-only --loop none --no-stage-rooflines --no-workload-stats > /tmp/log.txt 2>&1)
f=$(find /tmp/prof -name "*kernel_trace.csv" | head -1); python tools/prof_summary.py seq $f composite_fwd $O/r06h_seq_late$v.txt > /dev/null; echo "== late=$v"; sed -n 12,40p $O/r06h_seq_late$v.txt
done
