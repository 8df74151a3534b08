#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_sort.py tests/test_hip_parity.py tests/test_metric_point_parity.py -x -q -m gpu 2>&1 | tail -2
for w in S-1080p-1M S-1080p-6M; do
  rm -rf /tmp/prof
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python $GRAFT_REPO_ROOT/bench.py --workload $w --steps 24 --warmup 4 --no-cpu-baseline --no-renderer-only --loop none --no-stage-rooflines --no-workload-stats > /tmp/log.txt 2>&1)
  f=$(find /tmp/prof -name "*kernel_stats.csv" | head -1)
  echo "== $w"; python tools/prof_summary.py stats $f 1 | grep "radix_\|TOTAL" | head -8
  f=$(find /tmp/prof -name "*kernel_trace.csv" | head -1); python tools/prof_summary.py seq $f composite_fwd gpurun_out/r06n_seq_$w.txt | tail -1
done
