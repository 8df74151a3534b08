#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
V=$GRAFT_REPO_ROOT/gaussian-splatting-lightning_amd/variants
GSPL_HIP_LIB=$V/libgspl_hip_match32.so timeout 600 python -m pytest tests/test_sort.py tests/test_hip_parity.py -x -q -m gpu -k "sort or binning or isect or bin" 2>&1 | tail -1
for rep in 1 2; do for v in base match32; do
  if [ $v = base ]; then unset GSPL_HIP_LIB; else export GSPL_HIP_LIB=$V/libgspl_hip_$v.so; fi
  rm -rf /tmp/prof
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python $GRAFT_REPO_ROOT/bench.py --steps 24 --warmup 4 --no-cpu-baseline --no-renderer-only --loop none --no-stage-rooflines --no-workload-stats > /tmp/log.txt 2>&1)
  f=$(find /tmp/prof -name "*kernel_stats.csv" | head -1)
  echo "== $v"; python tools/prof_summary.py stats $f 1 | grep "radix_scatter"
done; done
