#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
V=$GRAFT_REPO_ROOT/gaussian-splatting-lightning_amd/variants
for rep in 1 2; do
for v in base sh128 sh64; do
  if [ $v = base ]; then unset GSPL_HIP_LIB; else export GSPL_HIP_LIB=$V/libgspl_hip_$v.so; fi
  rm -rf /tmp/prof
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-renderer-only --loop none --no-workload-stats > /tmp/log.txt 2>&1)
  f=$(find /tmp/prof -name "*kernel_stats.csv" | head -1)
  echo "== $v"; tail -1 /tmp/log.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms/step', d['ms_per_step'], 'sh alone', d['stage_rooflines']['sh_fwd_alone']['ms'], 'beside binning', d['stage_rooflines']['sh_fwd_overlapped_with_binning']['ms'], 'binning', d['stage_rooflines']['binning']['ms'])"
  python tools/prof_summary.py stats $f 1 | grep "sh_fwd\|sh_bwd\|bin_keys"
done; done
