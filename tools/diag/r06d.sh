#!/bin/bash
# call d: early keys x throttled colour kernel (LDS padding)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out
export TMPDIR=/tmp
run() { python bench.py --steps 100 --warmup 10 --no-cpu-baseline --loop none --no-stage-rooflines --no-workload-stats --no-renderer-only 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 ms/step', d['ms_per_step'])"; }
for rep in 1 2; do
for ek in 0 1; do for kb in 0 64 96 160; do
  export GSPL_EARLY_KEYS=$ek; if [ $kb = 0 ]; then unset GSPL_SH_FWD_LDS_KB; else export GSPL_SH_FWD_LDS_KB=$kb; fi
  run "early=$ek lds=$kb"
done; done; done
export GSPL_EARLY_KEYS=1 GSPL_SH_FWD_LDS_KB=160
rm -rf /tmp/prof
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-renderer-only --loop none --no-stage-rooflines --no-workload-stats > /tmp/log.txt 2>&1)
f=$(find /tmp/prof -name "*kernel_trace.csv" | head -1); python tools/prof_summary.py seq $f composite_fwd $O/r06d_seq_on_160.txt > /dev/null; sed -n 12,40p $O/r06d_seq_on_160.txt
