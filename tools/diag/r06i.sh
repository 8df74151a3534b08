#!/bin/bash
# call i: kernel sequence of the step at S-1080p-6M (where does the binning's 1.45 ms go?)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out
export TMPDIR=/tmp
rm -rf /tmp/prof
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python $GRAFT_REPO_ROOT/bench.py --workload S-1080p-6M --steps 10 --warmup 3 --no-cpu-baseline --no-renderer-only --loop none --no-stage-rooflines --no-workload-stats > /tmp/log.txt 2>&1)
tail -1 /tmp/log.txt | cut -c1-300
f=$(find /tmp/prof -name "*kernel_trace.csv" | head -1); python tools/prof_summary.py seq $f composite_fwd $O/r06i_6M_seq.txt > /dev/null; cat $O/r06i_6M_seq.txt
