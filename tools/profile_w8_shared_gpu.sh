#!/bin/bash
# usage (on the GPU box): tools/profile_w8_shared_gpu.sh <out dir>
# The driver's launch line at world size 8 with the eight ranks SHARING the box's one GPU (--share-device, collectives staged through gloo:
# RCCL refuses two ranks on one device), under rocprofv3 --kernel-trace: the kernel sequence of one sharded step of rank 0 and of rank 5 —
# the 8-camera batched projection / SH launch, pack, the peer transport's put / signal / wait over eight IPC-mapped buffers, unpack,
# binning, compositing, and the way back.  Times are those of eight processes time-sharing one device: the SEQUENCE is the evidence, not
# the durations ("the W = 8 kernel sequence of one sharded step kept under profiles/").
root=$(cd "$(dirname "$0")/.." && pwd)
out=${1:-gpurun_out/w8}; case $out in /*) ;; *) out=$root/$out;; esac; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_w8
port=$(python -c "import socket;s=socket.socket();s.bind(('127.0.0.1',0));print(s.getsockname()[1])")
GSPL_BENCH_SOFT_EXIT=1 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_w8 -- python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
  --master-port $port $root/bench.py --gpus 8 --steps 6 --warmup 2 --workload S-800-100k --share-device --dist-backend gloo --no-workload-stats \
  > $out/w8_launch_line.txt 2> /tmp/log_w8.txt || tail -20 /tmp/log_w8.txt
tail -1 $out/w8_launch_line.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['n_gpus'], d['value'], d['ms_per_step'], d['config']['parallelism'])" | tee $out/w8_summary.txt
n=0
for f in $(find /tmp/prof_w8 -name "*kernel_trace.csv" | sort); do
  rows=$(grep -c composite_fwd $f)
  [ "$rows" -lt 4 ] && continue
  python $root/tools/prof_summary.py seq $f composite_fwd $out/w8_sequence_process_$n.txt 4 > /dev/null
  n=$((n+1))
done
ls $out
