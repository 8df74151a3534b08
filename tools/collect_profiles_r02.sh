#!/bin/bash
# usage (on the GPU box): tools/collect_profiles_r02.sh <tag>   -> gpurun_out/<tag>/...  (copy the files worth keeping to profiles/)
# bench lines (default, stage times, gsplat API, other workloads, 2 ranks on the shared GPU in both multi-GPU modes), rocprofv3
# kernel stats + last-step sequence at the metric workload and at S-1080p-6M, and the PMC passes (each in its own run).
tag=${1:-r02}
cd /root/repo
O=gpurun_out/$tag; mkdir -p $O
b() { python bench.py "$@" 2>/dev/null | tail -1; }
b > $O/${tag}_bench.json
b --no-cpu-baseline --stage-times > $O/${tag}_bench_stage_times.json
b --no-cpu-baseline --api gsplat > $O/${tag}_bench_gsplat.json
b --no-cpu-baseline --optimizer none > $O/${tag}_bench_no_optimizer.json
for w in S-800-100k S-1080p-6M S-garden-6M S-4k-2M S-1080p-1M-inside; do b --no-cpu-baseline --stage-times --workload $w >> $O/${tag}_bench_other_workloads.jsonl; done
b --no-cpu-baseline --parallelism sharded > $O/${tag}_bench_sharded_1gpu.json
for m in replicated sharded; do
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $((29510 + RANDOM % 400)) bench.py --gpus 2 --steps 20 --warmup 5 \
      --dist-backend gloo --share-device --parallelism $m 2>/dev/null | tail -1 > $O/${tag}_bench_${m}_2ranks_shared_gpu_gloo.json
done
cd /tmp && export TMPDIR=/tmp
prof() {   # name, bench args...
  local name=$1; shift
  rm -rf /tmp/prof_$name
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$name -- python /root/repo/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-renderer-only "$@" > /tmp/log_$name.txt 2>&1
  f=$(find /tmp/prof_$name -name "*kernel_stats.csv" | head -1); python /root/repo/tools/prof_summary.py stats $f 25 /root/repo/$O/${tag}_${name}kernel_stats.csv > /dev/null
  f=$(find /tmp/prof_$name -name "*kernel_trace.csv" | head -1); python /root/repo/tools/prof_summary.py seq $f composite_fwd /root/repo/$O/${tag}_${name}sequence.txt > /dev/null
}
prof ""
prof "6M_" --workload S-1080p-6M
pmc() {    # name, counters...
  local name=$1; shift
  rm -rf /tmp/pmc_$name
  rocprofv3 --pmc "$@" --output-format csv -d /tmp/pmc_$name -- python /root/repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-renderer-only > /tmp/logp_$name.txt 2>&1
  f=$(find /tmp/pmc_$name -name "*counter_collection.csv" | head -1); python /root/repo/tools/prof_summary.py pmc $f /root/repo/$O/${tag}_pmc_$name.csv > /dev/null
}
pmc FETCH_SIZE FETCH_SIZE
pmc WRITE_SIZE WRITE_SIZE
pmc SQ SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES
ls -la /root/repo/$O
