#!/bin/bash
# SQ counters of the backward kernel: in-tree library vs the paired-queue variant
cd /tmp && export TMPDIR=/tmp
mkdir -p /root/repo/gpurun_out/r03q
for v in base paired; do
  lib=/root/repo/gaussian-splatting-lightning_amd/libgspl_hip.so
  [ $v != base ] && lib=/root/repo/gaussian-splatting-lightning_amd/variants/libgspl_hip_$v.so
  for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA"; do
    rm -rf /tmp/pmc_$v
    GSPL_HIP_LIB=$lib rocprofv3 --pmc $set --output-format csv -d /tmp/pmc_$v -- python /root/repo/bench.py --steps 8 --warmup 8 --no-cpu-baseline --no-renderer-only --no-stage-rooflines --no-workload-stats > /tmp/logq.txt 2>&1
    f=$(find /tmp/pmc_$v -name "*counter_collection.csv" | head -1)
    python - "$f" $v <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    if "composite_bwd2_kernel" in r["Kernel_Name"]:
        a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
print(sys.argv[2], {k: round(v[0] / max(v[1], 1) / 1e6, 3) for k, v in acc.items()}, "(millions per launch)")
PY
  done
done
