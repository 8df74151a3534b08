#!/bin/bash
# The ONE scratch script of a gpurun call (rewritten per call; outputs under gpurun_out/<tag>/, the keepers are copied to profiles/).
tag=${1:-r10c}
cd /root/repo
O=/root/repo/gpurun_out/$tag; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
summ() { python - "$1" <<'PY'
import csv, sys, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = re.sub(r"\(.*", "", r["Kernel_Name"])[:80]
    if "radix" not in k: continue
    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in sorted(acc.items()):
    print(k)
    for c, v in sorted(cs.items()):
        print("   %-28s mean %16.1f  n %d" % (c, sum(v) / len(v), len(v)))
PY
}
for n in 6000000 1000000; do
  echo "== N=$n kernel stats"
  rm -rf /tmp/p; SORT_N=$n rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p -- python /root/repo/tools/micro/depth_sort_probe.py 2>&1 | grep "depth sort"
  f=$(find /tmp/p -name "*kernel_stats.csv" | head -1); python /root/repo/tools/prof_summary.py stats $f 12 /tmp/ks.csv | head -16
done > $O/${tag}_depth_sort_probe.txt 2>&1
pass() { rm -rf /tmp/q; SORT_N=6000000 rocprofv3 --pmc "$@" --output-format csv -d /tmp/q -- python /root/repo/tools/micro/depth_sort_probe.py > /tmp/lq.txt 2>&1; f=$(find /tmp/q -name "*counter_collection.csv" | head -1); echo "== pmc $*"; summ $f; }
{ pass SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS
  pass SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU
  pass FETCH_SIZE WRITE_SIZE
  pass SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_INSTS_VMEM
  pass TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_HIT_sum TCC_MISS_sum; } >> $O/${tag}_depth_sort_probe.txt 2>&1
cat $O/${tag}_depth_sort_probe.txt
