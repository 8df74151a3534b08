#!/bin/bash
# usage (on the GPU box): tools/pmc_quick.sh "<bench args>" <kernel substring> <counter>...   (one rocprofv3 --pmc pass; per-kernel means)
args=$1; pat=$2; shift 2
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmcq
rocprofv3 --pmc "$@" --output-format csv -d /tmp/pmcq -- python /root/repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-renderer-only $args > /tmp/logq.txt 2>&1
f=$(find /tmp/pmcq -name "*counter_collection.csv" | head -1)
python - "$f" "$pat" <<'PY'
import csv, sys, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if sys.argv[2] not in k: continue
    k = re.sub(r"\(.*", "", k)[:90]
    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    for c, v in cs.items():
        print("   %-28s mean %14.1f  n %d" % (c, sum(v) / len(v), len(v)))
PY
