#!/usr/bin/env python
"""Condense rocprofv3 CSV outputs into small per-kernel summaries (for profiles/).

  python tools/prof_summary.py stats  <kernel_stats.csv> <n_launches_per_kernel> [out.csv]
  python tools/prof_summary.py pmc    <counter_collection.csv> [out.csv]
"""
import collections
import csv
import re
import sys


def short(n):
    n = re.sub(r"rocprim::ROCPRIM_\d+_NS::", "rocprim::", n)
    m = re.match(r"(void )?(gspl::\w+(<[^>]*>)?)", n)
    if m:
        return m.group(2)
    for k in ("radix_sort_onesweep_iteration", "radix_sort_onesweep_global_offsets", "scan_impl", "init_lookback_scan_state",
              "radix_sort_block_sort", "radix_sort_merge", "block_sort", "merge"):
        if k in n:
            t = re.search(r"unsigned (long|int), unsigned int", n)
            return "rocprim::" + k + ("<" + t.group(0) + ">" if t else "")
    m = re.search(r"at::native::(\w+)<[^,]*,\s*at::native::([\w:]+)", n)
    if m:
        return "at::%s<%s>" % (m.group(1), m.group(2)[:30])
    return n[:70]


def stats(path, launches, out=None):
    rows = list(csv.DictReader(open(path)))
    lines = ["kernel,calls,avg_us,ms_per_step,pct"]
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    for r in rows:
        lines.append('"%s",%s,%.1f,%.4f,%s' % (short(r["Name"]), r["Calls"], float(r["AverageNs"]) / 1e3,
                                               float(r["TotalDurationNs"]) / launches / 1e6, r["Percentage"]))
    lines.append('"TOTAL",,,%.4f,100' % (tot / launches / 1e6))
    text = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(text)
    print("\n".join(lines[:34] + lines[-1:]))


def pmc(path, out=None):
    rows = list(csv.DictReader(open(path)))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    counters = sorted({c for d in agg.values() for c in d})
    lines = ["kernel,launches," + ",".join(counters)]
    for k, d in sorted(agg.items(), key=lambda kv: -sum(sum(v) for v in kv[1].values())):
        n = max(len(v) for v in d.values())
        lines.append('"%s",%d,' % (k, n) + ",".join("%.1f" % (sum(d[c]) / len(d[c])) if d.get(c) else "" for c in counters))
    text = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(text)
    print(text[:4000])


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2], int(sys.argv[3]), sys.argv[4] if len(sys.argv) > 4 else None)
    else:
        pmc(sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else None)
