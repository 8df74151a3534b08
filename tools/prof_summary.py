#!/usr/bin/env python
"""Condense rocprofv3 CSV outputs into small per-kernel summaries (for profiles/).

  python tools/prof_summary.py stats  <kernel_stats.csv> <n_launches_per_kernel> [out.csv]
  python tools/prof_summary.py pmc    <counter_collection.csv> [out.csv]
  python tools/prof_summary.py seq    <kernel_trace.csv> <anchor kernel substring> [out.txt] [k]
      time-ordered kernel sequence of one step (from the k-th launch of the anchor kernel to the next; default: the last full step —
      in a bench.py run that is a step of the INSTRUMENTED pass, with its three event records; give k = warm-up + steps / 2 for a step
      of the timed region): start offset, duration, idle gap before the launch (us)
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


def seq(path, anchor, out=None, k=None):
    rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(rows) if anchor in r["Kernel_Name"]]
    a, b = (idx[-2], idx[-1]) if k is None else (idx[k], idx[k + 1])
    t0 = int(rows[a]["Start_Timestamp"])
    lines, prev_end, busy = [], None, 0
    for r in rows[a:b]:
        st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        gap = 0.0 if prev_end is None else (st - prev_end) / 1e3
        lines.append("%9.1f %8.1f %7.1f  %s" % ((st - t0) / 1e3, (en - st) / 1e3, gap, short(r["Kernel_Name"])))
        prev_end = en
        busy += en - st
    span = (int(rows[b]["Start_Timestamp"]) - t0) / 1e3
    lines.append("step span %.1f us, kernels busy %.1f us, %d launches%s" % (span, busy / 1e3, b - a,
                 "" if k is None else " (step %d of %d in the trace)" % (k, len(idx))))
    text = "start_us   dur_us  gap_us  kernel\n" + "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(text)
    print(text)


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
    if sys.argv[1] == "seq":
        seq(sys.argv[2], sys.argv[3], sys.argv[4] if len(sys.argv) > 4 else None, int(sys.argv[5]) if len(sys.argv) > 5 else None)
    elif sys.argv[1] == "stats":
        stats(sys.argv[2], int(sys.argv[3]), sys.argv[4] if len(sys.argv) > 4 else None)
    else:
        pmc(sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else None)
