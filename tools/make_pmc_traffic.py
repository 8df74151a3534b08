#!/usr/bin/env python
"""profiles/<tag>_pmc_traffic.json from the PMC summaries of one collection (tools/collect_profiles_r06.sh):

    python tools/make_pmc_traffic.py <dir with <tag>_pmc_FETCH_SIZE.csv, <tag>_pmc_WRITE_SIZE.csv[, <tag>_pmc_SQ.csv]> <tag> <workload/api> [out.json]

bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: FETCH_SIZE / WRITE_SIZE are in KiB, the factor 2 is the gfx950 FETCH_SIZE
correction of MI355X_MICROARCH.md (HBM section); WRITE_SIZE uncalibrated (atomics count as writes).  The file records the ABI version
and a fingerprint of the compositing kernels' sources of the tree it was made in: bench.py refuses it for any other code state."""
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def table(path):
    out = {}
    if os.path.exists(path):
        for r in csv.DictReader(open(path)):
            out[r["kernel"]] = {k: float(v) for k, v in r.items() if k != "kernel" and v != ""}
    return out


def is_follow_up(full):
    """The second launch of a SEGMENTED backward (composite_bwd2_kernel<..., 2>): it belongs to the call whose first launch is <..., 1>."""
    m = re.match(r"gspl::composite_bwd2_kernel<(.*)>", full)
    return bool(m) and m.group(1).split(",")[-1].strip() == "2" and len(m.group(1).split(",")) >= 6


def main():
    d, tag, key = sys.argv[1], sys.argv[2], sys.argv[3]
    out_path = sys.argv[4] if len(sys.argv) > 4 else os.path.join(ROOT, "profiles", f"{tag}_pmc_traffic.json")
    import bench
    import gspl_amd  # noqa: F401
    from gspl_amd import _lib
    fetch, write, sq = (table(os.path.join(d, f"{tag}_pmc_{n}.csv")) for n in ("FETCH_SIZE", "WRITE_SIZE", "SQ"))
    doc = {"_comment": "HBM traffic per launch from rocprofv3 PMC passes (separate --pmc FETCH_SIZE / WRITE_SIZE / SQ runs of bench.py; "
                       "tools/collect_profiles_r06.sh + tools/make_pmc_traffic.py): bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024, the factor 2 "
                       "being the gfx950 FETCH_SIZE correction of MI355X_MICROARCH.md; WRITE_SIZE uncalibrated (atomics are counted as writes); "
                       "means over the launches of the run (the steps cycle through the camera set).",
           "_measured_on": {"abi_version": _lib.ABI_VERSION, "kernel_source_sha16": bench.kernel_source_sha16(),
                            "summaries": [f"profiles/{tag}_pmc_FETCH_SIZE.csv", f"profiles/{tag}_pmc_WRITE_SIZE.csv", f"profiles/{tag}_pmc_SQ.csv"]},
           key: {}}
    # One entry per kernel TEMPLATE, per CALL of the entry point that launches it: the instantiations of a template (the plain and the
    # segmented compositing backward, round 6: a view with a tail takes <..., 1> + <..., 2>, the others <..., 0>) are weighted by their
    # launches, and the segmented form's second launch is added to its call instead of counting as one — what `roofline.avg_ms` brackets.
    groups = {}
    for full, f in fetch.items():
        m = re.match(r"gspl::(composite_\w+_kernel)", full)
        if not m or full not in write:
            continue
        groups.setdefault(m.group(1), []).append(full)
    for name, fulls in groups.items():
        calls = sum(fetch[x].get("launches", 1.0) for x in fulls if not is_follow_up(x))
        if calls <= 0:
            continue
        fs = sum(fetch[x]["FETCH_SIZE"] * fetch[x].get("launches", 1.0) for x in fulls) / calls
        ws = sum(write[x]["WRITE_SIZE"] * write[x].get("launches", fetch[x].get("launches", 1.0)) for x in fulls) / sum(
            write[x].get("launches", fetch[x].get("launches", 1.0)) for x in fulls if not is_follow_up(x))
        entry = {"kernel": " + ".join(x.replace("gspl::", "") for x in sorted(fulls)), "fetch_size_kb": round(fs, 1), "write_size_kb": round(ws, 1),
                 "traffic_bytes": int(round((2 * fs + ws) * 1024)), "source": f"profiles/{tag}_pmc_*.csv",
                 "calls": int(calls), "launches": {x.replace("gspl::", ""): int(fetch[x].get("launches", 1)) for x in sorted(fulls)}}
        main_inst = max((x for x in fulls if not is_follow_up(x)), key=lambda x: fetch[x].get("launches", 1.0))
        if main_inst in sq:
            entry["sq"] = {k.lower(): v for k, v in sq[main_inst].items() if k != "launches"}
            entry["sq_of"] = main_inst.replace("gspl::", "")
        doc[key][name] = entry
    json.dump(doc, open(out_path, "w"), indent=1)
    print(out_path, list(doc[key]))


if __name__ == "__main__":
    main()
