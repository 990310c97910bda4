"""Summarise rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) into profiles/rNN_pmc_traffic.json.

usage: pmc_summarize.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json> [profiled command]
Counter unit is KB; on gfx950 reads are bytes = 2 * FETCH_SIZE * 1024 (MI355X_MICROARCH.md, HBM section), writes
WRITE_SIZE * 1024.  Per-kernel means over the dispatches of the profiled command (tools/profile_eval.py M).
"""
import collections
import csv
import json
import re
import sys


def per_kernel(path, counter):
    acc = collections.defaultdict(lambda: [0, 0.0])
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            name = re.sub(r"^void ", "", r["Kernel_Name"])
            name = re.sub(r"\(.*$", "", name)
            a = acc[name]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    return acc


def main():
    fetch, write, out = sys.argv[1:4]
    cmd = sys.argv[4] if len(sys.argv) > 4 else "python3 tools/profile_eval.py M"
    fa, wa = per_kernel(fetch, "FETCH_SIZE"), per_kernel(write, "WRITE_SIZE")
    kernels = {}
    for name in fa:
        if not name.startswith("hbegp::"):
            continue
        n = fa[name][0]
        kernels[name] = {
            "dispatches": n,
            "fetch_bytes_per_dispatch": 2.0 * 1024.0 * fa[name][1] / n,
            "write_bytes_per_dispatch": 1024.0 * wa[name][1] / max(1, wa[name][0]) if name in wa else None,
        }
    doc = {
        "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in two separate passes over `" + cmd + "` "
                "(n=4096 d=8 f64). Counter unit KB. gfx950 correction applied to reads: bytes = 2 * FETCH_SIZE * 1024 "
                "(MI355X_MICROARCH.md HBM section); WRITE_SIZE * 1024 used as is.",
        "kernels": kernels,
    }
    with open(out, "w") as f:
        json.dump(doc, f, indent=1)
    for k, v in kernels.items():
        print(k, v)


if __name__ == "__main__":
    main()
