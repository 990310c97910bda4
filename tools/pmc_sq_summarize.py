"""Summarise rocprofv3 --pmc SQ/GRBM passes over tools/profile_eval.py into profiles/rNN_pmc_sq.json.

usage: pmc_sq_summarize.py <out.json> <label>=<counter_collection.csv> ...
Per kernel (mean per dispatch): every counter collected, plus
  mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs)   (fp64 16x16x4 MFMA = 64 busy cycles)
GRBM_GUI_ACTIVE is summed over the 8 XCDs by rocprofv3 (MI355X_MICROARCH.md, DVFS note)."""
import collections
import csv
import json
import re
import sys


def load(path):
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    with open(path) as f:
        for r in csv.DictReader(f):
            name = re.sub(r"\(.*$", "", re.sub(r"^void ", "", r["Kernel_Name"]))
            a = acc[name][r["Counter_Name"]]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    return acc


def main():
    out = sys.argv[1]
    doc = {"note": __doc__.split("usage")[0].strip() + " Command: python3 tools/profile_eval.py M (n=4096 d=8 f64, one evaluation stream); "
                   "labels: dag = HBEGP_DAG=1 (task-queue launch, 256 workgroups), launches = HBEGP_DAG=0 (launch-per-product path).",
           "runs": {}}
    for spec in sys.argv[2:]:
        label, path = spec.split("=", 1)
        acc = load(path)
        kernels = {}
        for name, ctrs in acc.items():
            if not name.startswith("hbegp::"):
                continue
            row = {c: v[1] / max(1, v[0]) for c, v in ctrs.items()}
            row["dispatches"] = max(v[0] for v in ctrs.values())
            if row.get("GRBM_GUI_ACTIVE", 0) > 0 and "SQ_VALU_MFMA_BUSY_CYCLES" in row:
                row["mfma_busy_frac"] = row["SQ_VALU_MFMA_BUSY_CYCLES"] / (row["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
            if row.get("SQ_WAVE_CYCLES", 0) > 0:
                for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
                    if c in row:
                        row[c + "_frac_of_wave_cycles"] = row[c] / row["SQ_WAVE_CYCLES"]
            kernels[name] = row
        doc["runs"][label] = kernels
    json.dump(doc, open(out, "w"), indent=1)
    for label, ks in doc["runs"].items():
        for k, v in ks.items():
            if "mfma_busy_frac" in v:
                print(label, k, "dispatches", v["dispatches"], "mfma_busy_frac %.3f" % v["mfma_busy_frac"],
                      "lds_conflict/idx %.4f" % (v.get("SQ_LDS_BANK_CONFLICT", 0) / max(1.0, v.get("SQ_LDS_IDX_ACTIVE", 1))))


if __name__ == "__main__":
    main()
