"""Per-kernel averages of the rocprofv3 --pmc passes (one directory per counter group, devtools/gpu_pmc.sh)
-> profiles/rNN/pmc/pmc_summary.csv.   usage: python profiles/pmc_summary.py gpurun_out/pmc profiles/r01/pmc"""
import collections
import csv
import glob
import os
import sys


def main(src, dst):
    vals = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(src, "*", "pmc_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            vals[r["Kernel_Name"][:120]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    counters = sorted({c for k in vals.values() for c in k})
    os.makedirs(dst, exist_ok=True)
    with open(os.path.join(dst, "pmc_summary.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "launches"] + counters)
        for k in sorted(vals):
            n = max(len(v) for v in vals[k].values())
            w.writerow([k, n] + [round(sum(vals[k][c]) / len(vals[k][c]), 1) if vals[k][c] else "" for c in counters])


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
