"""Sums rocprofv3 counter_collection CSVs per kernel name and counter.
usage: python tools/pmc_summary.py <dir-with-pass-subdirs> [kernel-substring]"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def short(name):
    for k in ("k_traverse", "k_extend", "k_shade", "k_bin", "k_raygen", "k_accumulate", "k_resolve", "k_intersect"):
        if k in name:
            return k
    return name[:40]


def main(root, want=None):
    sums = defaultdict(lambda: defaultdict(float))
    counts = defaultdict(lambda: defaultdict(int))
    for path in sorted(glob.glob(os.path.join(root, "*", "*_counter_collection.csv"))):
        with open(path) as f:
            for row in csv.DictReader(f):
                k = short(row["Kernel_Name"])
                if want and want not in k:
                    continue
                sums[k][row["Counter_Name"]] += float(row["Counter_Value"])
                counts[k][row["Counter_Name"]] += 1
    out = {k: {c: v for c, v in d.items()} for k, d in sums.items()}
    for k in out:
        out[k]["_dispatches"] = max(counts[k].values())
    print(json.dumps(out, indent=1, sort_keys=True))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
