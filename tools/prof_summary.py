#!/usr/bin/env python3
"""Condenses a tools/prof.sh output directory into a per-kernel table (durations + PMC sums)."""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name):
    for k in ("encode_small_kernel", "encode_kernel", "decode_slots_kernel", "decode_stream_kernel", "gather_kernel", "scan_", "generate_"):
        if k in name:
            return k.rstrip("_")
    return name[:40]


def main(out):
    dur = defaultdict(list)
    for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            dur[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    print("== kernel durations (ms): name count avg min max")
    for k, v in sorted(dur.items()):
        print(f"{k:28s} {len(v):4d} {sum(v)/len(v):10.3f} {min(v):10.3f} {max(v):10.3f}")
    pmc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(out, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            pmc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("== PMC counters: per-dispatch average")
    for k in sorted(pmc):
        if k.startswith("generate") or k.startswith("scan"):
            continue
        print(f"-- {k}")
        for c, v in sorted(pmc[k].items()):
            print(f"   {c:28s} {sum(v)/len(v):18.1f}  (n={len(v)})")
    for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
        print("== rocprofv3 --stats:", os.path.relpath(f, out))
        print(open(f).read())


if __name__ == "__main__":
    main(sys.argv[1])
