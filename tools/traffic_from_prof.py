#!/usr/bin/env python3
"""Turns a tools/prof.sh output directory into profiles/<tag>_traffic.json: per-launch HBM
traffic of the encode and decode kernels from the FETCH_SIZE / WRITE_SIZE PMC passes.

Units and corrections follow /opt/skills/guides/MI355X_MICROARCH.md section HBM:
rocprofv3 reports both in KiB; on gfx950 FETCH_SIZE tallies 128-byte requests at 64 bytes,
i.e. it shows HALF the bytes of a 16-byte-per-lane streaming read, so it is doubled here;
WRITE_SIZE is exact for 16-byte-per-lane stores.  (Calibration in this repo's own pattern:
the decode kernel must read every compressed byte at least once and its raw FETCH_SIZE is
0.55x that byte count, which the doubling turns into 1.11x.)
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def main(out_dir, tag, gib):
    vals = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(out_dir, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"]
            key = "encode" if "encode_kernel" in name else "decode" if "decode_slots_kernel" in name else None
            if key and r["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
                vals[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    res = {"source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), tools/prof_run.py --gib {gib}",
           "input_gib": gib, "fetch_correction": 2.0}
    for k, d in vals.items():
        f = sum(d["FETCH_SIZE"]) / max(1, len(d["FETCH_SIZE"])) * 1024.0
        w = sum(d["WRITE_SIZE"]) / max(1, len(d["WRITE_SIZE"])) * 1024.0
        res[k] = {"fetch_size_bytes_raw": f, "write_size_bytes": w, "hbm_bytes_per_launch": 2.0 * f + w}
    path = os.path.join("profiles", f"{tag}_traffic.json")
    json.dump(res, open(path, "w"), indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], float(sys.argv[3]))
