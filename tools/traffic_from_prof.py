#!/usr/bin/env python3
"""Turns a tools/prof.sh output directory into profiles/<tag>_traffic.json: per-launch HBM
traffic of the encode and decode kernels from the FETCH_SIZE / WRITE_SIZE PMC passes, plus the
issue-side counters that show what really binds them (vector-pipe busy and wait share of the
wavefronts' cycles, vector instructions per symbol step).

Units and corrections follow /opt/skills/guides/MI355X_MICROARCH.md section HBM:
rocprofv3 reports both in KiB; on gfx950 FETCH_SIZE tallies 128-byte requests at 64 bytes,
i.e. it shows HALF the bytes of a 16-byte-per-lane streaming read, so it is doubled here;
WRITE_SIZE is exact for 16-byte-per-lane stores.  (Calibration in this repo's own pattern:
the decode kernel must read every compressed byte at least once and its raw FETCH_SIZE is
0.55x that byte count, which the doubling turns into 1.11x.)

The record is stamped with the sha256 of the kernel sources it was taken from; bench.py quotes
it only while that stamp matches the sources that are built.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(out_dir, tag, gib, kind="uniform"):
    from bench import kernel_source_stamp
    vals = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(out_dir, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"]
            key = ("encode" if "encode_kernel" in name or "encode_small_kernel" in name else "decode" if "decode_slots_kernel" in name else
                   "decode_stream" if "decode_stream_kernel" in name else "gather" if "gather_kernel" in name else None)
            if key:
                vals[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    res = {"source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), tools/prof_run.py --gib {gib} --kind {kind}",
           "input_gib": gib, "kind": kind, "fetch_correction": 2.0, "kernel_source_sha256_16": kernel_source_stamp(ROOT)}
    symbol_steps = gib * (1 << 30) / 64.0            # one step = 64 lanes x one byte each
    for k, d in vals.items():
        avg = lambda c: sum(d[c]) / max(1, len(d[c]))   # noqa: E731
        f, w = avg("FETCH_SIZE") * 1024.0, avg("WRITE_SIZE") * 1024.0
        res[k] = {"fetch_size_bytes_raw": f, "write_size_bytes": w, "hbm_bytes_per_launch": 2.0 * f + w}
        if d.get("SQ_WAVE_CYCLES"):
            # per WAVEFRONT: share of its resident cycles in which it issued a vector instruction / sat in a wait
            res[k]["valu_busy"] = avg("SQ_ACTIVE_INST_VALU") / avg("SQ_WAVE_CYCLES")
            res[k]["wait_frac"] = avg("SQ_WAIT_ANY") / avg("SQ_WAVE_CYCLES")
        if d.get("SQ_ACTIVE_INST_VALU") and d.get("GRBM_GUI_ACTIVE"):
            # per SIMD: vector-issue quad-cycles of all wavefronts over the cycles the 1024 SIMDs had (the encoder keeps
            # four wavefronts on a SIMD, so its per-wavefront figure is a quarter of what the SIMD sees)
            # (rocprofv3 sums GRBM_GUI_ACTIVE over the 8 XCDs; the decoder, one wavefront per SIMD, is the cross-check:
            # both ways of counting give it the same 0.70)
            res[k]["valu_busy_per_simd"] = avg("SQ_ACTIVE_INST_VALU") * 4.0 / (avg("GRBM_GUI_ACTIVE") / 8.0 * 1024.0)
        if d.get("SQ_INSTS_VALU"):
            res[k]["valu_insts_per_symbol_step"] = avg("SQ_INSTS_VALU") / symbol_steps
        if d.get("SQ_INSTS_LDS"):
            res[k]["lds_insts_per_symbol_step"] = avg("SQ_INSTS_LDS") / symbol_steps
    path = os.path.join(ROOT, "profiles", f"{tag}_traffic.json" if kind == "uniform" and gib == 8 else f"{tag}_traffic_{kind}_{gib:g}gib.json")
    json.dump(res, open(path, "w"), indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], float(sys.argv[3]), *(sys.argv[4:5]))
