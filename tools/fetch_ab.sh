#!/usr/bin/env bash
# FETCH_SIZE / WRITE_SIZE of decode_slots_kernel for a list of library builds (same method as bench.py live_traffic)
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/fab_$c
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/fab_$c -- python3 $GRAFT_REPO_ROOT/tools/prof_run.py --gib 8 --only both --reps 2 --lib $GRAFT_REPO_ROOT/gpuar_amd/lib/exp/$lib.so > /tmp/fab.log 2>&1
    python3 - "$lib" "$c" <<'PY'
import csv, glob, sys
lib, c = sys.argv[1:3]
v = {}
for f in glob.glob(f"/tmp/fab_{c}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = "decode" if "decode_slots" in r["Kernel_Name"] else "encode" if "encode_kernel" in r["Kernel_Name"] else None
        if k and r["Counter_Name"] == c:
            v.setdefault(k, []).append(float(r["Counter_Value"]) * 1024 / 1e9)
print(f"{lib:12s} {c:10s}", {k: [round(x, 3) for x in a] for k, a in v.items()})
PY
  done
done
