#!/usr/bin/env bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate rocprofv3 --pmc passes) and kernel time of encode_kernel for several
# experiment builds of the library: tools/pmc_libs.sh [--gib G] NAME...   (gpuar_amd/lib/exp/NAME.so, tools/exp_build.sh)
# Run on the GPU box from the repository root.  How the cache-policy A/B of the coder's stores was measured
# (profiles/r05_encoder_attribution.txt section 3).
gib=8
if [ "$1" = "--gib" ]; then gib="$2"; shift 2; fi
repo=$(pwd); out=$repo/gpurun_out/pmc_libs; rm -rf "$out"; mkdir -p "$out"; cd /tmp; export TMPDIR=/tmp
for lib in "$@"; do
    [ -f "$repo/gpuar_amd/lib/exp/$lib.so" ] || { echo "missing gpuar_amd/lib/exp/$lib.so" >&2; exit 2; }
    for c in FETCH_SIZE WRITE_SIZE; do
        rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$out/${lib}_$c" -- python3 "$repo/tools/prof_run.py" --gib "$gib" --reps 2 --only encode \
            --lib "$repo/gpuar_amd/lib/exp/$lib.so" > "$out/${lib}_$c.log" 2>&1
    done
    echo "$lib counted" >> "$out/progress.log"
done
cd "$repo"
python3 - "$gib" <<'PY'
import csv, glob, collections, sys
gib = float(sys.argv[1])
n = int(gib * (1 << 30))
for d in sorted(glob.glob('gpurun_out/pmc_libs/*/')):
    acc = collections.defaultdict(list)
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if 'encode_kernel' in r['Kernel_Name']:
                acc[r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in sorted(acc.items()):
        b = sum(v) / len(v) * 1024 * (2 if k == 'FETCH_SIZE' else 1)          # KiB units; FETCH_SIZE doubled on gfx950 (MI355X_MICROARCH.md)
        print(f"{d.split('/')[-2]:24s} {k:10s} {b / 1e9:8.3f} GB per launch = {b / n:6.3f} x the {gib:g} GiB of input")
PY
