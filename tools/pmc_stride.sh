#!/usr/bin/env bash
# FETCH_SIZE / WRITE_SIZE of the codec kernels for two builds that differ in the slot stride (8704 vs 8832 bytes):
# is the encoder's write amplification an L2 set-aliasing effect of the 8704-byte stride?  (It is not: DESIGN.md 4.2.)
# Usage (GPU box, repository root):  bash tools/pmc_stride.sh
#   expects gpuar_amd/lib/exp/base.so and gpuar_amd/lib/exp/slot8832.so
#   (tools/exp_build.sh base "" ; tools/exp_build.sh slot8832 "-DGPUAR_SLOT_BYTES=8832u")
for lib in base slot8832; do [ -f "gpuar_amd/lib/exp/$lib.so" ] || { echo "missing gpuar_amd/lib/exp/$lib.so (see the usage note in this script)" >&2; exit 2; }; done
repo=$(pwd); out=$repo/gpurun_out/pmc_stride; rm -rf $out; mkdir -p $out; cd /tmp; export TMPDIR=/tmp
for v in base:8704 slot8832:8832; do lib=${v%%:*}; sl=${v##*:};
 for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/${lib}_$c -- python3 $repo/tools/prof_run.py --gib 4 --reps 2 --lib $repo/gpuar_amd/lib/exp/$lib.so --slot $sl > $out/${lib}_$c.log 2>&1
 done
done
cd $repo; python3 - <<'PY'
import csv,glob,collections
for d in sorted(glob.glob('gpurun_out/pmc_stride/*/')):
    acc=collections.defaultdict(list)
    for f in glob.glob(d+'/**/*counter_collection.csv',recursive=True):
        for r in csv.DictReader(open(f)):
            acc[(r['Kernel_Name'][:30],r['Counter_Name'])].append(float(r['Counter_Value']))
    for k,v in sorted(acc.items()):
        if 'code' in k[0]: print(d.split('/')[-2],k,sum(v)/len(v)*1024/1e9,'GB (KiB units; FETCH x2 needed)')
PY
