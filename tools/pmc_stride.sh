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
