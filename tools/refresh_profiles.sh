#!/usr/bin/env bash
# Regenerates everything under profiles/ for a round tag (default r04), in three stages that each fit one gpurun call
# (a call is limited to 20 minutes).  Run on the GPU box from the repository root:
#     gpurun --timeout 1100 -- 'bash tools/refresh_profiles.sh r04 uniform'
#     gpurun --timeout 1100 -- 'bash tools/refresh_profiles.sh r04 others'
#     gpurun --timeout 1100 -- 'bash tools/refresh_profiles.sh r04 bench'
# then copy gpurun_out/profiles_<tag>/* into profiles/ (gpurun_out/ is what travels back).
#   uniform : kernel trace + stats and all PMC passes on the bench workload (uniform 8 GiB): encode, compaction,
#             decode (slots), decode (stream) -> <tag>_kernel_stats.csv, <tag>_rocprofv3_summary.txt, <tag>_traffic.json
#   others  : the traffic / instruction-count passes for text(1) and zipf(1) 8 GiB and for the 64 MiB case
#   bench   : bench.py lines (uniform with CPU baseline, text, zipf), hardware probes, kernel time by stream kind and by
#             occupancy, CLI wall times with the pipeline's own timeline
set -u
tag="${1:-r04}"
stage="${2:-uniform}"
out="gpurun_out/profiles_$tag"
mkdir -p "$out"
case "$stage" in
uniform)
    rm -rf "gpurun_out/prof_$tag"
    timeout -k 10 1000 bash tools/prof.sh "$tag" --gib 8 --reps 2 > /dev/null 2>&1
    python3 tools/traffic_from_prof.py "gpurun_out/prof_$tag" "$tag" 8 uniform > /dev/null
    cp "profiles/${tag}_traffic.json" "$out/"
    cp "gpurun_out/prof_$tag/summary.txt" "$out/${tag}_rocprofv3_summary.txt"
    stats="$(grep -l encode_kernel gpurun_out/prof_$tag/trace/*/*_kernel_stats.csv | head -1)"
    [ -n "$stats" ] && cp "$stats" "$out/${tag}_kernel_stats.csv"
    ;;
others)
    for spec in text:8 zipf:8 uniform:0.0625; do
        kind="${spec%%:*}"; gib="${spec##*:}"
        name="${tag}_${kind}_${gib}gib"
        rm -rf "gpurun_out/prof_$name"
        timeout -k 10 330 bash tools/prof.sh "$name" --quick --gib "$gib" --reps 2 --kind "$kind" > /dev/null 2>&1
        python3 tools/traffic_from_prof.py "gpurun_out/prof_$name" "$tag" "$gib" "$kind" > /dev/null
        cp "profiles/${tag}_traffic_${kind}_${gib}gib.json" "$out/" 2>/dev/null
        cp "gpurun_out/prof_$name/summary.txt" "$out/${name}_rocprofv3_summary.txt"
        echo "$name done" >> "$out/progress.log"
    done
    ;;
bench)
    # the stdout line (<= 4 KB) and, next to it, the full object it was cut out of
    timeout -k 10 400 python3 bench.py --detail-file "$out/${tag}_bench_detail.json" > "$out/${tag}_bench.json" 2> "$out/${tag}_bench.err"
    # the kernel statistics of that very command (a shorter run; TMPDIR for the profiler's scratch files)
    rm -rf "gpurun_out/prof_bench_$tag"
    ( export TMPDIR=/tmp; timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "gpurun_out/prof_bench_$tag" -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-live-traffic --detail-file /tmp/bench_detail_under_rocprof.json > /dev/null 2>&1 )
    stats="$(find "gpurun_out/prof_bench_$tag" -name '*kernel_stats.csv' | head -1)"
    [ -n "$stats" ] && cp "$stats" "$out/${tag}_bench_kernel_stats.csv"
    for kind in text zipf; do
        timeout -k 10 200 python3 bench.py --kind "$kind" --seed 1 --gib-per-gpu 8 --no-cpu-baseline --no-small-config --no-by-kind --no-live-traffic \
            --detail-file "$out/${tag}_bench_${kind}_detail.json" > "$out/${tag}_bench_$kind.json" 2>/dev/null
    done
    for probe in valu_probe lds_probe ldsbw_probe placement_probe lat_probe stride_probe active_probe mix_probe halfexec_probe xlane_probe copy_probe latsearch_probe; do
        [ -x "tools/$probe.bin" ] && timeout -k 10 200 "./tools/$probe.bin" > "$out/${tag}_$probe.txt" 2>&1
    done
    [ -x tools/io_probe.bin ] && timeout -k 10 200 ./tools/io_probe.bin /tmp 2 > "$out/${tag}_io_probe.txt" 2>&1
    timeout -k 10 200 python3 tools/kind_timing.py --gib 2 --kinds uniform,text,zipf,zeros > "$out/${tag}_kind_timing.txt" 2>&1
    for p in 64 1024 16384 65536; do
        printf "%6d packets: " "$p" >> "$out/${tag}_occupancy_timing.txt"
        timeout -k 10 100 python3 tools/kind_timing.py --gib "$(python3 -c "print($p*8192/2**30)")" --kinds uniform | cut -c1-150 >> "$out/${tag}_occupancy_timing.txt" 2>&1
    done
    timeout -k 10 300 bash tools/cli_timing.sh 8 /tmp > "$out/${tag}_cli_timing.txt" 2>&1
    ;;
*)
    echo "unknown stage $stage (uniform | others | bench)" >&2
    exit 2
    ;;
esac
ls -la "$out"
