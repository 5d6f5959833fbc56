#!/usr/bin/env bash
# Regenerates everything under profiles/ for a round tag (default r02).  Run on the GPU box from the
# repository root, e.g.   gpurun --timeout 1500 -- 'bash tools/refresh_profiles.sh r02'
# then copy gpurun_out/profiles_<tag>/* into profiles/ (gpurun_out/ is what travels back).
set -u
tag="${1:-r02}"
out="gpurun_out/profiles_$tag"
rm -rf "gpurun_out/prof_$tag" "$out"
mkdir -p "$out"
timeout 900 bash tools/prof.sh "$tag" --gib 8 --reps 2 > /dev/null 2>&1
python3 tools/traffic_from_prof.py "gpurun_out/prof_$tag" "$tag" 8 > /dev/null      # writes profiles/<tag>_traffic.json (bench.py reads it)
cp "profiles/${tag}_traffic.json" "$out/"
cp "gpurun_out/prof_$tag/summary.txt" "$out/${tag}_rocprofv3_summary.txt"
stats="$(grep -l encode_kernel gpurun_out/prof_$tag/trace/*/*_kernel_stats.csv | head -1)"
[ -n "$stats" ] && cp "$stats" "$out/${tag}_kernel_stats.csv"
timeout 900 python3 bench.py > "$out/${tag}_bench.json" 2> "$out/${tag}_bench.err"
for kind in text zipf; do
    timeout 600 python3 bench.py --kind "$kind" --seed 1 --gib-per-gpu 8 --no-cpu-baseline --no-small-config > "$out/${tag}_bench_$kind.json" 2>/dev/null
done
for probe in valu_probe lds_probe placement_probe lat_probe stride_probe; do
    [ -x "tools/$probe.bin" ] && timeout 200 "./tools/$probe.bin" > "$out/${tag}_$probe.txt" 2>&1
done
timeout 300 python3 tools/kind_timing.py --gib 2 --kinds uniform,text,zipf,zeros > "$out/${tag}_kind_timing.txt" 2>&1
for p in 64 1024 16384 65536; do
    printf "%6d packets: " "$p" >> "$out/${tag}_occupancy_timing.txt"
    timeout 120 python3 tools/kind_timing.py --gib "$(python3 -c "print($p*8192/2**30)")" --kinds uniform | cut -c1-150 >> "$out/${tag}_occupancy_timing.txt" 2>&1
done
timeout 600 bash tools/cli_timing.sh 8 > "$out/${tag}_cli_timing.txt" 2>&1
ls -la "$out"
