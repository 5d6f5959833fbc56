#!/usr/bin/env bash
# Which HIP API calls does `gpuar c` / `gpuar d` make?  rocprofv3 --hip-trace --stats of the CLI on a 2 GiB file
# (no counters in the same run).  The steady state must hold no hipDeviceSynchronize and no NULL-stream copy: every
# lane works on its own non-blocking stream and reads its own status word.
# Usage (GPU box, repository root): bash tools/cli_hip_trace.sh [GiB, default 2]   -> gpurun_out/cli_hip_trace/summary.txt
set -u
G=${1:-2}
repo="$(pwd)"
out="$repo/gpurun_out/cli_hip_trace"
rm -rf "$out"; mkdir -p "$out"
python3 - <<PY
from gpuar_amd import synth
n = int($G * (1 << 30))
with open("/tmp/t.dat", "wb") as f:
    step = 1 << 28
    for off in range(0, n, step):
        synth.uniform(42, min(step, n - off), offset=off).tofile(f)
PY
cd /tmp && export TMPDIR=/tmp GPUAR_NO_FAST_EXIT=1
rocprofv3 --hip-trace --stats --output-format csv -d "$out/c" -- "$repo/gpuar_amd/bin/gpuar" c --in=/tmp/t.dat --out=/tmp/t.gip --batch=8192 > "$out/c.log" 2>&1
rocprofv3 --hip-trace --stats --output-format csv -d "$out/d" -- "$repo/gpuar_amd/bin/gpuar" d --in=/tmp/t.gip --out=/tmp/t.back --batch=8192 > "$out/d.log" 2>&1
cmp /tmp/t.dat /tmp/t.back && echo "roundtrip-ok" > "$out/roundtrip.txt"
cd "$repo"
{
  echo "== rocprofv3 --hip-trace --stats -- gpuar c|d, $G GiB uniform(42), --batch=8192 (32 chunks of 64 MiB over 3 lanes)"
  for w in c d; do
    echo "-- gpuar $w: HIP API calls (name, calls, total ns, ...)"
    f="$(ls "$out/$w"/*/*hip_api_stats.csv 2>/dev/null | head -1)"
    [ -n "$f" ] && cut -d, -f1-4 "$f"
    echo "   hipDeviceSynchronize calls: $(grep -c hipDeviceSynchronize "$out/$w"/*/*hip_api_trace.csv 2>/dev/null || true)"
  done
  cat "$out/roundtrip.txt" 2>/dev/null
} > "$out/summary.txt"
cat "$out/summary.txt"
rm -f /tmp/t.dat /tmp/t.gip /tmp/t.back
