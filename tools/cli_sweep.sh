#!/usr/bin/env bash
# Wall time of `gpuar c` / `gpuar d` on an 8 GiB page-cached file for several chunk sizes (--batch, packets per chunk).
# Usage (GPU box, repository root):  bash tools/cli_sweep.sh
set -u
B=gpuar_amd/bin/gpuar; D=/tmp; G=8
python3 - <<PY
from gpuar_amd import synth
n = int($G * (1 << 30))
with open("$D/u.dat", "wb") as f:
    step = 1 << 28
    for off in range(0, n, step):
        synth.uniform(42, min(step, n - off), offset=off).tofile(f)
PY
t() { local s=$(date +%s%N); "$@" > "$D/cli.log" 2>&1; local rc=$?; local e=$(date +%s%N); local ms=$(( (e - s) / 1000000 )); printf "%6d ms  %6.2f GB/s  rc=%d  %s\n" "$ms" "$(python3 -c "print($G * 1.073741824 / ($ms / 1000.0))")" "$rc" "$*"; }
for b in 4096 8192 16384 32768; do
  rm -f $D/u.gip $D/u.back
  t $B c --batch=$b --in=$D/u.dat --out=$D/u.gip
  t $B d --batch=$b --in=$D/u.gip --out=$D/u.back
done
cmp $D/u.dat $D/u.back && echo roundtrip-ok
rm -f $D/u.dat $D/u.gip $D/u.back
