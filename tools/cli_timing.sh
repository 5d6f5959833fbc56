#!/usr/bin/env bash
# End-to-end wall time of the gpuar CLI on the GPU box (file in the page cache -> GPU -> new file):
# uniform(42), compress / decompress, with and without the packet-offset index, fresh output and overwritten output.
# Usage (via gpurun): bash tools/cli_timing.sh [GiB, default 8] [directory, default $TMPDIR or /tmp]
set -u
G=${1:-8}
D=${2:-${TMPDIR:-/tmp}}
B=gpuar_amd/bin/gpuar
echo "== $G GiB uniform(42) in $D ($(df -T "$D" | tail -1 | awk '{print $2}'))"
python3 - <<PY
from gpuar_amd import synth
n = int($G * (1 << 30))
with open("$D/u.dat", "wb") as f:
    step = 1 << 28
    for off in range(0, n, step):
        synth.uniform(42, min(step, n - off), offset=off).tofile(f)
PY
t() { local s=$(date +%s%N); "$@" > "$D/cli.log" 2>&1; local rc=$?; local e=$(date +%s%N); local ms=$(( (e - s) / 1000000 )); printf "%6d ms  %6.2f GB/s  rc=%d  %s | %s\n" "$ms" "$(python3 -c "print($G * 1.073741824 / ($ms / 1000.0))")" "$rc" "$*" "$(grep -E 'Compute time|I/O time' "$D/cli.log" | tr -s ' ' | tr '\n' ' ')"; }
rm -f $D/u.gip $D/u_idx.gip $D/u.back $D/u.back2
t $B c --in=$D/u.dat --out=$D/u.gip
t $B c --in=$D/u.dat --out=$D/u.gip
t $B c --index --in=$D/u.dat --out=$D/u_idx.gip
t $B d --in=$D/u.gip --out=$D/u.back
t $B d --in=$D/u.gip --out=$D/u.back
t $B d --in=$D/u_idx.gip --out=$D/u.back2
t $B d --in=$D/u_idx.gip --out=$D/u.back2
GPUAR_NO_MMAP=1 t $B c --in=$D/u.dat --out=$D/u_nommap.gip
echo "-- the pipeline's own timeline (GPUAR_TRACE=1): where the wall time goes"
GPUAR_TRACE=1 $B c --in=$D/u.dat --out=$D/u.gip 2>&1 | tr -d '\r' | grep -E "^\[gpuar|\[gpuar" | sed -e 's/^.*\[gpuar/[gpuar/' | sed -e 's/^/   c: /'
GPUAR_TRACE=1 $B d --in=$D/u.gip --out=$D/u.back 2>&1 | tr -d '\r' | grep -E "\[gpuar" | sed -e 's/^.*\[gpuar/[gpuar/' | sed -e 's/^/   d: /'
cmp $D/u.gip $D/u_nommap.gip && echo "unmapped-input file identical"
cmp $D/u.dat $D/u.back && cmp $D/u.dat $D/u.back2 && echo roundtrip-ok
rm -f $D/u.dat $D/u.gip $D/u_idx.gip $D/u.back $D/u.back2 $D/u_nommap.gip
