#!/usr/bin/env bash
# End-to-end wall time of the gpuar CLI on the GPU box: 1 GiB uniform(42), compress / decompress,
# with and without the packet-offset index.  Usage (via gpurun): bash tools/cli_timing.sh
set -u
B=gpuar_amd/bin/gpuar
D=${TMPDIR:-/tmp}
python3 - <<PY
from gpuar_amd import synth
synth.uniform(42, 1 << 30).tofile("$D/u1g.dat")
PY
t() { local s=$(date +%s%N); "$@" > "$D/cli.log" 2>&1; local rc=$?; local e=$(date +%s%N); printf "%5d ms  rc=%d  %s | %s\n" "$(( (e - s) / 1000000 ))" "$rc" "$*" "$(grep -E 'Compute time|I/O time' "$D/cli.log" | tr -s ' ' | tr '\n' ' ')"; }
t $B c --in=$D/u1g.dat --out=$D/u1g.gip
t $B c --in=$D/u1g.dat --out=$D/u1g.gip
t $B c --index --in=$D/u1g.dat --out=$D/u1g_idx.gip
t $B d --in=$D/u1g.gip --out=$D/u1g.back
t $B d --in=$D/u1g.gip --out=$D/u1g.back
t $B d --in=$D/u1g_idx.gip --out=$D/u1g.back2
t $B d --in=$D/u1g_idx.gip --out=$D/u1g.back2
cmp $D/u1g.dat $D/u1g.back && cmp $D/u1g.dat $D/u1g.back2 && echo roundtrip-ok
rm -f $D/u1g.dat $D/u1g.gip $D/u1g_idx.gip $D/u1g.back $D/u1g.back2
