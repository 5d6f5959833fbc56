#!/usr/bin/env bash
# The CLI's own timeline of one 8 GiB compression, twice (GPUAR_TRACE=1): HIP runtime start, buffer allocation, first
# registration of the mapped input, first piece at the writer, end of the writer.  Usage (via gpurun): bash tools/cli_trace_once.sh
set -u
D=/tmp; B=gpuar_amd/bin/gpuar
python3 - <<PY
from gpuar_amd import synth
n = 8 << 30
with open("$D/u.dat", "wb") as f:
    step = 1 << 28
    for off in range(0, n, step):
        synth.uniform(42, min(step, n - off), offset=off).tofile(f)
PY
for i in 1 2; do
s=$(date +%s%N); GPUAR_TRACE=1 $B c --in=$D/u.dat --out=$D/u.gip > $D/log 2>&1; e=$(date +%s%N); echo "wall $(( (e - s) / 1000000 )) ms"; grep "gpuar" $D/log | tr -d '\r' | sed -e 's/^.*\[gpuar/[gpuar/' | head -14
done
