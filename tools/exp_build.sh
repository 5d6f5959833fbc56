#!/usr/bin/env bash
# tools/exp_build.sh NAME "-DEXP_A=1 -DEXP_B=2"  -> gpuar_amd/lib/exp/NAME.so   (experiment builds for tools/ab_timing.sh)
# Builds from the working tree with extra flags.  A failed build removes the stale library and exits non-zero,
# so ab_timing.sh can never time yesterday's binary under today's name.
set -u
if [ $# -lt 1 ]; then echo "usage: $0 NAME [\"extra hipcc flags\"]" >&2; exit 2; fi
root="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
arch="${GPUAR_ARCH:-gfx950}"
hipcc="${HIPCC:-/opt/rocm/bin/hipcc}"
out="$root/gpuar_amd/lib/exp/$1.so"
mkdir -p "$root/gpuar_amd/lib/exp" "$root/build"
rm -f "$out"
[ -f "$root/build/host_codec.o" ] || make -C "$root/gpuar_amd/csrc" "$root/build/host_codec.o" > /dev/null || exit 1
# a timing switch of the GPUAR_EXP_* family takes pieces out of a kernel: the build must say that it is an experiment
marker=""
case "${2:-}" in *GPUAR_EXP_*) marker="-DGPUAR_EXPERIMENT_BUILD=1" ;; esac
log="$(mktemp)"
( cd "$root/gpuar_amd/csrc" && "$hipcc" --offload-arch="$arch" -O3 -std=c++17 -fPIC -I"$root/include" -Wno-unused-function \
    -mllvm -phi-node-folding-threshold=64 -mllvm -two-entry-phi-node-folding-threshold=64 ${2:-} $marker -shared \
    -o "$out" "$root/build/host_codec.o" gpuar_kernels.hip ) > "$log" 2>&1
rc=$?
grep -E "error|warning: v" "$log"
rm -f "$log"
if [ $rc -ne 0 ]; then rm -f "$out"; echo "exp_build: $1 FAILED (rc $rc)" >&2; exit $rc; fi
echo "built $out"
