#!/usr/bin/env bash
# tools/exp_asm.sh NAME "-DEXP_A=1"  -> build/asm/NAME.s (gfx950 assembly of the kernels with those flags) + instruction counts per kernel
set -u
root="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
mkdir -p "$root/build/asm"
( cd "$root/gpuar_amd/csrc" && "${HIPCC:-/opt/rocm/bin/hipcc}" --offload-arch="${GPUAR_ARCH:-gfx950}" -O3 -std=c++17 -fPIC -I"$root/include" -Wno-unused-function \
    -mllvm -phi-node-folding-threshold=64 -mllvm -two-entry-phi-node-folding-threshold=64 ${2:-} --offload-device-only -S \
    -o "$root/build/asm/$1.s" gpuar_kernels.hip ) || exit 1
echo "$root/build/asm/$1.s"
