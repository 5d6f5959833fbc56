#!/usr/bin/env bash
# Cuts decode_slots_kernel out of build/asm/gpuar_kernels.s (after `make -C gpuar_amd/csrc asm`) into build/asm/dec.s.
root="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
src="$root/build/asm/gpuar_kernels.s"
[ -f "$src" ] || { echo "run: make -C gpuar_amd/csrc asm" >&2; exit 2; }
awk '/^_ZN5gpuar19decode_slots_kernel[^:]*:/,/s_endpgm/' "$src" > "$root/build/asm/dec.s"
wc -l "$root/build/asm/dec.s"
