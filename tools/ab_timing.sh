#!/usr/bin/env bash
# A/B timing of experiment builds: tools/ab_timing.sh [--gib G] lib1.so lib2.so ...   (run on the GPU box)
gib=2
if [ "$1" = "--gib" ]; then gib="$2"; shift 2; fi
for lib in "$@"; do
    printf "%-40s " "$(basename "$lib")"
    python3 tools/kind_timing.py --gib "$gib" --kinds uniform --lib "$lib" 2>&1 | grep uniform | cut -c1-110
done
