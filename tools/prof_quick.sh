#!/usr/bin/env bash
# Quick issue-side profile of the two codec kernels (kernel trace + two PMC passes) on the GPU box.
# Usage: tools/prof_quick.sh TAG [prof_run args...]      -> gpurun_out/prof_TAG/summary.txt
set -u
tag="$1"; shift
repo="$(pwd)"
out="$repo/gpurun_out/prof_$tag"
rm -rf "$out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python3 "$repo/tools/prof_run.py" "$@" > "$out/trace.log" 2>&1
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_WAVES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM" \
           "SQ_INST_LEVEL_SMEM SQ_INSTS_FLAT SQ_ACTIVE_INST_FLAT SQ_IFETCH SQ_IFETCH_LEVEL SQ_CYCLES SQ_BUSY_CU_CYCLES SQ_ITEMS"; do
    name="$(echo "$set" | tr ' ' '_' | cut -c1-40)"
    rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$out/pmc_$name" -- python3 "$repo/tools/prof_run.py" "$@" > "$out/pmc_$name.log" 2>&1
done
cd "$repo"
python3 tools/prof_summary.py "$out" > "$out/summary.txt" 2>&1
grep -A40 "decode_slots_kernel$" "$out/summary.txt" | head -60
