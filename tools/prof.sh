#!/usr/bin/env bash
# Runs on the GPU box (via gpurun): kernel trace + stats, then PMC passes, for tools/prof_run.py.
# Usage: tools/prof.sh TAG [--quick] [prof_run args...]       (--quick: the traffic and instruction-count passes only)
# Counters are collected in their own runs with --kernel-trace only (never together with a sys/hip trace).
set -u
tag="$1"; shift
quick=0
if [ "${1:-}" = "--quick" ]; then quick=1; shift; fi
repo="$(pwd)"
out="$repo/gpurun_out/prof_$tag"
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python3 "$repo/tools/prof_run.py" "$@" > "$out/trace.log" 2>&1
sets=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS"
      "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
      "SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_WAVES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM"
      "GRBM_GUI_ACTIVE FETCH_SIZE" "WRITE_SIZE GRBM_COUNT")
if [ $quick -eq 1 ]; then sets=("SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS" "GRBM_GUI_ACTIVE FETCH_SIZE" "WRITE_SIZE GRBM_COUNT"); fi
for set in "${sets[@]}"; do
    name="$(echo "$set" | tr ' ' '_' | cut -c1-40)"
    rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$out/pmc_$name" -- python3 "$repo/tools/prof_run.py" "$@" > "$out/pmc_$name.log" 2>&1
    echo "pass $name done" >> "$out/progress.log"
done
cd "$repo"
python3 tools/prof_summary.py "$out" > "$out/summary.txt" 2>&1
cat "$out/summary.txt"
