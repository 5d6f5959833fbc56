#!/usr/bin/env bash
# tools/exp_build.sh NAME "-DEXP_A=1 -DEXP_B=2"  -> gpuar_amd/lib/exp/NAME.so   (experiment builds for tools/ab_timing.sh)
mkdir -p /root/repo/gpuar_amd/lib/exp
cd /root/repo/gpuar_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I/root/repo/include -Wno-unused-function \
  -mllvm -phi-node-folding-threshold=64 -mllvm -two-entry-phi-node-folding-threshold=64 $2 -shared \
  -o /root/repo/gpuar_amd/lib/exp/$1.so /root/repo/build/host_codec.o gpuar_kernels.hip 2>&1 | grep -E "error|warning: v" 
exit 0
