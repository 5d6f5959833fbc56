#!/usr/bin/env bash
# oracle/build_ref.sh -- builds oracle/_ref/libgpuar_ref.so from the REFERENCE
# sources where they lie under /root/reference (never copied into this repo).
#
# What is compiled: /root/reference/src/gpuar_kernel.cu as *host* C++ with g++.
# Every codec function in it is `__host__ __device__` plain C++.  Two line
# ranges cannot be built in this image and are left out of the translation
# unit by line number (nothing is rewritten, no stand-in header is provided):
#   453-460  initConstantRange()  -- calls cudaMemcpyToSymbol (no CUDA runtime)
#   894-954  the two __global__ kernels and their <<<>>> launchers
# The CUDA headers the file includes (<cuda_runtime.h> via src/gpu.h) are the
# real ones that ship in this image with triton's NVIDIA backend; under g++
# they turn __host__/__device__ into ignored attributes.  `-include math.h`
# supplies ceil(), which nvcc would have provided implicitly.
#
# The reference's CLI (main.cpp / cpu_compressor.cpp / compressor.cpp) is NOT
# built: its constructors call cudaMallocHost, and there is no libcudart here;
# writing a stand-in for it is off the table.  oracle/ref_driver.cpp (our code)
# drives the reference's arCompress/arDecompress packet by packet instead.
set -euo pipefail
here="$(cd "$(dirname "$0")" && pwd)"
ref="${GPUAR_REFERENCE:-/root/reference}"
out="$here/_ref"
if [ ! -f "$ref/src/gpuar_kernel.cu" ]; then
    echo "build_ref: $ref not present (GPU box?) -- keeping prebuilt $out" >&2
    exit 0
fi
cudainc="$(python3 - <<'PY'
import importlib.util, os
spec = importlib.util.find_spec("triton")
base = os.path.dirname(spec.origin) if spec else ""
print(os.path.join(base, "backends", "nvidia", "include"))
PY
)"
if [ ! -f "$cudainc/cuda_runtime.h" ]; then
    echo "build_ref: no cuda_runtime.h in this image -- reference unbuildable" >&2
    exit 0
fi
mkdir -p "$out"
flags="-std=c++11 -O3 -w -fPIC -include math.h -I$cudainc -I$ref/common -I$ref/src"
sed -e '453,460d' -e '894,954d' "$ref/src/gpuar_kernel.cu" \
    | g++ -x c++ $flags -c - -o "$out/ref_codec.o"
g++ $flags -c "$here/ref_driver.cpp" -o "$out/ref_driver.o"
# -Bsymbolic: the reference defines extern "C" read()/write() helpers
# (src/gpuar_kernel.cu:18-74); bind them inside the library so libc's do not win.
g++ -shared -Wl,-Bsymbolic -o "$out/libgpuar_ref.so" "$out/ref_codec.o" "$out/ref_driver.o" -lpthread
rm -f "$out/ref_codec.o" "$out/ref_driver.o"
echo "build_ref: built $out/libgpuar_ref.so"
