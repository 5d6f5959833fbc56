"""A COMPILED consumer of the drop-in boundary (build container only; skipped where /root/reference is absent).

INTEGRATION.md says switching the reference's host code to this library is "a link change".  This test is that
change in miniature: a C++ translation unit that includes the REFERENCE's own header
(/root/reference/src/gpuar.h:59-86 -- C names, C++ reference parameters), calls
initializeAdaptiveProbabilityRangeList / arCompress / arDecompress exactly as the reference's callers do
(src/cpu_compressor.cpp:59-60,159-160; src/main.cpp:27-43), and links -lgpuar_host instead of the reference's
kernel object.  The TU below is this repository's code; nothing of the reference is copied -- its header is
included where it lies, with the same real CUDA include path oracle/build_ref.sh uses.  No GPU.
"""
import importlib.util
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("GPUAR_REFERENCE", "/root/reference")

CONSUMER = r"""
#include "gpuar.h"            // the reference's header: AdaptiveProbabilityRange, probability_t, the extern "C" prototypes
#include <cstdio>
#include <cstring>
#include <vector>

static int fail(const char *what) { std::printf("FAIL %s\n", what); return 1; }

int main() {
    // "hello" -> 09 00 05 00 68 65 08 23 e3 (SURVEY.md section 8(b)), model re-initialised per packet as every caller does
    const uint8_t hello[5] = {'h', 'e', 'l', 'l', 'o'};
    const uint8_t want[9] = {0x09, 0x00, 0x05, 0x00, 0x68, 0x65, 0x08, 0x23, 0xe3};
    AdaptiveProbabilityRange range;
    probability_t total = 0;
    uint8_t packet[COMPRESSED_PACKET_SIZE];
    initializeAdaptiveProbabilityRangeList(&range, total);          // C++ reference argument, as src/cpu_compressor.cpp:59
    if (total != 256) return fail("initial total");
    const uint16_t clen = arCompress(hello, 5, packet, range, total);   // src/cpu_compressor.cpp:60
    if (clen != 9 || std::memcmp(packet, want, 9) != 0) return fail("hello vector");
    if (total != 261) return fail("total after five symbols");
    if (getCompressedSize == nullptr) return fail("unreachable");   // (a reference prototype this TU does not call: still declared)

    uint8_t back[UNCOMPRESSED_PACKET_SIZE];
    initializeAdaptiveProbabilityRangeList(&range, total);
    const uint16_t ulen = arDecompress(packet, clen, back, range, total);   // src/cpu_compressor.cpp:159-160
    if (ulen != 5 || std::memcmp(back, hello, 5) != 0) return fail("hello round trip");

    // a full 8192-byte packet of a skewed source, and the single byte d7 -> 06 00 01 00 d7 40
    std::vector<uint8_t> in(UNCOMPRESSED_PACKET_SIZE);
    uint32_t s = 12345;
    for (auto &b : in) { s = s * 1664525u + 1013904223u; b = static_cast<uint8_t>((s >> 24) & (s >> 16) & 0xFF); }
    initializeAdaptiveProbabilityRangeList(&range, total);
    const uint16_t c2 = arCompress(in.data(), UNCOMPRESSED_PACKET_SIZE, packet, range, total);
    if (c2 < PACKET_HEADER_LENGTH || c2 > COMPRESSED_PACKET_SIZE) return fail("packet length");
    if ((packet[0] | packet[1] << 8) != c2 || (packet[2] | packet[3] << 8) != UNCOMPRESSED_PACKET_SIZE) return fail("packet header");
    initializeAdaptiveProbabilityRangeList(&range, total);
    if (arDecompress(packet, c2, back, range, total) != UNCOMPRESSED_PACKET_SIZE || std::memcmp(back, in.data(), in.size()) != 0)
        return fail("8192-byte round trip");
    const uint8_t d7 = 0xd7, want1[6] = {0x06, 0x00, 0x01, 0x00, 0xd7, 0x40};
    initializeAdaptiveProbabilityRangeList(&range, total);
    if (arCompress(&d7, 1, packet, range, total) != 6 || std::memcmp(packet, want1, 6) != 0) return fail("d7 vector");
    std::printf("OK hello clen=%u skewed clen=%u\n", clen, c2);
    return 0;
}
"""


def cuda_include_dir():
    spec = importlib.util.find_spec("triton")
    if not spec:
        return None
    d = os.path.join(os.path.dirname(spec.origin), "backends", "nvidia", "include")
    return d if os.path.exists(os.path.join(d, "cuda_runtime.h")) else None


@pytest.mark.parametrize("library", ["gpuar_host", "gpuar_hip"])
def test_reference_header_consumer_links_and_runs(tmp_path, library):
    if not os.path.exists(os.path.join(REF, "src", "gpuar.h")):
        pytest.skip("/root/reference absent (GPU box): the consumer needs the reference's own header")
    cudainc = cuda_include_dir()
    if not cudainc:
        pytest.skip("no cuda_runtime.h in this image: the reference's header cannot be parsed")
    libdir = os.path.join(ROOT, "gpuar_amd", "lib")
    if not os.path.exists(os.path.join(libdir, f"lib{library}.so")):
        import __graft_entry__ as g
        g.build()
    src = tmp_path / "consumer.cpp"
    src.write_text(CONSUMER)
    exe = tmp_path / "consumer"
    # the same flags oracle/build_ref.sh compiles the reference's own sources with; -lgpuar_* is the "link change"
    cmd = ["g++", "-std=c++11", "-O1", "-w", "-include", "math.h", f"-I{cudainc}", f"-I{REF}/common", f"-I{REF}/src",
           str(src), "-o", str(exe), f"-L{libdir}", f"-l{library}", f"-Wl,-rpath,{libdir}"]
    if library == "gpuar_hip":          # the GPU library needs the HIP runtime at load time (no device is touched by these calls)
        cmd += ["-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib"]
    built = subprocess.run(cmd, capture_output=True, text=True)
    assert built.returncode == 0, built.stderr[-3000:]
    run = subprocess.run([str(exe)], capture_output=True, text=True)
    assert run.returncode == 0 and run.stdout.startswith("OK "), (run.returncode, run.stdout, run.stderr)
