"""The per-lane codec (gpuar_amd/csrc/lane_codec.h) under AddressSanitizer + UBSan on the CPU.

GPU sanitizers are not available on the pool, so the same source the kernels run per lane is
exercised here with the host build: every golden fixture through encode + decode, plus malformed
packets, with shifts, indices and buffer bounds checked by the sanitizers."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)

DRIVER = r'''
import ctypes as C, numpy as np, sys
sys.path.insert(0, %(tests)r); sys.path.insert(0, %(root)r)
from test_oracle_golden import REFV, case_input
from oracle import oracle as O
lib = C.CDLL(%(so)r)
u8p = C.POINTER(C.c_uint8); u64p = C.POINTER(C.c_uint64)
lib.emu_encode_slots.argtypes = [u8p, C.c_size_t, u8p]
lib.emu_decode_stream.argtypes = [u8p, u64p, C.c_size_t, u8p]
P = O.PortOracle()
for c in REFV:
    data = case_input(c)
    npk = (data.size + 8191) // 8192
    slots = np.zeros(max(npk, 1) * 8704, dtype=np.uint8)
    lib.emu_encode_slots(data.ctypes.data_as(u8p), data.size, slots.ctypes.data_as(u8p))
    parts, offs = [], [0]
    for p in range(npk):
        cl = int(slots[p * 8704]) | (int(slots[p * 8704 + 1]) << 8)
        parts.append(slots[p * 8704:p * 8704 + cl]); offs.append(offs[-1] + cl)
    stream = np.concatenate(parts); offs = np.asarray(offs, dtype=np.uint64)
    assert np.array_equal(stream, P.encode_stream(data)), c["name"]
    padded = np.concatenate([stream, np.zeros(16, dtype=np.uint8)]); out = np.zeros(npk * 8192, dtype=np.uint8)
    lib.emu_decode_stream(padded.ctypes.data_as(u8p), offs.ctypes.data_as(u64p), npk, out.ctypes.data_as(u8p))
    assert np.array_equal(out[:data.size], data), c["name"]
rng = np.random.default_rng(3)
for t in range(30):
    blob = rng.integers(0, 256, 9000, dtype=np.uint8)
    blob[0:2] = np.frombuffer((8984).to_bytes(2, "little"), dtype=np.uint8)
    offs = np.asarray([0, 8984], dtype=np.uint64); out = np.zeros(8192, dtype=np.uint8)
    lib.emu_decode_stream(blob.ctypes.data_as(u8p), offs.ctypes.data_as(u64p), 1, out.ctypes.data_as(u8p))
print("SANITIZED-OK")
'''


def test_lane_codec_is_clean_under_asan_ubsan(tmp_path):
    libasan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(libasan) or not os.path.exists(libasan):
        pytest.skip("libasan not available")
    so = str(tmp_path / "liblane_emulation_asan.so")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                           "-shared", "-fPIC", "-fconstexpr-ops-limit=100000000", "-fconstexpr-loop-limit=1000000",
                           "-Wno-unknown-pragmas", "-I", os.path.join(ROOT, "include"), "-o", so,
                           os.path.join(HERE, "lane_emulation.cpp")])
    env = dict(os.environ, LD_PRELOAD=libasan, ASAN_OPTIONS="detect_leaks=0")
    r = subprocess.run([sys.executable, "-c", DRIVER % {"tests": HERE, "root": ROOT, "so": so}],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "SANITIZED-OK" in r.stdout, r.stderr[-3000:]


HOST_DRIVER = r'''
import ctypes as C, numpy as np, sys
sys.path.insert(0, %(tests)r); sys.path.insert(0, %(root)r)
from gpuar_amd import synth
from oracle import oracle as O
lib = C.CDLL(%(so)r)
u8p = C.POINTER(C.c_uint8); u16p = C.POINTER(C.c_uint16)
lib.arCompress.restype = C.c_uint16; lib.arCompress.argtypes = [u8p, C.c_uint16, u8p, u16p, u16p]
lib.arDecompress.restype = C.c_uint16; lib.arDecompress.argtypes = [u8p, C.c_uint16, u8p, u16p, u16p]
lib.initializeAdaptiveProbabilityRangeList.argtypes = [u16p, u16p]
P = O.PortOracle()
def fresh():
    r = np.zeros(257, dtype=np.uint16); t = C.c_uint16(0)
    lib.initializeAdaptiveProbabilityRangeList(r.ctypes.data_as(u16p), C.byref(t)); return r, t
rng = np.random.default_rng(11)
for kind in synth.KINDS:
    for seed, n in [(1, 0), (2, 1), (3, 17), (4, 4097), (5, 8192)]:
        data = synth.generate(kind, seed, n)
        want = P.encode_packet(data.tobytes())
        src = data.copy() if n else np.zeros(1, dtype=np.uint8)          # exactly n readable bytes
        out = np.zeros(len(want), dtype=np.uint8)                        # exactly clen writable bytes
        r, t = fresh()
        got = lib.arCompress(src.ctypes.data_as(u8p), n, out.ctypes.data_as(u8p), r.ctypes.data_as(u16p), C.byref(t))
        assert got == len(want) and out.tobytes() == want, (kind, seed, n)
        pkt = np.frombuffer(want, dtype=np.uint8).copy()                 # exactly clen readable bytes
        back = np.zeros(max(n, 1), dtype=np.uint8)                       # exactly ulen writable bytes
        r, t = fresh()
        m = lib.arDecompress(pkt.ctypes.data_as(u8p), pkt.size, back.ctypes.data_as(u8p), r.ctypes.data_as(u16p), C.byref(t))
        assert m == n and back[:n].tobytes() == data.tobytes()
        for _ in range(20):                                              # damaged body: must stay inside both buffers
            bad = pkt.copy()
            if bad.size > 4:
                bad[int(rng.integers(4, bad.size))] ^= 1 << int(rng.integers(0, 8))
            r, t = fresh()
            lib.arDecompress(bad.ctypes.data_as(u8p), bad.size, back.ctypes.data_as(u8p), r.ctypes.data_as(u16p), C.byref(t))
print("SANITIZED-OK")
'''


def test_host_codec_is_clean_under_asan_ubsan(tmp_path):
    libasan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(libasan) or not os.path.exists(libasan):
        pytest.skip("libasan not available")
    so = str(tmp_path / "libgpuar_host_asan.so")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                           "-shared", "-fPIC", "-I", os.path.join(ROOT, "include"), "-o", so,
                           os.path.join(ROOT, "gpuar_amd", "csrc", "host", "host_codec.cpp")])
    env = dict(os.environ, LD_PRELOAD=libasan, ASAN_OPTIONS="detect_leaks=0")
    r = subprocess.run([sys.executable, "-c", HOST_DRIVER % {"tests": HERE, "root": ROOT, "so": so}],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "SANITIZED-OK" in r.stdout, r.stderr[-3000:]
