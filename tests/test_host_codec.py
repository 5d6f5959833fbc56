"""The host-callable packet codec (include/gpuar_host.h, libgpuar_host.so and
the same three symbols in libgpuar_hip.so) against the reference's own
arCompress / arDecompress / initializeAdaptiveProbabilityRangeList
(/root/reference/src/gpuar.h:73,75,76).

Anchors: the tiny vectors of SURVEY.md section 8(b), tests/golden/ref_vectors.json,
tests/golden/model_vectors.json (caller-owned model carried across calls, from
oracle/_ref), the C restatement in oracle/, and -- where the build container has
it -- oracle/_ref live on seeded inputs, including corrupted packets.  No GPU.
"""
import ctypes as C
import hashlib
import json
import os

import numpy as np
import pytest

from gpuar_amd import host as HC
from gpuar_amd import synth

GOLD = os.path.join(os.path.dirname(__file__), "golden")
with open(os.path.join(GOLD, "model_vectors.json")) as f:
    MODELV = json.load(f)
with open(os.path.join(GOLD, "ref_vectors.json")) as f:
    REFV = json.load(f)["cases"]


@pytest.fixture(scope="module", autouse=True)
def built():
    if not os.path.exists(HC.LIB_PATH):
        import __graft_entry__ as g
        g.build()


def ranges_of(hexstr):
    return np.frombuffer(bytes.fromhex(hexstr), dtype="<u2").astype(np.uint16)


def test_initial_model_is_the_references():
    m = HC.Model()
    assert m.total == MODELV["initial_total"] == 256
    assert np.array_equal(m.ranges, ranges_of(MODELV["initial_ranges_hex"]))
    assert m.ranges[0] == 0 and m.ranges[256] == 256 and m.ranges[1] == 1 and m.ranges[192] == 64


def test_survey_tiny_vectors():
    assert HC.encode_packet(b"hello").hex() == "0900050068650823e3"
    assert HC.encode_packet(b"\xd7").hex() == "06000100d740"
    zeros = HC.encode_packet(bytes(8192))
    assert len(zeros) == 210 and zeros[:4].hex() == "d2000020" and zeros[-1] == 0x04
    assert HC.decode_packet(bytes.fromhex("0900050068650823e3")) == b"hello"
    assert HC.decode_packet(zeros) == bytes(8192)


@pytest.mark.parametrize("case", MODELV["cases"], ids=lambda c: f"{c['kind']}_s{c['seed']}")
def test_model_carried_across_calls(case):
    enc, dec = HC.Model(), HC.Model()
    for seg in case["segments"]:
        data = synth.generate(case["kind"], case["seed"], seg["n"], offset=seg["offset"]).tobytes()
        pkt = HC.encode_packet(data, enc)
        assert len(pkt) == seg["clen"]
        assert hashlib.md5(pkt).hexdigest() == seg["packet_md5"]
        if "packet_hex" in seg:
            assert pkt.hex() == seg["packet_hex"]
        assert enc.total == seg["total_after"]
        assert np.array_equal(enc.ranges, ranges_of(seg["ranges_after_hex"]))
        assert HC.decode_packet(pkt, dec) == data
        assert dec.total == enc.total and np.array_equal(dec.ranges, enc.ranges)


def test_single_packet_golden_cases():
    for c in REFV:
        if c["kind"] not in synth.KINDS or c["n"] > 8192:
            continue
        data = synth.generate(c["kind"], c["seed"], c["n"]).tobytes()
        pkt = HC.encode_packet(data)
        assert len(pkt) == c["stream_len"] and hashlib.md5(pkt).hexdigest() == c["stream_md5"], c["name"]
        assert HC.decode_packet(pkt) == data


def test_fixture_streams_packet_by_packet(port_oracle):
    data = np.fromfile(os.path.join(GOLD, "adversarial_midpoint.in.bin"), dtype=np.uint8).tobytes()
    want = open(os.path.join(GOLD, "adversarial_midpoint.stream.bin"), "rb").read()
    assert HC.encode_packet(data) == want            # the long pending-run packet
    assert HC.decode_packet(want) == data
    stream = open(os.path.join(GOLD, "uniform_s1_n65539_keep.stream.bin"), "rb").read()
    src = synth.uniform(1, 65539).tobytes()
    off = 0
    for p in range(9):
        clen = int.from_bytes(stream[off:off + 2], "little")
        chunk = src[p * 8192:(p + 1) * 8192]
        assert HC.encode_packet(chunk) == stream[off:off + clen]
        assert HC.decode_packet(stream[off:off + clen]) == chunk
        off += clen
    assert off == len(stream)


def test_matches_c_restatement_on_ragged_sizes(port_oracle):
    for kind in synth.KINDS:
        for seed, n in [(21, 0), (22, 1), (23, 2), (24, 15), (25, 16), (26, 17), (27, 255), (28, 4095), (29, 8191), (30, 8192)]:
            data = synth.generate(kind, seed, n).tobytes()
            pkt = HC.encode_packet(data)
            assert pkt == port_oracle.encode_packet(data), (kind, seed, n)
            assert HC.decode_packet(pkt) == data
    for b in (0, 0x41, 0xFF):
        data = bytes([b]) * 8192
        assert HC.encode_packet(data) == port_oracle.encode_packet(data)


def test_reads_exactly_size_bytes_and_writes_exactly_clen():
    # input sits at the very end of a buffer, output is guarded by sentinels
    lib = HC.load()
    data = synth.text(9, 1000)
    out = np.full(8704 + 64, 0xA5, dtype=np.uint8)
    m = HC.Model()
    n = lib.arCompress(data.ctypes.data_as(C.POINTER(C.c_uint8)), data.size, out.ctypes.data_as(C.POINTER(C.c_uint8)),
                       m.ranges.ctypes.data_as(C.POINTER(C.c_uint16)), C.byref(m._total))
    assert 4 < n < 1000 and (out[n:] == 0xA5).all()
    assert int(out[0]) | (int(out[1]) << 8) == n and int(out[2]) | (int(out[3]) << 8) == 1000


def test_same_symbols_in_both_libraries():
    from gpuar_amd import hip as H
    if not os.path.exists(H.LIB_PATH):
        pytest.skip("libgpuar_hip.so not built")
    big = C.CDLL(H.LIB_PATH)
    for name in HC.EXPORTS:
        assert hasattr(big, name), name
    u16p, u8p = C.POINTER(C.c_uint16), C.POINTER(C.c_uint8)
    big.arCompress.restype = C.c_uint16
    big.arCompress.argtypes = [u8p, C.c_uint16, u8p, u16p, u16p]
    big.initializeAdaptiveProbabilityRangeList.argtypes = [u16p, u16p]
    ranges = np.zeros(257, dtype=np.uint16)
    total = C.c_uint16(0)
    big.initializeAdaptiveProbabilityRangeList(ranges.ctypes.data_as(u16p), C.byref(total))
    src = np.frombuffer(b"hello", dtype=np.uint8).copy()
    out = np.zeros(64, dtype=np.uint8)
    n = big.arCompress(src.ctypes.data_as(u8p), 5, out.ctypes.data_as(u8p), ranges.ctypes.data_as(u16p), C.byref(total))
    assert out[:n].tobytes().hex() == "0900050068650823e3" and total.value == 261


def test_live_against_reference_with_carried_models(ref_oracle):
    rng = np.random.default_rng(5)
    for trial in range(40):
        kind = synth.KINDS[trial % 3]
        sizes = [int(x) for x in rng.integers(0, 4000, size=int(rng.integers(2, 5)))]
        r, t = ref_oracle.model_init()
        r2, t2 = ref_oracle.model_init()
        enc, dec = HC.Model(), HC.Model()
        off = 0
        for n in sizes:
            data = synth.generate(kind, trial + 1, n, offset=off).tobytes()
            off += n
            want, r, t = ref_oracle.encode_packet_model(data, r, t)
            assert HC.encode_packet(data, enc) == want
            assert enc.total == t and np.array_equal(enc.ranges, r)
            back, r2, t2 = ref_oracle.decode_packet_model(want, r2, t2)
            assert HC.decode_packet(want, dec) == back == data
            assert dec.total == t2 and np.array_equal(dec.ranges, r2)


def test_live_against_reference_on_corrupted_packets(ref_oracle):
    # a damaged packet must decode to the same bytes (and stop at the same place) as the reference
    rng = np.random.default_rng(9)
    for trial in range(200):
        kind = synth.KINDS[trial % 3]
        data = synth.generate(kind, trial + 3, int(rng.integers(1, 3000))).tobytes()
        pkt = bytearray(ref_oracle.encode_packet(data))
        for _ in range(int(rng.integers(1, 4))):
            pkt[int(rng.integers(4, len(pkt)))] ^= 1 << int(rng.integers(0, 8))
        assert HC.decode_packet(bytes(pkt)) == ref_oracle.decode_packet(bytes(pkt)), trial
