"""CPU check of the per-lane codec the gfx950 kernels run (gpuar_amd/csrc/lane_codec.h).

The header is compiled for the host (g++) into a small test-only library and
driven one packet at a time; outputs must equal the oracle's byte for byte on
every golden fixture.  This is how the left-count tree, the partial modelers,
the closed-form renormalisation, the bit-field-mask emission, the reciprocal
division and the division-free symbol search are validated where there is no
GPU; the GPU parity tests (tests/test_gpu_parity.py) then check the real kernels.
"""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from test_oracle_golden import REFV, case_input

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
u8p = C.POINTER(C.c_uint8)
u64p = C.POINTER(C.c_uint64)


@pytest.fixture(scope="session")
def emu():
    out_dir = os.path.join(HERE, "_build")
    os.makedirs(out_dir, exist_ok=True)
    so = os.path.join(out_dir, "liblane_emulation.so")
    srcs = [os.path.join(HERE, "lane_emulation.cpp"), os.path.join(ROOT, "gpuar_amd", "csrc", "lane_codec.h")]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-fconstexpr-ops-limit=100000000",
                               "-fconstexpr-loop-limit=1000000", "-Wno-unknown-pragmas",
                               "-I", os.path.join(ROOT, "include"), "-o", so, srcs[0]])
    lib = C.CDLL(so)
    lib.emu_encode_slots.restype = C.c_int
    lib.emu_encode_slots.argtypes = [u8p, C.c_size_t, u8p]
    lib.emu_encode_slots_e3.restype = C.c_int
    lib.emu_encode_slots_e3.argtypes = [u8p, C.c_size_t, u8p]
    lib.emu_encode_slots_phased.restype = C.c_int
    lib.emu_encode_slots_phased.argtypes = [u8p, C.c_size_t, u8p]
    lib.emu_encode_slots_split.restype = C.c_int
    lib.emu_encode_slots_split.argtypes = [u8p, C.c_size_t, u8p]
    lib.emu_encode_slots_device_rule.restype = C.c_int
    lib.emu_encode_slots_device_rule.argtypes = [u8p, C.c_size_t, u8p]
    lib.emu_device_rule_rare_events.restype = C.c_uint64
    lib.emu_device_rule_rare_events.argtypes = []
    lib.emu_decode_stream.restype = C.c_int
    lib.emu_decode_stream.argtypes = [u8p, u64p, C.c_size_t, u8p]
    lib.emu_check_recip.restype = C.c_uint64
    lib.emu_check_recip.argtypes = [C.c_uint32]
    lib.emu_check_renorm_count.restype = C.c_uint64
    lib.emu_check_renorm_count.argtypes = [C.c_uint32]
    return lib


@pytest.fixture(scope="session")
def emu_small_slots():
    """The same emulation built with 1024-byte slots, so ordinary data drives CoderLane into overflow."""
    out_dir = os.path.join(HERE, "_build")
    os.makedirs(out_dir, exist_ok=True)
    so = os.path.join(out_dir, "liblane_emulation_slot1024.so")
    srcs = [os.path.join(HERE, "lane_emulation.cpp"), os.path.join(ROOT, "gpuar_amd", "csrc", "lane_codec.h")]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-fconstexpr-ops-limit=100000000",
                               "-fconstexpr-loop-limit=1000000", "-Wno-unknown-pragmas", "-DGPUAR_SLOT_BYTES=1024u",
                               "-I", os.path.join(ROOT, "include"), "-o", so, srcs[0]])
    lib = C.CDLL(so)
    lib.emu_encode_slots.restype = C.c_int
    lib.emu_encode_slots.argtypes = [u8p, C.c_size_t, u8p]
    return lib


@pytest.fixture(scope="session")
def emu_d1_regs():
    """The emulation built with -DGPUAR_TOP_D1_REGS: the top modeler keeps depth 1 of the tree in a register (round 5's A/B
    build, 1.9 % slower on the GPU and therefore not the default -- profiles/r05_encoder_attribution.txt); as long as the flag
    is in the tree its output is pinned like the default's."""
    out_dir = os.path.join(HERE, "_build")
    os.makedirs(out_dir, exist_ok=True)
    so = os.path.join(out_dir, "liblane_emulation_d1regs.so")
    srcs = [os.path.join(HERE, "lane_emulation.cpp"), os.path.join(ROOT, "gpuar_amd", "csrc", "lane_codec.h")]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-fconstexpr-ops-limit=100000000",
                               "-fconstexpr-loop-limit=1000000", "-Wno-unknown-pragmas", "-DGPUAR_TOP_D1_REGS",
                               "-I", os.path.join(ROOT, "include"), "-o", so, srcs[0]])
    lib = C.CDLL(so)
    for name in ("emu_encode_slots", "emu_encode_slots_phased"):
        getattr(lib, name).restype = C.c_int
        getattr(lib, name).argtypes = [u8p, C.c_size_t, u8p]
    return lib


def emu_encode(lib, data: np.ndarray):
    npk = (data.size + 8191) // 8192
    slots = np.zeros(max(npk, 1) * 8704, dtype=np.uint8)
    src = np.ascontiguousarray(data)
    ov = lib.emu_encode_slots(src.ctypes.data_as(u8p), src.size, slots.ctypes.data_as(u8p))
    return slots, npk, ov


def slots_to_stream(slots, npk):
    parts, offs = [], [0]
    for p in range(npk):
        clen = int(slots[p * 8704]) | (int(slots[p * 8704 + 1]) << 8)
        parts.append(slots[p * 8704:p * 8704 + clen])
        offs.append(offs[-1] + clen)
    stream = np.concatenate(parts) if parts else np.zeros(0, dtype=np.uint8)
    return stream, np.asarray(offs, dtype=np.uint64)


def emu_decode(lib, stream, offs, npk):
    padded = np.concatenate([stream, np.zeros(16, dtype=np.uint8)])
    out = np.zeros(max(npk, 1) * 8192, dtype=np.uint8)
    bad = lib.emu_decode_stream(padded.ctypes.data_as(u8p), offs.ctypes.data_as(u64p), npk, out.ctypes.data_as(u8p))
    return out, bad


@pytest.mark.parametrize("c", REFV, ids=lambda c: c["name"])
def test_phased_modelers_give_the_same_slots(emu, c):
    """The encode kernel's roles run a phase apart: the low modeler adds onto the top modeler's parts and never sees
    the phase behind its own (PartialModeler::prime / step_last / `onto`).  Same slots as the straight emulation."""
    data = np.ascontiguousarray(case_input(c))
    want, npk, ov = emu_encode(emu, data)
    got = np.zeros_like(want)
    assert emu.emu_encode_slots_phased(data.ctypes.data_as(u8p), data.size, got.ctypes.data_as(u8p)) == ov
    assert np.array_equal(got, want)


@pytest.mark.parametrize("c", REFV, ids=lambda c: c["name"])
def test_latency_mode_roles_give_the_same_slots(emu, c):
    """encode_small_kernel's five roles: the tree dealt 3 + 3 + (0, 7, tail) to three modelers that add onto each
    other, the coder cut into IntervalLane | SinkLane joined by one word per symbol (agreed bits, e, u) and the last lower bound.
    Same slots as the straight emulation -- the adversarial packet that owes 2396 underflow bits included."""
    data = np.ascontiguousarray(case_input(c))
    want, npk, ov = emu_encode(emu, data)
    got = np.zeros_like(want)
    assert emu.emu_encode_slots_split(data.ctypes.data_as(u8p), data.size, got.ctypes.data_as(u8p)) == ov
    assert np.array_equal(got, want)


@pytest.mark.parametrize("c", REFV, ids=lambda c: c["name"])
def test_carry_form_coder_equals_the_owed_bits_form(emu, c):
    """CarryCoderLane (encode_kernel's coder since round 4: the lower bound as a 64-bit window of one long binary fraction,
    carries running through the bits not yet stored and -- rarely -- back into memory) against CoderLane, which mirrors
    the reference's 16-bit bounds and owed underflow bits (writeEncodedBits :321-367): the same slots byte for byte, the
    packet that owes 2396 bits (a carry through 75 stored dwords) included."""
    data = np.ascontiguousarray(case_input(c))
    want = np.zeros(((data.size + 8191) // 8192) * 8704, dtype=np.uint8)
    ov = emu.emu_encode_slots_e3(data.ctypes.data_as(u8p), data.size, want.ctypes.data_as(u8p))
    got, npk, ov2 = emu_encode(emu, data)
    assert ov == ov2 == 0 and np.array_equal(got, want)


def test_carry_form_coder_on_streams_that_keep_carrying(emu, port_oracle):
    """Inputs built to make carries travel: two symbols whose boundary sits at the interval's midpoint (long runs of
    owed bits resolved both ways), and few-symbol alphabets at every skew."""
    rng = np.random.default_rng(29)
    for trial in range(40):
        n = int(rng.integers(2000, 8193))
        if trial % 4 == 0:
            data = np.where(rng.random(n) < 0.5, 0x7F, 0x80).astype(np.uint8)
        elif trial % 4 == 1:
            data = rng.choice(np.array([0x7F, 0x80, 0x00, 0xFF], dtype=np.uint8), n, p=[0.45, 0.45, 0.05, 0.05])
        elif trial % 4 == 2:
            k = int(rng.integers(2, 6))
            data = rng.choice(rng.integers(0, 256, k).astype(np.uint8), n)
        else:
            data = np.repeat(rng.integers(0, 256, n // 64 + 1).astype(np.uint8), 64)[:n]
        slots, npk, ov = emu_encode(emu, np.ascontiguousarray(data))
        stream, _ = slots_to_stream(slots, npk)
        assert ov == 0 and np.array_equal(stream, port_oracle.encode_stream(data)), trial


def carry_heavy_inputs():
    """Packets whose dwords keep being carried into, and dwords of 32 ones that wait for their verdict (the coder's rare path):
    midpoint pairs, midpoint pairs with strangers, few-symbol alphabets, long runs (the test below counts how often the rare
    path really ran)."""
    rng = np.random.default_rng(31)
    for trial in range(120):
        n = int(rng.integers(3000, 8193))
        kind = trial % 5
        if kind == 0:
            data = np.where(rng.random(n) < 0.5, 0x7F, 0x80).astype(np.uint8)
        elif kind == 1:
            data = rng.choice(np.array([0x7F, 0x80, 0x00, 0xFF], dtype=np.uint8), n, p=[0.47, 0.47, 0.03, 0.03])
        elif kind == 2:
            data = np.full(n, 0x7F, dtype=np.uint8)
            data[rng.random(n) < float(rng.uniform(0.3, 0.7))] = 0x80
            data[::int(rng.integers(50, 400))] = int(rng.integers(0, 256))
        elif kind == 3:
            data = rng.choice(rng.integers(0, 256, int(rng.integers(2, 5))).astype(np.uint8), n)
        else:
            data = np.repeat(rng.integers(0, 256, n // 32 + 1).astype(np.uint8), 32)[:n]
        yield trial, np.ascontiguousarray(data)


def test_device_store_rule_equals_leave_on_streams_that_keep_carrying(emu, port_oracle):
    """The GPU's store region + rare path (hand-written text in CarryCoderLane::shift_and_store: store `cache + over`, find the
    rare lanes with one compare against `key`, rewind `at` / let the waiting dwords go, key = nff ? 0 : ~0; `held` kept 16
    too high) restated in C++ (tests/lane_emulation.cpp, DeviceRuleCoder) against take_top() + leave(), which is what every other CPU
    test pins against the oracle: the same slots on every golden case -- the packet that owes 2396 bits included -- and on
    120 packets built to keep carrying.  (the CPU emulation never executed the device's rule.)"""
    cases = [(c["name"], np.ascontiguousarray(case_input(c))) for c in REFV] + list(carry_heavy_inputs())
    for name, data in cases:
        npk = (data.size + 8191) // 8192
        got = np.zeros(max(npk, 1) * 8704, dtype=np.uint8)
        ov = emu.emu_encode_slots_device_rule(data.ctypes.data_as(u8p), data.size, got.ctypes.data_as(u8p))
        want, _, ov2 = emu_encode(emu, data)
        assert ov == ov2 == 0 and np.array_equal(got, want), name
        if not isinstance(name, str) and name % 10 == 0:
            stream, _ = slots_to_stream(got, npk)
            assert np.array_equal(stream, port_oracle.encode_stream(data)), name
    events = emu.emu_device_rule_rare_events()
    undecided, filled = events & 0xFFFFFFFF, events >> 32
    assert undecided >= 100 and filled >= 100, (undecided, filled)        # the rare path did run: dwords of ones waited and were let go


@pytest.mark.parametrize("c", REFV, ids=lambda c: c["name"])
def test_depth_one_in_a_register_gives_the_same_slots(emu, emu_d1_regs, c):
    """PartialModeler with kHead == 3 (depth 1's two nodes packed in one register, picked by a bit-field extract, counted by a
    shift-add) against the LDS-resident form, straight and in the kernel's phases of eight."""
    data = np.ascontiguousarray(case_input(c))
    want, npk, ov = emu_encode(emu, data)
    for fn in (emu_d1_regs.emu_encode_slots, emu_d1_regs.emu_encode_slots_phased):
        got = np.zeros(max(npk, 1) * 8704, dtype=np.uint8)
        assert fn(data.ctypes.data_as(u8p), data.size, got.ctypes.data_as(u8p)) == ov == 0
        assert np.array_equal(got, want)


def test_renormalisation_count_in_one_clz_equals_the_loop(emu):
    """The decoder's n = e + u from ONE count of leading zeros (lane_codec.h renorm_count) against the reference's
    renormalisation loop, for EVERY pair lo <= hi < 65536 (2^31 pairs, a few seconds of C)."""
    assert emu.emu_check_renorm_count(1) == 0


def test_reciprocal_table_is_exact(emu):
    # every total, every 97th multiple boundary on both sides, plus the extremes
    assert emu.emu_check_recip(97) == 0


@pytest.mark.parametrize("c", REFV, ids=lambda c: c["name"])
def test_lane_codec_matches_reference_fixture(emu, port_oracle, c):
    data = case_input(c)
    slots, npk, ov = emu_encode(emu, data)
    assert ov == 0
    stream, offs = slots_to_stream(slots, npk)
    assert stream.size == c["stream_len"]
    assert np.array_equal(stream, port_oracle.encode_stream(data))
    # the lane decoder reads the back-to-back stream, so packet starts are unaligned
    out, bad = emu_decode(emu, stream, offs, npk)
    assert bad == 0
    assert np.array_equal(out[:data.size], data)


def test_lane_codec_random_packets(emu, port_oracle):
    rng = np.random.default_rng(7)
    for trial in range(60):
        n = int(rng.integers(1, 3 * 8192))
        if trial % 3 == 0:
            data = rng.integers(0, 256, n, dtype=np.uint8)
        elif trial % 3 == 1:
            data = (rng.geometric(0.05, n) % 256).astype(np.uint8)
        else:
            data = rng.choice(np.array([0, 1, 254, 255], dtype=np.uint8), n)
        slots, npk, ov = emu_encode(emu, data)
        stream, offs = slots_to_stream(slots, npk)
        assert ov == 0 and np.array_equal(stream, port_oracle.encode_stream(data)), trial
        out, bad = emu_decode(emu, stream, offs, npk)
        assert bad == 0 and np.array_equal(out[:n], data), trial


def test_lane_decoder_survives_garbage(emu):
    """Malformed packets must neither crash nor write outside their 8192-byte output."""
    rng = np.random.default_rng(11)
    flagged = 0
    for trial in range(40):
        blob = rng.integers(0, 256, 9000, dtype=np.uint8)
        blob[0:2] = np.frombuffer(int(9000 - 16).to_bytes(2, "little"), dtype=np.uint8)
        blob[2:4] = np.frombuffer(int(rng.integers(0, 65536)).to_bytes(2, "little"), dtype=np.uint8)
        offs = np.asarray([0, 9000 - 16], dtype=np.uint64)
        out = np.zeros(8192 + 64, dtype=np.uint8)
        out[8192:] = 0xA5
        flagged += emu.emu_decode_stream(blob.ctypes.data_as(u8p), offs.ctypes.data_as(u64p), 1, out.ctypes.data_as(u8p))
        assert np.all(out[8192:] == 0xA5)
    assert flagged > 0          # impossible lengths and out-of-model code values are reported


def test_slot_overflow_is_flagged_and_contained(emu_small_slots, port_oracle):
    """GPUAR_STATUS_SLOT_OVERFLOW path (CoderLane::put clamp + finish): a packet that outgrows its slot is
    reported, its stores never leave the slot, its header says clen = slot size, and the bytes that did fit
    are the true prefix of the packet.  A packet that fits next to it is untouched by the neighbour's overflow."""
    slot = 1024
    rng = np.random.default_rng(3)
    big = rng.integers(0, 256, 8192, dtype=np.uint8)          # ~8250 bytes of output: overflows a 1024-byte slot
    small = np.zeros(8192, dtype=np.uint8)                    # 210 bytes: fits
    data = np.concatenate([big, small, big[:5000]])
    guard = 64
    buf = np.full(3 * slot + guard, 0xA5, dtype=np.uint8)
    ov = emu_small_slots.emu_encode_slots(data.ctypes.data_as(u8p), data.size, buf.ctypes.data_as(u8p))
    assert ov == 1
    assert (buf[3 * slot:] == 0xA5).all()                      # nothing beyond the last slot
    want0 = port_oracle.encode_packet(big.tobytes())
    want1 = port_oracle.encode_packet(small.tobytes())
    # packet 0 overflowed: clen clamped to the slot size, ulen kept, the body up to the last dword is the true prefix
    assert int(buf[0]) | (int(buf[1]) << 8) == slot and int(buf[2]) | (int(buf[3]) << 8) == 8192
    assert buf[4:slot - 4].tobytes() == want0[4:slot - 4]
    # packet 1 fits and is exact, although its neighbours overflowed
    assert buf[slot:slot + len(want1)].tobytes() == want1
    assert int(buf[2 * slot]) | (int(buf[2 * slot + 1]) << 8) == slot
