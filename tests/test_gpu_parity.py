"""GPU parity: the HIP kernels, called through the C ABI, against the oracle.

Bit-exact is the bar (integer/byte work): packet slots, compacted packet
stream, per-packet lengths, decoded bytes.  Run on the GPU box with
`python -m pytest tests -m gpu`.  Nothing here reads /root/reference.
"""
import ctypes as C
import hashlib
import json
import os

import numpy as np
import pytest

from gpuar_amd import synth
from test_oracle_golden import REFV, SURVEY, case_input, md5

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def H():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from gpuar_amd import hip
    hip.load()          # raises if the HIP library is missing: no fallback
    return hip


@pytest.fixture(scope="module")
def oracle():
    """The checker of every parity test below -- and it must be the one this box is expected to have: the reference's own codec
    (oracle/_ref) wherever the committed golden vectors pin its binary.  A box that lacks it fails HERE, red, instead of
    quietly comparing the kernels with our own port."""
    from oracle import oracle as O
    codec = O.require_best()
    assert codec.kind == O.expected_kind()
    return codec


def test_the_checker_on_this_box_is_the_reference_codec(oracle):
    from oracle import oracle as O
    if os.environ.get("GPUAR_ALLOW_PORT_CHECKER") == "1":
        pytest.skip("GPUAR_ALLOW_PORT_CHECKER=1: the port was accepted knowingly")
    assert O.pinned_checker_sha256(), "tests/golden/ref_vectors.json lost its checker pin"
    assert oracle.kind == "reference" and O.file_sha256(O.REF_LIB_PATH) == O.pinned_checker_sha256()


@pytest.fixture(params=["latency", "throughput"])
def encode_mode(request):
    """gpuar_hip_encode sends small inputs to the five-role latency-mode kernel and large ones to the three-role
    throughput kernel (GPUAR_ENCODE_MODE pins the choice): the small parity cases run through BOTH kernels."""
    old = os.environ.get("GPUAR_ENCODE_MODE")
    os.environ["GPUAR_ENCODE_MODE"] = request.param
    yield request.param
    if old is None:
        del os.environ["GPUAR_ENCODE_MODE"]
    else:
        os.environ["GPUAR_ENCODE_MODE"] = old


def oracle_slots(codec, data: np.ndarray):
    """The packet slots the oracle says `data` codes to: (slots[npk, 8704] zero-padded, clen[npk], total bytes).
    Built from the best oracle on the box (the reference's own codec when oracle/_ref is there)."""
    stream = codec.encode_stream(data)
    npk = (data.size + 8191) // 8192
    slots = np.zeros((max(npk, 1), 8704), dtype=np.uint8)
    lens = np.zeros(npk, dtype=np.int64)
    off = 0
    for p in range(npk):
        c = int(stream[off]) | (int(stream[off + 1]) << 8)
        slots[p, :c] = stream[off:off + c]
        lens[p] = c
        off += c
    assert off == stream.size
    return slots, lens, int(stream.size)


def gpu_stream(H, data: np.ndarray):
    """encode + compact on the GPU; returns (stream bytes, offsets, slots tensor, npk)."""
    npk = H.packet_count(data.size)
    if data.size == 0:
        return np.zeros(0, dtype=np.uint8), np.zeros(1, dtype=np.int64), None, 0
    d_in = torch.from_numpy(np.ascontiguousarray(data)).cuda()
    d_slots = H.encode(d_in)
    d_stream, d_off = H.compact(d_slots, npk)
    torch.cuda.synchronize()
    offs = d_off.cpu().numpy()
    return d_stream[:int(offs[-1])].cpu().numpy(), offs, d_slots, npk


@pytest.mark.parametrize("c", REFV, ids=lambda c: c["name"])
def test_encode_matches_reference_fixture(H, c, encode_mode):
    data = case_input(c)
    stream, offs, d_slots, npk = gpu_stream(H, data)
    assert H.status() == 0
    assert stream.size == c["stream_len"]
    assert md5(stream.tobytes()) == c["stream_md5"]
    assert list(np.diff(offs)) == c["clens"]
    keep = os.path.join(os.path.dirname(__file__), "golden", c["name"] + ".stream.bin")
    if os.path.exists(keep):
        assert stream.tobytes() == open(keep, "rb").read()
    back = H.decode(d_slots, npk)[:data.size].cpu().numpy()
    assert np.array_equal(back, data)
    assert H.status() == 0


@pytest.mark.parametrize("s", SURVEY["streams"], ids=lambda s: f"{s['kind']}-{s['seed']}-{s['n']}")
def test_encode_matches_reference_cli_streams(H, s, encode_mode):
    """BASELINE.json configs[1]: 64 MiB stand-in for data/random_64m.dat on one GPU,
    packet stream byte-equal (md5) to the reference's --host output; plus the small cases."""
    n = s["n"]
    d_in = H.generate(s["kind"], s["seed"], n)
    if n <= (1 << 20):
        assert md5(d_in.cpu().numpy().tobytes()) == s["input_md5"]
    npk = H.packet_count(n)
    d_slots = H.encode(d_in)
    d_stream, d_off = H.compact(d_slots, npk)
    total = int(d_off[-1].item())
    assert total + 20 == s["gip_bytes"]
    assert md5(d_stream[:total].cpu().numpy().tobytes()) == s["stream_md5"]
    d_back = H.decode_stream(d_stream, d_off, npk)
    assert torch.equal(d_back[:n], d_in)
    assert H.status() == 0


def test_device_generators_match_numpy(H):
    for kind in synth.KINDS:
        for off, n in [(0, 1), (0, 100003), (8192 * 3, 70001), (8, 4096)]:
            got = H.generate(kind, 9, n, offset=off).cpu().numpy()
            assert np.array_equal(got, synth.generate(kind, 9, n, offset=off)), (kind, off, n)


def test_shader_clock_samples_of_the_throughput_kernels(H):
    """gpuar_hip_clock_samples: every 64th workgroup of encode_kernel and of the decode kernels leaves the shader clock's and
    the 100 MHz clock's counters at its start and its end; their ratio is the clock the kernel ran at (bench.py's
    roofline_valu at the measured clock).  512 MiB = 1024 groups = 16 sampled workgroups per launch; the result must be a
    plausible MI355X shader clock, reading resets the samples, and the latency kernel does not sample."""
    n = 512 << 20
    d_in = H.generate("uniform", 3, n)
    npk = H.packet_count(n)
    H.shader_clock_mhz("encode"), H.shader_clock_mhz("decode")          # reset
    assert H.shader_clock_mhz("encode") == (None, 0) and H.shader_clock_mhz("decode") == (None, 0)
    d_slots = H.encode(d_in, mode="throughput")
    d_out = H.decode(d_slots, npk)
    enc_mhz, enc_n = H.shader_clock_mhz("encode", reset=False)
    dec_mhz, dec_n = H.shader_clock_mhz("decode")
    assert enc_n == 16 and dec_n == 16, (enc_n, dec_n)
    assert 1000.0 < enc_mhz < 2600.0 and 1000.0 < dec_mhz < 2600.0, (enc_mhz, dec_mhz)
    assert H.shader_clock_mhz("encode") == (enc_mhz, enc_n)             # (reset=False above left them in place; this read clears them)
    assert H.shader_clock_mhz("encode") == (None, 0) and H.shader_clock_mhz("decode") == (None, 0)
    H.encode(d_in[:64 << 20], mode="latency")
    assert H.shader_clock_mhz("encode") == (None, 0)
    assert torch.equal(d_out[:n], d_in) and H.status() == 0
    lib = H.load()
    assert lib.gpuar_hip_clock_samples(2, (C.c_uint64 * 1024)(), 0) == -2 and lib.gpuar_hip_clock_samples(0, None, 0) == -2


def test_device_copy_copies(H):
    """gpuar_hip_copy (bench.py's measured HBM roof) moves exactly the bytes it is given, for sizes around its
    four-quads-per-thread loop and its grid cap."""
    for n in (16, 16 * 255, 16 * 1024 * 4 + 16, (8192 * 1024 * 16) + 48, 64 << 20):
        src = H.generate("uniform", 3, n + 32)
        dst = torch.zeros(n + 32, dtype=torch.uint8, device="cuda")
        H.device_copy(src, dst, n)
        assert torch.equal(dst[:n], src[:n]) and int(dst[n:].sum()) == 0, n
    with pytest.raises(H.GpuarError):
        H.device_copy(src, dst, 24)                      # not a multiple of 16


@pytest.mark.parametrize("kind", ["uniform", "zipf", "text", "zeros"])
@pytest.mark.parametrize("n", [1, 15, 16, 17, 8191, 8192, 8193, 64 * 8192, 64 * 8192 + 1, 200 * 8192 + 4097])
def test_slots_equal_oracle(H, oracle, kind, n, encode_mode):
    data = synth.generate(kind, 21, n)
    want_slots, lens, total = oracle_slots(oracle, data)
    d_slots = H.encode(torch.from_numpy(data).cuda())
    got = d_slots.cpu().numpy()
    npk = H.packet_count(n)
    for p in range(npk):
        clen = int(lens[p])
        assert np.array_equal(got[p * 8704:p * 8704 + clen], want_slots[p, :clen]), (kind, n, p)
    back = H.decode(d_slots, npk)[:n].cpu().numpy()
    assert np.array_equal(back, data)
    assert H.status() == 0


@pytest.fixture(scope="module")
def oracle_port():
    from oracle import oracle as O
    return O.PortOracle()


def test_reference_named_executors(H):
    """garCompressExecutor / garDecompressExecutor / initConstantRange: same names,
    argument meaning and NULL-stream behaviour as src/gpuar.h:74,77,78."""
    lib = H.load()
    n = 5 * 8192 + 99
    data = synth.zipf(3, n)
    d_in = torch.from_numpy(data).cuda()
    npk = H.packet_count(n)
    d_slots = torch.zeros(npk * 8704, dtype=torch.uint8, device="cuda")
    d_out = torch.zeros(npk * 8192, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    lib.initConstantRange()
    lib.garCompressExecutor(d_in.data_ptr(), n, d_slots.data_ptr(), (npk + 31) // 32)
    lib.garDecompressExecutor(d_slots.data_ptr(), npk * 8704, d_out.data_ptr(), (npk + 31) // 32)
    torch.cuda.synchronize()
    assert lib.gpuar_hip_last_error() == 0
    assert np.array_equal(d_out[:n].cpu().numpy(), data)
    from oracle import oracle as O
    want, _ = O.PortOracle().encode_slots(data)
    got = d_slots.cpu().numpy()
    for p in range(npk):
        clen = int(want[p * 8704]) | (int(want[p * 8704 + 1]) << 8)
        assert np.array_equal(got[p * 8704:p * 8704 + clen], want[p * 8704:p * 8704 + clen])


def test_argument_errors(H):
    lib = H.load()
    buf = torch.zeros(8704 * 2 + 64, dtype=torch.uint8, device="cuda")
    assert lib.gpuar_hip_encode(buf.data_ptr() + 1, 100, buf.data_ptr(), None, None) == -1      # misaligned input
    assert lib.gpuar_hip_encode(None, 100, buf.data_ptr(), None, None) == -2
    assert lib.gpuar_hip_encode(buf.data_ptr(), 0, buf.data_ptr(), None, None) == 0              # empty input: nothing to do
    lib.garCompressExecutor(buf.data_ptr() + 1, 100, buf.data_ptr(), 1)
    assert lib.gpuar_hip_last_error() == -1 and lib.gpuar_hip_last_error() == 0
    # sizes whose launch would need 2^32 threads or more are refused as arguments, before any launch: the copy
    # at 64 GiB (one 16-byte quad per thread), the compaction at 2^24 packets (one 256-thread workgroup per packet)
    assert lib.gpuar_hip_copy(buf.data_ptr(), buf.data_ptr(), 64 << 30, None) == -2
    assert lib.gpuar_hip_compact(buf.data_ptr(), 1 << 24, buf.data_ptr(), buf.data_ptr(), None) == -2
    torch.cuda.synchronize()
    assert lib.gpuar_hip_last_error() == 0 and H.status() == 0


def test_malformed_packets_are_flagged_not_fatal(H):
    rng = np.random.default_rng(5)
    npk = 64
    slots = rng.integers(0, 256, npk * 8704, dtype=np.uint8)
    for p in range(npk):
        slots[p * 8704:p * 8704 + 2] = np.frombuffer(int(8000).to_bytes(2, "little"), dtype=np.uint8)
        ulen = 8192 if p % 2 else 60000          # half the packets claim an impossible length
        slots[p * 8704 + 2:p * 8704 + 4] = np.frombuffer(int(ulen).to_bytes(2, "little"), dtype=np.uint8)
    d_out = torch.full((npk * 8192 + 4096,), 0xA5, dtype=torch.uint8, device="cuda")
    H.decode(torch.from_numpy(slots).cuda(), npk, d_out)
    flags = H.status()
    assert flags & H.STATUS_BAD_PACKET
    assert bool((d_out[npk * 8192:] == 0xA5).all())
    assert H.status() == 0                        # status is read-and-clear


def test_status_word_per_launch(H):
    """Per-call status (include/gpuar_hip.h `d_status`): two decode launches in flight on two streams, one over
    malformed packets and one over good ones, each with its own device word -- the flag lands in the word of the
    launch that met the bad packet, the other stays 0, and the device's fallback word is not touched."""
    assert H.status() == 0
    n = 200 * 8192
    data = synth.zipf(9, n)
    npk = H.packet_count(n)
    good = H.encode(torch.from_numpy(data).cuda())
    bad = good.clone()
    view = bad.view(npk, 8704)
    view[7, 2] = 0xFF                       # ulen = 0xFFxx > 8192
    view[7, 3] = 0xFF
    view[150, 4:200] = 0xFF                 # a code value no symbol owns somewhere down the packet, most likely
    words = torch.zeros(4, dtype=torch.int32, device="cuda")
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    out_bad = H.decode(bad, npk, stream=s1, d_status=words[0:1])
    out_good = H.decode(good, npk, stream=s2, d_status=words[1:2])
    d_stream, d_off = H.compact(good, npk)
    out_stream = H.decode_stream(d_stream, d_off, npk, d_status=words[2:3])
    H.encode(torch.from_numpy(data).cuda(), d_status=words[3:4])
    torch.cuda.synchronize()
    w = words.cpu().numpy()
    assert w[0] & H.STATUS_BAD_PACKET and w[1] == 0 and w[2] == 0 and w[3] == 0
    assert H.status() == 0                                   # nothing went to the fallback word
    assert np.array_equal(out_good[:n].cpu().numpy(), data) and np.array_equal(out_stream[:n].cpu().numpy(), data)
    # every packet but the two damaged ones still decodes to its bytes (a bad packet garbles only its own 8192)
    ok = np.ones(npk, dtype=bool)
    ok[[7, 150]] = False
    got = out_bad.cpu().numpy().reshape(npk, 8192)
    assert np.array_equal(got[ok], data.reshape(npk, 8192)[ok])
    assert lib_misaligned_status(H) == -1


def lib_misaligned_status(H):
    buf = torch.zeros(8704 + 64, dtype=torch.uint8, device="cuda")
    return H.load().gpuar_hip_decode(buf.data_ptr(), 1, buf.data_ptr(), buf.data_ptr() + 2, None)


def test_executor_decodes_every_slot_that_starts_inside_size(H):
    """garDecompressExecutor's `size` (src/gpuar_kernel.cu:916-934): a slot is decoded when it STARTS inside
    `size` -- the last one may be cut short right behind its packet -- and nothing at or behind source + size is
    read: the buffer ends exactly there, with a canary page of 0xEE behind it that must not influence anything."""
    lib = H.load()
    n = 66 * 8192 + 1000                     # two wavefronts, the second nearly empty, last packet short
    data = synth.text(12, n)
    npk = H.packet_count(n)
    slots = H.encode(torch.from_numpy(data).cuda())
    last_clen = int(slots[(npk - 1) * 8704].item()) | (int(slots[(npk - 1) * 8704 + 1].item()) << 8)
    size = (npk - 1) * 8704 + last_clen      # the caller's buffer ends with the last packet
    d_out = torch.full((npk * 8192,), 0x5A, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    lib.garDecompressExecutor(slots.data_ptr(), size, d_out.data_ptr(), (npk + 31) // 32)
    torch.cuda.synchronize()
    assert lib.gpuar_hip_last_error() == 0 and H.status() == 0
    assert np.array_equal(d_out[:n].cpu().numpy(), data)
    assert bool((d_out[n:] == 0x5A).all())
    # one byte less and the last slot no longer starts inside: floor -> it is left alone
    d_out.fill_(0x5A)
    lib.garDecompressExecutor(slots.data_ptr(), (npk - 1) * 8704, d_out.data_ptr(), (npk + 31) // 32)
    torch.cuda.synchronize()
    assert np.array_equal(d_out[:(npk - 1) * 8192].cpu().numpy(), data[:(npk - 1) * 8192])
    assert bool((d_out[(npk - 1) * 8192:] == 0x5A).all())


@pytest.mark.parametrize("shift", [0, 4, 8, 12])
def test_decode_stream_at_every_allowed_alignment(H, oracle, shift):
    """gpuar_hip_decode_stream documents a 4-byte aligned stream pointer: the same stream at base + 0/4/8/12
    (the wavefront's 16-byte piece grid is then skewed against the packets), with short packets in the MIDDLE of
    the stream (lanes of different lengths inside one wavefront: the PLAIN variant of the symbol step and the
    partial-block tail, away from the end of the file), a 1-byte packet among them."""
    rng = np.random.default_rng(77 + shift)
    lens = [8192] * 130
    for at, ln in ((3, 100), (40, 8191), (64, 1), (65, 4097), (100, 63), (101, 64), (102, 65), (129, 5000)):
        lens[at] = ln
    packets = [synth.zipf(100 + i, ln) if i % 3 else rng.integers(0, 256, ln, dtype=np.uint8) for i, ln in enumerate(lens)]
    stream = np.concatenate([oracle.encode_stream(p) for p in packets])
    offs = np.zeros(len(lens) + 1, dtype=np.int64)
    at = 0
    for i, p in enumerate(packets):
        offs[i] = at
        at += int(stream[at]) | (int(stream[at + 1]) << 8)
    offs[len(lens)] = at
    assert at == stream.size
    raw = torch.zeros(stream.size + 64, dtype=torch.uint8, device="cuda")
    base = (-raw.data_ptr()) % 16 + shift                    # 16-byte aligned + shift
    raw[base:base + stream.size] = torch.from_numpy(stream).cuda()
    d_stream = raw[base:base + stream.size]
    assert d_stream.data_ptr() % 16 == shift
    word = torch.zeros(1, dtype=torch.int32, device="cuda")
    d_out = torch.full((len(lens) * 8192,), 0xC3, dtype=torch.uint8, device="cuda")
    H.decode_stream(d_stream, torch.from_numpy(offs).cuda(), len(lens), d_out, d_status=word)
    torch.cuda.synchronize()
    assert int(word.item()) == 0
    got = d_out.cpu().numpy().reshape(len(lens), 8192)
    for i, p in enumerate(packets):
        assert np.array_equal(got[i, :p.size], p), (shift, i, p.size)
        assert bool((got[i, p.size:] == 0xC3).all()), (shift, i)      # a short packet writes only its ulen bytes


def test_round_trip_properties_at_scale(H):
    """1 GiB text (BASELINE.json configs[2]) on one GPU: size-independent checks --
    decode(encode(x)) == x, slot lengths sum to the compacted size, compaction is
    a pure gather (stream decodes to the same bytes), ratio in the expected band."""
    n = 1 << 30
    d_in = H.generate("text", 1, n)
    npk = H.packet_count(n)
    d_slots = H.encode(d_in)
    d_stream, d_off = H.compact(d_slots, npk)
    total = int(d_off[-1].item())
    clen = d_slots.view(npk, 8704)[:, 0].to(torch.int64) | (d_slots.view(npk, 8704)[:, 1].to(torch.int64) << 8)
    assert int(clen.sum().item()) == total
    assert torch.equal(torch.cumsum(clen, 0), d_off[1:])
    assert 0.66 < total / n < 0.69            # measured ratio 0.675 for text
    d_back = H.decode(d_slots, npk)
    assert torch.equal(d_back[:n], d_in)
    del d_back
    d_back2 = H.decode_stream(d_stream, d_off, npk)
    assert torch.equal(d_back2[:n], d_in)
    assert H.status() == 0
    # oracle spot check: 32 packets from the middle
    from oracle import oracle as O
    p0 = npk // 2
    host = d_in[p0 * 8192:(p0 + 32) * 8192].cpu().numpy()
    want = O.require_best().encode_stream(host)
    got = d_stream[int(d_off[p0].item()):int(d_off[p0 + 32].item())].cpu().numpy()
    assert np.array_equal(got, want)


def test_compaction_offsets_beyond_4gib(H):
    """Maximum sizes: 5 GiB of uniform data compacts to > 4 GiB, so packet offsets must be 64-bit
    all the way (the reference's container cannot represent this; SURVEY.md section 7 risk 4)."""
    n = 5 << 30
    d_in = H.generate("uniform", 7, n)
    npk = H.packet_count(n)
    d_slots = H.encode(d_in)
    d_stream, d_off = H.compact(d_slots, npk)
    total = int(d_off[-1].item())
    assert total > (1 << 32)
    clen = d_slots.view(npk, 8704)[:, 0].to(torch.int64) | (d_slots.view(npk, 8704)[:, 1].to(torch.int64) << 8)
    assert int(clen.sum().item()) == total
    assert bool((d_off[1:] - d_off[:-1] == clen).all())
    # the last packets decode from the compacted stream (offsets above 4 GiB in use)
    del d_slots
    d_back = H.decode_stream(d_stream, d_off, npk)
    assert torch.equal(d_back[n - (1 << 20):n], d_in[n - (1 << 20):])
    assert torch.equal(d_back[:1 << 20], d_in[:1 << 20])
    assert H.status() == 0


def test_many_mixed_packets_against_oracle(H, oracle, encode_mode):
    """2048 packets, every one from a different source model (uniform, k-symbol, geometric, long
    runs, ramps, near-midpoint pairs, constant, constant with a few late strangers): slot-for-slot equality with the
    oracle, then decode."""
    rng = np.random.default_rng(20261003)
    npk = 2048
    data = np.empty(npk * 8192, dtype=np.uint8)
    for p in range(npk):
        view = data[p * 8192:(p + 1) * 8192]
        mode = p % 8
        if mode == 0:
            view[:] = rng.integers(0, 256, 8192, dtype=np.uint8)
        elif mode == 1:
            view[:] = rng.integers(0, int(rng.integers(1, 9)), 8192, dtype=np.uint8) * int(rng.integers(1, 32))
        elif mode == 2:
            view[:] = (rng.geometric(float(rng.uniform(0.02, 0.5)), 8192) % 256).astype(np.uint8)
        elif mode == 3:
            view[:] = np.repeat(rng.integers(0, 256, 8192 // 32, dtype=np.uint8), 32)
        elif mode == 4:
            view[:] = (np.arange(8192) * int(rng.integers(1, 255))) % 256
        elif mode == 5:
            view[:] = rng.choice(np.array([0x7F, 0x80], dtype=np.uint8), 8192)
        elif mode == 6:
            view[:] = int(rng.integers(0, 256))
            if p % 16 == 14:        # a constant packet with a few strangers late in it: the narrowest intervals there are
                late = rng.integers(6000, 8192, 12)      # (width 1 or 2 of a range just above 2^14, total near 2^13)
                view[late] = rng.integers(0, 256, 12, dtype=np.uint8)
        else:
            view[:] = np.sort(rng.integers(0, 256, 8192, dtype=np.uint8))
    want, want_len, total = oracle_slots(oracle, data)
    d_slots = H.encode(torch.from_numpy(data).cuda())
    got = d_slots.cpu().numpy()
    got_len = got.reshape(npk, 8704)[:, 0].astype(np.int64) | (got.reshape(npk, 8704)[:, 1].astype(np.int64) << 8)
    assert np.array_equal(want_len, got_len)
    mask = np.arange(8704)[None, :] < want_len[:, None]
    assert np.array_equal(got.reshape(npk, 8704)[mask], want[mask])
    d_stream, d_off = H.compact(d_slots, npk)
    assert int(d_off[-1].item()) == total
    assert np.array_equal(H.decode(d_slots, npk).cpu().numpy(), data)
    assert np.array_equal(H.decode_stream(d_stream, d_off, npk).cpu().numpy(), data)
    assert H.status() == 0


REAL_TEXT_GLOBS = ("*.md", "bench.py", "__graft_entry__.py", "gpuar_amd/*.py", "gpuar_amd/csrc/*.hip", "gpuar_amd/csrc/*.h",
                   "gpuar_amd/csrc/Makefile", "gpuar_amd/csrc/host/*", "include/*.h", "oracle/*.c", "oracle/*.py", "oracle/*.cpp",
                   "oracle/*.sh", "tests/*.py", "tests/*.cpp", "tools/*.hip", "tools/*.py", "tools/*.sh", "tools/*.cpp", "tools/*.md",
                   "profiles/*.txt", "profiles/*.md")


def real_text(min_bytes):
    """This repository's own text -- sources, documents, profile summaries, in sorted path order -- repeated until it is at
    least `min_bytes` long; copy k has every 1009th byte from position k on moved by k (still 7-bit), so no two copies hold
    the same packets.  (The repository on the GPU box is the snapshot gpurun pushes: the same files, no .git.)"""
    import glob
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = sorted({f for pat in REAL_TEXT_GLOBS for f in glob.glob(os.path.join(root, pat)) if os.path.isfile(f)})
    base = np.concatenate([np.fromfile(f, dtype=np.uint8) for f in files])
    assert base.size > (1 << 20), "the repository's text shrank below 1 MiB?"
    copies = []
    for k in range((min_bytes + base.size - 1) // base.size):
        c = base.copy()
        c[k::1009] = (c[k::1009].astype(np.uint32) + k) % 128
        copies.append(c)
    return np.concatenate(copies), len(files)


def test_real_text_against_oracle(H, oracle, encode_mode):
    """Real bytes, once (SURVEY.md section 6's third probe row was source text; BASELINE.md section 2 row 3: ratio 0.5717): every
    other GPU input is synthetic.  >= 32 MiB of this repository's own text through both encode kernels: the whole stream
    byte-equal to the oracle's, both decoders, the ratio inside the band real source text lands in."""
    data, n_files = real_text(32 << 20)
    assert data.size >= (32 << 20) and n_files > 50
    want = oracle.encode_stream(data)
    stream, offs, d_slots, npk = gpu_stream(H, data)
    assert stream.size == want.size and np.array_equal(stream, want), "packet stream differs from the oracle on real text"
    ratio = (stream.size + 20) / data.size
    assert 0.55 < ratio < 0.70, ratio
    d_stream, d_off = H.compact(d_slots, npk)
    assert np.array_equal(H.decode(d_slots, npk).cpu().numpy()[:data.size], data)
    assert np.array_equal(H.decode_stream(d_stream, d_off, npk).cpu().numpy()[:data.size], data)
    assert H.status() == 0


def test_packets_that_keep_carrying_against_oracle(H, oracle, encode_mode):
    """The coder's rare path (a leaving dword of 32 ones that has to wait for its carry, waiting dwords let go -- hand-written
    behind a scalar branch in the store region, lane_codec.h) with every lane of a wavefront in a different state of it:
    512 packets of two symbols either side of the interval's midpoint, few-symbol alphabets at every skew, and long
    constant stretches, of ragged lengths; slot for slot against the oracle."""
    rng = np.random.default_rng(2904)
    npk = 512
    data = np.zeros(npk * 8192, dtype=np.uint8)
    for p in range(npk):
        view = data[p * 8192:(p + 1) * 8192]
        kind = p % 4
        if kind == 0:
            view[:] = np.where(rng.random(8192) < 0.5, 0x7F, 0x80).astype(np.uint8)
        elif kind == 1:
            view[:] = rng.choice(np.array([0x7F, 0x80, 0x00, 0xFF], dtype=np.uint8), 8192, p=[0.45, 0.45, 0.05, 0.05])
        elif kind == 2:
            view[:] = rng.choice(rng.integers(0, 256, int(rng.integers(2, 6))).astype(np.uint8), 8192)
        else:
            view[:] = np.repeat(rng.integers(0, 256, 8192 // 64).astype(np.uint8), 64)
    n = npk * 8192 - 3001                               # the last packet is a short one
    data = data[:n]
    want, want_len, total = oracle_slots(oracle, data)
    d_slots = H.encode(torch.from_numpy(data).cuda())
    got = d_slots.cpu().numpy().reshape(npk, 8704)
    got_len = got[:, 0].astype(np.int64) | (got[:, 1].astype(np.int64) << 8)
    assert np.array_equal(want_len, got_len)
    mask = np.arange(8704)[None, :] < want_len[:, None]
    assert np.array_equal(got[mask], want[mask])
    assert np.array_equal(H.decode(d_slots, npk).cpu().numpy()[:n], data)
    assert H.status() == 0


def test_packets_that_drain_the_stream_window_fast_against_oracle(H, oracle):
    """The decoder refills its stream window every SECOND symbol and asks for ring pieces by where its reader stands
    (gpuar_kernels.hip, "the stream reader", "the stream ring"): both rest on a symbol taking at most 16 stream bits.  These
    packets make lanes run at the top of that for thousands of symbols while their neighbours idle: a model trained on one
    byte (or on a few) and then fed bytes it has never seen -- 12-13 bits a symbol --, switch points and lengths ragged, next to
    constant packets (0.2 bits a symbol) and uniform ones (8).  Slots against the oracle; both decoders back to the bytes."""
    rng = np.random.default_rng(6006)
    npk = 192
    data = np.zeros(npk * 8192, dtype=np.uint8)
    for p in range(npk):
        view = data[p * 8192:(p + 1) * 8192]
        kind = p % 6
        if kind == 0:                                       # one byte for a while, then every other byte in turn
            cut = int(rng.integers(512, 7000))
            view[:cut] = rng.integers(0, 256)
            view[cut:] = (np.arange(8192 - cut) * 37 + int(rng.integers(0, 256))).astype(np.uint8)
        elif kind == 1:                                     # a few bytes, then bytes drawn from the unseen rest, then back
            few = rng.integers(0, 256, 3).astype(np.uint8)
            a, b = sorted(int(v) for v in rng.integers(256, 7900, 2))
            view[:] = rng.choice(few, 8192)
            rest = np.setdiff1d(np.arange(256, dtype=np.uint8), few)
            view[a:b] = rng.choice(rest, b - a)
        elif kind == 2:
            view[:] = rng.integers(0, 256)                  # constant: the slowest reader
        elif kind == 3:
            view[:] = rng.integers(0, 256, 8192)            # uniform
        elif kind == 4:                                     # bursts of rare bytes in a constant packet
            view[:] = 0x20
            for _ in range(int(rng.integers(1, 40))):
                at = int(rng.integers(0, 8100))
                view[at:at + int(rng.integers(1, 90))] = rng.integers(0, 256, 1)
        else:                                               # a ramp over all byte values, then a constant tail
            cut = int(rng.integers(300, 8000))
            view[:cut] = (np.arange(cut) % 256).astype(np.uint8)
            view[cut:] = 0xFF
    n = npk * 8192 - 777
    data = data[:n]
    want, want_len, total = oracle_slots(oracle, data)
    assert 7500 < want_len.max() <= 8704                      # some packets really are long, none outgrows its slot
    d_slots = H.encode(torch.from_numpy(data).cuda())
    got = d_slots.cpu().numpy().reshape(npk, 8704)
    got_len = got[:, 0].astype(np.int64) | (got[:, 1].astype(np.int64) << 8)
    assert np.array_equal(want_len, got_len)
    mask = np.arange(8704)[None, :] < want_len[:, None]
    assert np.array_equal(got[mask], want[mask])
    assert np.array_equal(H.decode(d_slots, npk).cpu().numpy()[:n], data)
    d_stream, d_off = H.compact(d_slots, npk)
    assert np.array_equal(H.decode_stream(d_stream, d_off, npk).cpu().numpy()[:n], data)
    assert H.status() == 0


@pytest.mark.parametrize("seed", [7, 8])
def test_round_trips_over_ten_source_models(H, oracle, seed, encode_mode):
    """4096 packets from ten source models -- uniform, few symbols, geometric, runs, ramps, midpoint pairs, a constant with
    1-40 strangers late in the packet (intervals of width 1-3), sorted, one dominant symbol, zipf -- through both decoders;
    every 16th packet's stream against the reference codec."""
    rng = np.random.default_rng(seed)
    npk = 4096
    data = np.empty(npk * 8192, dtype=np.uint8)
    for p in range(npk):
        v = data[p * 8192:(p + 1) * 8192]
        m = p % 10
        if m == 0:
            v[:] = rng.integers(0, 256, 8192, dtype=np.uint8)
        elif m == 1:
            v[:] = rng.integers(0, int(rng.integers(1, 9)), 8192, dtype=np.uint8) * int(rng.integers(1, 32))
        elif m == 2:
            v[:] = (rng.geometric(float(rng.uniform(0.02, 0.5)), 8192) % 256).astype(np.uint8)
        elif m == 3:
            v[:] = np.repeat(rng.integers(0, 256, 8192 // 32, dtype=np.uint8), 32)
        elif m == 4:
            v[:] = (np.arange(8192) * int(rng.integers(1, 255))) % 256
        elif m == 5:
            v[:] = rng.choice(np.array([0x7F, 0x80], dtype=np.uint8), 8192)
        elif m == 6:
            v[:] = int(rng.integers(0, 256))
            late = rng.integers(4000, 8192, int(rng.integers(1, 40)))
            v[late] = rng.integers(0, 256, late.size, dtype=np.uint8)
        elif m == 7:
            v[:] = np.sort(rng.integers(0, 256, 8192, dtype=np.uint8))
        elif m == 8:
            k = int(rng.integers(2, 6))
            v[:] = rng.choice(rng.integers(0, 256, k, dtype=np.uint8), 8192, p=np.array([0.97] + [0.03 / (k - 1)] * (k - 1)))
        else:
            v[:] = (rng.zipf(1.3, 8192) % 256).astype(np.uint8)
    d_in = torch.from_numpy(data).cuda()
    d_slots = H.encode(d_in)
    assert H.status() == 0
    assert torch.equal(H.decode(d_slots, npk), d_in)
    d_stream, d_off = H.compact(d_slots, npk)
    assert torch.equal(H.decode_stream(d_stream, d_off, npk), d_in)
    assert H.status() == 0
    off = d_off.cpu().numpy()
    got = d_stream[:int(off[-1])].cpu().numpy()
    for p in range(0, npk, 16):
        want = oracle.encode_stream(data[p * 8192:(p + 1) * 8192])
        assert np.array_equal(got[off[p]:off[p + 1]], want), f"packet {p} (source model {p % 10})"


@pytest.fixture(scope="module")
def small_slot_lib():
    """libgpuar_hip.so built with 1024-byte slots (tests/_build/): ordinary data then drives packets past the end of
    their slots, which no input can do with the real 8704 bytes -- the only way to run the coders' end-of-slot paths
    (the clamped store region, GPUAR_STATUS_SLOT_OVERFLOW) on the device."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out_dir = os.path.join(root, "tests", "_build")
    os.makedirs(out_dir, exist_ok=True)
    so = os.path.join(out_dir, "libgpuar_hip_slot1024.so")
    srcs = [os.path.join(root, "gpuar_amd", "csrc", f) for f in ("gpuar_kernels.hip", "lane_codec.h")]
    if not os.path.exists(so) or any(os.path.getmtime(x) > os.path.getmtime(so) for x in srcs):
        host_o = os.path.join(root, "build", "host_codec.o")
        if not os.path.exists(host_o):
            subprocess.check_call(["make", "-C", os.path.join(root, "gpuar_amd", "csrc"), host_o])
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", f"-I{root}/include",
                               "-Wno-unused-function", "-mllvm", "-phi-node-folding-threshold=64", "-mllvm",
                               "-two-entry-phi-node-folding-threshold=64", "-DGPUAR_SLOT_BYTES=1024u", "-shared", "-o", so,
                               host_o, srcs[0]], cwd=os.path.dirname(srcs[0]))
    lib = C.CDLL(so)
    lib.gpuar_hip_encode_mode.restype = C.c_int
    lib.gpuar_hip_encode_mode.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    return lib


@pytest.mark.parametrize("mode", [1, 2], ids=["throughput", "latency"])
def test_slot_overflow_is_flagged_and_contained_on_the_device(small_slot_lib, oracle, mode):
    """With 1024-byte slots: packets of random bytes (~8260 bytes of output) outgrow their slots, a packet of zeros
    (210 bytes) between them fits.  The launch reports GPUAR_STATUS_SLOT_OVERFLOW in its own status word, nothing is
    written beyond the slot array or into a neighbour's slot, an overflowed slot's header says clen = slot size and
    ulen, what it holds up to its last two dwords is the true prefix of the packet, and the packet that fits is exact."""
    slot = 1024
    rng = np.random.default_rng(5)
    groups = 3                                              # 192 packets: three workgroups, every lane of the first busy
    data = rng.integers(0, 256, groups * 64 * 8192, dtype=np.uint8)
    fits = [1, 64, 130]                                     # these packets are zeros
    for p in fits:
        data[p * 8192:(p + 1) * 8192] = 0
    npk = groups * 64
    d_in = torch.from_numpy(data).cuda()
    guard = 4096
    d_slots = torch.full((npk * slot + guard,), 0xA5, dtype=torch.uint8, device="cuda")
    word = torch.zeros(1, dtype=torch.int32, device="cuda")
    rc = small_slot_lib.gpuar_hip_encode_mode(d_in.data_ptr(), data.size, d_slots.data_ptr(), word.data_ptr(), None, mode)
    torch.cuda.synchronize()
    assert rc == 0 and int(word.item()) & 1, (rc, int(word.item()))
    got = d_slots.cpu().numpy()
    assert (got[npk * slot:] == 0xA5).all()                 # nothing beyond the last slot
    zero_pkt = np.frombuffer(oracle.encode_packet(bytes(8192)), dtype=np.uint8)
    for p in fits:
        assert np.array_equal(got[p * slot:p * slot + zero_pkt.size], zero_pkt), p
        assert (got[p * slot + zero_pkt.size + 3:(p + 1) * slot] == 0xA5).all(), p      # (the tail store may round up to a dword)
    for p in (0, 2, 63, 65, 191):
        want = np.frombuffer(oracle.encode_packet(data[p * 8192:(p + 1) * 8192].tobytes()), dtype=np.uint8)
        s0 = got[p * slot:(p + 1) * slot]
        assert int(s0[0]) | (int(s0[1]) << 8) == slot and int(s0[2]) | (int(s0[3]) << 8) == 8192, p
        assert np.array_equal(s0[4:slot - 8], want[4:slot - 8]), p


def bench_line_and_detail(r, tmp_path):
    """(stdout line, full object) of a finished bench.py child.  The line is read the way the DRIVER reads it -- out of the last
    6000 bytes of stdout + "---- stderr ----" + stderr -- and held to its size rule; the deep fields come from the detail file
    the line names (the run was given --detail-file under tmp_path)."""
    import bench_stub
    d = bench_stub.line_from_driver_tail(r.stdout, r.stderr)
    assert len([l for l in r.stdout.splitlines() if l.strip()]) >= 1
    assert [l for l in r.stdout.splitlines() if l.startswith("{")][-1:] == [l for l in r.stdout.splitlines() if l.startswith('{"metric"')][-1:]
    assert d["detail"] == str(tmp_path / "detail.json"), d["detail"]
    full = json.load(open(d["detail"]))
    assert abs(full["value"] - d["value"]) <= 1e-5 * full["value"] and full["roofline"]["kernel"] == d["roofline"]["kernel"]
    return d, full


def test_bench_two_rank_flow_on_one_gpu(tmp_path):
    """bench.py --gpus 2 through torch.distributed.run, both ranks on this box's one GPU (gloo control
    plane, GPUAR_OVERSUBSCRIBE_DEVICES=1): rank r codes bytes [r*B, (r+1)*B) of the stream, the JSON line
    aggregates both, every rank's round trip must pass.  (Real multi-GPU runs use RCCL, one GPU per rank.)"""
    import json
    import socket
    import subprocess
    import sys
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GPUAR_OVERSUBSCRIBE_DEVICES="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1",
           "--gib-per-gpu", "0.25", "--total-gib", "0.5", "--no-cpu-baseline", "--no-small-config", "--no-live-traffic", "--detail-file", str(tmp_path / "detail.json")]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    d, _ = bench_line_and_detail(r, tmp_path)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["roundtrip_equal"] is True and d["oracle_prefix_match"] is True
    assert abs(d["compression_ratio"] - 1.00804) < 1e-3
    assert d["value"] > 0 and d["roofline"]["bound"] == "hbm" and d["roofline"]["frac"] > 0


def _self_launched_bench(extra, tmp_path):
    """`python bench.py --gpus 2 ...` with NO launcher around it: bench.py must start its two ranks itself
    (fresh child processes, before it touches the GPU) and relay rank 0's JSON line."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["GPUAR_OVERSUBSCRIBE_DEVICES"] = "1"
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1",
           "--no-cpu-baseline", "--no-small-config", "--no-live-traffic", "--detail-file", str(tmp_path / "detail.json"), *extra]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    return bench_line_and_detail(r, tmp_path)


def test_bench_launches_its_own_ranks(tmp_path):
    line, d = _self_launched_bench(["--gib-per-gpu", "0.25", "--total-gib", "0.5"], tmp_path)
    assert d["n_gpus"] == 2 and d["n_ranks_seen"] == 2 and d["scaling"] == "weak"
    assert d["roundtrip_equal"] is True and d["oracle_prefix_match"] is True and d["device_status"] == 0
    assert abs(d["compression_ratio"] - 1.00804) < 1e-3
    assert abs(d["value"] * d["ms_per_step"] * 1e-3 - 2 * 0.25 * 1.073741824) < 1e-6    # both ranks' bytes over the time
    # the stdout line: the same run in numbers only (per-rank min/max, no arrays), with the extras of a world of N in short form
    assert line["n_gpus"] == 2 and line["n_ranks_seen"] == 2 and line["roundtrip_equal"] is True and line["checker"] == d["checker"]
    assert set(line["per_rank"]) == {"encode_ms_min", "encode_ms_max", "decode_ms_min", "decode_ms_max"}
    assert line["other_scaling"]["scaling"] == "strong" and line["other_scaling"]["roundtrip_equal"] is True
    assert line["gather_probe"]["cheaper"] == d["gather_probe"]["cheaper"] and line["gather_probe"]["rccl"] is False
    # what an 8-GPU run must carry in the same line: per-rank kernel times, the other scaling mode,
    # and the staged-copy vs gather measurement
    pr = d["per_rank"]
    assert 0 < pr["encode_ms_min"] <= pr["encode_ms_max"] and 0 < pr["decode_ms_min"] <= pr["decode_ms_max"]
    assert len(pr["compressed_bytes"]) == 2
    o = d["other_scaling"]
    assert o["scaling"] == "strong" and o["roundtrip_equal"] is True and o["value"] > 0 and o["total_bytes"] == 1 << 29
    g = d["gather_probe"]
    assert g["ranks"] == 2 and g["bytes_per_rank"] > 0 and g["staged_d2h_ms"] > 0 and g["gather_then_d2h_ms"] > 0
    assert g["cheaper"] in ("staged hipMemcpyAsync", "RCCL gather")
    for k in ("compact_ms", "encode_plus_compact_ms", "decode_stream_ms", "roofline_compact", "roofline_decode_stream"):
        assert k in d, k
    assert d["decode_stream_roundtrip_equal"] is True


def test_bench_prints_its_line_when_a_scaling_extra_raises(tmp_path):
    """The N > 1 extras (gather probe, the other scaling mode) must never cost the run its line: when one of them RAISES on
    rank 0 (a transport error, out of memory) the line of the timed pass is printed with `scaling_extras` saying what failed,
    and the run ends -- without the closing barrier, which the other rank (still inside the extras' collectives) would never
    reach -- with exit code 0."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(GPUAR_OVERSUBSCRIBE_DEVICES="1", GPUAR_TEST_FAIL_EXTRAS="0")
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--gib-per-gpu", "0.125",
           "--total-gib", "0.25", "--no-cpu-baseline", "--no-small-config", "--no-live-traffic", "--extras-timeout", "20", "--detail-file", str(tmp_path / "detail.json")]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600, cwd=str(tmp_path))
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, (r.returncode, r.stdout[-500:], r.stderr[-1500:])
    d, full = bench_line_and_detail(r, tmp_path)
    assert "GPUAR_TEST_FAIL_EXTRAS" in full["scaling_extras"] and "other_scaling" not in full
    assert d["n_gpus"] == 2 and d["n_ranks_seen"] == 2 and d["roundtrip_equal"] is True and d["value"] > 0
    assert "GPUAR_TEST_FAIL_EXTRAS" in d["scaling_extras"] and "other_scaling" not in d and "gather_probe" not in d


def test_bench_four_rank_dry_run_on_one_gpu(tmp_path):
    """The widest dry run this pool allows: a one-GPU box admits SIX processes on its card at once (a seventh gets the
    whole run killed by its process guard); this test process is one of them and the launcher's agent another, so
    `bench.py --gpus 4` is self-launched with all four ranks oversubscribed onto the one GPU -- the N > 2 flow of the line
    the driver's 8-GPU run will print: n_ranks_seen, one compressed size per rank, per-rank kernel times, the other
    scaling mode and the gather probe, all over four ranks.  (Eight ranks: the CPU rehearsal in tests/test_sharding_gloo.py.)"""
    import subprocess
    import sys
    world = 4
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["GPUAR_OVERSUBSCRIBE_DEVICES"] = "1"
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", str(world), "--steps", "1", "--warmup", "1",
           "--gib-per-gpu", "0.125", "--total-gib", "0.5", "--no-cpu-baseline", "--no-small-config", "--no-live-traffic", "--detail-file", str(tmp_path / "detail.json")]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-2000:]
    line, d = bench_line_and_detail(r, tmp_path)
    assert line["n_gpus"] == world and line["n_ranks_seen"] == world and "compressed_bytes" not in line["per_rank"]
    assert d["n_gpus"] == world and d["n_ranks_seen"] == world and d["scaling"] == "weak"
    assert d["roundtrip_equal"] is True and d["device_status"] == 0
    assert len(d["per_rank"]["compressed_bytes"]) == world and all(c > 0 for c in d["per_rank"]["compressed_bytes"])
    assert abs(d["value"] * d["ms_per_step"] * 1e-3 - world * 0.125 * 1.073741824) < 1e-6
    assert d["other_scaling"]["scaling"] == "strong" and d["other_scaling"]["roundtrip_equal"] is True
    assert d["other_scaling"]["total_bytes"] == world * (1 << 27)
    assert d["gather_probe"]["ranks"] == world and d["gather_probe"]["cheaper"] in ("staged hipMemcpyAsync", "RCCL gather")
    assert "hbm_copy_peak" in d and d["roofline"]["peak_measured_copy"] > 0


def test_rccl_preflight_at_world_size_one(tmp_path):
    """The first 8-GPU run must not be the first RCCL run: bench.py with --force-collectives brings up
    init_process_group("nccl", device_id=...) -- RCCL -- on this box's one GPU as a world of ONE rank and routes its barrier,
    the MAX of the elapsed time, the MIN of the sample size and the all_gather of the per-rank rows through it, with tensors
    on the device.  What a world of one cannot rehearse is the point-to-point leg of the gather probe (batch_isend_irecv to
    rank 0 needs a peer): that leg is skipped and the record says how many ranks it saw."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "GPUAR_OVERSUBSCRIBE_DEVICES")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--force-collectives", "--steps", "2", "--warmup", "1",
           "--gib-per-gpu", "0.25", "--no-cpu-baseline", "--no-small-config", "--no-live-traffic", "--no-by-kind", "--init-timeout", "240",
           "--detail-file", str(tmp_path / "detail.json")]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900, cwd=str(tmp_path))
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    line, d = bench_line_and_detail(r, tmp_path)
    assert line["collectives"]["backend"] == "nccl" and line["collectives"]["calls"] >= 6 and line["gather_probe"]["rccl"] is True
    c = d["collectives"]
    assert c["backend"] == "nccl" and c["through_torch_distributed"] is True and c["world"] == 1
    # the first barrier behind init, two around the timed steps, eight inside the gather probe's timing loop ...
    assert c["calls"]["barrier"] >= 3 and c["calls"]["all_reduce_max"] >= 1 and c["calls"]["all_reduce_min"] >= 1 and c["calls"]["all_gather"] >= 1
    assert d["n_gpus"] == 1 and d["n_ranks_seen"] == 1 and d["roundtrip_equal"] is True and d["oracle_prefix_match"] is True
    g = d["gather_probe"]
    assert g["ranks"] == 1 and g["transport"].startswith("RCCL") and g["staged_d2h_ms"] > 0 and g["gather_then_d2h_ms"] > 0


def test_bench_refuses_more_ranks_than_gpus_before_starting_any(tmp_path):
    """`bench.py --gpus 8` on a box with one GPU and no oversubscription hook: one clear line, non-zero, within seconds --
    no rank is spawned, no rendezvous is waited for."""
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "GPUAR_OVERSUBSCRIBE_DEVICES")}
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=300, cwd=str(tmp_path))
    assert r.returncode != 0 and "nothing was started" in r.stderr and "--gpus 8" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert time.time() - t0 < 120
    # the same under a launcher that hands this node more ranks than it has GPUs: the rank says so before any rendezvous
    env.update(WORLD_SIZE="8", RANK="5", LOCAL_RANK="5", MASTER_ADDR="127.0.0.1", MASTER_PORT="1")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8"], capture_output=True, text=True, env=env,
                       timeout=300, cwd=str(tmp_path))
    assert r.returncode != 0 and "nothing was started" in r.stderr


def test_bench_line_carries_text_and_zipf_next_to_uniform(tmp_path):
    """One driver-timed line for every single-GPU BASELINE workload: after the uniform pass, text(1) and
    zipf(1) at the same size, each with its own times, ratio, round trip, oracle prefix and roofline objects."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "GPUAR_OVERSUBSCRIBE_DEVICES")}
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--steps", "2", "--warmup", "1", "--gib-per-gpu", "0.5", "--by-kind-steps", "2",
           "--no-cpu-baseline", "--no-small-config", "--detail-file", str(tmp_path / "detail.json")]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-3000:]
    line, d = bench_line_and_detail(r, tmp_path)
    assert set(line["by_kind"]) == {"text", "zipf"} and all(k["ok"] is True and k["frac"] > 0 for k in line["by_kind"].values())
    # the coder kernels' HBM bytes were MEASURED in this run (two rocprofv3 --pmc child runs behind the timed pass): the line says
    # "live", and the figure is the algorithmic N + C and a little more, not less and not a multiple
    assert "error" not in d["traffic_live"], d["traffic_live"]
    for key in ("roofline", "roofline_encode"):
        assert line[key]["traffic_from"] == "live", line[key]
        algo = d[key]["algorithmic_bytes_per_launch"]
        assert 0.97 * algo < line[key]["traffic"] < 1.6 * algo, (key, line[key]["traffic"], algo)
    assert d["roofline_decode"]["traffic_from"] == "live" and "traffic (rocprofv3" in d["roofline"]["measured_live"]
    # ... and so were the issue-side counters (the third pass): instructions per symbol step and the vector pipe's share
    assert "issue_counters_error" not in d["traffic_live"], d["traffic_live"]
    assert line["roofline"]["counters_from"] == "live" and line["roofline_encode"]["counters_from"] == "live"
    assert 70 < line["roofline_encode"]["valu_insts_per_step"] < 90 and 70 < d["roofline_decode"]["valu_insts_per_symbol_step"] < 95
    assert 0.3 < d["roofline_decode"]["roofline_valu"]["frac"] < 1.0 and 0.5 < line["roofline_encode"]["valu_frac"] < 1.05
    assert abs(line["by_kind"]["text"]["ratio"] - d["by_kind"]["text"]["compression_ratio"]) < 1e-5
    assert d["roofline"]["kernel"] in ("decode_slots_kernel", "encode_kernel") and d["roofline_encode"]["kernel"] == "encode_kernel"
    assert d["collectives"]["through_torch_distributed"] is False
    assert set(d["by_kind"]) == {"text", "zipf"}
    for kind, lo, hi in (("text", 0.66, 0.69), ("zipf", 0.77, 0.80)):
        k = d["by_kind"][kind]
        assert k["roundtrip_equal"] is True and k["oracle_prefix_match"] is True and k["device_status"] == 0, kind
        assert lo < k["compression_ratio"] < hi, (kind, k["compression_ratio"])
        assert k["encode_ms"] > 0 and k["decode_ms"] > 0 and k["value"] > 0 and k["steps"] == 2
        assert k["roofline"]["bound"] == "hbm" and k["roofline"]["peak_measured_copy"] > 0 and 0 < k["roofline"]["frac"] < 1
        assert k["roofline"]["algorithmic_bytes_per_launch"] == (1 << 29) + round(k["compression_ratio"] * (1 << 29)) - 20


def test_bench_strong_scaling_splits_one_stream(tmp_path):
    """configs[3] in miniature: a fixed total split into contiguous packet ranges, one per rank."""
    line, d = _self_launched_bench(["--scaling", "strong", "--total-gib", "0.5", "--gib-per-gpu", "0.25"], tmp_path)
    assert line["scaling"] == "strong" and line["n_ranks_seen"] == 2
    assert d["n_gpus"] == 2 and d["n_ranks_seen"] == 2 and d["scaling"] == "strong"
    assert d["roundtrip_equal"] is True and d["oracle_prefix_match"] is True
    assert abs(d["value"] * d["ms_per_step"] * 1e-3 - 0.5 * 1.073741824) < 1e-6           # the total, not per rank
    assert abs(d["compression_ratio"] - 1.00804) < 1e-3


def _shard_at_true_offset(H, oracle, kind, seed, offset, n, ratio_band):
    """One rank's shard of a BASELINE config, generated on the device at its true offset in the global
    stream: encode, compact, decode both ways; the whole shard round-trips, and three windows of 64 packets
    (first, middle, last) are byte-equal to what the oracle makes of the same bytes generated on the host."""
    d_in = H.generate(kind, seed, n, offset=offset)
    npk = H.packet_count(n)
    d_slots = H.encode(d_in)
    d_stream, d_off = H.compact(d_slots, npk)
    total = int(d_off[-1].item())
    assert ratio_band[0] < total / n < ratio_band[1]
    d_back = H.decode(d_slots, npk)
    assert torch.equal(d_back[:n], d_in)
    del d_back, d_slots
    d_back2 = H.decode_stream(d_stream, d_off, npk)
    assert torch.equal(d_back2[:n], d_in)
    del d_back2
    assert H.status() == 0
    for p0 in (0, npk // 2 - 32, npk - 64):
        host = synth.generate(kind, seed, 64 * 8192, offset=offset + p0 * 8192)
        assert np.array_equal(d_in[p0 * 8192:(p0 + 64) * 8192].cpu().numpy(), host), (kind, p0)      # same bytes on both sides
        want = oracle.encode_stream(host)
        got = d_stream[int(d_off[p0].item()):int(d_off[p0 + 64].item())].cpu().numpy()
        assert got.size == want.size and np.array_equal(got, want), (kind, p0)


def test_config3_last_rank_of_uniform_8gib_over_8_gpus(H, oracle):
    """BASELINE.json configs[3]: uniform(42) 8 GiB sharded over 8 GPUs -- rank 7 codes bytes [7 GiB, 8 GiB)."""
    from gpuar_amd import sharding
    off, n = sharding.plan_shards(8 << 30, 8)[7]
    assert (off, n) == (7 << 30, 1 << 30)
    _shard_at_true_offset(H, oracle, "uniform", 42, off, n, (1.0075, 1.0085))


def test_config4_last_rank_of_zipf_64gib_over_8_gpus(H, oracle):
    """BASELINE.json configs[4]: zipf(1) 64 GiB over 8 GPUs -- rank 7 codes bytes [56 GiB, 64 GiB), 8 GiB per GPU."""
    from gpuar_amd import sharding
    off, n = sharding.weak_shard(8 << 30, 7)
    assert (off, n) == (56 << 30, 8 << 30)
    _shard_at_true_offset(H, oracle, "zipf", 1, off, n, (0.78, 0.80))
