"""ctypes front-end for the CPU oracle -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  Two libraries can sit behind it:

  * ``port``      -- oracle/libgpuar_oracle.so, the plain-C restatement in
                     oracle/arcodec_oracle.c (always available; built by make).
  * ``reference`` -- oracle/_ref/libgpuar_ref.so, the reference's unmodified
                     arCompress/arDecompress (oracle/build_ref.sh; prebuilt file
                     travels to the GPU box, /root/reference does not).
"""
from __future__ import annotations

import ctypes as C
import hashlib
import json
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
PACKET_IN = 8192
PACKET_SLOT = 8704
HEADER_LEN = 20

REF_LIB_PATH = os.path.join(HERE, "_ref", "libgpuar_ref.so")
GOLDEN_JSON = os.path.join(os.path.dirname(HERE), "tests", "golden", "ref_vectors.json")

_u8p = C.POINTER(C.c_uint8)


class CheckerMismatch(RuntimeError):
    """oracle/_ref/libgpuar_ref.so is not the file the golden vectors were made with."""


def file_sha256(path: str) -> str:
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for block in iter(lambda: f.read(1 << 20), b""):
            h.update(block)
    return h.hexdigest()


def pinned_checker_sha256():
    """sha256 of oracle/_ref/libgpuar_ref.so as recorded by tests/golden/make_golden.py, or None if unrecorded."""
    try:
        with open(GOLDEN_JSON) as f:
            return json.load(f).get("checker", {}).get("sha256")
    except (OSError, ValueError):
        return None


def golden_cases():
    with open(GOLDEN_JSON) as f:
        return json.load(f)["cases"]


def golden_case_input(c) -> np.ndarray:
    """The input bytes of one case of tests/golden/ref_vectors.json (generator parameters, or a file next to it)."""
    from gpuar_amd import synth          # the integer-only generators of SURVEY.md section 8(d) (data, not codec)
    k = c["kind"]
    if k in synth.KINDS:
        return synth.generate(k, c["seed"], c["n"])
    if k == "const":
        return np.full(c["n"], c["byte"], dtype=np.uint8)
    if k == "ramp":
        return (np.arange(c["n"]) % 256).astype(np.uint8)
    if k == "tile":
        t = np.frombuffer(bytes.fromhex(c["tile_hex"]), dtype=np.uint8)
        return np.tile(t, c["n"] // t.size)
    if k == "file":
        return np.fromfile(os.path.join(os.path.dirname(GOLDEN_JSON), c["file"]), dtype=np.uint8)
    raise KeyError(k)


def replay_golden(codec, cases=None):
    """Runs every golden vector through `codec`; returns the names of the cases whose stream differs (empty: all equal).
    The vectors were produced by the pinned checker from the reference's sources, so a library that reproduces all of
    them -- stream md5, length and every packet's clen -- codes like the pinned one."""
    wrong = []
    for c in (golden_cases() if cases is None else cases):
        stream = codec.encode_stream(golden_case_input(c))
        ok = stream.size == c["stream_len"] and hashlib.md5(stream.tobytes()).hexdigest() == c["stream_md5"]
        if ok and "clens" in c:
            off, clens = 0, []
            while off < stream.size:
                clens.append(int(stream[off]) | (int(stream[off + 1]) << 8))
                off += clens[-1]
            ok = clens == c["clens"]
        if not ok:
            wrong.append(c["name"])
    return wrong


def build(force: bool = False) -> None:
    """Compile the C restatement (gcc) and, when /root/reference exists, oracle/_ref."""
    so = os.path.join(HERE, "libgpuar_oracle.so")
    src = os.path.join(HERE, "arcodec_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O3", "-Wall", "-shared", "-fPIC", "-o", so, src])
    ref_so = os.path.join(HERE, "_ref", "libgpuar_ref.so")
    driver = os.path.join(HERE, "ref_driver.cpp")
    if os.path.isdir("/root/reference/src") and (force or not os.path.exists(ref_so)
                                                 or os.path.getmtime(ref_so) < os.path.getmtime(driver)):
        subprocess.check_call(["bash", os.path.join(HERE, "build_ref.sh")])


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(_u8p)


class _Codec:
    kind = "?"

    def __init__(self, lib, prefix):
        self._lib = lib
        p = prefix
        self._enc_pkt = getattr(lib, p + "encode_packet")
        self._enc_pkt.restype = C.c_size_t
        self._enc_pkt.argtypes = [_u8p, C.c_uint16, _u8p]
        self._enc_stream = getattr(lib, p + "encode_stream")
        self._enc_stream.restype = C.c_size_t
        self._enc_stream.argtypes = [_u8p, C.c_size_t, _u8p]

    # -- packets -----------------------------------------------------------
    def encode_packet(self, data) -> bytes:
        a = np.frombuffer(bytes(data), dtype=np.uint8) if not isinstance(data, np.ndarray) else np.ascontiguousarray(data)
        assert a.size <= PACKET_IN
        src = np.zeros(max(a.size, 1) + 32, dtype=np.uint8)
        src[:a.size] = a
        out = np.zeros(2 * PACKET_IN + 64, dtype=np.uint8)
        n = self._enc_pkt(_ptr(src), a.size, _ptr(out))
        return out[:n].tobytes()

    # -- streams (packets back to back, no 20-byte header) -----------------
    def encode_stream(self, data) -> np.ndarray:
        a = np.ascontiguousarray(np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else data)
        npk = (a.size + PACKET_IN - 1) // PACKET_IN
        out = np.empty(npk * PACKET_SLOT + 64, dtype=np.uint8)
        src = np.zeros(a.size + 32, dtype=np.uint8)
        src[:a.size] = a
        n = self._enc_stream(_ptr(src), a.size, _ptr(out))
        return out[:n].copy()

    # -- the same with the packets fanned out over `threads` host threads (bench.py's all-cores CPU row).
    #    Generic version: one contiguous packet range per Python thread, each ONE long C call (ctypes releases the GIL).
    def encode_stream_mt(self, data, threads: int) -> np.ndarray:
        from concurrent.futures import ThreadPoolExecutor
        a = np.ascontiguousarray(data)
        npk = (a.size + PACKET_IN - 1) // PACKET_IN
        per = (npk + threads - 1) // threads * PACKET_IN
        parts = [a[t * per:(t + 1) * per] for t in range(threads) if t * per < a.size]
        with ThreadPoolExecutor(len(parts)) as pool:
            return np.concatenate(list(pool.map(self.encode_stream, parts)))

    def decode_stream_mt(self, stream, n_out: int, threads: int) -> np.ndarray:
        return self.decode_stream(stream, n_out)              # (the port has no threaded decoder: single thread)


class PortOracle(_Codec):
    kind = "port"

    def __init__(self):
        build()
        lib = C.CDLL(os.path.join(HERE, "libgpuar_oracle.so"))
        super().__init__(lib, "oracle_")
        lib.oracle_decode_packet.restype = C.c_size_t
        lib.oracle_decode_packet.argtypes = [_u8p, C.c_size_t, _u8p, C.c_size_t]
        lib.oracle_decode_stream.restype = C.c_size_t
        lib.oracle_decode_stream.argtypes = [_u8p, C.c_size_t, _u8p, C.c_size_t]
        lib.oracle_encode_slots.restype = C.c_size_t
        lib.oracle_encode_slots.argtypes = [_u8p, C.c_size_t, _u8p]
        lib.oracle_decode_slots.restype = None
        lib.oracle_decode_slots.argtypes = [_u8p, C.c_size_t, _u8p]

    def decode_packet(self, pkt: bytes) -> bytes:
        a = np.frombuffer(pkt, dtype=np.uint8).copy()
        out = np.zeros(PACKET_IN, dtype=np.uint8)
        n = self._lib.oracle_decode_packet(_ptr(a), a.size, _ptr(out), out.size)
        return out[:n].tobytes()

    def decode_stream(self, stream, n_out: int) -> np.ndarray:
        a = np.ascontiguousarray(stream)
        out = np.zeros(n_out + PACKET_IN, dtype=np.uint8)
        n = self._lib.oracle_decode_stream(_ptr(a), a.size, _ptr(out), out.size)
        if n == C.c_size_t(-1).value:
            raise ValueError("malformed packet stream")
        return out[:n].copy()

    def encode_slots(self, data: np.ndarray):
        """Fixed 8704-byte slots as garCompress lays them out; returns (slots, sum clen)."""
        a = np.ascontiguousarray(data)
        npk = (a.size + PACKET_IN - 1) // PACKET_IN
        slots = np.zeros(npk * PACKET_SLOT, dtype=np.uint8)
        total = self._lib.oracle_encode_slots(_ptr(a), a.size, _ptr(slots))
        if total == C.c_size_t(-1).value:
            raise OverflowError("a packet outgrew its 8704-byte slot")
        return slots, total

    def decode_slots(self, slots: np.ndarray, n_packets: int) -> np.ndarray:
        a = np.ascontiguousarray(slots)
        out = np.zeros(n_packets * PACKET_IN, dtype=np.uint8)
        self._lib.oracle_decode_slots(_ptr(a), n_packets, _ptr(out))
        return out


class ReferenceOracle(_Codec):
    kind = "reference"

    def __init__(self, path: str = None, expect_sha256: str = "pinned", allow_replay: bool = None):
        """`expect_sha256`: "pinned" = the hash tests/golden/ref_vectors.json records (the default: a checker that is not
        the pinned file is not trusted on its name); a hex string = that hash; None = no check (make_golden.py only, while
        it produces the file's pin).
        `allow_replay`: what happens to a file with ANOTHER hash.  Where the reference's sources are present (the build
        container: /root/reference) the library can legitimately have been rebuilt -- another gcc or binutils gives other
        bytes -- so it is put to the test instead: every golden vector is replayed through it, and it is accepted
        only if it reproduces all of them (`validated_by` says which way a checker was admitted).  Where the sources are
        absent (the GPU box, which can only have the pushed binary) the pin is the binary's hash and nothing else.
        None = decide by the presence of /root/reference/src."""
        build()
        path = path or REF_LIB_PATH
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        want = pinned_checker_sha256() if expect_sha256 == "pinned" else expect_sha256
        if expect_sha256 == "pinned" and want is None:
            raise CheckerMismatch(f"{GOLDEN_JSON} records no checker sha256: run tests/golden/make_golden.py")
        if allow_replay is None:
            allow_replay = os.path.isdir("/root/reference/src")
        needs_replay = False
        self.validated_by = "no check asked for" if want is None else "sha256 pin"
        if want is not None:
            have = file_sha256(path)
            if have != want:
                if not allow_replay:
                    raise CheckerMismatch(f"{path}: sha256 {have} is not the pinned {want} -- this is not the library the golden "
                                          "vectors were produced with (rebuilt from other sources, damaged or replaced); rebuild "
                                          "it with oracle/build_ref.sh or re-pin with tests/golden/make_golden.py")
                needs_replay = True
        lib = C.CDLL(path)
        super().__init__(lib, "ref_")
        if needs_replay:
            wrong = replay_golden(self)
            if wrong:
                raise CheckerMismatch(f"{path}: not the pinned file (sha256 {have}) and it does not reproduce the golden vectors "
                                      f"{wrong[:5]}: not the reference's codec")
            self.validated_by = f"replay of {len(golden_cases())} golden vectors (sha256 {have[:16]}... is not the pinned file's)"
        lib.ref_decode_packet.restype = C.c_size_t
        lib.ref_decode_packet.argtypes = [_u8p, C.c_size_t, _u8p]
        lib.ref_decode_stream.restype = C.c_size_t
        lib.ref_decode_stream.argtypes = [_u8p, C.c_size_t, _u8p]
        self._have_mt = hasattr(lib, "ref_encode_stream_mt")
        if self._have_mt:
            lib.ref_encode_stream_mt.restype = C.c_size_t
            lib.ref_encode_stream_mt.argtypes = [_u8p, C.c_size_t, _u8p, C.c_uint]
            lib.ref_decode_stream_mt.restype = C.c_size_t
            lib.ref_decode_stream_mt.argtypes = [_u8p, C.c_size_t, _u8p, C.c_uint]

    def encode_stream_mt(self, data, threads: int) -> np.ndarray:
        """Native threads (oracle/ref_driver.cpp): one contiguous packet range per thread."""
        if not self._have_mt:
            return super().encode_stream_mt(data, threads)
        a = np.ascontiguousarray(data)
        npk = (a.size + PACKET_IN - 1) // PACKET_IN
        out = np.empty(npk * PACKET_SLOT + 64, dtype=np.uint8)
        src = np.zeros(a.size + 32, dtype=np.uint8)
        src[:a.size] = a
        n = self._lib.ref_encode_stream_mt(_ptr(src), a.size, _ptr(out), threads)
        return out[:n]

    def decode_stream_mt(self, stream, n_out: int, threads: int) -> np.ndarray:
        if not self._have_mt:
            return self.decode_stream(stream, n_out)
        a = np.ascontiguousarray(stream)
        out = np.zeros(n_out + PACKET_IN + 64, dtype=np.uint8)
        n = self._lib.ref_decode_stream_mt(_ptr(a), a.size, _ptr(out), threads)
        if n == C.c_size_t(-1).value:
            raise ValueError("malformed packet stream")
        return out[:n]

    def decode_packet(self, pkt: bytes) -> bytes:
        a = np.frombuffer(pkt, dtype=np.uint8).copy()
        out = np.zeros(65536 + 64, dtype=np.uint8)
        n = self._lib.ref_decode_packet(_ptr(a), a.size, _ptr(out))
        return out[:n].tobytes()

    # -- caller-owned model state (what the reference's own callers pass) ----
    def _model_api(self):
        lib = self._lib
        if not hasattr(lib, "_model_ready"):
            u16p = C.POINTER(C.c_uint16)
            lib.ref_model_init.restype = None
            lib.ref_model_init.argtypes = [u16p, u16p]
            lib.ref_encode_packet_model.restype = C.c_size_t
            lib.ref_encode_packet_model.argtypes = [_u8p, C.c_uint16, _u8p, u16p, u16p]
            lib.ref_decode_packet_model.restype = C.c_size_t
            lib.ref_decode_packet_model.argtypes = [_u8p, C.c_size_t, _u8p, u16p, u16p]
            lib._model_ready = True
        return lib

    def model_init(self):
        """(ranges[257] u16 Fenwick array, total) after initializeAdaptiveProbabilityRangeList."""
        lib = self._model_api()
        ranges = np.zeros(257, dtype=np.uint16)
        total = C.c_uint16(0)
        lib.ref_model_init(ranges.ctypes.data_as(C.POINTER(C.c_uint16)), C.byref(total))
        return ranges, int(total.value)

    def encode_packet_model(self, data, ranges: np.ndarray, total: int):
        """arCompress from the given model state; returns (packet, ranges', total')."""
        lib = self._model_api()
        a = np.ascontiguousarray(np.frombuffer(bytes(data), dtype=np.uint8))
        src = np.zeros(a.size + 32, dtype=np.uint8)
        src[:a.size] = a
        out = np.zeros(4 * PACKET_IN, dtype=np.uint8)
        r = np.array(ranges, dtype=np.uint16)
        t = C.c_uint16(total)
        n = lib.ref_encode_packet_model(_ptr(src), a.size, _ptr(out), r.ctypes.data_as(C.POINTER(C.c_uint16)), C.byref(t))
        return out[:n].tobytes(), r, int(t.value)

    def decode_packet_model(self, pkt: bytes, ranges: np.ndarray, total: int):
        lib = self._model_api()
        a = np.frombuffer(pkt, dtype=np.uint8).copy()
        out = np.zeros(65536 + 64, dtype=np.uint8)
        r = np.array(ranges, dtype=np.uint16)
        t = C.c_uint16(total)
        n = lib.ref_decode_packet_model(_ptr(a), a.size, _ptr(out), r.ctypes.data_as(C.POINTER(C.c_uint16)), C.byref(t))
        return out[:n].tobytes(), r, int(t.value)

    def decode_stream(self, stream, n_out: int) -> np.ndarray:
        a = np.ascontiguousarray(stream)
        out = np.zeros(n_out + PACKET_IN + 64, dtype=np.uint8)
        n = self._lib.ref_decode_stream(_ptr(a), a.size, _ptr(out))
        if n == C.c_size_t(-1).value:
            raise ValueError("malformed packet stream")
        return out[:n].copy()


def have_reference() -> bool:
    return os.path.exists(REF_LIB_PATH)


def best():
    """The strongest checker available: the reference build if present, else the port."""
    return ReferenceOracle() if have_reference() else PortOracle()


def expected_kind() -> str:
    """Which checker a box is EXPECTED to have.  tests/golden/ref_vectors.json is committed and records the sha256 of the
    oracle/_ref/libgpuar_ref.so that made it (oracle/build_ref.sh: the reference's own codec); that binary travels with
    every push to a GPU box.  So wherever the pin exists the checker must be "reference" -- a box without the binary would
    otherwise turn "bit-exact vs the reference" into "bit-exact vs our port" without a red test.  GPUAR_ALLOW_PORT_CHECKER=1
    says out loud that the port is acceptable (a checkout that never had /root/reference)."""
    if os.environ.get("GPUAR_ALLOW_PORT_CHECKER") == "1" or pinned_checker_sha256() is None:
        return "port"
    return "reference"


def require_best():
    """best(), or an error when it is weaker than expected_kind() says this box must have."""
    want = expected_kind()
    if want == "reference" and not have_reference():
        raise CheckerMismatch(f"{REF_LIB_PATH} is missing but tests/golden/ref_vectors.json pins it: the parity claims of this box would "
                              "be against the port, not the reference's codec.  Build it where /root/reference exists (oracle/build_ref.sh) "
                              "and push it, or set GPUAR_ALLOW_PORT_CHECKER=1 to accept the port knowingly.")
    codec = best()
    if want == "reference" and codec.kind != "reference":
        raise CheckerMismatch(f"checker is {codec.kind!r}, expected 'reference'")
    return codec


# -- container (20-byte header) as the reference writes it -------------------
# src/file_header.hpp:19-36,61-72; zeros where the reference leaves stack garbage.
def gip_header(n_uncompressed: int, n_stream: int) -> bytes:
    h = bytearray(HEADER_LEN)
    h[0:3] = bytes([0, 1, 0])
    h[4:12] = int(n_uncompressed).to_bytes(8, "little")
    h[12:20] = int(HEADER_LEN + n_stream).to_bytes(8, "little")
    return bytes(h)
