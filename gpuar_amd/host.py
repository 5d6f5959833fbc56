"""ctypes binding of the host-callable packet codec (include/gpuar_host.h).

Mirrors what the reference's CPU compressor does around its own
arCompress/arDecompress (src/cpu_compressor.cpp:59-60,159-160): the caller owns
a model (257-entry Fenwick array + running total), initialises it, and codes
one packet per call.  Loads gpuar_amd/lib/libgpuar_host.so (no HIP runtime
needed) and fails loudly if it is missing.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "lib", "libgpuar_host.so")
EXPORTS = ["initializeAdaptiveProbabilityRangeList", "arCompress", "arDecompress"]
MODEL_ENTRIES = 257
PACKET_BYTES = 8192
SLOT_BYTES = 8704

_u8p = C.POINTER(C.c_uint8)
_u16p = C.POINTER(C.c_uint16)
_lib = None


class HostCodecError(RuntimeError):
    pass


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HostCodecError(f"{LIB_PATH} not built: run `python -c 'import __graft_entry__ as g; g.build()'`")
        lib = C.CDLL(LIB_PATH)
        lib.initializeAdaptiveProbabilityRangeList.restype = None
        lib.initializeAdaptiveProbabilityRangeList.argtypes = [_u16p, _u16p]
        lib.arCompress.restype = C.c_uint16
        lib.arCompress.argtypes = [_u8p, C.c_uint16, _u8p, _u16p, _u16p]
        lib.arDecompress.restype = C.c_uint16
        lib.arDecompress.argtypes = [_u8p, C.c_uint16, _u8p, _u16p, _u16p]
        _lib = lib
    return _lib


class Model:
    """AdaptiveProbabilityRange + cumulativeProb, owned by the caller (src/gpuar.h:42-48)."""

    def __init__(self, ranges=None, total=None):
        self.ranges = np.zeros(MODEL_ENTRIES, dtype=np.uint16)
        self._total = C.c_uint16(0)
        if ranges is None:
            self.reset()
        else:
            self.ranges[:] = np.asarray(ranges, dtype=np.uint16)
            self._total.value = int(total)

    def reset(self):
        load().initializeAdaptiveProbabilityRangeList(self.ranges.ctypes.data_as(_u16p), C.byref(self._total))

    @property
    def total(self) -> int:
        return int(self._total.value)


def encode_packet(data, model: Model | None = None, capacity: int = 4 * PACKET_BYTES) -> bytes:
    """arCompress: one packet (u16 clen, u16 ulen, bitstream) from at most 8192 bytes."""
    a = np.ascontiguousarray(np.frombuffer(bytes(data), dtype=np.uint8))
    if a.size > PACKET_BYTES:
        raise ValueError("a packet holds at most 8192 bytes")
    m = model or Model()
    src = a if a.size else np.zeros(1, dtype=np.uint8)
    out = np.zeros(capacity, dtype=np.uint8)
    n = load().arCompress(src.ctypes.data_as(_u8p), a.size, out.ctypes.data_as(_u8p),
                          m.ranges.ctypes.data_as(_u16p), C.byref(m._total))
    return out[:n].tobytes()


def decode_packet(packet: bytes, model: Model | None = None) -> bytes:
    """arDecompress: the bytes one packet codes."""
    a = np.frombuffer(bytes(packet), dtype=np.uint8).copy()
    if a.size < 4:
        raise ValueError("a packet starts with a 4-byte header")
    m = model or Model()
    out = np.zeros(65536, dtype=np.uint8)
    n = load().arDecompress(a.ctypes.data_as(_u8p), min(a.size, 65535), out.ctypes.data_as(_u8p),
                            m.ranges.ctypes.data_as(_u16p), C.byref(m._total))
    return out[:n].tobytes()
