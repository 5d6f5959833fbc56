"""ctypes bindings to libgpuar_hip.so (include/gpuar_hip.h).

There is no CPU fallback here: if the HIP library is missing or a call fails,
these functions raise.  Device buffers are torch uint8 CUDA tensors (torch is
used for allocation and stream handles only).
"""
from __future__ import annotations

import ctypes as C
import os

PACKET = 8192
SLOT = 8704
HEADER_LEN = 20
STATUS_SLOT_OVERFLOW = 0x1
STATUS_BAD_PACKET = 0x2
KIND_ID = {"uniform": 0, "zipf": 1, "text": 2}

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libgpuar_hip.so")
EXPORTS = [
    "initConstantRange", "garCompressExecutor", "garDecompressExecutor",
    "gpuar_hip_packet_count", "gpuar_hip_encode", "gpuar_hip_encode_mode", "gpuar_hip_decode", "gpuar_hip_compact",
    "gpuar_hip_decode_stream", "gpuar_hip_status", "gpuar_hip_last_error", "gpuar_hip_error_string",
    "gpuar_hip_version", "gpuar_hip_abi_version", "gpuar_hip_generate", "gpuar_hip_copy", "gpuar_hip_clock_samples",
]
CLOCK_SLOTS = 256                    # GPUAR_CLOCK_SLOTS
ABI_VERSION = 2                      # GPUAR_HIP_ABI_VERSION of the header these bindings were written against
MODE_ID = {"auto": 0, "throughput": 1, "latency": 2}     # GPUAR_MODE_*

_lib = None


class GpuarError(RuntimeError):
    pass


def load() -> C.CDLL:
    """Loads the HIP library; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise GpuarError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                         "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    lib = C.CDLL(LIB_PATH)
    # a library built from older sources lacks newer exports: say "rebuild" instead of dying with an AttributeError half-way
    # through the declarations below
    missing = [name for name in EXPORTS if not hasattr(lib, name)]
    if missing:
        raise GpuarError(f"{LIB_PATH} lacks {missing}: it was built from older sources -- rebuild the library")
    vp, sz, u32 = C.c_void_p, C.c_size_t, C.c_uint32
    lib.initConstantRange.restype = None
    lib.initConstantRange.argtypes = []
    lib.garCompressExecutor.restype = None
    lib.garCompressExecutor.argtypes = [vp, sz, vp, u32]
    lib.garDecompressExecutor.restype = None
    lib.garDecompressExecutor.argtypes = [vp, sz, vp, u32]
    lib.gpuar_hip_packet_count.restype = sz
    lib.gpuar_hip_packet_count.argtypes = [sz]
    lib.gpuar_hip_encode.restype = C.c_int
    lib.gpuar_hip_encode.argtypes = [vp, sz, vp, vp, vp]
    lib.gpuar_hip_encode_mode.restype = C.c_int
    lib.gpuar_hip_encode_mode.argtypes = [vp, sz, vp, vp, vp, C.c_int]
    lib.gpuar_hip_decode.restype = C.c_int
    lib.gpuar_hip_decode.argtypes = [vp, sz, vp, vp, vp]
    lib.gpuar_hip_compact.restype = C.c_int
    lib.gpuar_hip_compact.argtypes = [vp, sz, vp, vp, vp]
    lib.gpuar_hip_decode_stream.restype = C.c_int
    lib.gpuar_hip_decode_stream.argtypes = [vp, vp, sz, vp, vp, vp]
    lib.gpuar_hip_status.restype = C.c_int
    lib.gpuar_hip_status.argtypes = [C.POINTER(C.c_uint32)]
    lib.gpuar_hip_last_error.restype = C.c_int
    lib.gpuar_hip_last_error.argtypes = []
    lib.gpuar_hip_error_string.restype = C.c_char_p
    lib.gpuar_hip_error_string.argtypes = [C.c_int]
    lib.gpuar_hip_version.restype = C.c_char_p
    lib.gpuar_hip_version.argtypes = []
    lib.gpuar_hip_abi_version.restype = C.c_int
    lib.gpuar_hip_abi_version.argtypes = []
    lib.gpuar_hip_generate.restype = C.c_int
    lib.gpuar_hip_generate.argtypes = [C.c_int, C.c_uint64, C.c_uint64, sz, vp, vp]
    lib.gpuar_hip_copy.restype = C.c_int
    lib.gpuar_hip_copy.argtypes = [vp, vp, sz, vp]
    lib.gpuar_hip_clock_samples.restype = C.c_int
    lib.gpuar_hip_clock_samples.argtypes = [C.c_int, C.POINTER(C.c_uint64), C.c_int]
    if lib.gpuar_hip_abi_version() != ABI_VERSION:
        raise GpuarError(f"{LIB_PATH} speaks ABI {lib.gpuar_hip_abi_version()}, these bindings {ABI_VERSION}: rebuild the library")
    # a build with timing switches (tools/exp_build.sh -DGPUAR_EXP_...) decodes / encodes garbage by design: it is loaded only from
    # where the timing tools put it (LIB_PATH pointed there by --lib), never as the product library
    if b"EXPERIMENT" in lib.gpuar_hip_version() and os.path.abspath(LIB_PATH) == os.path.join(_HERE, "lib", "libgpuar_hip.so"):
        raise GpuarError(f"{LIB_PATH} is an experiment build ({lib.gpuar_hip_version().decode()}): rebuild the product library with make")
    _lib = lib
    return lib


def _mode_id(mode, env_var) -> int:
    """GPUAR_MODE_* for a call: the caller's `mode` ("auto" | "throughput" | "latency"), else the environment variable
    that tests and tools use to pin the encode kernel (GPUAR_ENCODE_MODE; there is no decode mode) -- read HERE, in the
    Python shim, never inside the library --, else auto."""
    name = mode if mode is not None else os.environ.get(env_var, "auto")
    if name not in MODE_ID:
        raise GpuarError(f"unknown kernel mode {name!r} (auto, throughput, latency)")
    return MODE_ID[name]


def _check(code: int, what: str) -> None:
    if code != 0:
        raise GpuarError(f"{what}: {load().gpuar_hip_error_string(code).decode()} (code {code})")


def _stream_handle(stream=None) -> int:
    import torch
    s = stream if stream is not None else torch.cuda.current_stream()
    return int(s.cuda_stream)


def packet_count(n_bytes: int) -> int:
    return (n_bytes + PACKET - 1) // PACKET


def _require_cuda_u8(t, name):
    import torch
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.uint8 and t.is_contiguous()):
        raise GpuarError(f"{name} must be a contiguous uint8 CUDA tensor")


def _status_ptr(d_status):
    """Device address of a caller-owned status word (a 1-element int32/uint32 CUDA tensor), or None."""
    if d_status is None:
        return None
    import torch
    if not (isinstance(d_status, torch.Tensor) and d_status.is_cuda and d_status.element_size() == 4 and d_status.numel() >= 1):
        raise GpuarError("d_status must be a CUDA tensor of one 32-bit word")
    return d_status.data_ptr()


def encode(d_in, d_slots=None, stream=None, d_status=None, mode=None):
    """Encode the bytes of `d_in` into 8704-byte packet slots (garCompress layout).  `d_status`: this launch's
    own status word (a zeroed 1-element int32 CUDA tensor); None = the device's fallback word (status()).
    `mode`: "auto" | "throughput" | "latency" (gpuar_hip_encode_mode); None = $GPUAR_ENCODE_MODE, else auto."""
    import torch
    _require_cuda_u8(d_in, "d_in")
    n = d_in.numel()
    npk = packet_count(n)
    if d_slots is None:
        d_slots = torch.empty(max(npk, 1) * SLOT, dtype=torch.uint8, device=d_in.device)
    _require_cuda_u8(d_slots, "d_slots")
    if d_slots.numel() < npk * SLOT:
        raise GpuarError("d_slots too small")
    _check(load().gpuar_hip_encode_mode(d_in.data_ptr(), n, d_slots.data_ptr(), _status_ptr(d_status), _stream_handle(stream),
                                        _mode_id(mode, "GPUAR_ENCODE_MODE")), "gpuar_hip_encode_mode")
    return d_slots


def decode(d_slots, n_packets: int, d_out=None, stream=None, d_status=None):
    """Decode `n_packets` slots into n_packets*8192 output bytes."""
    import torch
    _require_cuda_u8(d_slots, "d_slots")
    if d_slots.numel() < n_packets * SLOT:
        raise GpuarError("d_slots too small")
    if d_out is None:
        d_out = torch.empty(max(n_packets, 1) * PACKET, dtype=torch.uint8, device=d_slots.device)
    _require_cuda_u8(d_out, "d_out")
    if d_out.numel() < n_packets * PACKET:
        raise GpuarError("d_out too small")
    _check(load().gpuar_hip_decode(d_slots.data_ptr(), n_packets, d_out.data_ptr(), _status_ptr(d_status), _stream_handle(stream)), "gpuar_hip_decode")
    return d_out


def compact(d_slots, n_packets: int, d_stream=None, d_offsets=None, stream=None):
    """Slots -> (back-to-back packet stream buffer, u64 offsets[n_packets+1]) on the device."""
    import torch
    _require_cuda_u8(d_slots, "d_slots")
    if d_stream is None:
        d_stream = torch.empty(max(n_packets, 1) * SLOT, dtype=torch.uint8, device=d_slots.device)
    if d_offsets is None:
        d_offsets = torch.empty(n_packets + 1, dtype=torch.int64, device=d_slots.device)
    _check(load().gpuar_hip_compact(d_slots.data_ptr(), n_packets, d_stream.data_ptr(), d_offsets.data_ptr(),
                                    _stream_handle(stream)), "gpuar_hip_compact")
    return d_stream, d_offsets


def decode_stream(d_stream, d_offsets, n_packets: int, d_out=None, stream=None, d_status=None):
    import torch
    _require_cuda_u8(d_stream, "d_stream")
    if d_out is None:
        d_out = torch.empty(max(n_packets, 1) * PACKET, dtype=torch.uint8, device=d_stream.device)
    _check(load().gpuar_hip_decode_stream(d_stream.data_ptr(), d_offsets.data_ptr(), n_packets, d_out.data_ptr(),
                                          _status_ptr(d_status), _stream_handle(stream)), "gpuar_hip_decode_stream")
    return d_out


def status() -> int:
    """Reads and clears the device's FALLBACK status word -- what launches without a d_status of their own
    reported (synchronises the device)."""
    flags = C.c_uint32(0)
    _check(load().gpuar_hip_status(C.byref(flags)), "gpuar_hip_status")
    return int(flags.value)


def generate(kind: str, seed: int, n: int, offset: int = 0, device="cuda", out=None, stream=None):
    """Synthetic stream bytes [offset, offset+n) generated on the device (offset % 8 == 0)."""
    import torch
    if out is None:
        out = torch.empty(n, dtype=torch.uint8, device=device)
    if kind == "zeros":
        out[:n].zero_()
        return out
    _check(load().gpuar_hip_generate(KIND_ID[kind], seed & 0xFFFFFFFFFFFFFFFF, offset, n, out.data_ptr(),
                                     _stream_handle(stream)), "gpuar_hip_generate")
    return out


def device_copy(d_src, d_dst, n_bytes: int = None, stream=None):
    """Plain 16-byte-per-lane device copy (gpuar_hip_copy): the measured HBM roof of bench.py."""
    _require_cuda_u8(d_src, "d_src")
    _require_cuda_u8(d_dst, "d_dst")
    n = d_src.numel() if n_bytes is None else n_bytes
    if d_dst.numel() < n or d_src.numel() < n:
        raise GpuarError("copy: buffer too small")
    _check(load().gpuar_hip_copy(d_src.data_ptr(), d_dst.data_ptr(), n, _stream_handle(stream)), "gpuar_hip_copy")
    return d_dst


def gip_header(n_uncompressed: int, n_stream: int) -> bytes:
    """20-byte container header (src/file_header.hpp:19-36,61-72): version 0.1.0,
    sizes little-endian in the 8-byte slots (low 4 bytes = what the reference
    writes; it leaves the other bytes uninitialised, we write the high half /
    zeros so files over 4 GiB stay representable)."""
    h = bytearray(HEADER_LEN)
    h[0:3] = bytes([0, 1, 0])
    h[4:12] = int(n_uncompressed).to_bytes(8, "little")
    h[12:20] = int(HEADER_LEN + n_stream).to_bytes(8, "little")
    return bytes(h)


def shader_clock_mhz(which: str, reset: bool = True):
    """The shader clock while the last launches of the throughput encode kernel (which = "encode") or of a decode kernel
    ("decode") ran, from the samples their workgroups left (gpuar_hip_clock_samples): (MHz, number of samples), or
    (None, 0) when no sampled workgroup has run to its end since the last reset.  Synchronises the device."""
    buf = (C.c_uint64 * (4 * CLOCK_SLOTS))()
    _check(load().gpuar_hip_clock_samples({"encode": 0, "decode": 1}[which], buf, 1 if reset else 0), "gpuar_hip_clock_samples")
    shader = real = n = 0
    for i in range(CLOCK_SLOTS):
        s0, r0, s1, r1 = buf[4 * i:4 * i + 4]
        if r0 and r1 > r0 and s1 > s0:              # a workgroup that started and ended since the reset
            shader, real, n = shader + (s1 - s0), real + (r1 - r0), n + 1
    return (shader / real * 100.0, n) if real else (None, 0)
