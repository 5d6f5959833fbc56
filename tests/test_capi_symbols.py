"""The C-ABI library loads on a machine without a GPU and exports every symbol
include/gpuar_hip.h declares (no compute calls here)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    from gpuar_amd import hip as H
    if not os.path.exists(H.LIB_PATH):
        g.build()
    return H.load()


def declared_symbols(header="gpuar_hip.h"):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = "\n".join(l for l in text.splitlines() if not l.lstrip().startswith("#"))
    return sorted(set(re.findall(r"\b(\w+)\s*\([^;{]*\)\s*;", text)))


def test_header_symbols_are_exported(lib):
    from gpuar_amd import hip as H
    names = declared_symbols()
    assert set(names) == set(H.EXPORTS), (names, H.EXPORTS)
    for n in names:
        assert hasattr(lib, n), n


def declared_arity(header="gpuar_hip.h"):
    """name -> number of parameters, read off the header's prototypes"""
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = "\n".join(l for l in text.splitlines() if not l.lstrip().startswith("#"))
    out = {}
    for name, params in re.findall(r"\b(\w+)\s*\(([^;{]*)\)\s*;", text):
        params = params.strip()
        out[name] = 0 if params in ("", "void") else params.count(",") + 1
    return out


def test_bindings_pin_every_signature(lib):
    """A binding with a different number of arguments than the header's prototype is how a stream handle ends up
    where a status word is expected: the ctypes argtypes are checked against the header, and the
    library's ABI number against the header's."""
    from gpuar_amd import hip as H
    for name, arity in declared_arity().items():
        fn = getattr(lib, name)
        assert fn.argtypes is not None and len(fn.argtypes) == arity, (name, arity, fn.argtypes)
    header = open(os.path.join(ROOT, "include", "gpuar_hip.h")).read()
    abi = int(re.search(r"#define\s+GPUAR_HIP_ABI_VERSION\s+(\d+)", header).group(1))
    assert lib.gpuar_hip_abi_version() == abi == H.ABI_VERSION
    assert f"gpuar-hip 0.{abi} ".encode() in lib.gpuar_hip_version()


def test_library_reads_no_environment(lib):
    """The kernel choice is an argument of gpuar_hip_encode_mode; GPUAR_ENCODE_MODE is honoured by the Python shim only."""
    import subprocess
    from gpuar_amd import hip as H
    undefined = subprocess.run(["nm", "-D", "--undefined-only", H.LIB_PATH], capture_output=True, text=True, check=True).stdout
    assert "getenv" not in undefined, "libgpuar_hip.so imports getenv"
    assert lib.gpuar_hip_encode_mode(None, 8192, None, None, None, 7) == -2      # unknown mode: GPUAR_ERR_ARGUMENT, before anything else
    assert H._mode_id("latency", "GPUAR_ENCODE_MODE") == 2 and H._mode_id(None, "GPUAR_NO_SUCH_VARIABLE") == 0
    with pytest.raises(H.GpuarError):
        H._mode_id("fast", "GPUAR_ENCODE_MODE")


def test_host_codec_symbols_are_exported_by_both_libraries(lib):
    import ctypes
    from gpuar_amd import host as HC
    names = declared_symbols("gpuar_host.h")
    assert set(names) == set(HC.EXPORTS), (names, HC.EXPORTS)
    small = ctypes.CDLL(HC.LIB_PATH)
    for n in names:
        assert hasattr(small, n) and hasattr(lib, n), n


def test_host_only_calls(lib):
    assert lib.gpuar_hip_packet_count(0) == 0
    assert lib.gpuar_hip_packet_count(1) == 1
    assert lib.gpuar_hip_packet_count(8192) == 1
    assert lib.gpuar_hip_packet_count(8193) == 2
    assert b"gfx950" in lib.gpuar_hip_version()
    assert lib.gpuar_hip_error_string(0) == b"ok"
    assert b"align" in lib.gpuar_hip_error_string(-1)
    assert lib.gpuar_hip_last_error() == 0


def test_missing_library_fails_loudly(monkeypatch):
    from gpuar_amd import hip as H
    monkeypatch.setattr(H, "_lib", None)
    monkeypatch.setattr(H, "LIB_PATH", "/nonexistent/libgpuar_hip.so")
    with pytest.raises(H.GpuarError):
        H.load()


def test_header_layout_helper():
    from gpuar_amd import hip as H
    h = H.gip_header(65539, 66067)
    assert len(h) == 20 and h[0:3] == b"\x00\x01\x00"
    assert int.from_bytes(h[4:8], "little") == 65539
    assert int.from_bytes(h[12:16], "little") == 66087


def test_the_product_library_is_not_an_experiment_build():
    """The GPUAR_EXP_* timing switches take pieces out of the kernels (WRONG output by design).  Such a build only compiles with
    GPUAR_EXPERIMENT_BUILD (tools/exp_build.sh), says so in its version string, and gpuar_amd/hip.py refuses it as the product."""
    import ctypes as C
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = C.CDLL(os.path.join(root, "gpuar_amd", "lib", "libgpuar_hip.so"))
    lib.gpuar_hip_version.restype = C.c_char_p
    assert b"EXPERIMENT" not in lib.gpuar_hip_version()
    src = open(os.path.join(root, "gpuar_amd", "csrc", "gpuar_kernels.hip")).read()
    guard = src[src.index("#if (defined(GPUAR_EXP_"):src.index('#error "a GPUAR_EXP_* switch without GPUAR_EXPERIMENT_BUILD')]
    switches = set(re.findall(r"GPUAR_EXP_[A-Z0-9_]+", src)) - {"GPUAR_EXP_STORES"}
    assert switches and all(f"defined({s})" in guard for s in switches), sorted(switches)     # every switch is under the guard
    # the header the host build shares carries no wrong-output branch at all
    assert "GPUAR_EXP_" not in open(os.path.join(root, "gpuar_amd", "csrc", "lane_codec.h")).read().replace("GPUAR_EXP_* ", "")
    assert "GPUAR_EXPERIMENT_BUILD" in open(os.path.join(root, "tools", "exp_build.sh")).read()
