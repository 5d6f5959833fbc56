"""The `gpuar` CLI and the C++ Compressor/CPUCompressor classes behind it (--host mode), on CPU.

Mirrors the reference's own manual check (README.md:12-29: compress, decompress,
compare) and pins the produced .gip against the reference's outputs: header
bytes 0-2, 4-7, 12-15 and every byte from offset 20 on.
"""
import hashlib
import os
import subprocess

import numpy as np
import pytest

from gpuar_amd import synth
from test_oracle_golden import SURVEY

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "gpuar_amd", "bin", "gpuar")


@pytest.fixture(scope="module")
def cli():
    if not os.path.exists(CLI):
        import __graft_entry__ as g
        g.build()
    return CLI


def run(cli, *args):
    return subprocess.run([cli, *args], capture_output=True, text=True, timeout=600)


@pytest.mark.parametrize("s", [s for s in SURVEY["streams"] if not s.get("slow")],
                         ids=lambda s: f"{s['kind']}-{s['seed']}-{s['n']}")
def test_host_cli_matches_reference_files(cli, tmp_path, s):
    src, gip, back = tmp_path / "in.dat", tmp_path / "out.gip", tmp_path / "back.dat"
    data = synth.generate(s["kind"], s["seed"], s["n"])
    data.tofile(src)
    r = run(cli, "c", "--host", f"--in={src}", f"--out={gip}")
    assert r.returncode == 0, r.stderr
    assert "Attention: execute kernel code on host." in r.stdout and "Compression ratio" in r.stdout
    blob = open(gip, "rb").read()
    assert len(blob) == s["gip_bytes"]
    assert blob[0:3] == b"\x00\x01\x00"
    assert int.from_bytes(blob[4:8], "little") == s["n"]
    assert int.from_bytes(blob[12:16], "little") == s["gip_bytes"]
    assert hashlib.md5(blob[20:]).hexdigest() == s["stream_md5"]
    # `--in F` (space) spelling, multi-threaded decode
    r = run(cli, "d", "--host", "--threads", "0", "--in", str(gip), "--out", str(back))
    assert r.returncode == 0, r.stderr
    assert open(back, "rb").read() == data.tobytes()


def test_host_cli_empty_and_errors(cli, tmp_path):
    src, gip, back = tmp_path / "empty", tmp_path / "e.gip", tmp_path / "e.back"
    src.write_bytes(b"")
    assert run(cli, "c", "--host", f"--in={src}", f"--out={gip}").returncode == 0
    assert len(gip.read_bytes()) == 20                      # header only
    assert run(cli, "d", "--host", f"--in={gip}", f"--out={back}").returncode == 0
    assert back.read_bytes() == b""
    r = run(cli, "c", "--host", f"--in={tmp_path / 'missing'}", f"--out={gip}")
    assert r.returncode == 1 and "Can not open input file" in r.stderr
    bad = tmp_path / "bad.gip"
    bad.write_bytes(b"\x09\x09\x09" + bytes(40))
    r = run(cli, "d", "--host", f"--in={bad}", f"--out={back}")
    assert r.returncode == 1 and "Incorrect file format" in r.stderr
    r = run(cli, "c", "--host")
    assert r.returncode == 1 and "Please specify the input file name" in r.stderr
    assert "Usage: gpuar" in run(cli, "--help").stdout


def test_gpu_mode_has_no_silent_cpu_fallback(cli, tmp_path):
    """Without --host the CLI must use the GPU or fail loudly (the reference falls back silently, src/main.cpp:142)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    src = tmp_path / "in.dat"
    src.write_bytes(b"hello")
    r = run(cli, "c", f"--in={src}", f"--out={tmp_path / 'o.gip'}")
    assert r.returncode == 1 and "No HIP device" in r.stderr


def test_host_only_binary_needs_no_gpu_runtime(cli, tmp_path):
    """gpuar-host: the same CLI built without the GPU classes -- links neither
    libgpuar_hip.so nor libamdhip64 (GPU-runtime-free --host)."""
    exe = CLI + "-host"
    assert os.path.exists(exe)
    deps = subprocess.run(["ldd", exe], capture_output=True, text=True).stdout
    assert "amdhip64" not in deps and "gpuar_hip" not in deps
    src, gip, back = tmp_path / "in.dat", tmp_path / "out.gip", tmp_path / "back.dat"
    data = synth.text(3, 100000)
    data.tofile(src)
    r = subprocess.run([exe, "c", "--host", f"--in={src}", f"--out={gip}"], capture_output=True, text=True,
                       env={"PATH": os.environ.get("PATH", ""), "LD_LIBRARY_PATH": ""})
    assert r.returncode == 0, r.stderr
    blob = gip.read_bytes()
    assert len(blob) == 67807 and hashlib.md5(blob[20:]).hexdigest() == "02f5848ec18fa461dd34f35359f85456"
    assert run(exe, "d", "--host", f"--in={gip}", f"--out={back}").returncode == 0
    assert back.read_bytes() == data.tobytes()
    r = run(exe, "c", f"--in={src}", f"--out={gip}")          # no --host: refuses, does not fall back
    assert r.returncode == 1 and "no GPU path" in r.stderr


def test_index_trailer_is_a_pure_suffix(cli, tmp_path):
    """--index appends the packet-offset index behind the stream: bytes up to the header's size field are
    those written without it, the header still says 20 + packet bytes (what the reference reads), and both
    files decode -- with one thread and with all, through gpuar and gpuar-host."""
    src, plain, indexed = tmp_path / "in.dat", tmp_path / "plain.gip", tmp_path / "indexed.gip"
    data = synth.zipf(7, 5 * 8192 + 123)
    data.tofile(src)
    assert run(cli, "c", "--host", f"--in={src}", f"--out={plain}").returncode == 0
    assert run(cli, "c", "--host", "--index", f"--in={src}", f"--out={indexed}").returncode == 0
    a, b = plain.read_bytes(), indexed.read_bytes()
    assert b[:len(a)] == a and int.from_bytes(b[12:20], "little") == len(a)
    trailer = b[len(a):]
    assert trailer[:4] == b"GIPX" and trailer[-4:] == b"XPIG"
    assert int.from_bytes(trailer[4:8], "little") == 1 and int.from_bytes(trailer[8:16], "little") == 6
    assert int.from_bytes(trailer[-12:-4], "little") == len(trailer) and len(trailer) % 4 == 0
    clens = [int.from_bytes(trailer[16 + 2 * i:18 + 2 * i], "little") for i in range(6)]
    assert sum(clens) == len(a) - 20
    off = 20
    for c in clens:                                   # the stored lengths are the packets' own headers
        assert int.from_bytes(a[off:off + 2], "little") == c
        off += c
    for exe, threads in ((cli, "1"), (cli, "0"), (cli + "-host", "0")):
        back = tmp_path / "back.dat"
        r = run(exe, "d", "--host", "--threads", threads, f"--in={indexed}", f"--out={back}")
        assert r.returncode == 0, r.stderr
        assert back.read_bytes() == data.tobytes()
    # a damaged trailer is ignored (lengths no longer add up): the reader falls back to the header walk
    broken = bytearray(b)
    broken[len(a) + 16] ^= 1
    (tmp_path / "broken.gip").write_bytes(bytes(broken))
    assert run(cli, "d", "--host", f"--in={tmp_path / 'broken.gip'}", f"--out={tmp_path / 'b.dat'}").returncode == 0
    assert (tmp_path / "b.dat").read_bytes() == data.tobytes()
    # empty input: header + an index of zero packets
    (tmp_path / "empty").write_bytes(b"")
    assert run(cli, "c", "--host", "--index", f"--in={tmp_path / 'empty'}", f"--out={tmp_path / 'e.gip'}").returncode == 0
    e = (tmp_path / "e.gip").read_bytes()
    assert len(e) == 20 + 28 and int.from_bytes(e[12:20], "little") == 20
    assert run(cli, "d", "--host", f"--in={tmp_path / 'e.gip'}", f"--out={tmp_path / 'e.back'}").returncode == 0
    assert (tmp_path / "e.back").read_bytes() == b""


def test_config0_full_64mib_through_the_products_cpu_compressor(cli, tmp_path):
    """BASELINE.json configs[0]: data/random_64m.dat (stand-in: uniform(42), 64 MiB) encoded and decoded by
    the product's own CPUCompressor (`gpuar --host`), md5-checked as /root/reference/README.md:12-29 does.
    Packet stream md5 = what the reference's --host CLI wrote for the same bytes."""
    s = [s for s in SURVEY["streams"] if s.get("slow")][0]
    assert s["n"] == 64 << 20 and s["stream_md5"] == "c01b5d124681f6fc7264574e57548cdb"
    src, gip, back = tmp_path / "random_64m.dat", tmp_path / "random_64m.gip", tmp_path / "random_64m.back"
    synth.generate(s["kind"], s["seed"], s["n"]).tofile(src)
    r = run(cli, "c", "--host", "--threads=0", f"--in={src}", f"--out={gip}")
    assert r.returncode == 0, r.stderr
    blob = gip.read_bytes()
    assert len(blob) == s["gip_bytes"] == 67648304
    assert int.from_bytes(blob[4:8], "little") == s["n"] and int.from_bytes(blob[12:16], "little") == s["gip_bytes"]
    assert hashlib.md5(blob[20:]).hexdigest() == s["stream_md5"]
    del blob
    r = run(cli, "d", "--host", "--threads=0", f"--in={gip}", f"--out={back}")
    assert r.returncode == 0, r.stderr
    assert hashlib.md5(back.read_bytes()).hexdigest() == s["input_md5"] == "c9f0253b284172e8c4456be3234de256"


def reference_style_header(blob: bytes) -> bytes:
    """What the reference's writer leaves in a .gip header: only the low 4 bytes of each size are set,
    bytes 3, 8-11 and 16-19 are whatever was on its stack (src/file_header.hpp:31-36)."""
    h = bytearray(blob[:20])
    h[3] = 0x5A
    h[8:12] = b"\xde\xad\xbe\xef"
    h[16:20] = b"\x13\x37\xc0\xde"
    return bytes(h) + blob[20:]


def test_host_decodes_files_with_the_references_uninitialised_header_bytes(cli, tmp_path):
    src, gip, ref_gip, back = tmp_path / "in.dat", tmp_path / "a.gip", tmp_path / "ref.gip", tmp_path / "back.dat"
    data = synth.text(12, 3 * 8192 + 777)
    data.tofile(src)
    assert run(cli, "c", "--host", f"--in={src}", f"--out={gip}").returncode == 0
    ref_gip.write_bytes(reference_style_header(gip.read_bytes()))
    r = run(cli, "d", "--host", f"--in={ref_gip}", f"--out={back}")
    assert r.returncode == 0, r.stderr
    assert back.read_bytes() == data.tobytes()
    assert f"Uncompressed file size {data.size} bytes" in r.stdout


def test_host_decode_streams_in_windows(cli, tmp_path):
    """--host d walks the stream in windows of 4096 packets (bounded memory): a file longer than one
    window, with a ragged tail, with and without the index trailer."""
    src, gip, idx, back = tmp_path / "in.dat", tmp_path / "a.gip", tmp_path / "i.gip", tmp_path / "back.dat"
    data = synth.zipf(5, (4096 + 300) * 8192 + 99)
    data.tofile(src)
    assert run(cli, "c", "--host", "--threads=0", f"--in={src}", f"--out={gip}").returncode == 0
    assert run(cli, "c", "--host", "--threads=0", "--index", f"--in={src}", f"--out={idx}").returncode == 0
    for f in (gip, idx):
        assert run(cli, "d", "--host", "--threads=0", f"--in={f}", f"--out={back}").returncode == 0
        assert hashlib.md5(back.read_bytes()).hexdigest() == hashlib.md5(data.tobytes()).hexdigest()
    cut = tmp_path / "cut.gip"
    cut.write_bytes(gip.read_bytes()[:-50])
    r = run(cli, "d", "--host", "--threads=0", f"--in={cut}", f"--out={back}")
    assert r.returncode == 1
