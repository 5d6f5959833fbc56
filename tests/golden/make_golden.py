#!/usr/bin/env python3
"""Regenerates tests/golden/ref_vectors.json and tests/golden/adversarial_*.bin.

Run in the build container only (needs /root/reference to build oracle/_ref):

    python tests/golden/make_golden.py

Every expected output below comes from the REFERENCE's own arCompress
(/root/reference/src/gpuar_kernel.cu:487-531) through oracle/_ref -- not from
this repo's restatement.  The committed files are data: inputs (or generator
parameters) and the reference's outputs for them.
"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from gpuar_amd import synth  # noqa: E402
from oracle import oracle as O  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def md5(b) -> str:
    return hashlib.md5(bytes(b)).hexdigest()


def clens_of(stream: np.ndarray):
    out, off = [], 0
    while off < stream.size:
        c = int(stream[off]) | (int(stream[off + 1]) << 8)
        out.append(c)
        off += c
    return out


def midpoint_hugger(n: int) -> np.ndarray:
    """An input that keeps the coder's interval straddling 0x7FFF|0x8000 as long
    as it can, so underflow ('pending') bits pile up: exercises the encoder's
    long-pending path and the decoder's underflow branch.  Built with a small
    pure-Python model of the spec in SURVEY.md Appendix B."""
    cnt = [1] * 256
    total = 256
    lo, hi = 0, 0xFFFF
    out = bytearray()
    best_pending = pending = 0
    for _ in range(n):
        rng = hi - lo + 1
        cum = 0
        pick = None
        for s in range(256):
            nlo = lo + (cum * rng) // total
            nhi = lo + ((cum + cnt[s]) * rng) // total - 1
            if nlo <= 0x7FFF and nhi >= 0x8000:
                pick = (s, nlo, nhi)
                break
            cum += cnt[s]
        if pick is None:            # no symbol straddles: take the one just below the midpoint
            cum = 0
            for s in range(256):
                nlo = lo + (cum * rng) // total
                nhi = lo + ((cum + cnt[s]) * rng) // total - 1
                if nhi >= 0x7FFF - 64 or s == 255:
                    pick = (s, nlo, nhi)
                    break
                cum += cnt[s]
        s, lo, hi = pick
        lo &= 0xFFFF
        hi &= 0xFFFF
        cnt[s] += 1
        total += 1
        out.append(s)
        while True:
            if (hi ^ lo) & 0x8000 == 0:
                pending = 0
            elif (lo & 0x4000) and not (hi & 0x4000):
                pending += 1
                lo &= 0x3FFF
                hi |= 0x4000
            else:
                break
            lo = (lo << 1) & 0xFFFF
            hi = ((hi << 1) | 1) & 0xFFFF
        best_pending = max(best_pending, pending)
    return np.frombuffer(bytes(out), dtype=np.uint8).copy(), best_pending


def narrowest_greedy(n: int) -> np.ndarray:
    """Always codes a symbol of count 1 (the least probable), cycling so counts
    stay as flat as possible: the longest packets the model can be driven to."""
    return (np.arange(n) % 256).astype(np.uint8)


def main():
    ref = O.ReferenceOracle(expect_sha256=None)      # the file being pinned: nothing to check it against yet
    cases = []

    def add(name, data: np.ndarray, spec, keep_stream=False):
        stream = ref.encode_stream(data)
        back = ref.decode_stream(stream, data.size)
        assert np.array_equal(back, data), name
        c = {"name": name, "n": int(data.size), "input_md5": md5(data.tobytes()),
             "stream_len": int(stream.size), "stream_md5": md5(stream.tobytes()),
             "clens": clens_of(stream)}
        c.update(spec)
        if data.size <= 64:
            c["input_hex"] = data.tobytes().hex()
            c["stream_hex"] = stream.tobytes().hex()
        if keep_stream:
            with open(os.path.join(HERE, name + ".stream.bin"), "wb") as f:
                f.write(stream.tobytes())
        cases.append(c)

    for kind in synth.KINDS:
        for seed, n in [(1, 65539), (2, 8192), (3, 8191), (4, 8193), (5, 16384), (6, 100), (7, 1), (8, 17), (9, 4097), (11, 262144 + 5)]:
            add(f"{kind}_s{seed}_n{n}", synth.generate(kind, seed, n), {"kind": kind, "seed": seed})
    # single-symbol and two-symbol packets (extreme skew)
    for b in (0, 0x41, 0xFF):
        add(f"const_{b:02x}_8192", np.full(8192, b, dtype=np.uint8), {"kind": "const", "byte": b})
        add(f"const_{b:02x}_20000", np.full(20000, b, dtype=np.uint8), {"kind": "const", "byte": b})
    add("ramp_65536", narrowest_greedy(65536), {"kind": "ramp"})
    alt = np.tile(np.array([0x7F, 0x80], dtype=np.uint8), 4096)
    add("alt_7f80_8192", alt, {"kind": "tile", "tile_hex": "7f80"})
    hug, best_pending = midpoint_hugger(8192)
    with open(os.path.join(HERE, "adversarial_midpoint.in.bin"), "wb") as f:
        f.write(hug.tobytes())
    add("adversarial_midpoint", hug, {"kind": "file", "file": "adversarial_midpoint.in.bin",
                                      "max_pending_bits": int(best_pending)}, keep_stream=True)
    add("uniform_s1_n65539_keep", synth.uniform(1, 65539), {"kind": "uniform", "seed": 1}, keep_stream=True)

    # The checker binary itself is pinned: oracle/_ref/libgpuar_ref.so is git-ignored and cannot be rebuilt on the GPU
    # box (no /root/reference there), so "bit-exact vs the reference" on the box would otherwise rest on whatever file
    # was pushed.  ReferenceOracle() refuses a library whose sha256 is not the one recorded here, next to the
    # vectors that very file produced.
    with open(os.path.join(HERE, "ref_vectors.json"), "w") as f:
        json.dump({"_provenance": "expected outputs produced by oracle/_ref (the reference's unmodified "
                                  "arCompress/arDecompress); regenerate with tests/golden/make_golden.py",
                   "checker": {"file": "oracle/_ref/libgpuar_ref.so", "sha256": O.file_sha256(O.REF_LIB_PATH),
                               "built_by": "oracle/build_ref.sh from /root/reference/src/gpuar_kernel.cu + oracle/ref_driver.cpp"},
                   "cases": cases}, f, indent=0)
    print(f"wrote {len(cases)} cases; midpoint hugger max pending = {best_pending}")


if __name__ == "__main__":
    main()
