#!/usr/bin/env python3
"""Small fixed workload for rocprofv3: N encode + N decode launches over a resident buffer.

    rocprofv3 --kernel-trace --stats -d OUT -- python3 tools/prof_run.py [--gib 1] [--reps 3]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gib", type=float, default=1.0)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--kind", default="uniform")
    ap.add_argument("--only", default="all", choices=["all", "both", "encode", "decode"],
                    help="all = encode, compaction, decode (slots), decode (stream); both = the two coder kernels only")
    ap.add_argument("--slot", type=int, default=None, help="slot stride of an experiment build (-DGPUAR_SLOT_BYTES=...)")
    ap.add_argument("--lib", default=None, help="an experiment build of libgpuar_hip.so to profile instead of the product one")
    a = ap.parse_args()
    import torch
    from gpuar_amd import hip as H
    if a.lib:
        H.LIB_PATH = os.path.abspath(a.lib)
    if a.slot:
        H.SLOT = a.slot
    n = int(a.gib * (1 << 30)) // 8192 * 8192
    d_in = H.generate(a.kind, 42, n)
    npk = H.packet_count(n)
    d_slots = torch.empty(npk * H.SLOT, dtype=torch.uint8, device="cuda")
    d_out = torch.empty(npk * H.PACKET, dtype=torch.uint8, device="cuda")
    d_stream = torch.empty(npk * H.SLOT + 16, dtype=torch.uint8, device="cuda")
    d_off = torch.empty(npk + 1, dtype=torch.int64, device="cuda")
    H.encode(d_in, d_slots)
    for _ in range(a.reps):
        if a.only in ("all", "both", "encode"):
            H.encode(d_in, d_slots)
        if a.only == "all":
            H.compact(d_slots, npk, d_stream, d_off)
        if a.only in ("all", "both", "decode"):
            H.decode(d_slots, npk, d_out)
    torch.cuda.synchronize()
    # an experiment build is timed only while it still round-trips (a broken kernel must not produce a number);
    # only a build with another slot stride (--slot) lays its output out differently from what this check reads
    if a.only != "encode":
        assert torch.equal(d_out[:n], d_in), "decode(encode(x)) != x"
    if a.only == "all":
        for _ in range(a.reps):
            d_out.zero_()
            H.decode_stream(d_stream, d_off, npk, d_out)
        torch.cuda.synchronize()
        assert torch.equal(d_out[:n], d_in), "decode_stream(compact(encode(x))) != x"
    assert H.status() == 0
    print("prof_run ok", n, "bytes", npk, "packets", a.kind)


if __name__ == "__main__":
    main()
