#!/usr/bin/env python3
"""Kernel time of encode / decode on different stream kinds (same size): separates what the data
costs (stream bytes per symbol, refills, misses) from what the instruction stream costs.

    python3 tools/kind_timing.py [--gib 2] [--kinds uniform,text,zipf,zeros]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gib", type=float, default=2.0)
    ap.add_argument("--kinds", default="uniform,text,zipf,zeros")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--slot", type=int, default=None, help="slot stride the experiment build was compiled with (-DGPUAR_SLOT_BYTES=...)")
    ap.add_argument("--lib", default=None, help="an experiment build of libgpuar_hip.so to time instead of the product one")
    a = ap.parse_args()
    import torch
    from gpuar_amd import hip as H
    if a.lib:
        H.LIB_PATH = os.path.abspath(a.lib)
    if a.slot:
        H.SLOT = a.slot
    n = int(a.gib * (1 << 30)) // 8192 * 8192
    npk = H.packet_count(n)
    d_slots = torch.empty(npk * H.SLOT, dtype=torch.uint8, device="cuda")
    d_out = torch.empty(npk * H.PACKET, dtype=torch.uint8, device="cuda")
    for kind in a.kinds.split(","):
        d_in = H.generate(kind, 1, n)
        H.encode(d_in, d_slots)
        H.decode(d_slots, npk, d_out)
        torch.cuda.synchronize()
        res = {}
        for name, fn in (("encode", lambda: H.encode(d_in, d_slots)), ("decode", lambda: H.decode(d_slots, npk, d_out))):
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.reps)]
            for s, e in ev:
                s.record()
                fn()
                e.record()
            torch.cuda.synchronize()
            res[name] = min(s.elapsed_time(e) for s, e in ev)
        d_stream, d_off = H.compact(d_slots, npk)
        ratio = int(d_off[-1].item()) / n
        ok = bool(torch.equal(d_out[:n], d_in)) and H.status() == 0
        steps = npk / 64 * 8192                     # symbol steps in the launch
        waves_per_simd = max(1.0, npk / 64 / 1024)  # decode: one wavefront per SIMD at a time
        cyc = res["decode"] * 1e-3 * 2.4e9 / (8192 * waves_per_simd)
        print(f"{kind:8s} ratio {ratio:.4f}  encode {res['encode']:8.3f} ms  decode {res['decode']:8.3f} ms  "
              f"decode ~{cyc:6.0f} cycles/symbol-step at 2.4 GHz  round trip {'ok' if ok else 'FAILED'}", flush=True)
        del d_in, d_stream, d_off


if __name__ == "__main__":
    main()
