#!/usr/bin/env python3
"""Does the GPU overlap an encode kernel with a decode kernel of ANOTHER batch when they are put on two
streams?  (The decoder leaves ~40 % of each SIMD's vector issue slots idle; the encoder is vector-bound.)
Measures K pipelined steps -- encode(batch k+1) || decode(batch k), double-buffered slots -- against the
same K steps run back to back on one stream.  Exploration only: bench.py keeps the sequential definition."""
import argparse
import sys
import os
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gpuar_amd import hip as H

ap = argparse.ArgumentParser()
ap.add_argument("--gib", type=float, default=4.0)
ap.add_argument("--steps", type=int, default=6)
a = ap.parse_args()
n = int(a.gib * (1 << 30))
npk = H.packet_count(n)
d_in = H.generate("uniform", 42, n)
slots = [torch.empty(npk * H.SLOT, dtype=torch.uint8, device="cuda") for _ in range(2)]
outs = [torch.empty(npk * H.PACKET, dtype=torch.uint8, device="cuda") for _ in range(2)]
s_enc, s_dec = torch.cuda.Stream(), torch.cuda.Stream()

def sequential(k):
    for i in range(k):
        H.encode(d_in, slots[i & 1])
        H.decode(slots[i & 1], npk, outs[i & 1])

def pipelined(k):
    enc_done = [torch.cuda.Event() for _ in range(k)]
    dec_done = [torch.cuda.Event() for _ in range(k)]
    for i in range(k):
        if i >= 2:
            s_enc.wait_event(dec_done[i - 2])          # slots[i & 1] free again
        H.encode(d_in, slots[i & 1], stream=s_enc)
        enc_done[i].record(s_enc)
        s_dec.wait_event(enc_done[i])
        H.decode(slots[i & 1], npk, outs[i & 1], stream=s_dec)
        dec_done[i].record(s_dec)

for name, fn in (("sequential", sequential), ("pipelined", pipelined)):
    fn(2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn(a.steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ok = bool(torch.equal(outs[(a.steps - 1) & 1][:n], d_in))
    print(f"{name:10s}: {dt / a.steps * 1e3:8.2f} ms per step, {n * a.steps / dt / 1e9:7.1f} GB/s, last round trip ok = {ok}")
