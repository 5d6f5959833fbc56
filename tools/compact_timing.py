#!/usr/bin/env python3
"""Times device-side compaction (scan + gather_kernel) on an encoded stream: tools/compact_timing.py [--gib G] [--kind K]
[--lib path/to/experiment.so].  Output buffers are allocated once; the time is the best of five launches."""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402
from gpuar_amd import hip as H  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--gib", type=float, default=8.0)
ap.add_argument("--kind", default="uniform")
ap.add_argument("--seed", type=int, default=42)
ap.add_argument("--lib", default=None)
a = ap.parse_args()
if a.lib:
    H.LIB_PATH = os.path.abspath(a.lib)
n = int(a.gib * (1 << 30)) // H.PACKET * H.PACKET
d_in = H.generate(a.kind, a.seed, n)
npk = H.packet_count(n)
d_slots = H.encode(d_in)
del d_in
d_stream, d_off = H.compact(d_slots, npk)
want = d_stream.clone()
total = int(d_off[-1].item())
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(5)]
for s, e in ev:
    s.record()
    H.compact(d_slots, npk, d_stream, d_off)
    e.record()
torch.cuda.synchronize()
ms = min(s.elapsed_time(e) for s, e in ev)
ok = bool(torch.equal(d_stream[:total], want[:total]))
print(f"{os.path.basename(a.lib or 'product'):28s} {a.kind} {a.gib:g} GiB: compaction {ms:7.3f} ms  {2 * total / ms / 1e6:7.1f} GB/s read + write  "
      f"({2 * total / ms / 1e6 / 8000 * 100:4.1f} % of 8 TB/s)  same bytes: {ok}")
