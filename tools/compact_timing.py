import sys, os
sys.path.insert(0, os.getcwd())
import torch
from gpuar_amd import hip as H
n = 8 << 30
d_in = H.generate("uniform", 42, n)
npk = H.packet_count(n)
d_slots = H.encode(d_in)
d_stream, d_off = H.compact(d_slots, npk)
torch.cuda.synchronize()
for name, fn in (("compact", lambda: H.compact(d_slots, npk, d_stream, d_off) if False else H.compact(d_slots, npk)),):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(3)]
    for s, e in ev:
        s.record(); r = fn(); e.record()
    torch.cuda.synchronize()
    print(name, [round(s.elapsed_time(e), 3) for s, e in ev], "ms")
total = int(d_off[-1].item())
print("stream bytes", total, "GB/s (read+write)", 2 * total / 1e6 / min(s.elapsed_time(e) for s, e in ev))
