import sys, os, ctypes, time
sys.path.insert(0, '.')
import torch
from gpuar_amd import hip as H
H.LIB_PATH = os.path.join(os.path.dirname(H.LIB_PATH), sys.argv[1])
H.load()
n = 2 << 30
d_in = H.generate("uniform", 42, n)
npk = H.packet_count(n)
d_slots = H.encode(d_in)
d_out = torch.empty(npk * 8192, dtype=torch.uint8, device="cuda")
for _ in range(2): H.decode(d_slots, npk, d_out)
torch.cuda.synchronize()
a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(3): H.decode(d_slots, npk, d_out)
b.record(); torch.cuda.synchronize()
print(sys.argv[1], "decode ms", a.elapsed_time(b) / 3, "ok" if torch.equal(d_out[:n], d_in) else "MISMATCH(expected for diag)")
