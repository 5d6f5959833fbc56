import sys, os
sys.path.insert(0, os.getcwd())
import torch
from gpuar_amd import hip as H
for npk in (64, 48, 32, 16, 8, 4, 1):
    n = npk * 8192
    d_in = H.generate("uniform", 1, n)
    d_slots = H.encode(d_in)
    d_out = torch.empty(npk * 8192, dtype=torch.uint8, device="cuda")
    H.decode(d_slots, npk, d_out); torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(7)]
    for s, e in ev:
        s.record(); H.decode(d_slots, npk, d_out); e.record()
    torch.cuda.synchronize()
    t = min(s.elapsed_time(e) for s, e in ev)
    ok = torch.equal(d_out[:n], d_in)
    print(f"{npk:3d} live lanes in ONE wavefront: decode {t:.3f} ms = {t*1e-3*2.39e9/8192:.0f} cycles per symbol step  ok={ok}", flush=True)
