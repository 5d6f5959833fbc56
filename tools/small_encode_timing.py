import os, sys
sys.path.insert(0, os.getcwd())
import torch
from gpuar_amd import hip as H
H.LIB_PATH = os.path.abspath(sys.argv[1])
os.environ["GPUAR_ENCODE_MODE"] = "latency"
for packets in (64, 8192, 32768):
    n = packets * 8192
    d_in = H.generate("uniform", 42, n)
    npk = H.packet_count(n)
    d_slots = torch.empty(npk * H.SLOT, dtype=torch.uint8, device="cuda")
    H.encode(d_in, d_slots); torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(7)]
    for s, e in ev:
        s.record(); H.encode(d_in, d_slots); e.record()
    torch.cuda.synchronize()
    ms = min(s.elapsed_time(e) for s, e in ev)
    ok = bool(torch.equal(H.decode(d_slots, npk)[:n], d_in))
    print(f"{os.path.basename(sys.argv[1]):12s} {packets:6d} packets: latency-mode encode {ms:7.3f} ms {n / ms / 1e6:7.1f} GB/s  round trip {ok}", flush=True)
