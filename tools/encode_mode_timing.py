import os, sys
sys.path.insert(0, os.getcwd())
import torch
from gpuar_amd import hip as H
for packets in (64, 1024, 8192, 16384, 32768, 49152, 65536):
    n = packets * 8192
    d_in = H.generate("uniform", 42, n)
    npk = H.packet_count(n)
    d_slots = torch.empty(npk * H.SLOT, dtype=torch.uint8, device="cuda")
    ref = None
    row = []
    for mode in ("throughput", "latency"):
        os.environ["GPUAR_ENCODE_MODE"] = mode
        H.encode(d_in, d_slots); torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(5)]
        for s, e in ev:
            s.record(); H.encode(d_in, d_slots); e.record()
        torch.cuda.synchronize()
        ms = min(s.elapsed_time(e) for s, e in ev)
        d_stream, d_off = H.compact(d_slots, npk)
        cur = d_stream[:int(d_off[-1].item())].clone()
        same = True if ref is None else bool(torch.equal(ref, cur))
        ref = cur
        row.append(f"{mode} {ms:7.3f} ms {n / ms / 1e6:7.1f} GB/s{'' if same else '  STREAMS DIFFER'}")
    print(f"{packets:6d} packets ({n >> 20:5d} MiB): " + " | ".join(row), flush=True)
