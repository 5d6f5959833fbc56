"""Large-scale regression check on the GPU box: the slots an OLD build of the library produces against the current one's, byte
for byte where they are defined, on streams of every generator kind plus a stream built to exercise the coder's rare paths;
then both decoders on the result.   python3 tools/equal_slots_check.py OLD.so [GiB]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.getcwd())
import torch

from gpuar_amd import hip as H


def load(path):
    lib = C.CDLL(os.path.abspath(path))
    vp, sz = C.c_void_p, C.c_size_t
    lib.gpuar_hip_encode.restype = C.c_int
    lib.gpuar_hip_encode.argtypes = [vp, sz, vp, vp, vp]
    lib.gpuar_hip_decode.restype = C.c_int
    lib.gpuar_hip_decode.argtypes = [vp, sz, vp, vp, vp]
    return lib


def defined_mask(slots, npk):
    v = slots.view(npk, H.SLOT)
    clen = v[:, 0].to(torch.int64) | (v[:, 1].to(torch.int64) << 8)
    return clen


def main():
    old = load(sys.argv[1])
    gib = float(sys.argv[2]) if len(sys.argv) > 2 else 2.0
    n = int(gib * (1 << 30)) // H.PACKET * H.PACKET - 3000          # a short last packet
    npk = H.packet_count(n)
    s = int(torch.cuda.current_stream().cuda_stream)
    word = torch.zeros(1, dtype=torch.int32, device="cuda")
    kinds = ["uniform", "text", "zipf", "zeros", "carry"]
    for kind in kinds:
        if kind == "carry":
            g = torch.Generator(device="cuda").manual_seed(7)
            r = torch.rand(n, device="cuda", generator=g)
            d_in = torch.where(r < 0.48, torch.tensor(0x7F, dtype=torch.uint8, device="cuda"), torch.tensor(0x80, dtype=torch.uint8, device="cuda"))
            d_in = torch.where(r > 0.97, torch.tensor(0xFF, dtype=torch.uint8, device="cuda"), d_in).contiguous()
            del r
        else:
            d_in = H.generate(kind, 11, n)
        a = torch.zeros(npk * H.SLOT, dtype=torch.uint8, device="cuda")
        b = torch.zeros(npk * H.SLOT, dtype=torch.uint8, device="cuda")
        rc = old.gpuar_hip_encode(d_in.data_ptr(), n, a.data_ptr(), word.data_ptr(), s)
        assert rc == 0
        H.encode(d_in, b, d_status=word, mode="throughput")
        torch.cuda.synchronize()
        # the latency-mode kernel on the first 256 MiB: the same slots
        n_small = min(n, 256 << 20)
        npk_small = H.packet_count(n_small)
        c = torch.zeros(npk_small * H.SLOT, dtype=torch.uint8, device="cuda")
        H.encode(d_in[:n_small], c, d_status=word, mode="latency")
        torch.cuda.synchronize()
        cc = defined_mask(c, npk_small)
        col_s = torch.arange(H.SLOT, device="cuda")[None, :]
        small_same = bool(torch.equal(cc, defined_mask(a, npk)[:npk_small])) and not bool(
            ((c.view(npk_small, H.SLOT) != a.view(npk, H.SLOT)[:npk_small]) & (col_s < cc[:, None])).any())
        del c
        ca, cb = defined_mask(a, npk), defined_mask(b, npk)
        same_len = bool(torch.equal(ca, cb))
        same = True
        col = torch.arange(H.SLOT, device="cuda")[None, :]
        for p0 in range(0, npk, 1 << 16):                                # 64 Ki packets at a time
            p1 = min(npk, p0 + (1 << 16))
            differ = (a.view(npk, H.SLOT)[p0:p1] != b.view(npk, H.SLOT)[p0:p1]) & (col < ca[p0:p1, None])
            same = same and not bool(differ.any())
        out_new = H.decode(b, npk)
        out_old = torch.empty(npk * H.PACKET, dtype=torch.uint8, device="cuda")
        rc = old.gpuar_hip_decode(b.data_ptr(), npk, out_old.data_ptr(), word.data_ptr(), s)
        torch.cuda.synchronize()
        rt_new = bool(torch.equal(out_new[:n], d_in))
        rt_old = bool(torch.equal(out_old[:n], d_in))
        ratio = float(ca.sum().item()) / n
        print(f"{kind:8s} {n} bytes, {npk} packets, ratio {ratio:.4f}: lengths equal {same_len}, slots equal {same}, "
              f"latency-mode kernel on the first 256 MiB equal {small_same}, "
              f"new decoder round trip {rt_new}, old decoder on the new slots {rt_old}, status {int(word.item())}", flush=True)
        assert same_len and same and small_same and rt_new and rt_old and int(word.item()) == 0
        del a, b, out_new, out_old, d_in
        torch.cuda.empty_cache()
    print("ok")


if __name__ == "__main__":
    main()
