#!/usr/bin/env python3
"""bench.py -- encode+decode throughput of the GPUAR packet codec on MI355X.

Contract: `python bench.py --gpus N --steps K --warmup W` prints ONE JSON line
on rank 0.  For N > 1 the driver launches one process per GPU through
torch.distributed.run; ranks shard the input stream by contiguous packet
ranges (no data-path collective -- packets are independent),
so per-GPU work is fixed and the scaling is "weak".

A "step" is one pass of the hot path over one batch that is already resident
in HBM: the encode kernel over the rank's shard, then the decode kernel over
the slots it produced.  `value` = uncompressed bytes through that round trip
per second, summed over ranks; encode-only and decode-only rates, the
compression ratio and the round-trip check are reported next to it.

Workload (default): uniform(42) stream, 8 GiB per GPU -- the size
BASELINE.json's north_star quotes its single-GPU encode target on.  The 64 MiB
stand-in for data/random_64m.dat (configs[1]) is timed too and reported under
"small_config" (it cannot fill the chip: 8192 packets = 128 groups of 64).
"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# RCCL between processes needs dmabuf IPC on this pool (already exported there; harmless to repeat)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

GIB = 1 << 30
HBM_PEAK_GBPS = 8000.0      # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--kind", default="uniform", choices=["uniform", "zipf", "text"])
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--gib-per-gpu", type=float, default=8.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-small-config", action="store_true")
    ap.add_argument("--cpu-sample-mib", type=int, default=64)
    return ap.parse_args()


def timed_kernel_ms(fn, reps):
    """Average duration of `fn` (one kernel launch on torch's current stream) from HIP events."""
    import torch
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    return [a.elapsed_time(b) for a, b in ev]


def load_profiled_traffic(args, n_bytes):
    """HBM bytes per launch from the newest profiles/*_traffic.json (PMC passes of rocprofv3,
    tools/prof.sh + tools/traffic_from_prof.py), if one exists for this workload size and stream
    kind; counters cannot be collected from inside this process, so otherwise traffic is null."""
    import glob
    out = {}
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json")))
    if not files or args.kind != "uniform":
        return out
    try:
        t = json.load(open(files[-1]))
    except Exception:
        return out
    if abs(t.get("input_gib", 0) * GIB - n_bytes) > 1:
        return out
    for k in ("encode", "decode"):
        if k in t:
            out[k] = t[k]["hbm_bytes_per_launch"]
    out["source"] = os.path.basename(files[-1]) + ": " + t.get("source", "")
    return out


def cpu_baseline(kind, seed, sample_bytes):
    """Reference codec (oracle/_ref: the reference's own arCompress/arDecompress) or, if that
    build is absent, the C port, timed on ONE host core the way the reference times --host
    (model init + codec call only, src/cpu_compressor.cpp:58-62,157-161)."""
    from gpuar_amd import synth
    from oracle import oracle as O
    codec = O.best()
    data = synth.generate(kind, seed, sample_bytes)
    t0 = time.perf_counter()
    stream = codec.encode_stream(data)
    t1 = time.perf_counter()
    back = codec.decode_stream(stream, data.size)
    t2 = time.perf_counter()
    ok = bool((back == data).all())
    enc, dec = data.size / (t1 - t0) / 1e9, data.size / (t2 - t1) / 1e9
    res = {
        "value": data.size / (t2 - t0) / 1e9, "unit": "GB/s", "cores": 1, "kind": codec.kind,
        "sample": f"first {sample_bytes >> 20} MiB of the same {kind}({seed}) stream, encode then decode, 1 thread",
        "encode_GBps": enc, "decode_GBps": dec, "roundtrip_ok": ok, "host_cpus": os.cpu_count(),
    }
    # Extra row: the same codec with the packets fanned out over host threads --
    # packets are independent, so this is the fair "all of the host" ceiling.  8 MiB per thread.
    from concurrent.futures import ThreadPoolExecutor
    threads = max(1, min(os.cpu_count() or 1, 64))
    per = 8 << 20
    big = synth.generate(kind, seed, threads * per)
    chunks = [big[t * per:(t + 1) * per] for t in range(threads)]
    with ThreadPoolExecutor(threads) as pool:            # ctypes calls release the GIL
        t0 = time.perf_counter()
        streams = list(pool.map(codec.encode_stream, chunks))
        t1 = time.perf_counter()
        backs = list(pool.map(lambda sc: codec.decode_stream(sc[0], sc[1].size), zip(streams, chunks)))
        t2 = time.perf_counter()
    res["all_cores"] = {
        "cores": threads, "sample": f"{threads} x {per >> 20} MiB of the same stream, one contiguous packet range per thread",
        "value": big.size / (t2 - t0) / 1e9, "encode_GBps": big.size / (t1 - t0) / 1e9, "decode_GBps": big.size / (t2 - t1) / 1e9,
        "roundtrip_ok": all(bool((b == c).all()) for b, c in zip(backs, chunks)),
    }
    return res


def main():
    args = parse_args()
    import torch
    import torch.distributed as dist
    from gpuar_amd import hip as H

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    # Test hook: GPUAR_OVERSUBSCRIBE_DEVICES=1 lets several ranks share one physical GPU (rank -> device
    # rank % count) with gloo carrying the barrier/size exchange, so the N > 1 flow can be exercised on a
    # one-GPU box.  The driver's real runs use one GPU per rank and RCCL ("nccl").
    oversubscribe = os.environ.get("GPUAR_OVERSUBSCRIBE_DEVICES") == "1"
    local_dev = local_rank % torch.cuda.device_count() if oversubscribe else local_rank
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    ctl_dev = dev                      # where the control tensors of the collectives live
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if oversubscribe:
            dist.init_process_group("gloo")
            ctl_dev = torch.device("cpu")
        else:
            dist.init_process_group("nccl", device_id=dev)
    H.load()

    # ---- this rank's shard: contiguous packet range of the global stream ----
    from gpuar_amd import sharding
    offset, n = sharding.weak_shard(int(args.gib_per_gpu * GIB), rank)
    npk = H.packet_count(n)
    d_in = H.generate(args.kind, args.seed, n, offset=offset, device=dev)
    d_slots = torch.empty(npk * H.SLOT, dtype=torch.uint8, device=dev)
    d_out = torch.empty(npk * H.PACKET, dtype=torch.uint8, device=dev)

    def encode():
        H.encode(d_in, d_slots)

    def decode():
        H.decode(d_slots, npk, d_out)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        encode()
        decode()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        encode()
        decode()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=ctl_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- per-kernel durations (HIP events on the launch stream), untimed region ----
    reps = max(3, args.steps)
    enc_ms = timed_kernel_ms(encode, reps)
    dec_ms = timed_kernel_ms(decode, reps)
    enc_avg, dec_avg = sum(enc_ms) / len(enc_ms), sum(dec_ms) / len(dec_ms)

    # ---- correctness of what was just timed ----
    status = H.status()
    d_stream, d_off = H.compact(d_slots, npk)
    c_bytes = int(d_off[-1].item())
    roundtrip_equal = bool(torch.equal(d_out[:n], d_in))
    sample = min(n, 64 << 20)
    md5_in = hashlib.md5(d_in[:sample].cpu().numpy().tobytes()).hexdigest()
    md5_out = hashlib.md5(d_out[:sample].cpu().numpy().tobytes()).hexdigest()
    oracle_ok = None
    if rank == 0:
        from oracle import oracle as O
        host = d_in[:64 * H.PACKET].cpu().numpy()
        want = O.best().encode_stream(host)
        got = d_stream[:int(d_off[64].item())].cpu().numpy()
        oracle_ok = bool(got.size == want.size and (got == want).all())
    del d_stream

    ok_flags = torch.tensor([int(roundtrip_equal and status == 0 and md5_in == md5_out), c_bytes], dtype=torch.int64, device=ctl_dev)
    if world > 1:
        all_flags = [torch.zeros_like(ok_flags) for _ in range(world)]
        dist.all_gather(all_flags, ok_flags)
    else:
        all_flags = [ok_flags]
    all_ok = all(int(f[0].item()) == 1 for f in all_flags)
    c_total = sum(int(f[1].item()) for f in all_flags)

    result = None
    if rank == 0:
        total_bytes = n * world
        ms_per_step = elapsed / args.steps * 1e3
        value = total_bytes * args.steps / elapsed / 1e9
        # dominant kernel = the slower of the two; algorithmic bytes per launch = N read/written + C written/read
        dom = "decode" if dec_avg >= enc_avg else "encode"
        algo_bytes = n + c_bytes

        traffic = load_profiled_traffic(args, n)

        def roof(ms, which):
            a = algo_bytes / (ms * 1e-3) / 1e9
            return {"bound": "hbm", "achieved": a, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": a / HBM_PEAK_GBPS,
                    "algorithmic_bytes_per_launch": algo_bytes, "traffic": traffic.get(which)}

        result = {
            "metric": "encode+decode GB/s (uncompressed bytes through encode then decode, kernels only, data resident in HBM)",
            "value": value, "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u16 coder state / u32 intermediates", "data": "synthetic",
            "config": {"workload": f"{args.kind}({args.seed}) {args.gib_per_gpu:g} GiB per GPU, 8192-byte packets, "
                                   f"{npk} packets per GPU, shard = contiguous packet range, no collective",
                       "parallelism": f"packet-sharded x{world}"},
            "encode_GBps": n * world / (enc_avg * 1e-3) / 1e9,
            "decode_GBps": n * world / (dec_avg * 1e-3) / 1e9,
            "encode_ms": enc_avg, "decode_ms": dec_avg,
            "encode_read_frac_of_hbm_peak": n / (enc_avg * 1e-3) / 1e9 / HBM_PEAK_GBPS,
            "compression_ratio": (c_total + 20) / total_bytes,
            "roundtrip_equal": all_ok, "md5_sample_match": md5_in == md5_out, "md5_sample": md5_in,
            "oracle_prefix_match": oracle_ok, "device_status": status,
            "roofline": dict(roof(dec_avg if dom == "decode" else enc_avg, dom), kernel=f"{dom}_kernel"),
            "roofline_encode": roof(enc_avg, "encode"), "roofline_decode": roof(dec_avg, "decode"),
            "traffic_source": traffic.get("source"),
        }

    # ---- configs[1]: the 64 MiB stand-in for data/random_64m.dat, rank 0 only ----
    if rank == 0 and not args.no_small_config:
        m = 64 << 20
        s_in = H.generate("uniform", 42, m, device=dev)
        s_npk = H.packet_count(m)
        s_slots = torch.empty(s_npk * H.SLOT, dtype=torch.uint8, device=dev)
        s_out = torch.empty(m, dtype=torch.uint8, device=dev)
        e = timed_kernel_ms(lambda: H.encode(s_in, s_slots), 5)
        d = timed_kernel_ms(lambda: H.decode(s_slots, s_npk, s_out), 5)
        s_stream, s_off = H.compact(s_slots, s_npk)
        s_total = int(s_off[-1].item())
        result["small_config"] = {
            "workload": "uniform(42) 64 MiB (stand-in for data/random_64m.dat), 1 GPU, 8192 packets = 128 groups of 64",
            "encode_GBps": m / (min(e) * 1e-3) / 1e9, "decode_GBps": m / (min(d) * 1e-3) / 1e9,
            "gip_bytes": s_total + 20,
            "stream_md5": hashlib.md5(s_stream[:s_total].cpu().numpy().tobytes()).hexdigest(),
            "reference_stream_md5": "c01b5d124681f6fc7264574e57548cdb",
            "roundtrip_equal": bool(torch.equal(s_out, s_in)),
        }

    if rank == 0 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(args.kind, args.seed, args.cpu_sample_mib << 20)

    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0 and not all_ok:
        raise SystemExit("round trip FAILED")


if __name__ == "__main__":
    main()
