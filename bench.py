#!/usr/bin/env python3
"""bench.py -- encode+decode throughput of the GPUAR packet codec on MI355X.

Contract: `python bench.py --gpus N --steps K --warmup W` prints ONE JSON line
on rank 0.  For N > 1 there is one process per GPU: either the caller starts
them (`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`,
RANK/LOCAL_RANK/WORLD_SIZE in the environment) or -- when WORLD_SIZE is not set
-- this script starts them itself as child processes, before it has touched
the GPU, and relays rank 0's line.  Ranks shard the input stream by contiguous
packet ranges with no data-path collective (packets are independent, SURVEY.md
8(e)); torch.distributed (RCCL) carries only the barrier, the MAX of the elapsed
time and the gathering of sizes/flags.

    --scaling weak   (default)  every GPU codes --gib-per-gpu GiB: rank r owns bytes [r*B, (r+1)*B)
    --scaling strong            --total-gib GiB in all, split into N contiguous packet ranges
                                (BASELINE.json configs[3]: 8 GiB uniform over 8 GPUs)

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
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# RCCL between processes needs dmabuf IPC on this pool (already exported there; harmless to repeat)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

GIB = 1 << 30
HBM_PEAK_GBPS = 8000.0      # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
KERNEL_SOURCES = ("gpuar_amd/csrc/gpuar_kernels.hip", "gpuar_amd/csrc/lane_codec.h")


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--kind", default="uniform", choices=["uniform", "zipf", "text"])
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--gib-per-gpu", type=float, default=8.0, help="weak scaling: GiB each GPU codes")
    ap.add_argument("--total-gib", type=float, default=8.0, help="strong scaling: GiB split over all GPUs")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-small-config", action="store_true")
    ap.add_argument("--cpu-sample-mib", type=int, default=64)
    return ap.parse_args(argv)


def kernel_source_stamp(root=ROOT):
    """sha256 over the kernel sources: a traffic record is only quoted for the kernels it was taken from."""
    h = hashlib.sha256()
    for rel in KERNEL_SOURCES:
        with open(os.path.join(root, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def self_launch_command(argv, n_gpus, port):
    """The child command for `python bench.py --gpus N` with no launcher around it: one rank per GPU
    through torch.distributed.run, rendezvous on 127.0.0.1.  The children are fresh processes; this
    process has made no GPU call when it starts them (and never execs)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus),
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), *argv]


def self_launch(argv, n_gpus):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    child = subprocess.run(self_launch_command(argv, n_gpus, port), env=env, stdout=subprocess.PIPE, text=True)
    lines = [l for l in child.stdout.splitlines() if l.startswith("{")]
    for l in child.stdout.splitlines():
        if not l.startswith("{"):
            print(l, file=sys.stderr)
    if lines:
        print(lines[-1])
    return child.returncode if child.returncode or lines else 1


def plan_shard(args, world, rank):
    """(byte offset into the global stream, byte length) this rank codes."""
    from gpuar_amd import sharding
    if args.scaling == "strong":
        total = int(args.total_gib * GIB) // sharding.PACKET * sharding.PACKET
        return sharding.plan_shards(total, world)[rank]
    return sharding.weak_shard(int(args.gib_per_gpu * GIB), rank)


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


def load_profiled_traffic(kind, n_bytes, root=ROOT, stamp=None):
    """HBM bytes per launch (and the issue-side counters) from the newest profiles/*_traffic.json --
    the PMC passes of rocprofv3 (tools/prof.sh + tools/traffic_from_prof.py); counters cannot be
    collected from inside this process.  The record is used only if it was taken on this workload
    size and stream kind AND on the kernel sources that are built now (sha256 stamp); otherwise
    traffic is null and `traffic_source` says why."""
    import glob
    files = sorted(glob.glob(os.path.join(root, "profiles", "*_traffic.json")))
    if not files:
        return {"source": "no profiles/*_traffic.json"}
    name = os.path.basename(files[-1])
    try:
        t = json.load(open(files[-1]))
    except Exception as e:                      # noqa: BLE001 -- a damaged record is reported, not fatal
        return {"source": f"{name}: unreadable ({e})"}
    if t.get("kind", "uniform") != kind or abs(t.get("input_gib", 0) * GIB - n_bytes) > 1:
        return {"source": f"{name}: taken on {t.get('kind', 'uniform')} {t.get('input_gib')} GiB, not this workload"}
    stamp = stamp or kernel_source_stamp(root)
    if t.get("kernel_source_sha256_16") != stamp:
        return {"source": f"{name}: STALE -- taken on kernel sources {t.get('kernel_source_sha256_16')}, built sources are "
                          f"{stamp}; re-run tools/refresh_profiles.sh"}
    out = {"source": name + ": " + t.get("source", "")}
    for k in ("encode", "decode"):
        if k in t:
            out[k] = t[k]
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
    # Extra row: the same codec with the packets fanned out over EVERY host core
    # (one thread per logical CPU, one contiguous packet range each) -- packets are independent, so this is
    # the fair "all of the host" ceiling.
    from concurrent.futures import ThreadPoolExecutor
    threads = max(1, os.cpu_count() or 1)
    per = (4 << 20) if threads > 64 else (8 << 20)
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


def assemble_result(args, world, n_ranks_seen, shard_bytes, total_bytes, npk_rank0, elapsed, enc_ms, dec_ms,
                    c_bytes_rank0, c_total, all_ok, md5_in, md5_out, oracle_ok, status, traffic):
    """The JSON line, from plain numbers (no GPU objects): the driver's contract fields, the roofline of
    the dominant kernel and what was verified.  enc_ms / dec_ms are rank 0's average launch durations over
    its shard of `shard_bytes` bytes; elapsed is the MAX over ranks of the wall time of args.steps steps."""
    ms_per_step = elapsed / args.steps * 1e3
    value = total_bytes * args.steps / elapsed / 1e9
    # dominant kernel = the slower of the two; algorithmic bytes per launch = N read/written + C written/read
    dom = "decode" if dec_ms >= enc_ms else "encode"
    algo_bytes = shard_bytes + c_bytes_rank0

    def roof(ms, which):
        a = algo_bytes / (ms * 1e-3) / 1e9
        t = traffic.get(which) or {}
        r = {"bound": "hbm", "achieved": a, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": a / HBM_PEAK_GBPS,
             "algorithmic_bytes_per_launch": algo_bytes, "traffic": t.get("hbm_bytes_per_launch")}
        # the roof that actually binds (SURVEY.md section 7 risk 1): vector-issue and wait share of the wavefronts' cycles
        for k in ("valu_busy", "valu_busy_per_simd", "wait_frac", "valu_insts_per_symbol_step"):
            if k in t:
                r[k] = t[k]
        return r

    if args.scaling == "strong":
        work = (f"{args.kind}({args.seed}) {args.total_gib:g} GiB in all over {world} GPU(s), 8192-byte packets, "
                f"{npk_rank0} packets on rank 0, shard = contiguous packet range, no collective")
    else:
        work = (f"{args.kind}({args.seed}) {args.gib_per_gpu:g} GiB per GPU, 8192-byte packets, "
                f"{npk_rank0} packets per GPU, shard = contiguous packet range, no collective")
    return {
        "metric": "encode+decode GB/s (uncompressed bytes through encode then decode, kernels only, data resident in HBM)",
        "value": value, "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
        "dtype": "u16 coder state / u32 intermediates", "data": "synthetic",
        "config": {"workload": work, "parallelism": f"packet-sharded x{world}"},
        "n_ranks_seen": n_ranks_seen,
        # rank 0's kernel rate times the number of GPUs (shards are equal to within 64 packets)
        "encode_GBps": shard_bytes * world / (enc_ms * 1e-3) / 1e9,
        "decode_GBps": shard_bytes * world / (dec_ms * 1e-3) / 1e9,
        "encode_ms": enc_ms, "decode_ms": dec_ms,
        "encode_read_frac_of_hbm_peak": shard_bytes / (enc_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
        "compression_ratio": (c_total + 20) / total_bytes,
        "roundtrip_equal": all_ok, "md5_sample_match": md5_in == md5_out, "md5_sample": md5_in,
        "oracle_prefix_match": oracle_ok, "device_status": status,
        "roofline": dict(roof(dec_ms if dom == "decode" else enc_ms, dom), kernel=f"{dom}_kernel"),
        "roofline_encode": roof(enc_ms, "encode"), "roofline_decode": roof(dec_ms, "decode"),
        "traffic_source": traffic.get("source"),
    }


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # no launcher around us: become the launcher (children only; nothing here has touched the GPU)
        raise SystemExit(self_launch(argv, args.gpus))

    import torch
    import torch.distributed as dist
    from gpuar_amd import hip as H

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
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
    offset, n = plan_shard(args, world, rank)
    if n == 0:
        raise SystemExit(f"rank {rank} has no packets: too little data for {world} GPUs")
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
        k = min(64, npk)
        host = d_in[:min(n, k * H.PACKET)].cpu().numpy()
        want = O.best().encode_stream(host)
        got = d_stream[:int(d_off[k].item())].cpu().numpy()
        oracle_ok = bool(got.size == want.size and (got == want).all())
    del d_stream

    # one row per rank: [round trip ok, compressed bytes, shard bytes, 1]; the last column counts the ranks
    # the collective really saw
    ok_flags = torch.tensor([int(roundtrip_equal and status == 0 and md5_in == md5_out), c_bytes, n, 1], dtype=torch.int64, device=ctl_dev)
    if world > 1:
        all_flags = [torch.zeros_like(ok_flags) for _ in range(world)]
        dist.all_gather(all_flags, ok_flags)
    else:
        all_flags = [ok_flags]
    all_ok = all(int(f[0].item()) == 1 for f in all_flags)
    c_total = sum(int(f[1].item()) for f in all_flags)
    total_bytes = sum(int(f[2].item()) for f in all_flags)
    n_ranks_seen = sum(int(f[3].item()) for f in all_flags)

    result = None
    if rank == 0:
        traffic = load_profiled_traffic(args.kind, n)
        result = assemble_result(args, world, n_ranks_seen, n, total_bytes, npk, elapsed, enc_avg, dec_avg, c_bytes, c_total,
                                 all_ok, md5_in, md5_out, oracle_ok, status, traffic)

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
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0 and not all_ok:
        raise SystemExit("round trip FAILED")


if __name__ == "__main__":
    main()
