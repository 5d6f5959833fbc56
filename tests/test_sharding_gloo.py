"""Multi-GPU path on CPU: world_size-2 `gloo` processes shard one stream by packet
ranges, each codes its own range, the segments concatenated in rank order must equal
the single-process stream byte for byte -- and the timing/size reductions bench.py uses
(MAX over ranks, all_gather of segment sizes) must work.  The codec here is the oracle,
standing in for the device (this test is about the sharding, not the kernels)."""
import hashlib
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gpuar_amd import sharding, synth


def test_plan_shards_properties():
    for n in [0, 1, 8191, 8192, 8193, 64 * 8192, 64 * 8192 + 1, 1000003, 1 << 30]:
        for world in [1, 2, 3, 4, 8]:
            plan = sharding.plan_shards(n, world)
            assert len(plan) == world and sum(l for _, l in plan) == n
            at = 0
            for off, length in plan:
                assert off % 8192 == 0 and (length == 0 or off == at)
                if length:
                    at = off + length
            # every shard but the last non-empty one is a whole number of wavefronts of packets
            nonempty = [(o, l) for o, l in plan if l]
            for o, l in nonempty[:-1]:
                assert l % (64 * 8192) == 0
    assert sharding.weak_shard(8 << 30, 3) == (3 * (8 << 30), 8 << 30)


def _worker(rank, world, port, n, kind, seed, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as O
    off, length = sharding.plan_shards(n, world)[rank]
    shard = synth.generate(kind, seed, length, offset=off)            # generated independently per rank
    seg = O.PortOracle().encode_stream(shard)
    sizes = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([seg.size], dtype=torch.int64))
    t = torch.tensor([0.5 + rank], dtype=torch.float64)               # stand-in for the per-rank elapsed time
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.barrier()
    q.put((rank, seg.tobytes(), [int(s.item()) for s in sizes], float(t.item())))
    dist.destroy_process_group()


@pytest.mark.parametrize("kind,n", [("text", 200 * 8192 + 77), ("uniform", 130 * 8192)])
def test_two_rank_segments_concatenate_to_the_whole_stream(kind, n):
    from oracle import oracle as O
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, kind, 9, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    whole = O.PortOracle().encode_stream(synth.generate(kind, 9, n))
    joined = sharding.concat_segments(seg for _, seg, _, _ in got)
    assert hashlib.md5(joined).hexdigest() == hashlib.md5(whole.tobytes()).hexdigest()
    assert got[0][2] == got[1][2] == [len(got[0][1]), len(got[1][1])]     # all_gather of segment sizes
    assert got[0][3] == got[1][3] == 1.5                                  # MAX over ranks


def _worker8(rank, world, port, q):
    """One of EIGHT ranks of bench.py's control plane on CPU: bench.plan_shard for both scaling modes, bench.Control.gather_rows
    (the all_gather of one row per rank), its barrier and the MAX of the elapsed time -- the code paths `bench.py --gpus 8` takes around
    its kernels, here with the oracle standing in for the device on a small shard."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    from oracle import oracle as O
    cpu = torch.device("cpu")
    total = 8 * 64 * 8192 + 5000                                      # configs[3] in miniature: one stream over 8 ranks
    args = bench.parse_args(["--gpus", "8", "--scaling", "strong", "--total-gib", str(total / (1 << 30))])
    off, n = sharding.plan_shards(total, world)[rank]
    weak = bench.plan_shard(bench.parse_args(["--gpus", "8", "--gib-per-gpu", "0.125"]), world, rank)
    seg = O.PortOracle().encode_stream(synth.generate("uniform", 42, n, offset=off))
    ctl = bench.Control(dist, world, rank, cpu, True, "gloo", device_sync=lambda: None)      # (no device here: the barrier's GPU sync is a no-op)
    rows = ctl.gather_rows([1, int(seg.size), n, 1, 1000 + rank, 2000 + rank])
    slowest = ctl.max_over_ranks(0.25 * (rank + 1))
    fewest = ctl.min_over_ranks(100 + rank)
    ctl.barrier()
    rep = ctl.report()
    assert rep["backend"] == "gloo" and rep["world"] == world and rep["calls"] == {"barrier": 1, "all_reduce_max": 1, "all_reduce_min": 1, "all_gather": 1}
    assert fewest == 100
    line = None
    if rank == 0:
        # the exact object and line rank 0 of `bench.py --gpus 8` makes of these rows: per-rank figures, the other scaling mode
        # and the gather probe hung on it, cut to the stdout line by driver_line() and held to the driver's size rule
        import sys
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        import bench_stub
        full = bench_stub.full_result(world, "strong", rows=rows)
        full["collectives"] = rep
        line = bench.driver_line(full, "bench_detail.json")
        bench_stub.check_line(line, world)
    q.put((rank, rows, weak, slowest, hashlib.md5(seg.tobytes()).hexdigest(), args.scaling, line))
    dist.destroy_process_group()


def test_eight_rank_control_plane_on_cpu():
    """No 8-GPU node is available to the build (and a one-GPU box admits six GPU processes at most), so the eight-rank
    flow of bench.py is rehearsed here over gloo: every rank sees eight rows in rank order, the shards tile the stream,
    the MAX of the times is the slowest rank's, and the segments concatenate to the single-process stream."""
    from oracle import oracle as O
    world = 8
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker8, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    total = 8 * 64 * 8192 + 5000
    rows0 = got[0][1]
    assert all(g[1] == rows0 for g in got) and len(rows0) == world
    assert sum(r[3] for r in rows0) == world                           # n_ranks_seen
    assert sum(r[2] for r in rows0) == total and [r[4] for r in rows0] == [1000 + k for k in range(world)]
    assert [g[2] for g in got] == [(k << 27, 1 << 27) for k in range(world)]      # weak shards: rank r owns [r*B, (r+1)*B)
    assert all(g[3] == 2.0 for g in got)                               # MAX over ranks
    whole = O.PortOracle().encode_stream(synth.generate("uniform", 42, total))
    assert sum(r[1] for r in rows0) == whole.size                      # the segment sizes add up to the whole stream's
    # rank 0's stdout line, made of the rows the eight ranks really exchanged: it fits the driver's capture, is seen through
    # the driver's tail behind the launcher's chatter, and carries eight ranks' worth of figures as min/max, not as arrays
    import json
    import bench_stub
    line = got[0][6]
    assert line is not None and all(g[6] is None for g in got[1:])
    d = bench_stub.line_from_driver_tail("launcher chatter\n" * 500 + line + "\n", "warning\n" * 20)
    assert d["n_gpus"] == 8 and d["n_ranks_seen"] == 8 and d["scaling"] == "strong" and len(line) <= 4096
    assert d["per_rank"] == {"encode_ms_min": 1.0, "encode_ms_max": 1.007, "decode_ms_min": 2.0, "decode_ms_max": 2.007}
    assert d["collectives"] == {"backend": "gloo", "calls": 4}
    assert abs(d["compression_ratio"] - (whole.size + 20) / total) < 1e-6
