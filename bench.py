#!/usr/bin/env python3
"""bench.py -- encode+decode throughput of the GPUAR packet codec on MI355X.

Contract: `python bench.py --gpus N --steps K --warmup W` prints ONE JSON line
on rank 0 -- at most 4 KB, numbers only (driver_line()); the full object with
every roofline, the provenance of each counter and the per-rank arrays goes to
bench_detail.json next to this file, which the line names.  For N > 1 there is one process per GPU: either the caller starts
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
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# RCCL between processes needs dmabuf IPC on this pool (already exported there; harmless to repeat)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

GIB = 1 << 30
HBM_PEAK_GBPS = 8000.0      # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
KERNEL_SOURCES = ("gpuar_amd/csrc/gpuar_kernels.hip", "gpuar_amd/csrc/lane_codec.h")
LINE_LIMIT = 4096                   # bytes: the driver keeps ~8 KB of stdout + stderr; round 5's 22 KB line fell out of it
DETAIL_FILE = "bench_detail.json"   # everything the line leaves out (prose provenance, every roofline object, per-rank arrays)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--kind", default="uniform", choices=["uniform", "zipf", "text"])
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--gib-per-gpu", type=float, default=8.0, help="weak scaling: GiB each GPU codes")
    ap.add_argument("--total-gib", type=float, default=8.0, help="strong scaling: GiB split over all GPUs")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--extras-timeout", type=float, default=240.0,
                    help="N > 1: seconds the other-scaling pass and the gather probe may take before the line is printed without them")
    ap.add_argument("--no-small-config", action="store_true")
    ap.add_argument("--cpu-sample-mib", type=int, default=16, help="single-core CPU baseline sample")
    ap.add_argument("--no-scaling-extras", action="store_true",
                    help="N > 1: skip the other scaling mode's pass and the gather probe")
    ap.add_argument("--no-by-kind", action="store_true", help="N = 1: skip the text(1) and zipf(1) passes (BASELINE configs[2], [4])")
    ap.add_argument("--by-kind-steps", type=int, default=5, help="timed steps of each of the by-kind passes")
    ap.add_argument("--init-timeout", type=float, default=180.0,
                    help="seconds init_process_group and the first barrier may take before the run gives up with one clear line")
    ap.add_argument("--force-collectives", action="store_true",
                    help="route barrier / MAX / all_gather through torch.distributed (RCCL) even at world size 1: the RCCL preflight")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="N = 1: do not start the two rocprofv3 --pmc child runs that measure the coder kernels' HBM bytes on this box")
    ap.add_argument("--live-traffic-timeout", type=float, default=150.0, help="seconds each of the two counter passes may take")
    ap.add_argument("--detail-file", default=None,
                    help=f"where the full result object goes (default: {DETAIL_FILE} next to bench.py, and a copy under gpurun_out/ when that exists)")
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


def device_count_error(n_ranks, n_devices, oversubscribe):
    """The one line `--gpus N` dies with when the box has fewer GPUs than ranks (None: fine)."""
    if oversubscribe or n_ranks <= n_devices:
        return None
    return (f"bench.py: --gpus {n_ranks} asks for {n_ranks} ranks, one per GPU, but this node has {n_devices} visible GPU(s) "
            "(torch.cuda.device_count()); nothing was started.  (Tests that rehearse N ranks on one GPU set "
            "GPUAR_OVERSUBSCRIBE_DEVICES=1.)")


class Watchdog:
    """A timer around a step that may hang in a transport: when it fires, `line()` is printed and the process exits `code`."""

    def __init__(self, seconds, line, code):
        def fire():
            text = line() if callable(line) else line
            # stdout carries rank 0's JSON line and nothing else: any other line goes to stderr whatever the exit code
            print(text, file=sys.stdout if text.startswith("{") else sys.stderr, flush=True)
            os._exit(code)
        self.t = threading.Timer(seconds, fire)
        self.t.daemon = True
        self.t.start()

    def cancel(self):
        self.t.cancel()


class Control:
    """The control plane of a run: barrier, MAX over ranks, one int64 row per rank.  With one rank and no
    --force-collectives it is plain Python; otherwise every call goes through torch.distributed -- RCCL ("nccl") on a real
    run, gloo when GPUAR_OVERSUBSCRIBE_DEVICES=1 lets several ranks share one GPU in tests."""

    def __init__(self, dist, world, rank, ctl_dev, active, backend, device_sync=None):
        self.dist, self.world, self.rank, self.ctl_dev, self.active, self.backend = dist, world, rank, ctl_dev, active, backend
        self.device_sync = device_sync      # None: torch.cuda.synchronize (the CPU rehearsal of the control plane passes a no-op)
        self.on_rccl = active and backend == "nccl"
        self.calls = {"barrier": 0, "all_reduce_max": 0, "all_reduce_min": 0, "all_gather": 0}

    def barrier(self):
        import torch
        if self.active:
            self.dist.barrier()
            self.calls["barrier"] += 1
        (self.device_sync or torch.cuda.synchronize)()

    def _reduce(self, value, dtype, op, name):
        import torch
        if not self.active:
            return value
        t = torch.tensor([value], dtype=dtype, device=self.ctl_dev)
        self.dist.all_reduce(t, op=op)
        self.calls[name] += 1
        return t.item()

    def max_over_ranks(self, seconds):
        import torch
        return float(self._reduce(seconds, torch.float64, self.dist.ReduceOp.MAX if self.active else None, "all_reduce_max"))

    def min_over_ranks(self, count):
        import torch
        return int(self._reduce(count, torch.int64, self.dist.ReduceOp.MIN if self.active else None, "all_reduce_min"))

    def gather_rows(self, row):
        """all_gather of one int64 row per rank (rank order)."""
        import torch
        mine = torch.tensor(row, dtype=torch.int64, device=self.ctl_dev)
        if not self.active:
            return [mine.tolist()]
        rows = [torch.zeros_like(mine) for _ in range(self.world)]
        self.dist.all_gather(rows, mine)
        self.calls["all_gather"] += 1
        return [r.tolist() for r in rows]

    def report(self):
        return {"backend": self.backend if self.active else None, "world": self.world, "through_torch_distributed": self.active,
                "calls": dict(self.calls)}


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
    """HBM bytes per launch (and the issue-side counters) from the newest profiles/*_traffic*.json taken on THIS
    workload -- the PMC passes of rocprofv3 (tools/prof.sh + tools/traffic_from_prof.py); counters cannot be
    collected from inside this process.  A record is used only if it was taken on this workload size and stream kind
    AND on the kernel sources that are built now (sha256 stamp); otherwise traffic is null and `traffic_source`
    says why."""
    import glob
    files = sorted(glob.glob(os.path.join(root, "profiles", "*_traffic*.json")))
    if not files:
        return {"source": "no profiles/*_traffic*.json"}
    why = []
    stamp = stamp or kernel_source_stamp(root)
    for path in reversed(files):
        name = os.path.basename(path)
        try:
            t = json.load(open(path))
        except Exception as e:                      # noqa: BLE001 -- a damaged record is reported, not fatal
            why.append(f"{name}: unreadable ({e})")
            continue
        if t.get("kind", "uniform") != kind or abs(t.get("input_gib", 0) * GIB - n_bytes) > 1:
            why.append(f"{name}: taken on {t.get('kind', 'uniform')} {t.get('input_gib')} GiB, not this workload")
            continue
        if t.get("kernel_source_sha256_16") != stamp:
            why.append(f"{name}: STALE -- taken on kernel sources {t.get('kernel_source_sha256_16')}, built sources are "
                       f"{stamp}; re-run tools/refresh_profiles.sh")
            continue
        out = {"source": name + ": " + t.get("source", "")}
        for k in ("encode", "decode", "decode_stream", "gather"):
            if k in t:
                out[k] = t[k]
        return out
    # the reason that concerns THIS workload first (a stale record of it), then the others
    why.sort(key=lambda w: 0 if "STALE" in w else 1)
    return {"source": "; ".join(why[:3])}


LIVE_PASSES = (("FETCH_SIZE", "GRBM_GUI_ACTIVE"), ("WRITE_SIZE",),
               ("SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU", "SQ_INSTS_LDS"))


def live_traffic(kind, seed, n_bytes, timeout_s, root=ROOT):
    """HBM bytes per launch of the two coder kernels -- and their issue-side counters -- MEASURED ON THIS BOX in this run: fresh
    child processes `rocprofv3 --kernel-trace --pmc <counters> -- python3 tools/prof_run.py ...`, one per entry of LIVE_PASSES
    (FETCH_SIZE and WRITE_SIZE do not fit one pass, /opt/skills/guides/MI355X_MICROARCH.md; the SQ counters are a third; the third
    may fail without costing the first two), on the same stream kind and size as the timed pass, after it and with its buffers freed.  The program itself follows `--` (no shell, no env, no exec of this GPU-holding process); each pass
    has its own timer and any failure -- no rocprofv3, a time-out, a CSV that does not parse -- returns {"error": ...} and the
    line falls back to the replayed record.  Units and corrections as tools/traffic_from_prof.py: KiB, and on gfx950 FETCH_SIZE
    tallies 128-byte requests at 64 bytes (doubled here); WRITE_SIZE is exact for 16-byte-per-lane stores."""
    import csv
    import glob
    import shutil
    import tempfile
    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(prof):
        return {"error": "no rocprofv3 on this box"}
    if seed != 42:
        return {"error": "tools/prof_run.py generates seed 42 only"}
    gib = n_bytes / GIB
    sums, counts = {}, {}
    t0 = time.perf_counter()
    with tempfile.TemporaryDirectory(dir="/tmp") as d:
        issue_error = None
        for n_pass, counters in enumerate(LIVE_PASSES):
            out = os.path.join(d, f"pass{n_pass}")
            cmd = [prof, "--kernel-trace", "--pmc", *counters, "--output-format", "csv", "-d", out, "--",
                   sys.executable, os.path.join(root, "tools", "prof_run.py"), "--gib", repr(gib), "--kind", kind, "--only", "both", "--reps", "1"]
            why = None
            try:
                r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, timeout=timeout_s)
                if r.returncode != 0 or "prof_run ok" not in r.stdout:
                    why = f"{counters[0]} pass: rc {r.returncode}: {(r.stderr or r.stdout)[-200:]}"
            except (subprocess.TimeoutExpired, OSError) as e:
                why = f"{counters[0]} pass: {type(e).__name__}"
            if why:
                if n_pass < 2:
                    return {"error": why}
                issue_error = why                     # the traffic passes are in: the line quotes them, and says what became of this one
                break
            for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    name = row.get("Kernel_Name", "")
                    key = "encode" if ("encode_kernel" in name or "encode_small_kernel" in name) else "decode" if "decode_slots_kernel" in name else None
                    counter = row.get("Counter_Name")
                    if key and counter in counters:
                        sums[key, counter] = sums.get((key, counter), 0.0) + float(row["Counter_Value"])
                        counts[key, counter] = counts.get((key, counter), 0) + 1
    res = {"source": "live: rocprofv3 --pmc child runs of tools/prof_run.py (FETCH_SIZE + GRBM_GUI_ACTIVE | WRITE_SIZE | SQ_*) behind the timed "
                     "pass, on this box", "seconds": time.perf_counter() - t0, "fetch_correction": 2.0, "n_bytes": n_bytes}
    if issue_error:
        res["issue_counters_error"] = issue_error
    symbol_steps = n_bytes / 64.0                       # one step = 64 lanes x one byte each
    for key in ("encode", "decode"):
        if (key, "FETCH_SIZE") not in sums or (key, "WRITE_SIZE") not in sums:
            return {"error": f"no {key} kernel rows in the counter files"}
        avg = lambda c, key=key: sums[key, c] / counts[key, c]          # noqa: E731
        f, w = avg("FETCH_SIZE") * 1024.0, avg("WRITE_SIZE") * 1024.0
        res[key] = {"fetch_size_bytes_raw": f, "write_size_bytes": w, "hbm_bytes_per_launch": 2.0 * f + w, "launches": counts[key, "FETCH_SIZE"]}
        if (key, "SQ_INSTS_VALU") in sums:              # the same derivations as tools/traffic_from_prof.py
            res[key].update({"valu_insts_per_symbol_step": avg("SQ_INSTS_VALU") / symbol_steps,
                             "lds_insts_per_symbol_step": avg("SQ_INSTS_LDS") / symbol_steps,
                             "valu_busy": avg("SQ_ACTIVE_INST_VALU") / avg("SQ_WAVE_CYCLES"),
                             "wait_frac": avg("SQ_WAIT_ANY") / avg("SQ_WAVE_CYCLES")})
            if (key, "GRBM_GUI_ACTIVE") in sums:        # (rocprofv3 sums GRBM_GUI_ACTIVE over the 8 XCDs; SQ_* are in quad-cycles)
                res[key]["valu_busy_per_simd"] = avg("SQ_ACTIVE_INST_VALU") * 4.0 / (avg("GRBM_GUI_ACTIVE") / 8.0 * 1024.0)
    return res


ISSUE_COUNTERS = ("valu_insts_per_symbol_step", "lds_insts_per_symbol_step", "valu_busy", "valu_busy_per_simd", "wait_frac")


def apply_live_traffic(result, live, machine=None):
    """Puts counters measured in this run into the three coder rooflines and says so in one word each (`traffic_from`,
    `counters_from`): HBM bytes always, the issue-side counters (instructions per symbol step, vector-pipe busy, wait share) and
    the vector roof made of them when the third pass came in; what did not come in stays the replayed record's, labelled as before."""
    result["traffic_live"] = live
    if "error" in live:
        return False
    dom = "decode" if result["roofline"].get("kernel", "").startswith("decode") else "encode"
    for key, which in (("roofline_encode", "encode"), ("roofline_decode", "decode"), ("roofline", dom)):
        r, got = result[key], live[which]
        r["traffic"], r["traffic_from"] = got["hbm_bytes_per_launch"], "live"
        if "valu_insts_per_symbol_step" in got:
            for k in ISSUE_COUNTERS:
                if k in got:
                    r[k] = got[k]
            r["counters_from"] = "live"
            old = r.get("roofline_valu") or {}
            ms = r["algorithmic_bytes_per_launch"] / (r["achieved"] * 1e9) * 1e3          # the launch time the HBM figure was made of
            n_bytes = live.get("n_bytes")
            if n_bytes:
                r["roofline_valu"] = valu_roof(got["valu_insts_per_symbol_step"], n_bytes, ms,
                                               measured_mhz=old.get("shader_clock_measured_MHz") or r.get("shader_clock_measured_MHz"), **(machine or {}))
                r["roofline_valu"]["lane_ops_per_byte_from"] = "SQ_INSTS_VALU per symbol step of a wavefront (64 lanes, 64 bytes), counted in this run"
    return True


def usable_cpus():
    """Host CPUs this process may really use: the affinity mask, cut down by the cgroup's CPU quota
    (a container on a 256-thread host is typically given a share of it; os.cpu_count() says 256 anyway)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    how = "sched_getaffinity"
    try:                                                     # cgroup v2
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            q = max(1, int(int(quota) / int(period) + 0.5))
            if q < n:
                n, how = q, "cgroup cpu.max"
    except (OSError, ValueError):
        try:                                                 # cgroup v1
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0 and max(1, int(quota / period + 0.5)) < n:
                n, how = max(1, int(quota / period + 0.5)), "cgroup cfs_quota"
        except (OSError, ValueError):
            pass
    return max(1, n), how


def cpu_baseline(kind, seed, sample_bytes):
    """The CPU beside the GPU numbers (a reported baseline, not a target).
    Row 1: the reference codec (oracle/_ref: the reference's own arCompress/arDecompress; the C port if that build
    is absent) on ONE host core, timed the way the reference times --host (model init + codec call only,
    src/cpu_compressor.cpp:58-62,157-161).  Row 2 (`all_cores`): the same codec with the packets fanned out over
    the host's usable cores by NATIVE threads (oracle/ref_driver.cpp, one contiguous packet range per thread) --
    packets are independent, so this is the fair "all of the host" ceiling; `cores` is the affinity mask cut down
    by the cgroup quota, not os.cpu_count().  Row 3 (`product_host`): this repository's own `gpuar-host --host
    --threads=0` on a file of the same stream, wall time."""
    from gpuar_amd import synth
    from oracle import oracle as O
    codec = O.require_best()
    data = synth.generate(kind, seed, sample_bytes)
    t0 = time.perf_counter()
    stream = codec.encode_stream(data)
    t1 = time.perf_counter()
    back = codec.decode_stream(stream, data.size)
    t2 = time.perf_counter()
    ok = bool((back == data).all())
    enc, dec = data.size / (t1 - t0) / 1e9, data.size / (t2 - t1) / 1e9
    cores, how = usable_cpus()
    res = {
        "value": data.size / (t2 - t0) / 1e9, "unit": "GB/s", "cores": 1, "kind": codec.kind, "sample_mib": sample_bytes >> 20,
        "sample": f"first {sample_bytes >> 20} MiB of the same {kind}({seed}) stream, encode then decode, 1 thread",
        "encode_GBps": enc, "decode_GBps": dec, "roundtrip_ok": ok,
        "host_cpus_online": os.cpu_count(), "host_cpus_usable": cores, "host_cpus_usable_from": how,
    }
    if hasattr(codec, "encode_stream_mt"):
        per = 8 << 20
        big = synth.generate(kind, seed, cores * per)
        t0 = time.perf_counter()
        streams = codec.encode_stream_mt(big, cores)
        t1 = time.perf_counter()
        back = codec.decode_stream_mt(streams, big.size, cores)
        t2 = time.perf_counter()
        res["all_cores"] = {
            "cores": cores, "threads": "native (oracle/ref_driver.cpp), one contiguous packet range per thread",
            "sample": f"first {cores} x {per >> 20} MiB of the same stream",
            "value": big.size / (t2 - t0) / 1e9, "encode_GBps": big.size / (t1 - t0) / 1e9, "decode_GBps": big.size / (t2 - t1) / 1e9,
            "roundtrip_ok": bool((back == big).all()),
        }
        res["all_cores"]["speedup_over_one_core"] = res["all_cores"]["value"] / res["value"]
    res["product_host"] = product_host_baseline(kind, seed, cores)
    return res


def product_host_baseline(kind, seed, cores, mib_per_core=8):
    """This repository's own --host path (gpuar-host: lane_codec.h on the CPU, --threads=0 = all cores), wall time of
    the CLI on a file in the page cache -- file I/O included, as a user of the CLI sees it."""
    import tempfile
    from gpuar_amd import synth
    exe = os.path.join(ROOT, "gpuar_amd", "bin", "gpuar-host")
    if not os.path.exists(exe):
        return {"skipped": "gpuar_amd/bin/gpuar-host not built"}
    n = cores * (mib_per_core << 20)
    with tempfile.TemporaryDirectory() as d:
        src, gip, back = (os.path.join(d, x) for x in ("in.dat", "out.gip", "back.dat"))
        synth.generate(kind, seed, n).tofile(src)
        t0 = time.perf_counter()
        c = subprocess.run([exe, "c", "--host", "--threads=0", f"--in={src}", f"--out={gip}"], capture_output=True, text=True)
        t1 = time.perf_counter()
        dd = subprocess.run([exe, "d", "--host", "--threads=0", f"--in={gip}", f"--out={back}"], capture_output=True, text=True)
        t2 = time.perf_counter()
        ok = c.returncode == 0 and dd.returncode == 0 and open(back, "rb").read() == open(src, "rb").read()
    return {"what": "gpuar-host c|d --host --threads=0 (this repository's CPU path), CLI wall time incl. file I/O",
            "cores": cores, "sample": f"{n >> 20} MiB of the same stream", "value": n / (t2 - t0) / 1e9, "unit": "GB/s",
            "encode_GBps": n / (t1 - t0) / 1e9, "decode_GBps": n / (t2 - t1) / 1e9, "roundtrip_ok": ok}


def hbm_roof(bytes_per_launch, ms):
    a = bytes_per_launch / (ms * 1e-3) / 1e9
    return {"bound": "hbm", "achieved": a, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": a / HBM_PEAK_GBPS,
            "algorithmic_bytes_per_launch": bytes_per_launch, "traffic": None}


def measure_copy_peak(H, d_src, d_dst, n_bytes, reps=10):
    """The practical HBM roof on THIS box, measured in THIS run ("confirm a practical peak on the
    box with a device-to-device copy ... and report both"): a plain 16-byte-per-lane copy (gpuar_hip_copy) of the
    workload's own input buffer into its output buffer, `reps` launches timed with HIP events after two untimed ones;
    the rate counts the bytes read plus the bytes written."""
    n = n_bytes // 16 * 16
    for _ in range(2):
        H.device_copy(d_src, d_dst, n)
    ms = timed_kernel_ms(lambda: H.device_copy(d_src, d_dst, n), reps)
    best, avg = min(ms), sum(ms) / len(ms)
    return {"GBps": 2 * n / (avg * 1e-3) / 1e9, "best_GBps": 2 * n / (best * 1e-3) / 1e9, "ms_avg": avg, "reps": reps,
            "bytes_copied": n, "kernel": "copy_kernel (gpuar_hip_copy): 16 B per lane, one quad per thread, read + write counted",
            "frac_of_datasheet_peak": 2 * n / (avg * 1e-3) / 1e9 / HBM_PEAK_GBPS}


def annotate_roofs(result, copy_peak, traffic_source=None):
    """Every roofline* object of the line (the by_kind entries' too) gets the roof measured in this run next to the
    datasheet's, and says which of its fields were measured live and which are replayed from a stamped profile record."""
    replayed = ("traffic", "valu_busy", "valu_busy_per_simd", "wait_frac", "valu_insts_per_symbol_step", "lds_insts_per_symbol_step")
    source = traffic_source if traffic_source is not None else result.get("traffic_source")
    for key, r in result.items():
        if key == "by_kind" and isinstance(r, dict):
            for entry in r.values():
                annotate_roofs(entry, copy_peak)
            continue
        if not (key.startswith("roofline") and isinstance(r, dict)):
            continue
        if copy_peak:
            r["peak_measured_copy"] = copy_peak["GBps"]
            r["frac_of_measured"] = r["achieved"] / copy_peak["GBps"]
        r["measured_live"] = ("achieved, frac, peak_measured_copy, frac_of_measured (HIP events in this run)"
                              + ("; roofline_valu.shader_clock_measured_MHz (the kernel's own workgroups)" if "roofline_valu" in r else ""))
        have = [k for k in replayed if r.get(k) is not None and not (k == "traffic" and r.get("traffic_from") == "live")
                and not (k != "traffic" and r.get("counters_from") == "live")]
        if r.get("counters_from") == "live":
            r["measured_live"] += "; instructions per symbol step, vector-pipe busy, wait share and the vector roof made of them (a third --pmc pass)"
        if r.get("traffic_from") == "live":
            r["measured_live"] += "; traffic (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE child runs behind the timed pass, on this box)"
        if have:
            r["counters"] = (f"{', '.join(have)}: replayed from profiles/{source} -- rocprofv3 PMC passes cannot run inside "
                             "this process; the record is quoted only while its kernel-source stamp matches the built sources")
        else:
            r["counters"] = "none quoted (no profile record taken on this workload and these kernel sources)"


def side_kernels(H, d_in, d_slots, d_out, npk, n, reps):
    """The kernels either side of the two coder kernels, timed like them (HIP events on the launch stream,
    untimed region): the device-side compaction (scan + gather: the one HBM-bound piece, SURVEY.md 8(d) "include its
    time in t_kernel"), encode + compaction back to back (the device side of `gpuar c`), and decode_stream_kernel --
    the decoder `gpuar d` really runs, reading the compacted stream instead of the slots."""
    import torch
    d_stream = torch.empty(npk * H.SLOT + 16, dtype=torch.uint8, device=d_in.device)
    d_off = torch.empty(npk + 1, dtype=torch.int64, device=d_in.device)
    compact_ms = timed_kernel_ms(lambda: H.compact(d_slots, npk, d_stream, d_off), reps)

    def both():
        H.encode(d_in, d_slots)
        H.compact(d_slots, npk, d_stream, d_off)

    enc_compact_ms = timed_kernel_ms(both, reps)
    d_out.zero_()
    dstream_ms = timed_kernel_ms(lambda: H.decode_stream(d_stream, d_off, npk, d_out), reps)
    torch.cuda.synchronize()
    c = int(d_off[-1].item())
    ok = bool(torch.equal(d_out[:n], d_in))
    avg = lambda v: sum(v) / len(v)                                                   # noqa: E731
    out = {
        "compact_ms": avg(compact_ms), "encode_plus_compact_ms": avg(enc_compact_ms), "decode_stream_ms": avg(dstream_ms),
        "decode_stream_roundtrip_equal": ok,
        # compaction reads the C defined bytes of the slots and writes them once (+ 2 bytes of header per packet for the scan)
        "roofline_compact": dict(hbm_roof(2 * c + 2 * npk, avg(compact_ms)), kernel="scan_* + gather_kernel"),
        "roofline_decode_stream": dict(hbm_roof(n + c + 8 * npk, avg(dstream_ms)), kernel="decode_stream_kernel"),
        "encode_plus_compact_GBps": n / (avg(enc_compact_ms) * 1e-3) / 1e9,
        "decode_stream_GBps": n / (avg(dstream_ms) * 1e-3) / 1e9,
    }
    return out, d_stream, d_off, c


def gather_probe(ctl, d_stream, c_bytes):
    """north_star: "RCCL over xGMI only if a gather is measurably cheaper than staged hipMemcpyAsync" -- measured,
    on a bounded sample of every rank's compacted segment:
      staged  : every rank copies its segment to its OWN pinned host buffer over its own PCIe link, all at once
      gathered: every rank sends its segment to rank 0 (RCCL send/recv over xGMI), rank 0 copies the whole to the host
    Both leave all segments in host memory, ready for the ordered write.  Times are the MAX over ranks."""
    import torch
    dist, world, rank, on_rccl = ctl.dist, ctl.world, ctl.rank, ctl.on_rccl
    sample = ctl.min_over_ranks(min(c_bytes, 256 << 20)) // 4096 * 4096
    seg = d_stream[:sample]
    dev = seg.device
    h_own = torch.empty(sample, dtype=torch.uint8, pin_memory=True)
    h_all = torch.empty(sample * world, dtype=torch.uint8, pin_memory=True) if rank == 0 else None
    d_all = torch.empty(sample * world, dtype=torch.uint8, device=dev) if rank == 0 else None

    def sync():
        torch.cuda.synchronize()
        ctl.barrier()

    max_over_ranks = ctl.max_over_ranks

    def staged():
        h_own.copy_(seg, non_blocking=True)
        torch.cuda.synchronize()

    def gathered():
        if world > 1:
            if on_rccl:
                ops = []
                if rank == 0:
                    for r in range(1, world):
                        ops.append(dist.P2POp(dist.irecv, d_all[r * sample:(r + 1) * sample], r))
                else:
                    ops.append(dist.P2POp(dist.isend, seg, 0))
                for w in dist.batch_isend_irecv(ops):
                    w.wait()
            else:                   # test mode (gloo, ranks share one GPU): host tensors stand in for the xGMI hop
                if rank == 0:
                    for r in range(1, world):
                        buf = torch.empty(sample, dtype=torch.uint8)
                        dist.recv(buf, r)
                        d_all[r * sample:(r + 1) * sample].copy_(buf)
                else:
                    dist.send(seg.cpu(), 0)
        if rank == 0:
            d_all[:sample].copy_(seg)
            h_all.copy_(d_all, non_blocking=True)
        torch.cuda.synchronize()

    res = {}
    for name, fn in (("staged_d2h_ms", staged), ("gather_then_d2h_ms", gathered)):
        fn()                                                   # warm-up (first touch of the pinned pages, RCCL channels)
        best = None
        for _ in range(3):
            sync()
            t0 = time.perf_counter()
            fn()
            sync()
            dt = max_over_ranks(time.perf_counter() - t0)
            best = dt if best is None else min(best, dt)
        res[name] = best * 1e3
    res.update({
        "bytes_per_rank": sample, "ranks": world,
        "transport": "RCCL send/recv to rank 0" if on_rccl else "gloo on host tensors (test mode, ranks share one GPU)",
        "staged_GBps": sample * world / (res["staged_d2h_ms"] * 1e-3) / 1e9,
        "gathered_GBps": sample * world / (res["gather_then_d2h_ms"] * 1e-3) / 1e9,
        "cheaper": "staged hipMemcpyAsync" if res["staged_d2h_ms"] <= res["gather_then_d2h_ms"] else "RCCL gather",
    })
    return res


# What the hot path is really bound by (SURVEY.md section 7 risk 1, section 8(d) "report VALUBusy ... next to the HBM figure"): the
# integer vector pipes.  A gfx950 SIMD retires 16 lanes of one vector instruction per cycle (a wave64 instruction occupies it
# for four), so the chip's roof in lane-operations is CUs x 4 SIMDs x 16 lanes x shader clock; the path's demand is the
# vector instructions one wavefront issues per symbol step (64 lanes advance 64 bytes: instructions per step = lane-ops per
# byte), counted by rocprofv3 (SQ_INSTS_VALU) and replayed from the stamped profile record.
MI355X_CUS = 256
SHADER_CLOCK_GHZ = 2.4              # peak engine clock of the MI355X datasheet (/opt/skills/guides/MI355X_MICROARCH.md); the clock under
                                    # this load was measured at 2.2-2.4 GHz (tools/lat_probe.hip)
SMALL_GROUPS = 512                  # gpuar_hip_encode (GPUAR_MODE_AUTO) takes the latency kernel up to this many groups of 64 packets


def kernel_symbols(n_packets):
    """The symbols rocprofv3 lists for this launch size (profiles/*_kernel_stats.csv)."""
    groups = (n_packets + 63) // 64
    return {"encode": "encode_small_kernel" if groups <= SMALL_GROUPS else "encode_kernel", "decode": "decode_slots_kernel"}


def valu_roof(lane_ops_per_byte, n_bytes, ms, cus=MI355X_CUS, clock_ghz=SHADER_CLOCK_GHZ, measured_mhz=None):
    """roofline_valu: the vector-issue roof.  achieved = lane-ops per byte (replayed counter) x bytes per launch / launch time;
    `frac` is against the data sheet's clock, `frac_at_measured_clock` against the shader clock the kernel's own workgroups
    measured while it ran (s_memtime over s_memrealtime, gpuar_hip_clock_samples) -- under this load the chip does not hold
    its peak clock, so the second figure says how full the vector pipes really were."""
    peak = cus * 4 * 16 * clock_ghz * 1e9
    achieved = lane_ops_per_byte * n_bytes / (ms * 1e-3)
    r = {"bound": "valu", "lane_ops_per_byte": lane_ops_per_byte, "achieved_lane_ops_per_s": achieved, "peak_lane_ops_per_s": peak,
         "frac": achieved / peak, "unit": "lane-ops/s",
         "peak_from": f"{cus} CUs x 4 SIMDs x 16 lanes x {clock_ghz:g} GHz (datasheet peak engine clock)",
         "lane_ops_per_byte_from": "SQ_INSTS_VALU per symbol step of a wavefront (64 lanes, 64 bytes), replayed profile record"}
    if measured_mhz:
        at_measured = cus * 4 * 16 * measured_mhz * 1e6
        r.update({"shader_clock_measured_MHz": measured_mhz, "peak_lane_ops_per_s_at_measured_clock": at_measured,
                  "frac_at_measured_clock": achieved / at_measured,
                  "shader_clock_from": "s_memtime / s_memrealtime x 100 MHz over every 64th workgroup of this kernel's launches in this run"})
    return r


def coder_roof(algo_bytes, n_bytes, ms, counters, kernel, machine=None, measured_mhz=None):
    """A roofline object of one of the two coder kernels: the HBM figure the contract asks for, the counters replayed
    from the stamped profile record, and -- when that record holds the instruction count -- the roof that binds."""
    a = algo_bytes / (ms * 1e-3) / 1e9
    t = counters or {}
    r = {"bound": "hbm", "achieved": a, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": a / HBM_PEAK_GBPS,
         "algorithmic_bytes_per_launch": algo_bytes, "traffic": t.get("hbm_bytes_per_launch"), "kernel": kernel}
    # the roof that actually binds (SURVEY.md section 7 risk 1): vector-issue and wait share of the wavefronts' cycles
    for k in ("valu_busy", "valu_busy_per_simd", "wait_frac", "valu_insts_per_symbol_step", "lds_insts_per_symbol_step"):
        if k in t:
            r[k] = t[k]
    if "valu_insts_per_symbol_step" in t:
        r["roofline_valu"] = valu_roof(t["valu_insts_per_symbol_step"], n_bytes, ms, measured_mhz=measured_mhz, **(machine or {}))
    elif measured_mhz:
        r["shader_clock_measured_MHz"] = measured_mhz
    return r


def assemble_result(args, world, n_ranks_seen, shard_bytes, total_bytes, npk_rank0, elapsed, enc_ms, dec_ms,
                    c_bytes_rank0, c_total, all_ok, md5_in, md5_out, oracle_ok, status, traffic, machine=None, clocks=None, checker=None):
    """The full result object (bench_detail.json; driver_line() cuts the stdout line out of it), from plain numbers (no GPU objects): the driver's contract fields, the roofline of
    the dominant kernel and what was verified.  enc_ms / dec_ms are rank 0's average launch durations over
    its shard of `shard_bytes` bytes; elapsed is the MAX over ranks of the wall time of args.steps steps."""
    ms_per_step = elapsed / args.steps * 1e3
    value = total_bytes * args.steps / elapsed / 1e9
    # dominant kernel = the slower of the two; algorithmic bytes per launch = N read/written + C written/read
    dom = "decode" if dec_ms >= enc_ms else "encode"
    algo_bytes = shard_bytes + c_bytes_rank0
    symbols = kernel_symbols(npk_rank0)
    clocks = clocks or {}

    def roof(ms, which):
        return coder_roof(algo_bytes, shard_bytes, ms, traffic.get(which), symbols[which], machine, clocks.get(which))

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
        "oracle_prefix_match": oracle_ok, "checker": checker, "device_status": status,
        "roofline": roof(dec_ms if dom == "decode" else enc_ms, dom),
        "roofline_encode": roof(enc_ms, "encode"), "roofline_decode": roof(dec_ms, "decode"),
        "traffic_source": traffic.get("source"),
    }


def kind_result(kind, seed, n, steps, elapsed, enc_ms, dec_ms, c_bytes, roundtrip_equal, oracle_ok, status, traffic, machine=None, clocks=None):
    """One entry of `by_kind`: the same hot path on another of BASELINE.json's single-GPU workloads (configs[2] text,
    configs[4]'s stream kind zipf), timed in the same run as the headline pass and assembled from plain numbers."""
    symbols = kernel_symbols((n + 8191) // 8192)
    algo = n + c_bytes
    dom = "decode" if dec_ms >= enc_ms else "encode"
    clocks = clocks or {}
    return {
        "workload": f"{kind}({seed}) {n / GIB:g} GiB on 1 GPU, 8192-byte packets",
        "steps": steps, "ms_per_step": elapsed / steps * 1e3, "value": n * steps / elapsed / 1e9, "unit": "GB/s",
        "encode_ms": enc_ms, "decode_ms": dec_ms,
        "encode_GBps": n / (enc_ms * 1e-3) / 1e9, "decode_GBps": n / (dec_ms * 1e-3) / 1e9,
        "compression_ratio": (c_bytes + 20) / n, "roundtrip_equal": roundtrip_equal, "oracle_prefix_match": oracle_ok,
        "device_status": status,
        "roofline": coder_roof(algo, n, dec_ms if dom == "decode" else enc_ms, traffic.get(dom), symbols[dom], machine, clocks.get(dom)),
        "roofline_encode": coder_roof(algo, n, enc_ms, traffic.get("encode"), symbols["encode"], machine, clocks.get("encode")),
        "roofline_decode": coder_roof(algo, n, dec_ms, traffic.get("decode"), symbols["decode"], machine, clocks.get("decode")),
        "traffic_source": traffic.get("source"),
    }


def _r(x, digits=5):
    """A float cut to `digits` significant digits (the line is for reading; the detail file keeps every digit)."""
    if isinstance(x, bool) or not isinstance(x, float):
        return x
    return float(f"{x:.{digits}g}")


def _slim_roof(r):
    """A roofline object as numbers only: the contract's fields, the roof measured in this run and the vector-issue roof."""
    if not isinstance(r, dict):
        return None
    out = {"bound": r.get("bound"), "achieved": _r(r.get("achieved")), "peak": r.get("peak"), "unit": r.get("unit"), "frac": _r(r.get("frac")),
           "traffic": r.get("traffic"), "kernel": r.get("kernel")}
    if r.get("traffic") is not None:
        out["traffic_from"] = r.get("traffic_from", "replayed")          # one word: "live" (counters of this run) or "replayed"
    if r.get("peak_measured_copy") is not None:
        out["peak_measured_copy"] = _r(r["peak_measured_copy"])
        out["frac_of_measured"] = _r(r.get("frac_of_measured"))
    v = r.get("roofline_valu")
    if isinstance(v, dict):
        out["valu_frac"] = _r(v.get("frac_at_measured_clock", v.get("frac")))
        out["valu_insts_per_step"] = _r(v.get("lane_ops_per_byte"))
        if v.get("shader_clock_measured_MHz"):
            out["clock_MHz"] = _r(v["shader_clock_measured_MHz"])
    for k in ("valu_busy_per_simd", "wait_frac"):
        if r.get(k) is not None:
            out[k] = _r(r[k], 3)
    if out.get("valu_insts_per_step") is not None or out.get("valu_busy_per_simd") is not None:
        out["counters_from"] = r.get("counters_from", "replayed")
    return out


def driver_line(result, detail_name=DETAIL_FILE):
    """The ONE stdout line: the contract's fields and the numbers a reader needs, nothing that is prose or repeated
    (the reference prints a dozen numbers on one screen, src/main.cpp:174-182).  `result` is the full object
    (assemble_result + everything main() hangs on it); what is not copied here is in the detail file the line names.
    Bounded by construction: no per-rank arrays, no provenance strings, two roofline objects."""
    g = result.get
    line = {
        "metric": "encode+decode GB/s", "value": _r(g("value"), 6), "unit": g("unit"), "n_gpus": g("n_gpus"), "steps": g("steps"),
        "warmup": g("warmup"), "ms_per_step": _r(g("ms_per_step"), 6), "higher_is_better": True, "scaling": g("scaling"),
        "vs_baseline": None, "dtype": "u16", "data": "synthetic", "config": g("config"), "n_ranks_seen": g("n_ranks_seen"),
        "encode_GBps": _r(g("encode_GBps")), "decode_GBps": _r(g("decode_GBps")), "encode_ms": _r(g("encode_ms")), "decode_ms": _r(g("decode_ms")),
        "encode_read_frac_of_hbm_peak": _r(g("encode_read_frac_of_hbm_peak")), "compression_ratio": _r(g("compression_ratio"), 7),
        "roundtrip_equal": g("roundtrip_equal"), "md5_sample_match": g("md5_sample_match"), "oracle_prefix_match": g("oracle_prefix_match"),
        "checker": g("checker"), "device_status": g("device_status"),
        "roofline": _slim_roof(g("roofline")), "roofline_encode": _slim_roof(g("roofline_encode")),
    }
    for k in ("compact_ms", "encode_plus_compact_ms", "decode_stream_ms"):
        if g(k) is not None:
            line[k] = _r(g(k))
    if isinstance(g("roofline_compact"), dict):
        line["compact_frac"] = _r(g("roofline_compact").get("frac"))
    c = g("cpu_baseline")
    if isinstance(c, dict):
        cb = {"value": _r(c.get("value")), "unit": c.get("unit"), "cores": c.get("cores"), "kind": c.get("kind"),
              "sample": f"{c.get('sample_mib')} MiB prefix, encode+decode" if c.get("sample_mib") else str(c.get("sample"))[:60],
              "encode_GBps": _r(c.get("encode_GBps")), "decode_GBps": _r(c.get("decode_GBps")), "roundtrip_ok": c.get("roundtrip_ok")}
        for sub in ("all_cores", "product_host"):
            if isinstance(c.get(sub), dict) and "value" in c[sub]:
                cb[sub] = {"cores": c[sub].get("cores"), "value": _r(c[sub]["value"]), "encode_GBps": _r(c[sub].get("encode_GBps")),
                           "decode_GBps": _r(c[sub].get("decode_GBps"))}
        line["cpu_baseline"] = cb
    if isinstance(g("by_kind"), dict):
        line["by_kind"] = {
            kind: {"value": _r(e.get("value")), "encode_ms": _r(e.get("encode_ms")), "decode_ms": _r(e.get("decode_ms")),
                   "ratio": _r(e.get("compression_ratio"), 6), "frac": _r((e.get("roofline") or {}).get("frac")),
                   "ok": bool(e.get("roundtrip_equal") and e.get("oracle_prefix_match") and e.get("device_status") == 0)}
            for kind, e in g("by_kind").items()}
    s = g("small_config")
    if isinstance(s, dict):
        line["small_config"] = {"mib": 64, "encode_GBps": _r(s.get("encode_GBps")), "decode_GBps": _r(s.get("decode_GBps")),
                                "md5_match": s.get("stream_md5") == s.get("reference_stream_md5"), "roundtrip_equal": s.get("roundtrip_equal")}
    if g("n_gpus", 1) > 1 and isinstance(g("per_rank"), dict):
        line["per_rank"] = {k: _r(v) for k, v in g("per_rank").items() if not isinstance(v, (list, dict))}
    o = g("other_scaling")
    if isinstance(o, dict):
        line["other_scaling"] = {"scaling": o.get("scaling"), "value": _r(o.get("value")), "ms_per_step": _r(o.get("ms_per_step")),
                                 "roundtrip_equal": o.get("roundtrip_equal")}
    p = g("gather_probe")
    if isinstance(p, dict):
        line["gather_probe"] = {"staged_d2h_ms": _r(p.get("staged_d2h_ms")), "gather_then_d2h_ms": _r(p.get("gather_then_d2h_ms")),
                                "cheaper": p.get("cheaper"), "rccl": str(p.get("transport", "")).startswith("RCCL")}
    if g("scaling_extras"):
        line["scaling_extras"] = str(g("scaling_extras"))[:160]
    col = g("collectives")
    if isinstance(col, dict):
        line["collectives"] = {"backend": col.get("backend"), "calls": sum((col.get("calls") or {}).values())}
    line["detail"] = detail_name
    text = json.dumps(line, separators=(",", ":"))
    if len(text) > LINE_LIMIT:            # cannot happen with the fields above; if it ever does, the contract's fields survive
        for k in ("collectives", "gather_probe", "other_scaling", "per_rank", "small_config", "by_kind", "compact_frac"):
            line.pop(k, None)
            text = json.dumps(line, separators=(",", ":"))
            if len(text) <= LINE_LIMIT:
                break
    return text


def write_detail(result, path=None, root=ROOT):
    """The full object: to `path` (--detail-file) or, by default, next to bench.py and -- when that directory exists -- under
    gpurun_out/, so a gpurun call brings it home.  Returns the name the line quotes, or None when nothing could be written
    (a read-only tree): the line still prints."""
    targets = [path] if path else [os.path.join(root, DETAIL_FILE)] + (
        [os.path.join(root, "gpurun_out", DETAIL_FILE)] if os.path.isdir(os.path.join(root, "gpurun_out")) else [])
    written = None
    for t in targets:
        try:
            with open(t + ".tmp", "w") as f:
                json.dump(result, f, indent=1)
            os.replace(t + ".tmp", t)
            written = written or (path if path else DETAIL_FILE)
        except OSError:
            pass
    return written


def oracle_prefix_ok(H, d_in, n, npk, d_stream, d_off, packets=64):
    """The first `packets` packets of the compacted stream against the oracle on the same bytes (checker only, untimed)."""
    from oracle import oracle as O
    k = min(packets, npk)
    host = d_in[:min(n, k * H.PACKET)].cpu().numpy()
    want = O.require_best().encode_stream(host)
    got = d_stream[:int(d_off[k].item())].cpu().numpy()
    return bool(got.size == want.size and (got == want).all())


def checker_kind():
    """Which oracle the prefix checks of this run compared with: "reference" (oracle/_ref: the reference's own codec) or "port"."""
    from oracle import oracle as O
    return O.require_best().kind              # raises where the pinned reference binary is expected and missing


def run_pass(args, H, ctl, dev, steps, warmup):
    """One measurement of the hot path under args.scaling: this rank's shard generated on the device, W untimed
    warm-up steps, then exactly `steps` steps (encode kernel + decode kernel) bracketed by barrier +
    torch.cuda.synchronize(), MAX over ranks; then, untimed, the per-kernel durations and the checks of what was
    just timed.  Returns plain numbers plus the device buffers (for the side kernels and the gather probe)."""
    import torch
    world, rank = ctl.world, ctl.rank
    offset, n = plan_shard(args, world, rank)
    if n == 0:
        raise SystemExit(f"rank {rank} has no packets: too little data for {world} GPUs")
    npk = H.packet_count(n)
    d_in = H.generate(args.kind, args.seed, n, offset=offset, device=dev)
    d_slots = torch.empty(npk * H.SLOT, dtype=torch.uint8, device=dev)
    d_out = torch.empty(npk * H.PACKET, dtype=torch.uint8, device=dev)
    word = torch.zeros(1, dtype=torch.int32, device=dev)      # this rank's own status word (include/gpuar_hip.h d_status)

    def encode():
        H.encode(d_in, d_slots, d_status=word)

    def decode():
        H.decode(d_slots, npk, d_out, d_status=word)

    barrier = ctl.barrier                                     # dist.barrier() (when there is a process group) + torch.cuda.synchronize()
    for _ in range(warmup):
        encode()
        decode()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        encode()
        decode()
    barrier()
    elapsed = ctl.max_over_ranks(time.perf_counter() - t0)

    # ---- per-kernel durations (HIP events on the launch stream), untimed region ----
    reps = max(3, min(steps, 10))
    H.shader_clock_mhz("encode")                             # (reset: only the launches timed below are sampled)
    H.shader_clock_mhz("decode")
    enc_ms = timed_kernel_ms(encode, reps)
    dec_ms = timed_kernel_ms(decode, reps)
    enc_avg, dec_avg = sum(enc_ms) / len(enc_ms), sum(dec_ms) / len(dec_ms)
    # the shader clock those very launches ran at, from their own workgroups (None for a launch of the latency kernel,
    # which does not sample)
    clocks = {"encode": H.shader_clock_mhz("encode")[0], "decode": H.shader_clock_mhz("decode")[0]}

    # ---- correctness of what was just timed ----
    torch.cuda.synchronize()
    status = int(word.item()) | H.status()
    roundtrip_equal = bool(torch.equal(d_out[:n], d_in))
    sample = min(n, 64 << 20)
    md5_in = hashlib.md5(d_in[:sample].cpu().numpy().tobytes()).hexdigest()
    md5_out = hashlib.md5(d_out[:sample].cpu().numpy().tobytes()).hexdigest()
    return {"n": n, "npk": npk, "elapsed": elapsed, "enc_ms": enc_avg, "dec_ms": dec_avg, "status": status, "clocks": clocks,
            "roundtrip_equal": roundtrip_equal, "md5_in": md5_in, "md5_out": md5_out,
            "d_in": d_in, "d_slots": d_slots, "d_out": d_out}


def by_kind_pass(args, H, ctl, dev, kind, seed, machine):
    """text(1) / zipf(1) at the headline pass's size (N = 1): W = 1 warm-up step, --by-kind-steps timed steps bracketed
    like the headline pass, per-kernel durations from HIP events, round trip, compaction for the ratio, oracle prefix."""
    import torch
    a = argparse.Namespace(**vars(args))
    a.kind, a.seed, a.scaling = kind, seed, "weak"
    steps = max(1, args.by_kind_steps)
    P = run_pass(a, H, ctl, dev, steps, 1)
    n, npk = P["n"], P["npk"]
    d_stream, d_off = H.compact(P["d_slots"], npk)
    torch.cuda.synchronize()
    c_bytes = int(d_off[-1].item())
    ok = oracle_prefix_ok(H, P["d_in"], n, npk, d_stream, d_off)
    traffic = load_profiled_traffic(kind, n)
    res = kind_result(kind, seed, n, steps, P["elapsed"], P["enc_ms"], P["dec_ms"], c_bytes,
                      bool(P["roundtrip_equal"] and P["md5_in"] == P["md5_out"]), ok, P["status"], traffic, machine, P["clocks"])
    del P, d_stream, d_off
    torch.cuda.empty_cache()
    return res


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    oversubscribe = os.environ.get("GPUAR_OVERSUBSCRIBE_DEVICES") == "1"
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # no launcher around us: become the launcher (children only; nothing here has touched the GPU --
        # torch.cuda.device_count() does not initialise it on this image)
        import torch
        err = device_count_error(args.gpus, torch.cuda.device_count(), oversubscribe)
        if err:
            raise SystemExit(err)
        raise SystemExit(self_launch(argv, args.gpus))

    import torch
    import torch.distributed as dist
    from gpuar_amd import hip as H

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world
    n_devices = torch.cuda.device_count()
    # a launcher gave this node more ranks than it has GPUs: every rank says so and leaves before any rendezvous
    err = device_count_error(max(world, local_rank + 1), n_devices, oversubscribe)
    if err:
        raise SystemExit(err)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    # Test hook: GPUAR_OVERSUBSCRIBE_DEVICES=1 lets several ranks share one physical GPU (rank -> device
    # rank % count) with gloo carrying the barrier/size exchange, so the N > 1 flow can be exercised on a
    # one-GPU box.  The driver's real runs use one GPU per rank and RCCL ("nccl").
    local_dev = local_rank % n_devices if oversubscribe else local_rank
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    ctl_dev = dev                      # where the control tensors of the collectives live
    collectives = world > 1 or args.force_collectives
    backend = None
    if collectives:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:                  # (--force-collectives without a launcher: a world of one)
            with socket.socket() as sock:
                sock.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sock.getsockname()[1])
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        backend = "gloo" if oversubscribe else "nccl"
        # The rendezvous, RCCL's start-up and the first collective are where a broken fabric hangs: they run under a timer
        # of their own, and a rank that is not through in --init-timeout seconds says where it stood and exits 3.
        where = {"at": "init_process_group"}
        init_watchdog = Watchdog(args.init_timeout, lambda: (
            f"bench.py: rank {rank} of {world} not through {where['at']} ({backend}) within {args.init_timeout:g} s: giving up "
            "(MASTER_ADDR/MASTER_PORT reachable? HSA_ENABLE_IPC_MODE_LEGACY=0 exported? one GPU per rank?)"), 3)
        if oversubscribe:
            dist.init_process_group("gloo")
            ctl_dev = torch.device("cpu")
        else:
            dist.init_process_group("nccl", device_id=dev)
        ctl = Control(dist, world, rank, ctl_dev, True, backend)
        where["at"] = "the first barrier"
        ctl.barrier()
        init_watchdog.cancel()
    else:
        ctl = Control(dist, world, rank, ctl_dev, False, None)
    H.load()
    props = torch.cuda.get_device_properties(dev)
    machine = {"cus": int(getattr(props, "multi_processor_count", MI355X_CUS) or MI355X_CUS), "clock_ghz": SHADER_CLOCK_GHZ}

    # ---- the pass the contract's `value` comes from ----
    P = run_pass(args, H, ctl, dev, args.steps, args.warmup)
    n, npk = P["n"], P["npk"]
    # ---- the roof, measured on this box in this run: a plain device copy of the same buffer (rank 0's figure is quoted) ----
    copy_peak = measure_copy_peak(H, P["d_in"], P["d_out"], n) if rank == 0 else None
    # ---- the kernels either side (compaction, encode + compaction, decode from the stream), untimed region ----
    side, d_stream, d_off, c_bytes = side_kernels(H, P["d_in"], P["d_slots"], P["d_out"], npk, n, reps=5)
    oracle_ok = oracle_prefix_ok(H, P["d_in"], n, npk, d_stream, d_off) if rank == 0 else None

    # one row per rank: [ok, compressed bytes, shard bytes, 1, encode us, decode us]; column 3 counts the ranks the
    # collective really saw
    ok = int(P["roundtrip_equal"] and P["status"] == 0 and P["md5_in"] == P["md5_out"] and side["decode_stream_roundtrip_equal"])
    rows = ctl.gather_rows([ok, c_bytes, n, 1, int(P["enc_ms"] * 1e3), int(P["dec_ms"] * 1e3)])
    all_ok = all(r[0] == 1 for r in rows)
    c_total = sum(r[1] for r in rows)
    total_bytes = sum(r[2] for r in rows)
    n_ranks_seen = sum(r[3] for r in rows)

    result = None
    if rank == 0:
        traffic = load_profiled_traffic(args.kind, n)
        result = assemble_result(args, world, n_ranks_seen, n, total_bytes, npk, P["elapsed"], P["enc_ms"], P["dec_ms"], c_bytes, c_total,
                                 all_ok, P["md5_in"], P["md5_out"], oracle_ok, P["status"], traffic, machine, P["clocks"], checker=checker_kind())
        result["shader_clock_MHz"] = P["clocks"]
        result.update(side)
        for key, rec in (("roofline_compact", "gather"), ("roofline_decode_stream", "decode_stream")):
            if rec in traffic:                      # the PMC passes cover these kernels too (tools/prof_run.py --only all)
                result[key]["traffic"] = traffic[rec].get("hbm_bytes_per_launch")
        result["hbm_copy_peak"] = copy_peak
        result["per_rank"] = {
            "encode_ms_min": min(r[4] for r in rows) / 1e3, "encode_ms_max": max(r[4] for r in rows) / 1e3,
            "decode_ms_min": min(r[5] for r in rows) / 1e3, "decode_ms_max": max(r[5] for r in rows) / 1e3,
            "compressed_bytes": [r[1] for r in rows],
        }

    extras_failed = None

    def finish_line():
        """The stdout line (<= LINE_LIMIT bytes); the full object goes to bench_detail.json first."""
        annotate_roofs(result, copy_peak)
        result["collectives"] = ctl.report()
        return driver_line(result, write_detail(result, args.detail_file))

    # ---- N > 1: the OTHER scaling mode in the same run (configs[3] strong: 8 GiB over N; configs[4] weak: 8 GiB each),
    #      and the measurement behind "RCCL only if a gather is measurably cheaper than staged hipMemcpyAsync" ----
    if world > 1 and not args.no_scaling_extras:
        # These extras must never cost the run its line: if they are not through within --extras-timeout seconds (a
        # transport that hangs), every rank's own timer fires, rank 0 prints what the timed pass measured, and all exit.
        def line_without_extras():
            if rank != 0:
                return f"bench.py: rank {rank}: scaling extras not finished within {args.extras_timeout} s"
            result["scaling_extras"] = f"not finished within {args.extras_timeout} s: left out"
            return finish_line()
        watchdog = Watchdog(args.extras_timeout, line_without_extras, 0 if all_ok else 1)
        try:
            if os.environ.get("GPUAR_TEST_FAIL_EXTRAS") == str(rank):         # (tests: an extra that raises on this rank)
                raise RuntimeError("GPUAR_TEST_FAIL_EXTRAS")
            probe = gather_probe(ctl, d_stream, c_bytes)
            del d_stream, d_off, P
            torch.cuda.empty_cache()
            other = argparse.Namespace(**vars(args))
            other.scaling = "strong" if args.scaling == "weak" else "weak"
            Q = run_pass(other, H, ctl, dev, max(3, args.steps // 4), 1)
            q_rows = ctl.gather_rows([int(Q["roundtrip_equal"] and Q["status"] == 0), Q["n"]])
            if rank == 0:
                q_total = sum(r[1] for r in q_rows)
                q_steps = max(3, args.steps // 4)
                result["gather_probe"] = probe
                result["other_scaling"] = {
                    "scaling": other.scaling, "value": q_total * q_steps / Q["elapsed"] / 1e9, "unit": "GB/s",
                    "ms_per_step": Q["elapsed"] / q_steps * 1e3, "steps": q_steps, "total_bytes": q_total,
                    "encode_ms_rank0": Q["enc_ms"], "decode_ms_rank0": Q["dec_ms"], "roundtrip_equal": all(r[0] == 1 for r in q_rows),
                    "workload": (f"{args.kind}({args.seed}) {args.total_gib:g} GiB in all over {world} GPUs" if other.scaling == "strong"
                                 else f"{args.kind}({args.seed}) {args.gib_per_gpu:g} GiB per GPU"),
                }
                all_ok = all_ok and result["other_scaling"]["roundtrip_equal"]
            del Q
        except Exception as e:                                  # noqa: BLE001 -- an extra that RAISES (a transport error, out of memory)
            # must not cost the run its line either: rank 0 says what failed and goes on to print; the other ranks may be
            # sitting in a collective this rank has left, so nobody meets the closing barrier (extras_failed below)
            extras_failed = f"{type(e).__name__}: {e}"[:300]
            if rank == 0:
                result["scaling_extras"] = "failed and left out: " + extras_failed
            else:
                print(f"bench.py: rank {rank}: scaling extras failed: {extras_failed}", file=sys.stderr, flush=True)
        watchdog.cancel()
    else:
        if args.force_collectives and world == 1:            # the RCCL preflight (a world of one: the probe's collectives have no peer to miss)
            result["gather_probe"] = gather_probe(ctl, d_stream, c_bytes)
        del d_stream, d_off, P
    if not extras_failed:
        torch.cuda.empty_cache()

    # ---- BASELINE.json configs[2] and [4]'s stream kinds at the same size, N = 1 only (the other ranks would sit at a barrier) ----
    if rank == 0 and world == 1 and not args.no_by_kind:
        result["by_kind"] = {}
        for kind, seed in (("text", 1), ("zipf", 1)):
            if kind == args.kind and seed == args.seed:
                continue
            result["by_kind"][kind] = by_kind_pass(args, H, ctl, dev, kind, seed, machine)
            all_ok = all_ok and result["by_kind"][kind]["roundtrip_equal"] and result["by_kind"][kind]["device_status"] == 0

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
            "encode_kernel": "encode_small_kernel (latency mode: six roles per 64 packets; gpuar_hip_encode's choice up to 256 MiB)",
            "encode_GBps": m / (min(e) * 1e-3) / 1e9, "decode_GBps": m / (min(d) * 1e-3) / 1e9,
            "gip_bytes": s_total + 20,
            "stream_md5": hashlib.md5(s_stream[:s_total].cpu().numpy().tobytes()).hexdigest(),
            "reference_stream_md5": "c01b5d124681f6fc7264574e57548cdb",
            "roundtrip_equal": bool(torch.equal(s_out, s_in)),
        }

    if rank == 0 and world == 1 and not args.no_cpu_baseline:        # at N = 1 only: the other ranks would sit at the barrier
        result["cpu_baseline"] = cpu_baseline(args.kind, args.seed, args.cpu_sample_mib << 20)

    # ---- the coder kernels' HBM bytes measured on THIS box: two rocprofv3 --pmc child runs, N = 1 only, behind everything that
    #      is timed, this process's buffers freed first ----
    if rank == 0 and world == 1 and not args.no_live_traffic:
        torch.cuda.empty_cache()
        apply_live_traffic(result, live_traffic(args.kind, args.seed, n, args.live_traffic_timeout), machine)

    if rank == 0:
        print(finish_line(), flush=True)
    if extras_failed:                      # some rank may still sit in a collective of the extras: leave without the closing barrier
        os._exit(0 if all_ok or rank != 0 else 1)
    if ctl.active:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0 and not all_ok:
        raise SystemExit("round trip FAILED")


if __name__ == "__main__":
    main()
