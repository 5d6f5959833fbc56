"""Stub measurements run through bench.py's own functions: the FULL result object of a run (every extra a world of N
ranks hangs on it) without a GPU, and the driver's view of a finished run (the tail of stdout + stderr it keeps)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import bench  # noqa: E402

GIB = 1 << 30
C_UNIFORM = 8658985568
DRIVER_TAIL = 6000          # the driver keeps about 8 KB of "stdout \n---- stderr ----\n stderr"; the tests allow themselves 6000

LONG_SOURCE = ("r06_traffic.json: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), tools/prof_run.py --gib 8.0 --kind uniform; "
               "SQ_INSTS_VALU, SQ_ACTIVE_INST_VALU, SQ_WAIT_INST_ANY, SQ_BUSY_CYCLES in further passes") * 2


def full_result(world=1, scaling="weak", traffic=None, rows=None):
    """What main() holds on rank 0 just before it prints, assembled by the same functions from stub numbers: the timed pass,
    the side kernels, per-rank figures, and -- as the world has them -- by_kind / small_config / cpu_baseline (N = 1) or
    other_scaling / gather_probe (N > 1).  `rows` = the all_gather'ed per-rank rows, when a test has real ones."""
    argv = ["--gpus", str(world), "--steps", "20", "--warmup", "5", "--scaling", scaling]
    args = bench.parse_args(argv)
    n = 8 * GIB if scaling == "weak" else 8 * GIB // world
    c = C_UNIFORM if scaling == "weak" else C_UNIFORM // world
    t = traffic or {"source": LONG_SOURCE,
                    "encode": {"hbm_bytes_per_launch": 19821000000.0, "valu_insts_per_symbol_step": 77.9, "lds_insts_per_symbol_step": 16.9,
                               "valu_busy": 0.24, "valu_busy_per_simd": 0.949, "wait_frac": 0.05},
                    "decode": {"hbm_bytes_per_launch": 18660000000.0, "valu_insts_per_symbol_step": 83.25, "lds_insts_per_symbol_step": 5.27,
                               "valu_busy": 0.17, "valu_busy_per_simd": 0.693, "wait_frac": 0.16},
                    "gather": {"hbm_bytes_per_launch": 17700000000.0}, "decode_stream": {"hbm_bytes_per_launch": 18700000000.0}}
    rows = rows or [[1, c, n, 1, 18099 + r, 26351 + r] for r in range(world)]
    res = bench.assemble_result(args, world, sum(r[3] for r in rows), n, sum(r[2] for r in rows), n // 8192, elapsed=0.04451234567 * args.steps,
                                enc_ms=18.0991234567, dec_ms=26.3511234567, c_bytes_rank0=c, c_total=sum(r[1] for r in rows), all_ok=True,
                                md5_in="c9f0253b284172e8c4456be3234de256", md5_out="c9f0253b284172e8c4456be3234de256", oracle_ok=True, status=0,
                                traffic=t, clocks={"encode": 2381.123456, "decode": 2391.654321}, checker="reference")
    res["shader_clock_MHz"] = {"encode": 2381.123456, "decode": 2391.654321}
    res.update({"compact_ms": 3.3123456, "encode_plus_compact_ms": 21.7123456, "decode_stream_ms": 26.4123456, "decode_stream_roundtrip_equal": True,
                "roofline_compact": dict(bench.hbm_roof(2 * c, 3.3123456), kernel="scan_* + gather_kernel"),
                "roofline_decode_stream": dict(bench.hbm_roof(n + c, 26.4123456), kernel="decode_stream_kernel"),
                "encode_plus_compact_GBps": 395.123456, "decode_stream_GBps": 325.123456})
    res["hbm_copy_peak"] = {"GBps": 6234.188171326362, "best_GBps": 6257.357398124409, "ms_avg": 2.7557, "reps": 10, "bytes_copied": n,
                            "kernel": "copy_kernel (gpuar_hip_copy): 16 B per lane, one quad per thread, read + write counted"}
    res["per_rank"] = {"encode_ms_min": min(r[4] for r in rows) / 1e3, "encode_ms_max": max(r[4] for r in rows) / 1e3,
                       "decode_ms_min": min(r[5] for r in rows) / 1e3, "decode_ms_max": max(r[5] for r in rows) / 1e3,
                       "compressed_bytes": [r[1] for r in rows]}
    if world > 1:
        res["gather_probe"] = {"staged_d2h_ms": 5.123456789, "gather_then_d2h_ms": 41.123456789, "bytes_per_rank": 268435456, "ranks": world,
                               "transport": "RCCL send/recv to rank 0", "staged_GBps": 419.123456, "gathered_GBps": 52.123456,
                               "cheaper": "staged hipMemcpyAsync"}
        res["other_scaling"] = {"scaling": "strong" if scaling == "weak" else "weak", "value": 1234.56789, "unit": "GB/s", "ms_per_step": 6.123456789,
                                "steps": 5, "total_bytes": 8 * GIB, "encode_ms_rank0": 2.41234567, "decode_ms_rank0": 3.61234567,
                                "roundtrip_equal": True, "workload": f"uniform(42) 8 GiB in all over {world} GPUs"}
    else:
        res["by_kind"] = {}
        for kind, cb in (("text", 5819484000), ("zipf", 6769484000)):
            res["by_kind"][kind] = bench.kind_result(kind, 1, n, 5, elapsed=5 * 0.04431234567, enc_ms=17.9123456, dec_ms=26.4123456, c_bytes=cb,
                                                     roundtrip_equal=True, oracle_ok=True, status=0, traffic=t,
                                                     clocks={"encode": 2381.1, "decode": 2391.6})
        res["small_config"] = {
            "workload": "uniform(42) 64 MiB (stand-in for data/random_64m.dat), 1 GPU, 8192 packets = 128 groups of 64",
            "encode_kernel": "encode_small_kernel (latency mode: six roles per 64 packets; gpuar_hip_encode's choice up to 256 MiB)",
            "encode_GBps": 98.1516905204868, "decode_GBps": 40.90766656178297, "gip_bytes": 67648304,
            "stream_md5": "c01b5d124681f6fc7264574e57548cdb", "reference_stream_md5": "c01b5d124681f6fc7264574e57548cdb", "roundtrip_equal": True}
        res["cpu_baseline"] = {
            "value": 0.005135105790953162, "unit": "GB/s", "cores": 1, "kind": "reference", "sample_mib": 16,
            "sample": "first 16 MiB of the same uniform(42) stream, encode then decode, 1 thread",
            "encode_GBps": 0.012698059774229838, "decode_GBps": 0.008621747590253932, "roundtrip_ok": True,
            "host_cpus_online": 256, "host_cpus_usable": 16, "host_cpus_usable_from": "cgroup cpu.max",
            "all_cores": {"cores": 16, "threads": "native (oracle/ref_driver.cpp), one contiguous packet range per thread",
                          "sample": "first 16 x 8 MiB of the same stream", "value": 0.08012875254800815, "encode_GBps": 0.19461344594301547,
                          "decode_GBps": 0.13621150732070794, "roundtrip_ok": True, "speedup_over_one_core": 15.604109401051872},
            "product_host": {"what": "gpuar-host c|d --host --threads=0 (this repository's CPU path), CLI wall time incl. file I/O", "cores": 16,
                             "sample": "128 MiB of the same stream", "value": 0.3380580392916557, "unit": "GB/s",
                             "encode_GBps": 1.0360316638650884, "decode_GBps": 0.5017937936327498, "roundtrip_ok": True}}
    bench.annotate_roofs(res, res["hbm_copy_peak"])
    res["collectives"] = {"backend": "nccl" if world > 1 else None, "world": world, "through_torch_distributed": world > 1,
                          "calls": {"barrier": 31, "all_reduce_max": 9, "all_reduce_min": 1, "all_gather": 2} if world > 1 else
                                   {"barrier": 0, "all_reduce_max": 0, "all_reduce_min": 0, "all_gather": 0}}
    return res


def line_from_driver_tail(stdout, stderr, keep=DRIVER_TAIL):
    """The bench line as the DRIVER finds it: out of the last `keep` bytes of stdout + a stderr section.  Raises if the
    line did not survive whole (round 5: a 22 KB line lost its head and with it the run's number)."""
    tail = (stdout + "\n---- stderr ----\n" + stderr)[-keep:]
    lines = [l for l in tail.splitlines() if l.startswith('{"metric"')]
    assert lines, f"no whole bench line in the last {keep} bytes of the run's output (stdout {len(stdout)} B, stderr {len(stderr)} B)"
    line = lines[-1]
    assert len(line) <= bench.LINE_LIMIT, len(line)
    return json.loads(line)


CONTRACT_FIELDS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                   "data", "config", "roofline", "n_ranks_seen", "detail")


def check_line(line_text, world):
    """The size and shape rules of the stdout line."""
    assert len(line_text) <= bench.LINE_LIMIT, len(line_text)
    assert "\n" not in line_text
    d = json.loads(line_text)
    for k in CONTRACT_FIELDS:
        assert k in d, k
    assert d["n_gpus"] == world and d["unit"] == "GB/s" and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / 8000.0) < 1e-4
    # numbers only: no string in the line is longer than a workload description
    def longest(o):
        if isinstance(o, str):
            return len(o)
        if isinstance(o, dict):
            return max([longest(v) for v in o.values()] + [0])
        if isinstance(o, list):
            return max([longest(v) for v in o] + [len(o) * 8])      # (and no per-rank arrays)
        return 0
    assert longest(d) <= 200, longest(d)
    return d
