"""bench.py's host-side logic, run on CPU: the JSON line is assembled from stub measurements by the same
function the GPU run uses (so every contract field, the roofline arithmetic and the stale-traffic rule
are checked on code, not on a stored record), the sharding of both scaling modes, and the command
bench.py starts itself with when it is called as plain `python bench.py --gpus N`."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import bench  # noqa: E402

GIB = 1 << 30


def stub_line(argv, world=1, traffic=None):
    args = bench.parse_args(argv)
    n = 8 * GIB
    c = 8658985568
    return bench.assemble_result(args, world, world, n, n * world, n // 8192, elapsed=0.0709 * args.steps, enc_ms=24.5, dec_ms=46.3,
                                 c_bytes_rank0=c, c_total=c * world, all_ok=True, md5_in="x", md5_out="x", oracle_ok=True,
                                 status=0, traffic=traffic or {"source": "none"})


def test_json_line_has_the_contract_fields_and_consistent_arithmetic():
    d = json.loads(json.dumps(stub_line(["--steps", "20", "--warmup", "5"])))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "n_ranks_seen"):
        assert k in d, k
    assert d["unit"] == "GB/s" and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    assert d["steps"] == 20 and d["warmup"] == 5 and d["n_gpus"] == 1 and d["n_ranks_seen"] == 1
    assert abs(d["ms_per_step"] - 70.9) < 1e-6
    assert abs(d["value"] - 8 * GIB / 0.0709 / 1e9) < 1e-6            # value = bytes of all ranks / time
    r = d["roofline"]
    # dominant kernel = the slower one; achieved = (N + C) / its launch duration; frac = achieved / 8 TB/s
    assert r["kernel"] == "decode_kernel" and r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert r["algorithmic_bytes_per_launch"] == 8 * GIB + 8658985568
    assert abs(r["achieved"] - (8 * GIB + 8658985568) / 46.3e-3 / 1e9) < 1e-6
    assert abs(r["frac"] - r["achieved"] / 8000.0) < 1e-12
    assert r["traffic"] is None and d["traffic_source"] == "none"
    assert abs(d["encode_read_frac_of_hbm_peak"] - 8 * GIB / 24.5e-3 / 1e9 / 8000.0) < 1e-12
    assert abs(d["compression_ratio"] - (8658985568 + 20) / (8 * GIB)) < 1e-12


def test_json_line_aggregates_over_ranks_and_names_the_scaling_mode():
    d = stub_line(["--gpus", "4"], world=4)
    assert d["n_gpus"] == 4 and d["n_ranks_seen"] == 4 and d["config"]["parallelism"] == "packet-sharded x4"
    assert abs(d["value"] - 4 * 8 * GIB * 5 / (0.0709 * 5) / 1e9) < 1e-6
    assert abs(d["encode_GBps"] - 4 * 8 * GIB / 24.5e-3 / 1e9) < 1e-6
    s = stub_line(["--gpus", "8", "--scaling", "strong", "--total-gib", "8"], world=8)
    assert s["scaling"] == "strong" and "8 GiB in all over 8 GPU(s)" in s["config"]["workload"]


def test_traffic_is_quoted_only_for_the_kernels_it_was_measured_on(tmp_path):
    os.makedirs(tmp_path / "profiles")
    for rel in bench.KERNEL_SOURCES:
        os.makedirs(os.path.dirname(tmp_path / rel), exist_ok=True)
        (tmp_path / rel).write_text("kernel v1")
    stamp = bench.kernel_source_stamp(str(tmp_path))
    rec = {"source": "pmc", "input_gib": 8.0, "kind": "uniform", "kernel_source_sha256_16": stamp,
           "encode": {"hbm_bytes_per_launch": 2.1e10, "valu_busy": 0.89, "wait_frac": 0.05},
           "decode": {"hbm_bytes_per_launch": 2.0e10, "valu_busy": 0.60, "wait_frac": 0.27}}
    (tmp_path / "profiles" / "r09_traffic.json").write_text(json.dumps(rec))
    t = bench.load_profiled_traffic("uniform", 8 * GIB, root=str(tmp_path))
    assert t["decode"]["hbm_bytes_per_launch"] == 2.0e10 and "r09_traffic.json" in t["source"]
    d = stub_line([], traffic=t)
    assert d["roofline"]["traffic"] == 2.0e10 and d["roofline"]["valu_busy"] == 0.60 and d["roofline"]["wait_frac"] == 0.27
    assert d["roofline_encode"]["traffic"] == 2.1e10
    # other workload -> not quoted
    assert "encode" not in bench.load_profiled_traffic("zipf", 8 * GIB, root=str(tmp_path))
    assert "encode" not in bench.load_profiled_traffic("uniform", 1 * GIB, root=str(tmp_path))
    # a kernel source changes -> the record is stale: traffic null, and the line says why
    (tmp_path / bench.KERNEL_SOURCES[0]).write_text("kernel v2")
    t = bench.load_profiled_traffic("uniform", 8 * GIB, root=str(tmp_path))
    assert "encode" not in t and "STALE" in t["source"]
    assert stub_line([], traffic=t)["roofline"]["traffic"] is None


def test_traffic_record_is_picked_by_workload(tmp_path):
    """profiles/ may hold one record per stream kind and size: the one taken on THIS workload is quoted."""
    os.makedirs(tmp_path / "profiles")
    for rel in bench.KERNEL_SOURCES:
        os.makedirs(os.path.dirname(tmp_path / rel), exist_ok=True)
        (tmp_path / rel).write_text("kernel v1")
    stamp = bench.kernel_source_stamp(str(tmp_path))
    for name, kind, gib, hbm in (("r09_traffic.json", "uniform", 8.0, 1.9e10), ("r09_traffic_text_8gib.json", "text", 8.0, 1.4e10),
                                 ("r09_traffic_uniform_0.0625gib.json", "uniform", 0.0625, 1.5e8)):
        rec = {"source": "pmc", "input_gib": gib, "kind": kind, "kernel_source_sha256_16": stamp,
               "encode": {"hbm_bytes_per_launch": hbm}, "decode": {"hbm_bytes_per_launch": hbm}, "gather": {"hbm_bytes_per_launch": hbm}}
        (tmp_path / "profiles" / name).write_text(json.dumps(rec))
    assert bench.load_profiled_traffic("text", 8 * GIB, root=str(tmp_path))["decode"]["hbm_bytes_per_launch"] == 1.4e10
    assert bench.load_profiled_traffic("uniform", 8 * GIB, root=str(tmp_path))["gather"]["hbm_bytes_per_launch"] == 1.9e10
    assert bench.load_profiled_traffic("uniform", GIB // 16, root=str(tmp_path))["encode"]["hbm_bytes_per_launch"] == 1.5e8
    assert "decode" not in bench.load_profiled_traffic("zipf", 8 * GIB, root=str(tmp_path))


def test_usable_cpus_is_the_affinity_cut_by_the_cgroup_quota():
    n, how = bench.usable_cpus()
    assert 1 <= n <= (os.cpu_count() or 1) and how in ("sched_getaffinity", "cgroup cpu.max", "cgroup cfs_quota")
    assert n <= len(os.sched_getaffinity(0))


def test_committed_traffic_record_matches_the_committed_kernels_or_is_declared_stale():
    t = bench.load_profiled_traffic("uniform", 8 * GIB)
    assert ("decode" in t) or ("STALE" in t["source"]) or ("no profiles" in t["source"])


def test_sharding_of_both_scaling_modes():
    a = bench.parse_args(["--scaling", "strong", "--total-gib", "8"])
    shards = [bench.plan_shard(a, 8, r) for r in range(8)]
    assert sum(n for _, n in shards) == 8 * GIB and shards[0] == (0, GIB) and shards[7] == (7 * GIB, GIB)
    assert all(off % (64 * 8192) == 0 for off, _ in shards)
    a = bench.parse_args(["--gib-per-gpu", "8"])
    assert bench.plan_shard(a, 8, 7) == (56 * GIB, 8 * GIB)             # configs[4]: last rank of zipf 64 GiB


def test_self_launch_command_is_one_rank_per_gpu_on_loopback():
    cmd = bench.self_launch_command(["--gpus", "8", "--steps", "20"], 8, 29517)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "8"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29517"
    assert cmd[-4:] == ["--gpus", "8", "--steps", "20"] and cmd[-5].endswith("bench.py")


def test_every_roofline_object_carries_the_measured_roof_and_says_what_is_replayed():
    """SURVEY.md section 8(d) "report both": the datasheet peak AND the copy peak measured in the same run; counter-derived
    fields are marked as replayed from a stamped profile record, not passed off as measured live."""
    t = {"source": "r09_traffic.json: pmc", "decode": {"hbm_bytes_per_launch": 2.0e10, "valu_busy": 0.6},
         "encode": {"hbm_bytes_per_launch": 2.1e10}}
    d = stub_line([], traffic=t)
    d["roofline_compact"] = bench.hbm_roof(2 * 8658985568, 3.5)
    copy_peak = {"GBps": 6100.0}
    bench.annotate_roofs(d, copy_peak, t["source"])
    for key in ("roofline", "roofline_encode", "roofline_decode", "roofline_compact"):
        r = d[key]
        assert r["peak"] == 8000.0 and r["peak_measured_copy"] == 6100.0
        assert abs(r["frac_of_measured"] - r["achieved"] / 6100.0) < 1e-12 and r["frac_of_measured"] > r["frac"]
        assert "measured_live" in r and "counters" in r
    assert "replayed from profiles/r09_traffic.json" in d["roofline"]["counters"] and "valu_busy" in d["roofline"]["counters"]
    assert d["roofline_compact"]["counters"].startswith("none quoted")
    json.dumps(d)
