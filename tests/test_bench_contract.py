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
import bench_stub  # noqa: E402

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
    # the kernel is named by the symbol rocprofv3 lists, not by a nickname
    assert r["kernel"] == "decode_slots_kernel" and r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
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


def test_roofline_valu_says_the_binding_roof_outright():
    """SURVEY.md section 8(d) "report VALUBusy ... next to the HBM figure": every coder roofline that has the instruction count
    of its kernel (a replayed counter) also carries the vector-issue roof: lane-ops per byte x bytes / time against
    CUs x 4 SIMDs x 16 lanes x clock."""
    t = {"source": "r09_traffic.json: pmc",
         "encode": {"hbm_bytes_per_launch": 1.98e10, "valu_insts_per_symbol_step": 77.9, "valu_busy_per_simd": 0.949},
         "decode": {"hbm_bytes_per_launch": 1.87e10, "valu_insts_per_symbol_step": 83.25, "valu_busy_per_simd": 0.693}}
    args = bench.parse_args([])
    n, c = 8 * GIB, 8658985568
    d = bench.assemble_result(args, 1, 1, n, n, n // 8192, elapsed=0.0447 * args.steps, enc_ms=18.15, dec_ms=26.47, c_bytes_rank0=c,
                              c_total=c, all_ok=True, md5_in="x", md5_out="x", oracle_ok=True, status=0, traffic=t)
    peak = 256 * 4 * 16 * 2.4e9
    for key, ms, per_byte in (("roofline_encode", 18.15, 77.9), ("roofline_decode", 26.47, 83.25), ("roofline", 26.47, 83.25)):
        v = d[key]["roofline_valu"]
        assert v["bound"] == "valu" and v["lane_ops_per_byte"] == per_byte and v["peak_lane_ops_per_s"] == peak
        assert abs(v["achieved_lane_ops_per_s"] - per_byte * n / (ms * 1e-3)) < 1e3
        assert abs(v["frac"] - v["achieved_lane_ops_per_s"] / peak) < 1e-12
    # the figures of VERDICT r4: the encoder at 94 % of the vector-issue roof, the decoder at 69 %
    assert abs(d["roofline_encode"]["roofline_valu"]["frac"] - 0.94) < 0.01 and abs(d["roofline_decode"]["roofline_valu"]["frac"] - 0.69) < 0.01
    assert d["roofline_encode"]["kernel"] == "encode_kernel" and d["roofline_decode"]["kernel"] == "decode_slots_kernel"
    # no counter record -> no vector roof is made up
    assert "roofline_valu" not in stub_line([])["roofline"]
    # another machine shape is stated, not assumed
    m = bench.valu_roof(80.0, GIB, 10.0, cus=128, clock_ghz=2.0)
    assert m["peak_lane_ops_per_s"] == 128 * 4 * 16 * 2.0e9 and "128 CUs" in m["peak_from"] and "frac_at_measured_clock" not in m
    # the clock the kernel's own workgroups measured while it ran: the same achieved figure against THAT roof
    e = bench.assemble_result(args, 1, 1, n, n, n // 8192, elapsed=0.0447 * args.steps, enc_ms=18.15, dec_ms=26.47, c_bytes_rank0=c,
                              c_total=c, all_ok=True, md5_in="x", md5_out="x", oracle_ok=True, status=0, traffic=t,
                              clocks={"encode": 2250.0, "decode": 2350.0})
    ve, vd = e["roofline_encode"]["roofline_valu"], e["roofline_decode"]["roofline_valu"]
    assert ve["shader_clock_measured_MHz"] == 2250.0 and vd["shader_clock_measured_MHz"] == 2350.0
    assert abs(ve["frac_at_measured_clock"] - ve["achieved_lane_ops_per_s"] / (256 * 4 * 16 * 2250.0e6)) < 1e-12
    assert ve["frac"] < ve["frac_at_measured_clock"] <= 1.01 and abs(ve["frac_at_measured_clock"] - ve["frac"] * 2400 / 2250) < 1e-9
    assert e["roofline"]["roofline_valu"]["shader_clock_measured_MHz"] == 2350.0          # the dominant kernel is the decoder


def test_kernel_symbols_follow_the_launch_size():
    assert bench.kernel_symbols(8192)["encode"] == "encode_small_kernel"           # 64 MiB: 128 groups -> latency kernel
    assert bench.kernel_symbols(512 * 64)["encode"] == "encode_small_kernel" and bench.kernel_symbols(512 * 64 + 1)["encode"] == "encode_kernel"
    assert bench.kernel_symbols(1 << 20) == {"encode": "encode_kernel", "decode": "decode_slots_kernel"}


def test_by_kind_entries_have_the_shape_the_driver_line_promises():
    """BASELINE.json configs[2] (text) and configs[4]'s stream kind (zipf) ride in the same line as the uniform pass."""
    t = {"source": "r09_traffic_text_8gib.json: pmc", "encode": {"hbm_bytes_per_launch": 1.5e10, "valu_insts_per_symbol_step": 78.0},
         "decode": {"hbm_bytes_per_launch": 1.45e10, "valu_insts_per_symbol_step": 83.0}}
    n, c = 8 * GIB, 5819484000
    k = bench.kind_result("text", 1, n, 5, elapsed=5 * 0.0446, enc_ms=18.1, dec_ms=26.5, c_bytes=c, roundtrip_equal=True, oracle_ok=True,
                          status=0, traffic=t)
    for key in ("workload", "steps", "ms_per_step", "value", "unit", "encode_ms", "decode_ms", "encode_GBps", "decode_GBps", "compression_ratio",
                "roundtrip_equal", "oracle_prefix_match", "device_status", "roofline", "roofline_encode", "roofline_decode", "traffic_source"):
        assert key in k, key
    assert k["workload"].startswith("text(1) 8 GiB") and k["unit"] == "GB/s" and k["steps"] == 5
    assert abs(k["value"] - n / 0.0446 / 1e9) < 1e-6 and abs(k["compression_ratio"] - (c + 20) / n) < 1e-12
    r = k["roofline"]
    assert r["kernel"] == "decode_slots_kernel" and r["algorithmic_bytes_per_launch"] == n + c and r["traffic"] == 1.45e10
    assert abs(r["achieved"] - (n + c) / 26.5e-3 / 1e9) < 1e-6 and abs(r["frac"] - r["achieved"] / 8000.0) < 1e-12
    assert "roofline_valu" in r and "roofline_valu" in k["roofline_encode"]
    line = stub_line([])
    line["by_kind"] = {"text": k}
    bench.annotate_roofs(line, {"GBps": 6200.0})
    assert k["roofline"]["peak_measured_copy"] == 6200.0 and "r09_traffic_text_8gib.json" in k["roofline"]["counters"]
    assert line["roofline"]["counters"].startswith("none quoted")
    json.dumps(line)


def test_more_ranks_than_gpus_is_refused_in_one_line_before_anything_starts(monkeypatch, capsys):
    assert bench.device_count_error(8, 8, False) is None and bench.device_count_error(1, 1, False) is None
    assert bench.device_count_error(4, 1, True) is None                          # the tests' oversubscription hook
    msg = bench.device_count_error(8, 1, False)
    assert "--gpus 8" in msg and "1 visible GPU" in msg and "\n" not in msg
    # `python bench.py --gpus 8` on a box with fewer GPUs: exits non-zero with that line, and never builds a launch command
    import torch
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.delenv("GPUAR_OVERSUBSCRIBE_DEVICES", raising=False)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 1)
    monkeypatch.setattr(bench, "self_launch", lambda *a, **k: (_ for _ in ()).throw(AssertionError("ranks were started")))
    try:
        bench.main(["--gpus", "8"])
    except SystemExit as e:
        assert isinstance(e.code, str) and "nothing was started" in e.code
    else:
        raise AssertionError("bench.main did not exit")


def test_watchdog_prints_its_line_and_exits_with_its_code():
    """The timer around init_process_group / the first barrier (and around the N > 1 extras): a rank that hangs there ends
    with one clear line and the given exit code instead of sitting in the driver's run until its limit."""
    import subprocess
    code = ("import bench, time\n"
            "w = bench.Watchdog(0.2, lambda: 'bench.py: rank 0 of 8 not through init_process_group', 3)\n"
            "time.sleep(30)\n")
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=20)
    assert r.returncode == 3 and "not through init_process_group" in r.stderr
    code = "import bench, time\nw = bench.Watchdog(5, 'never', 3)\nw.cancel()\nprint('done')\n"
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=20)
    assert r.returncode == 0 and "done" in r.stdout


def test_control_plane_without_a_process_group_is_plain_python():
    """world = 1 without --force-collectives: no torch.distributed call is made; the report says so."""
    import torch
    c = bench.Control(None, 1, 0, torch.device("cpu"), False, None)
    assert c.max_over_ranks(1.5) == 1.5 and c.min_over_ranks(7) == 7 and c.gather_rows([1, 2, 3]) == [[1, 2, 3]]
    rep = c.report()
    assert rep["through_torch_distributed"] is False and rep["backend"] is None and sum(rep["calls"].values()) == 0


def test_the_stdout_line_fits_the_drivers_capture_at_world_1_and_world_8(tmp_path):
    """the round's number was lost because the line (22 KB) fell out of the driver's ~8 KB tail.  The line is
    now cut out of the full object by driver_line(): <= 4096 bytes with every extra a run can hang on it, at N = 1 (by_kind,
    small_config, cpu_baseline) and at N = 8 in both scaling modes (per-rank figures, the other scaling mode, the gather
    probe); numbers only; the full object goes to the detail file the line names."""
    for world, scaling in ((1, "weak"), (8, "weak"), (8, "strong"), (2, "weak"), (4, "strong")):
        full = bench_stub.full_result(world, scaling)
        path = str(tmp_path / f"detail_{world}_{scaling}.json")
        name = bench.write_detail(full, path)
        assert name == path
        text = bench.driver_line(full, name)
        d = bench_stub.check_line(text, world)
        assert len(json.dumps(full)) > 2 * len(text)                   # (the stub really is the long object)
        assert d["scaling"] == scaling and d["detail"] == path and d["checker"] == "reference"
        assert json.load(open(path)) == json.loads(json.dumps(full))   # nothing is lost: the detail file holds the whole object
        assert d["roofline"]["kernel"] == "decode_slots_kernel" and d["roofline"]["traffic"] == 18660000000.0
        assert d["roofline"]["traffic_from"] == "replayed"
        assert d["roofline_encode"]["kernel"] == "encode_kernel"
        if world == 1:
            assert set(d["by_kind"]) == {"text", "zipf"} and d["by_kind"]["text"]["ok"] is True and 0.67 < d["by_kind"]["text"]["ratio"] < 0.68
            assert d["small_config"]["md5_match"] is True and d["small_config"]["encode_GBps"] > d["small_config"]["decode_GBps"] > 0
            c = d["cpu_baseline"]
            assert c["kind"] == "reference" and c["cores"] == 1 and c["value"] > 0 and c["unit"] == "GB/s" and "16 MiB" in c["sample"]
            assert c["all_cores"]["cores"] == 16 and c["product_host"]["value"] > c["all_cores"]["value"]
            assert "per_rank" not in d and "other_scaling" not in d
        else:
            assert d["n_ranks_seen"] == world and "by_kind" not in d and "cpu_baseline" not in d
            assert d["per_rank"]["encode_ms_min"] <= d["per_rank"]["encode_ms_max"] and "compressed_bytes" not in d["per_rank"]
            assert d["other_scaling"]["scaling"] != scaling and d["gather_probe"]["cheaper"] == "staged hipMemcpyAsync" and d["gather_probe"]["rccl"] is True
            assert d["collectives"] == {"backend": "nccl", "calls": 43}
    # the driver's view: the tail of stdout + stderr, with launcher chatter before the line and warnings behind it
    text = bench.driver_line(bench_stub.full_result(8, "weak"))
    stdout = "W1004 torch.distributed.run: setting OMP_NUM_THREADS=1\n" * 40 + text + "\n"
    stderr = "/opt/amdgpu/share/libdrm/amdgpu.ids: No such file or directory\n" * 8
    got = bench_stub.line_from_driver_tail(stdout, stderr)
    assert got == json.loads(text)


def test_a_line_that_would_not_fit_sheds_its_extras_not_its_contract():
    full = bench_stub.full_result(1, "weak")
    full["by_kind"] = {f"kind{k}": dict(full["by_kind"]["text"]) for k in range(60)}      # (no run makes this; the guard is for the unforeseen)
    text = bench.driver_line(full)
    assert len(text) <= bench.LINE_LIMIT
    d = json.loads(text)
    assert "by_kind" not in d and d["value"] > 0 and d["roofline"]["frac"] > 0 and "cpu_baseline" in d


def test_watchdog_keeps_stdout_for_the_json_line_only():
    """a rank other than 0 whose extras timer fires must not put a non-JSON line on stdout (the one-line contract)."""
    import subprocess
    code = ("import bench, time\n"
            "w = bench.Watchdog(0.2, lambda: 'bench.py: rank 3: scaling extras not finished', 0)\n"
            "time.sleep(30)\n")
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=20)
    assert r.returncode == 0 and r.stdout == "" and "rank 3" in r.stderr
    code = ("import bench, time\n"
            "w = bench.Watchdog(0.2, lambda: '{\"metric\": 1}', 1)\n"
            "time.sleep(30)\n")
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=20)
    assert r.returncode == 1 and r.stdout.strip() == '{"metric": 1}'


def test_live_traffic_replaces_the_replayed_figure_and_says_so():
    """the traffic figure of the line is measured on the driver's box when the counter passes succeed ("live"),
    and the replayed record stays -- labelled "replayed" -- when they do not."""
    full = bench_stub.full_result(1, "weak")
    assert full["roofline"]["traffic"] == 18660000000.0
    live = {"source": "live: ...", "seconds": 41.0, "fetch_correction": 2.0, "n_bytes": 8 * GIB,
            "encode": {"fetch_size_bytes_raw": 4.3e9, "write_size_bytes": 1.12e10, "hbm_bytes_per_launch": 1.98e10, "launches": 2},
            "decode": {"fetch_size_bytes_raw": 4.7e9, "write_size_bytes": 8.6e9, "hbm_bytes_per_launch": 1.8e10, "launches": 1}}
    assert bench.apply_live_traffic(full, live) is True
    bench.annotate_roofs(full, full["hbm_copy_peak"])
    d = json.loads(bench.driver_line(full))
    assert d["roofline"]["traffic"] == 1.8e10 and d["roofline"]["traffic_from"] == "live"
    assert d["roofline_encode"]["traffic"] == 1.98e10 and d["roofline_encode"]["traffic_from"] == "live"
    assert "traffic (rocprofv3" in full["roofline"]["measured_live"] and "traffic," not in full["roofline"]["counters"].split(":")[0]
    assert "valu_busy" in full["roofline"]["counters"]                 # only two passes came in: the issue-side counters are the replayed ones
    assert d["roofline"]["counters_from"] == "replayed" and d["roofline"]["valu_insts_per_step"] == 83.25
    # all three passes: instructions per step, vector-pipe busy and the vector roof are this run's as well
    third = bench_stub.full_result(1, "weak")
    live3 = json.loads(json.dumps(live))
    live3["decode"].update({"valu_insts_per_symbol_step": 82.27, "lds_insts_per_symbol_step": 4.77, "valu_busy": 0.76, "valu_busy_per_simd": 0.757, "wait_frac": 0.115})
    live3["encode"].update({"valu_insts_per_symbol_step": 77.94, "lds_insts_per_symbol_step": 16.9, "valu_busy": 0.24, "valu_busy_per_simd": 0.943, "wait_frac": 0.5})
    assert bench.apply_live_traffic(third, live3) is True
    bench.annotate_roofs(third, third["hbm_copy_peak"])
    d3 = json.loads(bench.driver_line(third))
    r = d3["roofline"]
    assert r["counters_from"] == "live" and r["valu_insts_per_step"] == 82.27 and r["valu_busy_per_simd"] == 0.757 and r["wait_frac"] == 0.115
    ms = third["roofline"]["algorithmic_bytes_per_launch"] / (third["roofline"]["achieved"] * 1e9) * 1e3
    want = 82.27 * 8 * GIB / (ms * 1e-3) / (256 * 4 * 16 * 2391.654321e6)       # against the clock the kernel's workgroups measured (the stub's)
    assert abs(r["valu_frac"] - want) < 1e-4, (r["valu_frac"], want)
    assert third["roofline"]["counters"].startswith("none quoted") or "replayed" not in third["roofline"]["counters"].split(":")[0]
    assert "a third --pmc pass" in third["roofline"]["measured_live"] and d3["roofline_encode"]["counters_from"] == "live"
    assert len(bench.driver_line(third)) <= bench.LINE_LIMIT
    # a failed pass: nothing changes, the reason is kept in the detail object
    again = bench_stub.full_result(1, "weak")
    assert bench.apply_live_traffic(again, {"error": "FETCH_SIZE pass: TimeoutExpired"}) is False
    d = json.loads(bench.driver_line(again))
    assert d["roofline"]["traffic"] == 18660000000.0 and d["roofline"]["traffic_from"] == "replayed"
    assert again["traffic_live"]["error"].startswith("FETCH_SIZE")
    # and with neither: null, no label
    bare = bench_stub.full_result(1, "weak", traffic={"source": "none"})
    d = json.loads(bench.driver_line(bare))
    assert d["roofline"]["traffic"] is None and "traffic_from" not in d["roofline"]
