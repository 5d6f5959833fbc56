"""The JSON line bench.py printed on the MI355X (kept as profiles/r01_bench.json) carries every field the
driver's contract names, with the roofline and cpu_baseline objects; no GPU needed to check the record."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_recorded_bench_line_has_the_contract_fields():
    d = json.load(open(os.path.join(ROOT, "profiles", "r01_bench.json")))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "GB/s" and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["ms_per_step"] * d["value"] - 8 * 1.073741824 * 1e3) / (8 * 1.073741824 * 1e3) < 0.01   # value = bytes / time
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and r["traffic"] > r["algorithmic_bytes_per_launch"] > 0
    c = d["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] == 1 and c["value"] > 0 and "MiB" in c["sample"]
    assert c["all_cores"]["cores"] > 1 and c["all_cores"]["roundtrip_ok"] is True
    assert d["roundtrip_equal"] is True and d["oracle_prefix_match"] is True and d["device_status"] == 0
    # the bit-exactness anchors of the 64 MiB case
    assert d["small_config"]["stream_md5"] == d["small_config"]["reference_stream_md5"] == "c01b5d124681f6fc7264574e57548cdb"
