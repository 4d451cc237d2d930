"""CPU: the committed bench line (profiles/r01_final_bench.json, printed by `python bench.py` on an MI355X) carries
every field of the bench contract, and bench.py's tables cover the entry points it reports."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_bench_line_has_the_contract_fields():
    d = json.load(open(os.path.join(ROOT, "profiles", "r01_final_bench.json")))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["metric"].startswith("CW-tokens/sec") and d["unit"] == "CW-tokens/s" and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "bf16" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert r["traffic"] is None or r["traffic"] >= 0.9 * r["algorithmic_bytes_per_launch"]
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1
    # value = tokens per step / step time
    tokens = d["n_gpus"] * d["config"]["per_gpu_batch"] * d["config"]["seq_len"]
    assert abs(d["value"] - tokens / (d["ms_per_step"] * 1e-3)) < 1e-3 * d["value"]


def test_bench_tables_cover_reported_entry_points():
    sys.path.insert(0, ROOT)
    import bench
    d = json.load(open(os.path.join(ROOT, "profiles", "r01_final_bench.json")))
    for name, row in d["kernels"].items():
        B, T = d["config"]["per_gpu_batch"], d["config"]["seq_len"]
        has_bytes = bench.algorithmic_bytes(name, B, T, s=2)
        has_flops = bench.algorithmic_flops(name, B, T)
        assert has_bytes or has_flops, name
        assert (row["GB/s"] is not None) == bool(has_bytes)
