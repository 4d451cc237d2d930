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


def _args(**kw):
    import argparse
    d = dict(steps=10, warmup=3, dtype="bf16", ppo_rollouts=64, ppo_window=1024, ppo_steps=10)
    d.update(kw)
    return argparse.Namespace(**d)


def test_report_picks_the_dominant_kernel_and_prices_the_scan_backward_as_one_unit():
    """bench.py's reporting on a synthetic kernel-time table (no GPU): `roofline` = the libcwlt entry with the largest
    total time whatever bounds it, `roofline_hbm` / `roofline_mfma` the largest of each kind, the two attention-backward
    launches merged and priced by SURVEY 8(d)'s 7 * D * s bytes per token, wgrad FLOP/s from the declared per-call
    shapes."""
    sys.path.insert(0, ROOT)
    import bench
    B, T, R = 512, 1024, 512 * 1024
    flops = bench.wgrad_flops_per_step(B, T, in_linear=True) / 50      # a table without the one-pass front's kernels
    kt = {"cwlt_wgrad_bf16": (50, 0.72, flops),
          "cwlt_causal_linear_bwd_dkdv": (12, 1.0, None), "cwlt_causal_linear_bwd_dq": (12, 0.75, None),
          "cwlt_causal_linear_fwd": (12, 0.53, None), "cwlt_bias_gelu_dropout_bwd": (12, 1.29, None),
          "cwlt_heads_fwd": (1, 0.4, None)}
    out = bench.report(_args(), kt, B, T, 2, 1, 190.0, B * T / 0.19, 3.9, None, True, 112.0)
    assert out["roofline"]["kernel"] == "cwlt_wgrad_bf16" and out["roofline"]["bound"] == "mfma"
    assert abs(out["roofline"]["achieved"] - flops / 0.72e-3 / 1e12) < 0.1
    assert abs(out["roofline"]["frac"] - out["roofline"]["achieved"] / 2500.0) < 1e-3
    hb = out["roofline_hbm"]
    assert hb["kernel"] == bench.SCAN_BWD and hb["bound"] == "hbm"
    assert hb["algorithmic_bytes_per_launch"] == R * (7 * 512 * 2 + 32)
    assert abs(hb["avg_launch_ms"] - 1.75) < 1e-9 and abs(hb["achieved"] - hb["algorithmic_bytes_per_launch"] / 1.75e-3 / 1e9) < 0.1
    assert out["kernels"]["cwlt_causal_linear_bwd_dq"]["part_of"] == bench.SCAN_BWD
    assert "GB/s" not in out["kernels"]["cwlt_causal_linear_bwd_dq"]
    assert out["kernels"]["cwlt_wgrad_bf16"]["flop_per_step_declared"] == out["kernels"]["cwlt_wgrad_bf16"]["flop_per_step_shapes"]
    # with the one-pass input front there is no in_linear weight-gradient GEMM over the token rows: 49 calls per step
    kt2 = dict(kt)
    kt2["cwlt_wgrad_bf16"] = (49, 0.72, bench.wgrad_flops_per_step(B, T) / 49)
    kt2["cwlt_cw_embed_proj_fwd"] = (1, 0.42, None)
    kt2["cwlt_cw_embed_proj_bwd"] = (1, 0.40, None)
    out2 = bench.report(_args(), kt2, B, T, 2, 1, 190.0, B * T / 0.19, 3.9, None, True, 112.0)
    w2 = out2["kernels"]["cwlt_wgrad_bf16"]
    assert abs(w2["flop_per_step_declared"] - w2["flop_per_step_shapes"]) < 1e-6 * w2["flop_per_step_shapes"]
    assert w2["flop_per_step_shapes"] < out["kernels"]["cwlt_wgrad_bf16"]["flop_per_step_shapes"]
    assert abs(out2["kernels"]["cwlt_cw_embed_proj_fwd"]["GB/s"] - R * (48 + 512 * 2) / 0.42e-3 / 1e9) < 0.1
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in out, k
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in out["roofline"] and k in hb, k
    # a merged single-launch backward (what the kernel becomes) is priced the same way
    kt2 = dict(kt)
    del kt2["cwlt_causal_linear_bwd_dkdv"], kt2["cwlt_causal_linear_bwd_dq"]
    kt2["cwlt_causal_linear_bwd"] = (12, 1.2, None)
    out2 = bench.report(_args(), kt2, B, T, 2, 1, 190.0, B * T / 0.19, 3.9, None, True, 112.0)
    assert out2["kernels"][bench.SCAN_BWD]["GB/s"] == round(R * (7 * 512 * 2 + 32) / 1.2e-3 / 1e9, 1)


def test_ppo_report_block():
    sys.path.insert(0, ROOT)
    import bench
    ppo = {"env_steps_per_s": 90.0, "rollout_only_env_steps_per_s": 2400.0, "ms_per_iteration": 21333.0,
           "replica_spread": None, "hipgraph_rollout": True,
           "kernel_times": {"cwlt_wgrad_bf16": (1000, 3.0, 2e12), "cwlt_causal_linear_fwd": (500, 1.0, None)}}
    r = bench.ppo_report(_args(), ppo, 1)
    assert r["metric"] == "PPO env-steps/sec" and r["unit"] == "env-steps/s" and r["value"] == 90.0
    assert r["dominant_kernel"]["kernel"] == "cwlt_wgrad_bf16" and r["dominant_kernel"]["bound"] == "mfma"
    assert "workload" in r["config"] and r["config"]["rollouts_per_gpu"] == 64 and r["config"]["window"] == 1024
