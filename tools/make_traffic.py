"""profiles/traffic.json from two rocprofv3 counter passes over bench.py (GPU box, then run here):

    rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pmc_fetch -o f --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timer
    rocprofv3 --pmc WRITE_SIZE -d gpurun_out/pmc_write -o w --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timer
    python tools/make_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write [batch]

HBM bytes per LAUNCH of each C-ABI entry point = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: both counters are in KiB, and
FETCH_SIZE counts 16-byte coalesced reads at half size on gfx950 (MI355X_MICROARCH.md, HBM / rocprofv3 section).
An entry point that launches several kernels (wgrad + its reduce, backward + finalize) sums them."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

ENTRY = [("cla_fwd_bf16_kernel", "cwlt_causal_linear_fwd"), ("cla_bwd_sweep_bf16_kernel", "cwlt_causal_linear_bwd_sweep"), ("cla_bwd_dq_bf16_kernel", "cwlt_causal_linear_bwd_dq"),
         ("cla_bwd_dkdv_bf16_kernel", "cwlt_causal_linear_bwd_dkdv"), ("add_dropout_ln_fwd_kernel", "cwlt_add_dropout_layernorm_fwd"),
         ("add_dropout_ln_bwd_kernel", "cwlt_add_dropout_layernorm_bwd"), ("bias_gelu_dropout_fwd_kernel", "cwlt_bias_gelu_dropout_fwd"),
         ("bias_gelu_dropout_bwd_kernel", "cwlt_bias_gelu_dropout_bwd"), ("wgrad_kernel", "cwlt_wgrad_bf16"),
         ("wgrad_reduce_kernel", "cwlt_wgrad_bf16"), ("cw_embed_proj_bwd", "cwlt_cw_embed_proj_bwd"),
         ("cw_embed_proj_fwd", "cwlt_cw_embed_proj_fwd"), ("cw_embed_bwd", "cwlt_cw_embed_bwd"), ("cw_embed_fwd", "cwlt_cw_embed_fwd"),
         ("heads_fwd", "cwlt_heads_fwd"), ("heads_ce_bwd", "cwlt_heads_ce_bwd"), ("posenc_dropout_kernel", "cwlt_posenc_dropout"),
         ("gemm_ln_kernel", "cwlt_gemm_nt_bias_dropout_add_layernorm"), ("gemm_nt_mul_kernel<1", "cwlt_gemm_nt_bias_gelu_dropout"), ("gemm_nt_mul_kernel", "cwlt_gemm_nt_mul"),
         # gemm_bf16.hip's persistent kernel: EPI 4 / 8 are the FFN forms, 0..3 the plain projections (averaged over their shapes)
         ("gemm_bf16_kernel<4", "cwlt_gemm_nt_bias_gelu_dropout"), ("gemm_bf16_kernel<8", "cwlt_gemm_nt_mul"),
         ("gemm_bf16_kernel", "cwlt_gemm_bf16")]
MAIN = {"cwlt_wgrad_bf16": "wgrad_kernel"}      # launches counted by the main kernel of multi-kernel entry points


def read(d, counter):
    tot, n = defaultdict(float), defaultdict(int)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] != counter:
                continue
            name = row["Kernel_Name"]
            for pat, ent in ENTRY:
                if pat in name:
                    tot[ent] += float(row["Counter_Value"])
                    if MAIN.get(ent, pat) == pat or ent not in MAIN:
                        if ent not in MAIN or pat == MAIN[ent]:
                            n[ent] += 1
                    break
    return tot, n


def main():
    fd, wd = sys.argv[1], sys.argv[2]
    ft, fn = read(fd, "FETCH_SIZE")
    wt, wn = read(wd, "WRITE_SIZE")
    out = {}
    for ent in sorted(ft):
        if fn[ent] and wn.get(ent):
            out[ent] = int(round((2.0 * ft[ent] / fn[ent] + wt[ent] / wn[ent]) * 1024))
    # the attention backward as bench.py prices it: ONE unit per attention call = its two launches together
    if "cwlt_causal_linear_bwd_sweep" in out:
        out["cwlt_causal_linear_bwd"] = out["cwlt_causal_linear_bwd_sweep"]
    elif "cwlt_causal_linear_bwd_dkdv" in out and "cwlt_causal_linear_bwd_dq" in out:
        out["cwlt_causal_linear_bwd"] = out["cwlt_causal_linear_bwd_dkdv"] + out["cwlt_causal_linear_bwd_dq"]
    meta = {"batch": int(sys.argv[3]) if len(sys.argv) > 3 else 512, "seq": 1024, "dtype": "bf16",
            "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on bench.py --steps 2 --warmup 1; "
                      "bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 FETCH_SIZE half-count correction for 16-B "
                      "coalesced reads, MI355X_MICROARCH.md); tools/make_traffic.py",
            "per_launch_bytes": out}
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "traffic.json")
    json.dump(meta, open(path, "w"), indent=1)
    for k, v in out.items():
        print("%-36s %8.1f MB" % (k, v / 1e6))


if __name__ == "__main__":
    main()
