#!/usr/bin/env python3
"""Benchmark of the hot path: agent pretrain step of the CW Linear Transformer at repo dims.

    python bench.py --gpus N --steps K --warmup W          (N > 1: under torch.distributed.run, or plain -- the
                                                            script then starts the N ranks itself, bench_launch.py)

Workload = BASELINE.json configs[1]: full agent pretrain (d_model 512, 12 layers, 8 heads, d_ff 2048;
dqn vocabulary [56,135,18,87,18,25]) on synthetic (B, T=1024, 7-field) compound-word sequences,
bf16 activations / MFMA GEMMs, f32 scan state + statistics + master weights, dropout 0.1 live.
One step = forward + backward + (DP gradient all-reduce) + clip_grad_norm_(3) + Adam, i.e. the body of
the reference's loop at dqn_policy/agent_pretrain.py:546-565 -- inputs already resident in HBM.
Prints ONE JSON line (rank 0).  `value` = CW tokens/s of the whole job.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

if __name__ == "__main__":
    # `--gpus N` (N > 1) without a launcher's rank environment: start the ranks as a child torch.distributed.run and
    # relay its line and exit code -- before anything in this process could touch the GPU
    import bench_launch
    _rc = bench_launch.maybe_self_launch(__file__)
    if _rc is not None:
        raise SystemExit(_rc)

import torch  # noqa: E402

N_CLASS_DISK = [56, 135, 18, 3, 87, 18, 25]      # 7 on-disk fields incl. `type` (dropped: agent_pretrain.py:524-526)
HBM_PEAK_GBS = 8000.0                            # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_BF16_PEAK = 2.5e15                          # dense bf16
FLOP_PER_TOKEN = 236e6                           # fwd+bwd, repo dims
PPO_GROUP = 16                                   # rollouts stacked per PPO update pass (bench_ppo.py --group)


def synth_batch(B, T, seed, device):
    g = torch.Generator().manual_seed(seed)
    full = torch.stack([torch.randint(0, n, (2, B, T), generator=g) for n in N_CLASS_DISK], -1)   # (2,B,T,7)
    keep = [0, 1, 2, 4, 5, 6]
    x, y = full[0][..., keep].contiguous(), full[1][..., keep].contiguous()
    mask = torch.ones(B, T)
    mask[: B // 2, int(T * 0.9):] = 0.0            # last 10 % masked for half the rows
    return x.to(device), y.to(device), mask.to(device)


SCAN_BWD = "cwlt_causal_linear_bwd"              # the backward of one attention call, however many launches it takes
SCAN_BWD_PARTS = ("cwlt_causal_linear_bwd", "cwlt_causal_linear_bwd_sweep", "cwlt_causal_linear_bwd_dkdv",
                  "cwlt_causal_linear_bwd_dq")


def algorithmic_bytes(entry, B, T, D=512, F=2048, H=8, s=2, ncat=1216, W=384):
    """Algorithmic HBM bytes of ONE launch of a libcwlt entry point at this workload.
    s = bytes per activation element.  The attention backward is priced as ONE unit whatever its launch count:
    7 * D * s per token (Q, K, V, dOut read; dQ, dK, dV written) -- re-reads by a second kernel, and the one-sweep
    kernel's read of `out` (needed only for dden), are waste, not work.  Likewise the forward activation is priced at
    2 * F * s although, with the fused FFN backward, it also writes gd (a third stream, the backward's activation
    derivative): `traffic_GB/s` next to `achieved` shows what the kernel really moves."""
    R = B * T
    return {
        "cwlt_causal_linear_fwd": R * (4 * D * s + H * 4),
        SCAN_BWD: R * (7 * D * s + H * 4),
        "cwlt_add_dropout_layernorm_fwd": R * (4 * D * s + 8),
        "cwlt_add_dropout_layernorm_bwd": R * (5 * D * s + 8),
        # the out-projection's residual block (K = D): a, x read; s, y written
        "cwlt_gemm_nt_bias_dropout_add_layernorm": R * (D + 3 * D) * s,
        "cwlt_bias_gelu_dropout_fwd": R * 2 * F * s,
        "cwlt_bias_gelu_dropout_bwd": R * 3 * F * s,
        "cwlt_colsum": R * 3 * D * s,
        "cwlt_posenc_dropout": R * 2 * D * s,
        "cwlt_cw_embed_fwd": R * (48 + ncat * s),
        "cwlt_cw_embed_bwd": R * (48 + ncat * s),
        # the one-pass input front (ops.EmbedProjFn): ids in, (R, D) activations out / gradient in (tables are L2-resident)
        "cwlt_cw_embed_proj_fwd": R * (48 + D * s),
        "cwlt_cw_embed_proj_bwd": R * (48 + D * s),
        "cwlt_heads_fwd": R * (W * s + 48 + 4),
        "cwlt_heads_ce_bwd": R * (2 * W * s + 48 + 4),
    }.get(entry)


def wgrad_flops_per_step(B, T, D=512, F=2048, ncat=1216, W=384, in_linear=False):
    """FLOPs of all weight-gradient GEMMs of one step, shape by shape: per layer dW2 (D x F), dW1 (F x D), dWo (D x D),
    dWqkv (3D x D); once the fused head projection (384 x D) and -- only with the three-kernel input front
    (CWLT_EMBED_PROJ=0) -- in_linear (D x 1216): the one-pass front has no GEMM over the token rows.  (Cross-check of the
    per-call figures the launches themselves declare through ops._call(work=...).)"""
    R = B * T
    return 2.0 * R * (12 * (2 * D * F + D * D + 3 * D * D) + (D * ncat if in_linear else 0) + W * D)


def usable_cores():
    """CPU threads this process may really use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 64))


def log(msg):
    print("[bench] " + msg, file=sys.stderr, flush=True)


def cpu_baseline(seconds_budget=30.0):
    """The oracle (CPU restatement of the reference path, fp32, all host cores) on a bounded sample:
    B=1, T=1024 pretrain steps (fwd + bwd + clip + Adam, dropout live)."""
    from oracle import cw_model
    cores = usable_cores()
    torch.set_num_threads(cores)
    log("cpu baseline on %d threads ..." % cores)
    torch.manual_seed(0)
    net = cw_model.CWLinearTransformer([56, 135, 18, 87, 18, 25], 512, 12, 8, variant="dqn").train()
    opt = torch.optim.Adam(net.parameters(), lr=1e-4)
    x, y, mask = synth_batch(1, 1024, 1234, "cpu")

    def step():
        losses = net.train_step(x, y, mask)
        loss = sum(losses) / 6
        net.zero_grad()
        loss.backward()
        torch.nn.utils.clip_grad_norm_(net.parameters(), 3)
        opt.step()

    tw = time.perf_counter()
    step()                                            # warm-up (allocator, thread pool)
    log("cpu baseline warm-up step %.1f s" % (time.perf_counter() - tw))
    t0 = time.perf_counter()
    n = 0
    while n < 1 or (time.perf_counter() - t0 < seconds_budget * 0.5 and n < 8):
        step()
        n += 1
    dt = (time.perf_counter() - t0) / n
    return {"value": round(1024 / dt, 2), "unit": "CW-tokens/s", "cores": cores, "kind": "port",
            "sample": "%d steps of B=1 x T=1024 (fwd+bwd+clip+Adam, fp32, dropout 0.1), oracle/cw_model.py" % n}


def roofline_entry(name, kind, achieved, avg_ms, share, work_per_launch, traffic=None):
    peak = HBM_PEAK_GBS if kind == "hbm" else MFMA_BF16_PEAK / 1e12
    return {"kernel": name, "bound": kind, "achieved": round(achieved, 1), "peak": peak,
            "unit": "GB/s" if kind == "hbm" else "TFLOP/s", "frac": round(achieved / peak, 4), "traffic": traffic,
            ("algorithmic_bytes_per_launch" if kind == "hbm" else "flop_per_launch"): work_per_launch,
            "avg_launch_ms": round(avg_ms, 4), "share_of_step": round(share, 4)}


def kernel_table(kt, B, T, s, ms_per_step, n_sampled):
    """-> (kernels dict for the JSON line, list of roofline candidates (total ms per step, entry))."""
    kernels, cands = {}, []
    merged = dict(kt)
    parts = [k for k in SCAN_BWD_PARTS if k in kt]
    if parts:                                   # the attention backward as ONE unit: launches per call summed
        calls = max(kt[k][0] for k in parts)
        merged[SCAN_BWD] = (calls, sum(kt[k][0] * kt[k][1] for k in parts) / calls, None)
    for k, (c, m, work) in sorted(merged.items(), key=lambda kv: -kv[1][0] * kv[1][1]):
        own = k in kt
        if k in SCAN_BWD_PARTS and k != SCAN_BWD:
            kernels[k] = {"calls_per_step": c / n_sampled, "avg_ms": round(m, 4), "part_of": SCAN_BWD}
            continue
        ab = algorithmic_bytes(k, B, T, s=s)
        e = {"calls_per_step": c / n_sampled, "avg_ms": round(m, 4),
             "share_of_step": round(c * m / n_sampled / ms_per_step, 4)}
        if ab:
            e["GB/s"] = round(ab / (m * 1e-3) / 1e9, 1)
            cands.append((c * m / n_sampled, roofline_entry(k, "hbm", ab / (m * 1e-3) / 1e9, m,
                                                            c * m / n_sampled / ms_per_step, ab)))
        if work:
            e["TFLOP/s"] = round(work / (m * 1e-3) / 1e12, 1)
            cands.append((c * m / n_sampled, roofline_entry(k, "mfma", work / (m * 1e-3) / 1e12, m,
                                                            c * m / n_sampled / ms_per_step, work)))
        if not own:
            e["launches"] = parts
        kernels[k] = e
    return kernels, cands


def report(args, kt, B, T, s, world, ms_per_step, tokens_per_s, final_loss, replica_spread, tuned, hbm_peak):
    roofline = roofline_hbm = roofline_mfma = None
    kernels = {}
    if kt:
        n_sampled = len([i for i in range(args.steps) if i % 10 == 0])     # steps on which events were recorded
        kernels, cands = kernel_table(kt, B, T, s, ms_per_step, n_sampled)
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        traffic = {}
        if os.path.exists(tpath):
            tj = json.load(open(tpath))
            if tj.get("batch") == B and tj.get("seq") == T and tj.get("dtype") == args.dtype:
                traffic = tj.get("per_launch_bytes", {})
        for _, e in cands:
            e["traffic"] = traffic.get(e["kernel"])
            if e["traffic"] and e["bound"] == "hbm":
                # bytes the counters saw per launch / this run's launch time, and how far they exceed the algorithmic ones
                e["traffic_GB/s"] = round(e["traffic"] / (e["avg_launch_ms"] * 1e-3) / 1e9, 1)
                e["traffic_over_algorithmic"] = round(e["traffic"] / e["algorithmic_bytes_per_launch"], 3)
        cands.sort(key=lambda t: -t[0])
        # `roofline`: the libcwlt entry point with the largest total time in the step, whatever bounds it;
        # `roofline_hbm` / `roofline_mfma`: the largest one of each kind
        roofline = cands[0][1]
        roofline_hbm = next((e for _, e in cands if e["bound"] == "hbm"), None)
        roofline_mfma = next((e for _, e in cands if e["bound"] == "mfma"), None)
        if "cwlt_wgrad_bf16" in kt:                 # cross-check: declared per-call FLOPs vs the shape table
            c, m, work = kt["cwlt_wgrad_bf16"]
            kernels["cwlt_wgrad_bf16"]["flop_per_step_declared"] = work * c / n_sampled
            kernels["cwlt_wgrad_bf16"]["flop_per_step_shapes"] = wgrad_flops_per_step(
                B, T, in_linear="cwlt_cw_embed_proj_fwd" not in kernels)
    return {
        "metric": "CW-tokens/sec pretrain fwd+bwd @T=1024", "value": round(tokens_per_s, 1), "unit": "CW-tokens/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": "agent_pretrain step (fwd+bwd+clip+Adam), repo dims 512/12/8/2048, "
                               "synthetic CW tokens (B=%d/GPU, T=%d, 7 fields -> 6)" % (B, T),
                   "per_gpu_batch": B, "global_batch": B * world, "seq_len": T,
                   "parallelism": "dp%d" % world, "dropout": 0.1, "params": 38982227,
                   "gemm_table": str(tuned)},
        "final_loss": final_loss, "replica_spread": replica_spread, "hbm_peak_gb": hbm_peak,
        "model_mfma_frac": round(tokens_per_s / world * FLOP_PER_TOKEN / MFMA_BF16_PEAK, 4),
        "roofline": roofline, "roofline_hbm": roofline_hbm, "roofline_mfma": roofline_mfma, "kernels": kernels,
    }


def ppo_report(args, ppo, world):
    """BASELINE.json metric, second half: PPO env-steps/s on configs[2] (64 rollouts x window 1024 per GPU)."""
    R, W = args.ppo_rollouts, args.ppo_window
    kt = ppo["kernel_times"]
    dom = None
    if kt:
        tot = {k: c * m for k, (c, m, _) in kt.items()}
        k = max(tot, key=tot.get)
        c, m, work = kt[k]
        dom = {"kernel": k, "calls_per_iteration": c, "avg_ms": round(m, 4),
               "share_of_iteration": round(tot[k] / ppo["ms_per_iteration"], 4)}
        if work:
            dom.update({"bound": "mfma", "TFLOP/s": round(work / (m * 1e-3) / 1e12, 1),
                        "frac": round(work / (m * 1e-3) / MFMA_BF16_PEAK, 4)})
    return {"metric": "PPO env-steps/sec", "value": round(ppo["env_steps_per_s"], 2), "unit": "env-steps/s",
            "rollout_only_env_steps_per_s": round(ppo["rollout_only_env_steps_per_s"], 2),
            "ms_per_iteration": round(ppo["ms_per_iteration"], 1), "n_gpus": world, "scaling": "weak",
            "replica_spread": ppo["replica_spread"], "hbm_peak_gb": ppo.get("hbm_peak_gb"), "dominant_kernel": dom,
            "config": {"workload": "ppo_train iteration (ppo_policy/ppo_train.py:460-506): %d rollouts/GPU x window %d, "
                                   "EPISODES 30, PPO_STEPS %d, actor/critic 512/12/8, reward Longformer 512/12/8 w=512; "
                                   "env-step = actor greedy fwd + critic value + reward model + buffer write; update: "
                                   "select_udpate's actor pass on each rollout's last state only -- the one whose rows "
                                   "the reference returns (ppo_train.py:346; CWLT_PPO_SELECT_ALL=1 runs all 30) --, "
                                   "CE pass, critic pass, both backwards, two Adam steps; "
                                   "1 warm-up + 1 timed iteration" % (R, W, args.ppo_steps),
                       "rollouts_per_gpu": R, "window": W, "episodes": 30, "ppo_steps": args.ppo_steps,
                       "update_group": ppo.get("update_group", PPO_GROUP), "hipgraph_rollout": ppo["hipgraph_rollout"],
                       "select_pass": ppo.get("select_pass", "last-state")}}


def run_ppo_block(run, group, nranks, device, log, *a, **kw):
    """bench_ppo.run with the out-of-memory fallback: 16 rollouts x 30 states x window 1024 per update pass peak at
    ~207 GB of the 288; should a box have less to give, the block is measured with half the rollouts per pass rather than
    lost.  The retry happens OUTSIDE the except block (inside it the exception's traceback keeps the failed run's frames
    -- models, rollout buffers -- alive, and nothing would be freed), and under data parallelism every rank takes the
    same decision: the flag is all-reduced (MAX), so a rank that did fit retries with the smaller group as well (the
    update passes hold collectives; ranks with different pass counts would hang)."""
    import gc
    while True:
        oom = False
        try:
            res = run(*a, group=group, **kw)
        except torch.OutOfMemoryError:
            oom, res = True, None
        if nranks > 1:
            flag = torch.tensor([1.0 if oom else 0.0], device=device)
            torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MAX)
            oom = bool(flag.item() > 0)
        if not oom:
            return res, group
        res = None
        gc.collect()
        if torch.cuda.is_available():
            torch.cuda.empty_cache()
        if group <= 1:
            raise RuntimeError("ppo block: out of memory even at one rollout per update pass")
        log("ppo block: out of memory at %d rollouts per update pass, measuring with %d" % (group, group // 2))
        group //= 2


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1,
                    help="ranks = GPUs of this node; N > 1 outside a launcher starts torch.distributed.run itself")
    ap.add_argument("--dry-run-launch", action="store_true", help="print the rank launcher's command line and exit")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=512, help="sequences per GPU")
    ap.add_argument("--seq", type=int, default=1024)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timer", action="store_true")
    ap.add_argument("--tune-gemms", action="store_true", help="re-tune the hipBLASLt/rocBLAS table (GPU box)")
    ap.add_argument("--no-tuned-gemms", action="store_true")
    ap.add_argument("--no-ppo", action="store_true", help="skip the PPO env-steps/s block (BASELINE's second metric)")
    ap.add_argument("--ppo-rollouts", type=int, default=64)
    ap.add_argument("--ppo-window", type=int, default=1024)
    ap.add_argument("--ppo-steps", type=int, default=10)
    args = ap.parse_args()

    import rlmg_amd  # noqa: F401
    from rlmg_amd import dist as rdist, gemm_tuning, ops
    from rlmg_amd.dqn_policy import model

    rank, local, world = rdist.init_from_env()
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the cwlt hot path has no CPU implementation")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    if args.tune_gemms:
        gemm_tuning.tune(os.path.join(ROOT, "gpurun_out", "gemm_gfx950.csv"))
        tuned = "tuning"
    else:
        tuned = (not args.no_tuned_gemms) and gemm_tuning.enable()
    log("tuned GEMM table: %s" % tuned)
    torch.manual_seed(0)                               # identical replicas on every rank
    n_class = [56, 135, 18, 87, 18, 25]
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        net = model.LinearTransformer(n_class)
    net = net.to(dev).train()
    net.compute_dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    sync = rdist.GradSync(net.parameters())
    opt = torch.optim.Adam(net.parameters(), lr=1e-4, fused=True)
    B, T = args.batch, args.seq
    x, y, mask = synth_batch(B, T, 1234 + rank, dev)   # weak scaling: B sequences per GPU, distinct per rank
    torch.manual_seed(100 + rank)                      # dropout streams differ per rank

    def step():
        sync.zero_grad()
        losses = net.train_step(x, y, mask)
        loss = (losses[0] + losses[1] + losses[2] + losses[3] + losses[4] + losses[5]) / 6
        loss.backward()
        sync.finish()
        sync.clip_grad_norm_(3.0)
        opt.step()
        return loss

    def fence():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step()
        if i == 0:
            torch.cuda.synchronize()
            log("first step done")
            if args.tune_gemms:
                gemm_tuning.save(os.path.join(ROOT, "gpurun_out", "gemm_gfx950.csv"))
                log("tuned table written")
    fence()
    log("warm-up done, timing %d steps" % args.steps)
    # A pair of HIP timing events around every C-ABI call (~190 calls per step) makes the GPU drain between kernels:
    # measured +25 ms on a 106 ms step.  The per-kernel durations therefore come from every 10th timed step only
    # (step 0 of the default 10): still live, inside the timed region, on the launch stream; cost ~2 % of `value`.
    timing = rank == 0 and not args.no_kernel_timer
    ops.KernelTimer.reset(False)
    if timing:
        ops.KernelTimer.reserve(512 * len([i for i in range(args.steps) if i % 10 == 0]))
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        ops.KernelTimer.enabled = timing and i % 10 == 0
        loss = step()
    fence()
    dt = time.perf_counter() - t0
    ops.KernelTimer.enabled = False
    log("timed region %.3f s" % dt)
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = t.item()
    ms_per_step = 1e3 * dt / args.steps
    tokens_per_s = world * B * T * args.steps / dt
    replica_spread = None
    kt_pretrain = ops.KernelTimer.summary() if rank == 0 else {}
    ops.KernelTimer.reset(False)
    final_loss = round(float(loss.item()), 4)
    hbm_peak = round(torch.cuda.max_memory_allocated() / 1e9, 1)
    if world > 1:
        # data-parallel sanity (outside the timed region): every rank must hold the same parameters after the
        # all-reduced steps; spread = max over ranks - min over ranks of a parameter checksum
        chk = torch.stack([p.detach().double().sum() for p in net.parameters()]).sum().reshape(1)
        hi, lo = chk.clone(), chk.clone()
        torch.distributed.all_reduce(hi, op=torch.distributed.ReduceOp.MAX)
        torch.distributed.all_reduce(lo, op=torch.distributed.ReduceOp.MIN)
        replica_spread = float((hi - lo).item())
        log("replica parameter checksum spread: %.3e" % replica_spread)

    ppo = None
    if not args.no_ppo:
        # BASELINE.json's second metric (PPO env-steps/s, configs[2]: 64 rollouts x window 1024 per GPU), measured in
        # the same driver-run process after the pretrain region; the pretrain model and its optimizer are released first
        del net, opt, sync, x, y, mask, loss
        import gc
        gc.collect()
        torch.cuda.empty_cache()
        import bench_ppo
        log("ppo block: %d rollouts x window %d, EPISODES %d, PPO_STEPS %d" % (args.ppo_rollouts, args.ppo_window, 30,
                                                                              args.ppo_steps))
        ppo, ppo_group = run_ppo_block(bench_ppo.run, PPO_GROUP, world, dev, log,
                                       args.ppo_rollouts, args.ppo_window, 30, args.ppo_steps, iters=1, warmup=1,
                                       dtype=args.dtype, rank=rank, world=world, dev=dev, timer=True)
        ppo["update_group"] = ppo_group

    if rank == 0:
        s = 2 if args.dtype == "bf16" else 4
        kt = kt_pretrain
        out = report(args, kt, B, T, s, world, ms_per_step, tokens_per_s, final_loss, replica_spread, tuned, hbm_peak)
        if ppo is not None:
            out["ppo"] = ppo_report(args, ppo, world)
        if world == 1 and not args.no_cpu_baseline:
            cb = cpu_baseline()
            out["cpu_baseline"] = cb
            out["gpu_over_cpu"] = round(tokens_per_s / cb["value"], 1)
            if ppo is not None:
                import bench_ppo
                pc = bench_ppo.cpu_rollout_baseline(args.ppo_window)
                out["ppo"]["cpu_baseline"] = pc
                # the CPU figure is ONE rollout per step, the GPU figure args.ppo_rollouts rollouts in lock-step: both batch
                # sizes are stated next to the ratio, and the ratio per rollout in flight beside it
                out["ppo"]["gpu_over_cpu_rollout_only"] = {
                    "ratio": round(out["ppo"]["rollout_only_env_steps_per_s"] / pc["value"], 1),
                    "gpu_rollouts_per_step": args.ppo_rollouts, "cpu_rollouts_per_step": 1,
                    "ratio_per_gpu_rollout": round(out["ppo"]["rollout_only_env_steps_per_s"] / args.ppo_rollouts
                                                   / pc["value"], 2)}
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
