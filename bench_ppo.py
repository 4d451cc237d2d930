#!/usr/bin/env python3
"""PPO + IRL fine-tuning throughput (BASELINE.json configs[2]: R parallel rollouts, window W).

    python bench_ppo.py --gpus N --rollouts 64 --window 1024 --iters K --warmup W
    (N > 1: python -m torch.distributed.run --nproc-per-node N ... bench_ppo.py --gpus N ..., or plain: the script
     then starts the N ranks itself as a child torch.distributed.run, bench_launch.py)

One iteration = the body of the reference's outer loop (ppo_policy/ppo_train.py:460-506), generalised from
one rollout to R rollouts per GPU run in lock-step:
  rollout phase, EPISODES steps:  actor greedy forward on (R, W, 6) -> action / log-prob rows (reference
      indexing incl. its quirks) -> next state = first half of the old window + the W/2 action tokens ->
      critic value -> reward model (Longformer, band attention) -> GPU-resident buffer write
  update phase, PPO_STEPS inner steps: per rollout, `select_udpate` on its (EPISODES, W, 6) states (the actor pass on
      the last state only: the one whose rows the reference returns, ppo_train.py:346; the critic on all), ratio-clip
      surrogate + CE vs the expert windows, critic MSE; gradients accumulated over the R rollouts (--group of
      them stacked per network pass), one Adam step per net per inner step (data-parallel all-reduce across
      GPUs when N > 1).
env-step = one (rollout, step) pair.  Prints one JSON line: whole-iteration and rollout-only
env-steps/s.  bf16 activations, dropout live (the reference trains with nets in train() mode).
"""
import argparse
import contextlib
import io
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


def log(msg):
    print("[bench_ppo] " + msg, file=sys.stderr, flush=True)


def run(rollouts=64, window=1024, episodes=30, ppo_steps=10, iters=1, warmup=1, dtype="bf16", group=8,
        rank=0, world=1, dev=None, timer=False):
    """One PPO + IRL workload (see the module docstring) on an ALREADY initialised process group / device.
    -> dict(env_steps_per_s, rollout_only_env_steps_per_s, ms_per_iteration, replica_spread, kernel_times)
    `timer`: one extra, untimed iteration with every libcwlt call bracketed by HIP events (rank 0)."""
    import rlmg_amd  # noqa: F401
    from rlmg_amd import ops, rl_ops
    from rlmg_amd.ppo_policy import config as pcfg, ppo_train as P

    R, W, E = rollouts, window, episodes
    NA = W // 2
    G = max(1, group)
    P.N_ACTIONS = P.NUM_ACTION = NA
    P.N_STATES = P.WINDOW_SIZE = W
    n_token = [49, 19, 19, 89, 67, 25]
    if W + 2 > pcfg.DiscriConfig["MAX_SEQ"]:
        # BASELINE configs[4] (window 4096) exceeds the reward model's position table (2048 in the reference): the
        # weights are random here anyway, so the table is simply built large enough
        pcfg.DiscriConfig["MAX_SEQ"] = W + 2
        log("reward model position table enlarged to %d for window %d" % (W + 2, W))
    torch.manual_seed(0)
    with contextlib.redirect_stdout(io.StringIO()):
        agent = P.PPO(n_token, Pretrain=False)
    adt = torch.bfloat16 if dtype == "bf16" else torch.float32
    for net in (agent.actor_net, agent.critic_net, agent.eval_net):
        net.compute_dtype = adt
    g = torch.Generator().manual_seed(1234 + rank)
    expert = torch.stack([torch.randint(0, n, (R, W + E + W), generator=g) for n in n_token], -1).to(dev)
    mask = torch.ones(R, W + E + W, device=dev)
    torch.manual_seed(100 + rank)

    def iteration():
        torch.cuda.synchronize()            # queued update graphs of the previous iteration must not count as rollout
        t_roll0 = time.perf_counter()
        state = expert[:, :W].clone()
        states = torch.empty((E, R, W, 6), dtype=torch.int64, device=dev)
        logps = torch.empty((E, R, NA, 6), dtype=torch.float32, device=dev)
        values = torch.empty((E, R), dtype=torch.float32, device=dev)
        rewards = torch.empty((E, R), dtype=torch.float32, device=dev)
        with torch.no_grad():
            for t in range(E):
                # actor action -> next state -> critic value -> reward model: one hipGraph replay per step
                _, logp, state, value, reward = agent.rollout_step(state, mask[:, t:t + W])
                values[t], rewards[t] = value.reshape(R), reward.reshape(R)
                states[t], logps[t] = state, logp
        torch.cuda.synchronize()
        t_roll = time.perf_counter() - t_roll0
        old_int = logps.long()                                                  # buffer returns .long() (:122,135)
        rets, advs = [], []
        for r in range(R):
            ret, adv = rl_ops.ppo_returns_adv(rewards[:, r], values[:, r], P.DISCOUNT_FACTOR, True)
            rets.append(ret)
            advs.append(adv)
        for _ in range(ppo_steps):
            agent.update_rollouts(states, old_int, advs, rets, expert, mask, group=G)
        return t_roll

    def fence():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for i in range(warmup):
        iteration()
        log("warm-up iteration %d done" % i)
    fence()
    ops.KernelTimer.reset(False)
    t0 = time.perf_counter()
    t_roll = 0.0
    for i in range(iters):
        t_roll += iteration()
        log("iteration %d done" % i)
    fence()
    dt = time.perf_counter() - t0
    kernel_times = {}
    if timer:
        # one EXTRA iteration, outside the timed region, with a pair of HIP events around every libcwlt call (the
        # events make the GPU drain between kernels, so they must not sit inside the measured iterations)
        if rank == 0:
            ops.KernelTimer.reserve(8192)
            torch.cuda.synchronize()
            ops.KernelTimer.enabled = True
        iteration()
        fence()
        ops.KernelTimer.enabled = False
        kernel_times = ops.KernelTimer.summary() if rank == 0 else {}
        ops.KernelTimer.reset(False)
    replica_spread = None
    if world > 1:
        t = torch.tensor([dt, t_roll], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt, t_roll = t.tolist()
        # data-parallel sanity: actor and critic replicas must be identical after the all-reduced updates
        chk = torch.stack([p.detach().double().sum() for n_ in (agent.actor_net, agent.critic_net)
                           for p in n_.parameters()]).sum().reshape(1)
        hi, lo = chk.clone(), chk.clone()
        torch.distributed.all_reduce(hi, op=torch.distributed.ReduceOp.MAX)
        torch.distributed.all_reduce(lo, op=torch.distributed.ReduceOp.MIN)
        replica_spread = float((hi - lo).item())
        log("replica parameter checksum spread: %.3e" % replica_spread)
    steps = world * R * E * iters
    return {"env_steps_per_s": steps / dt, "rollout_only_env_steps_per_s": steps / t_roll,
            "ms_per_iteration": 1e3 * dt / iters, "replica_spread": replica_spread, "kernel_times": kernel_times,
            "hipgraph_rollout": bool(ops.GRAPHS_ENABLED), "tokens_per_update_pass": E * W * min(G, R),
            "select_pass": "all-states" if P.SELECT_ALL_STATES else "last-state",
            "hbm_peak_gb": round(torch.cuda.max_memory_allocated() / 1e9, 1)}


def cpu_rollout_baseline(window=1024, seconds_budget=20.0):
    """One environment step of the reference's rollout loop (ppo_train.py:475-496: actor greedy action + log-probs,
    next state, critic value, reward model) through the CPU oracle, fp32, all usable host cores -> env-steps/s."""
    import bench as _b
    from oracle import cw_model, discriminator as odisc, rl_math
    import rlmg_amd  # noqa: F401
    from rlmg_amd.ppo_policy import config as pcfg, model as pmodel
    cores = _b.usable_cores()
    torch.set_num_threads(cores)
    n_token = [49, 19, 19, 89, 67, 25]
    torch.manual_seed(0)
    actor = cw_model.CWLinearTransformer(n_token, 512, 12, 8, variant="actor").eval()
    critic = cw_model.CWLinearTransformer(n_token, 512, 12, 8, variant="critic").eval()
    old = dict(pcfg.DiscriConfig)
    pcfg.DiscriConfig["MAX_SEQ"] = max(old["MAX_SEQ"], window + 2)
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            reward = pmodel.LongFormer(n_token)               # CPU parameter container for the oracle's state dict
    finally:
        pcfg.DiscriConfig.update(old)
    sd = {k: v.detach() for k, v in reward.state_dict().items()}
    g = torch.Generator().manual_seed(1)
    state = torch.stack([torch.randint(0, n, (1, window), generator=g) for n in n_token], -1)
    m = torch.ones(1, window, dtype=torch.long)

    def step(state):
        with torch.no_grad():
            ys = actor.forward_output(actor.forward_hidden(state))
            action, _ = rl_math.ppo_choose_action(ys, window // 2)
            nxt = torch.cat((state[0, :window // 2], action), 0).unsqueeze(0)
            critic.value_produce(nxt)
            odisc.ppo_reward_forward(sd, nxt, m, 12, 8, 512)
        return nxt

    step(state)                                               # warm-up
    t0 = time.perf_counter()
    n = 0
    while n < 1 or (time.perf_counter() - t0 < seconds_budget * 0.5 and n < 8):
        state = step(state)
        n += 1
    dt = (time.perf_counter() - t0) / n
    return {"value": round(1.0 / dt, 3), "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": "%d rollout steps of ONE rollout x window %d (actor + critic + reward model forward, fp32), "
                      "oracle/{cw_model,rl_math,discriminator}.py" % (n, window)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1,
                    help="ranks = GPUs of this node; N > 1 outside a launcher starts torch.distributed.run itself")
    ap.add_argument("--dry-run-launch", action="store_true", help="print the rank launcher's command line and exit")
    ap.add_argument("--rollouts", type=int, default=64, help="parallel rollouts per GPU")
    ap.add_argument("--window", type=int, default=1024)
    ap.add_argument("--episodes", type=int, default=30)
    ap.add_argument("--ppo-steps", type=int, default=10)
    ap.add_argument("--iters", type=int, default=1)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--group", type=int, default=16, help="rollouts stacked per update pass (1 = one at a time)")
    ap.add_argument("--tune-gemms", action="store_true", help="extend the GEMM table with this workload's shapes")
    ap.add_argument("--no-graphs", action="store_true", help="launch the rollout step eagerly (no hipGraph)")
    ap.add_argument("--cpu-baseline", action="store_true", help="also time one oracle rollout step on the host cores")
    args = ap.parse_args()

    import rlmg_amd  # noqa: F401
    from rlmg_amd import dist as rdist, gemm_tuning, ops

    if args.no_graphs:
        ops.GRAPHS_ENABLED = False
    rank, local, world = rdist.init_from_env()
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if args.tune_gemms:
        gemm_tuning.tune(os.path.join(ROOT, "gpurun_out", "gemm_gfx950.csv"))
    elif args.rollouts * args.window >= 16384:
        gemm_tuning.enable()
    else:
        # launch-bound shapes (the reference's own 1 rollout x window 50): the table holds no entry for them and
        # TunableOp's per-call lookup costs ~3 us of host time on each of ~280 GEMM calls of an update that is
        # host-bound (tools/profile_host_update.py: 25.2 -> 22.3 ms per DQN.update without it)
        log("tuned GEMM table not loaded: %d token rows per pass are launch-bound" % (args.rollouts * args.window))
    R, W, E = args.rollouts, args.window, args.episodes
    res = run(R, W, E, args.ppo_steps, args.iters, args.warmup, args.dtype, args.group, rank, world, dev)
    if rank == 0:
        out = {
            "metric": "PPO env-steps/sec", "value": round(res["env_steps_per_s"], 2), "unit": "env-steps/s",
            "n_gpus": world, "steps": args.iters, "warmup": args.warmup, "ms_per_step": round(res["ms_per_iteration"], 1),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "rollout_only_env_steps_per_s": round(res["rollout_only_env_steps_per_s"], 2),
            "replica_spread": res["replica_spread"], "hbm_peak_gb": res["hbm_peak_gb"],
            "config": {"workload": "ppo_train iteration: %d rollouts/GPU x window %d, EPISODES %d, PPO_STEPS %d, "
                                   "actor/critic 512/12/8, reward Longformer 512/12/8 w=512" % (R, W, E, args.ppo_steps),
                       "hipgraph_rollout": res["hipgraph_rollout"], "update_group": max(1, args.group),
                       "select_pass": res["select_pass"],
                       "rollouts_per_gpu": R, "window": W, "episodes": E, "ppo_steps": args.ppo_steps,
                       "parallelism": "dp%d" % world}}
        if args.cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_rollout_baseline(W)
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
