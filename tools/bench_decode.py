"""Generation throughput: CW tokens/s of the recurrent decode step at the repo dims (512/12/8), one song.
    python tools/bench_decode.py [--tokens 512] [--no-graph]
Prints one JSON line per mode.  Includes the host-side numpy sampling (as the reference's loop does) and,
separately, the device-only rate (step without sampling)."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rlmg_amd  # noqa: E402,F401
from rlmg_amd import generation  # noqa: E402
from rlmg_amd.sampling import sample_cw  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tokens", type=int, default=512)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--songs", type=int, nargs="*", default=[8, 32], help="extra batched-songs runs (device only)")
    a = ap.parse_args()
    from rlmg_amd.dqn_policy import model
    n_class = [56, 135, 18, 87, 18, 25]
    torch.manual_seed(0)
    net = model.LinearTransformer(n_class, is_training=False).cuda().eval()
    if a.dtype == "bf16":
        net.compute_dtype = torch.bfloat16
    modes = [(g, f) for f in ([False, True] if a.dtype == "f32" else [False]) for g in ([False] if a.no_graph else [False, True])]
    for graph, fused in modes:
        sess = generation.DecodeSession(net, graph=graph, fused=fused)
        np.random.seed(0)
        tok = generation.INIT_CW[0]
        for _ in range(8):
            tok = sample_cw(sess.split(sess.step(tok)))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.tokens):
            tok = sample_cw(sess.split(sess.step(tok)))
        t1 = time.perf_counter()
        for _ in range(a.tokens):
            sess.step(tok)
        t2 = time.perf_counter()
        print(json.dumps({"metric": "decode CW-tokens/s (1 song)", "graph": graph, "fused": fused, "dtype": a.dtype,
                          "with_sampling": round(a.tokens / (t1 - t0), 1),
                          "device_only": round(a.tokens / (t2 - t1), 1),
                          "us_per_token_device": round((t2 - t1) / a.tokens * 1e6, 1)}), flush=True)
    if a.dtype == "f32":
        # device-resident loop (ppo_policy/inference.py style): Categorical draws on the GPU, one host sync per song
        n_tok = 8 * a.tokens
        generation.categorical_rollout(net, 16, carry_memory=True, graph=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        generation.categorical_rollout(net, n_tok, carry_memory=True, graph=True)
        t1 = time.perf_counter()
        print(json.dumps({"metric": "decode CW-tokens/s (1 song, sampling on the device)", "graph": True, "fused": True,
                          "tokens": n_tok, "with_sampling": round(n_tok / (t1 - t0), 1),
                          "us_per_token": round((t1 - t0) / n_tok * 1e6, 1)}), flush=True)
        # the DQN-side loop (inference_from_scratch) with its temperature / nucleus samplers on the device
        w2e = {"bar-beat": {i: "x" for i in range(n_class[2])}}
        w2e = {k: w2e.get(k, {}) for k in ("tempo", "chord", "bar-beat", "pitch", "duration", "velocity")}
        sess = generation.DecodeSession(net, graph=True)
        generation.inference_from_scratch(net, w2e, 10 ** 9, max_tokens=64, session=sess, device_sampling=True)
        t0 = time.perf_counter()
        res = generation.inference_from_scratch(net, w2e, 10 ** 9, max_tokens=n_tok, session=sess, device_sampling=True)
        t1 = time.perf_counter()
        print(json.dumps({"metric": "decode CW-tokens/s (1 song, inference_from_scratch, nucleus sampling on the device)",
                          "tokens": len(res), "with_sampling": round(len(res) / (t1 - t0), 1),
                          "us_per_token": round((t1 - t0) / len(res) * 1e6, 1)}), flush=True)
        for n in a.songs:
            sess = generation.DecodeSession(net, graph=True, fused=True, n_songs=n)
            tok = np.tile(generation.INIT_CW[0], (n, 1))
            for _ in range(8):
                sess.step(tok)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.tokens):
                sess.step(tok)
            t1 = time.perf_counter()
            print(json.dumps({"metric": "decode CW-tokens/s (%d songs in lock-step)" % n, "graph": True, "fused": True,
                              "device_only": round(n * a.tokens / (t1 - t0), 1),
                              "us_per_step_device": round((t1 - t0) / a.tokens * 1e6, 1)}), flush=True)


if __name__ == "__main__":
    main()
