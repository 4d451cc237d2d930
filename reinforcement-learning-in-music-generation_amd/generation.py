"""Token-by-token generation on the recurrent form of the encoder.

Surface of the reference's generation scripts (dqn_policy/testing-no-type-cp.py:126-223,
dqn_policy/agent_pretrain.py:636-706, ppo_policy/inference.py:78-160):
    res = inference_from_scratch(model, word2event, bar_cond)      # (n_tokens, 6) int64 numpy
    generate(model, word2event, ...)                               # songs + runtime_stats.json
`model` is a `LinearTransformer` / `Actor_Transformer` built with `is_training=False`.

`DecodeSession` is the device side of one song: the 12 x [S (1,H,64,64), Zs (1,H,64)] state lives in HBM
and is updated in place, the CW token is written into a static (1,1,6) buffer, and the whole per-token
step (embedding gather -> in_linear -> 12 recurrent layers -> final LN -> fused 6-head GEMV) is ONE
hipGraph replay; the only host traffic per token is 48 B of ids in and sum(n_token) f32 logits out.
Sampling stays on the host with numpy, in the reference's order of draws, so that a seeded
`np.random` reproduces the reference's token stream.
"""
import json
import os
import time

import numpy as np
import torch

from . import ops
from .sampling import sample_cw

INIT_CW = np.array([[0, 0, 1, 0, 0, 0]])          # "Bar" token, testing-no-type-cp.py:135-137


class DecodeSession:
    """Per-song decode state + the captured one-token step.  `step(ids) -> (sum n_token,) f32 numpy logits`."""

    def __init__(self, model, graph=None):
        if not getattr(model, "_recurrent", False):
            raise RuntimeError("generation needs a model built with is_training=False (recurrent encoder)")
        p = next(model.parameters())
        if not p.is_cuda:
            raise RuntimeError("rlmg_amd models run on the GPU only (no CPU fallback): call .cuda() first")
        self.model, self.dev = model, p.device
        self.n_token = list(model.n_token)
        self.width = sum(self.n_token)
        enc = model.transformer_encoder
        H = enc.layers[0].attention.n_heads
        d = model.d_model // H
        self.tok = torch.zeros((1, 1, len(self.n_token)), dtype=torch.int64, device=self.dev)
        self.memory = [[torch.zeros((1, H, d, d), dtype=torch.float32, device=self.dev),
                        torch.zeros((1, H, d), dtype=torch.float32, device=self.dev)] for _ in enc.layers]
        self._host_tok = torch.zeros((1, 1, len(self.n_token)), dtype=torch.int64).pin_memory()
        self._host_logits = torch.zeros((self.width,), dtype=torch.float32).pin_memory()
        self.use_graph = ops.GRAPHS_ENABLED if graph is None else bool(graph)
        self._graph, self._out = None, None
        self.n_steps = 0

    def reset(self):
        for S, Z in self.memory:
            S.zero_()
            Z.zero_()
        self.n_steps = 0

    def _device_step(self):
        """testing-no-type-cp.py:150 / :166 (`forward_hidden(input_, memory, is_training=False)`) followed by the six
        head projections of forward_output_sampling (dqn_policy/model.py:273-278) as one fused GEMV."""
        h, mem = self.model.forward_hidden(self.tok, self.memory, is_training=False)
        for (S, Z), (S2, Z2) in zip(self.memory, mem):
            if S2.data_ptr() != S.data_ptr() or Z2.data_ptr() != Z.data_ptr():
                raise RuntimeError("recurrent state must be updated in place")
        return self.model.fused_logits(h).float()[0, :self.width].contiguous()

    def _capture(self):
        saved = [(S.clone(), Z.clone()) for S, Z in self.memory]
        side = torch.cuda.Stream(device=self.dev)
        side.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(2):
                self._device_step()
        torch.cuda.current_stream(self.dev).wait_stream(side)
        torch.cuda.synchronize(self.dev)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g), torch.no_grad():
            out = self._device_step()
        for (S, Z), (S0, Z0) in zip(self.memory, saved):      # warm-up and capture ran the step: restore
            S.copy_(S0)
            Z.copy_(Z0)
        self._graph, self._out = g, out

    def step(self, ids):
        """Feed one CW token (6 ids), advance the state, return the next-token logits (host numpy, f32)."""
        if self.model.training:
            raise RuntimeError("generation runs in eval() mode (agent_pretrain.py:657)")
        self._host_tok.view(-1).copy_(torch.as_tensor(np.asarray(ids, dtype=np.int64).reshape(-1)))
        self.tok.copy_(self._host_tok, non_blocking=True)
        if self.use_graph:
            if self._graph is None:
                self._capture()
            self._graph.replay()
            out = self._out
        else:
            with torch.no_grad():
                out = self._device_step()
        self._host_logits.copy_(out, non_blocking=True)
        torch.cuda.current_stream(self.dev).synchronize()
        self.n_steps += 1
        return self._host_logits.numpy()

    def split(self, logits):
        outs, o = [], 0
        for n in self.n_token:
            outs.append(logits[o:o + n])
            o += n
        return outs


def inference_from_scratch(model, word2event, bar_cond, max_tokens=None, log=None, session=None):
    """testing-no-type-cp.py:126-179: start from the Bar token, sample until `bar_cond` bars have begun.
    `max_tokens` (not in the reference, whose loop is unbounded) caps the song length."""
    classes = list(word2event.keys())
    sess = session or DecodeSession(model)
    sess.reset()

    def show(cp, prefix=""):
        if log is not None:
            log(prefix + " | ".join("{:15s}".format(str(word2event[k][int(cp[i])])) for i, k in enumerate(classes)))

    final_res = []
    cnt_bar = 1
    logits = None
    for row in INIT_CW:
        show(row)
        final_res.append(row[None, ...])
        logits = sess.step(row)
    while True:
        next_arr = sample_cw(sess.split(logits))
        final_res.append(next_arr[None, ...])
        show(next_arr, "bar: %d  ==" % cnt_bar)
        logits = sess.step(next_arr)
        if word2event["bar-beat"][int(next_arr[2])] == "Bar":
            cnt_bar += 1
        if cnt_bar == bar_cond:
            break
        if max_tokens is not None and len(final_res) >= max_tokens:
            break
    return np.concatenate(final_res)


def generate(model, word2event, n_songs=1, bar_cond=17, path_gendir="./gen_midis", write_midi=None,
             max_tokens=None, stats_path="runtime_stats.json", log=print):
    """testing-no-type-cp.py:182-223 / agent_pretrain.py:663-706: generate `n_songs`, time them, write
    runtime_stats.json with the reference's keys.  `write_midi(res, path, word2event)` is the caller's MIDI writer
    (miditoolkit-based in the reference; out of scope here) -- when None the token array is saved as .npy."""
    os.makedirs(path_gendir, exist_ok=True)
    sess = DecodeSession(model)
    song_time_list, words_len_list = [], []
    for sidx in range(n_songs):
        start = time.time()
        res = inference_from_scratch(model, word2event, bar_cond, max_tokens=max_tokens, session=sess)
        if write_midi is not None:
            write_midi(res, os.path.join(path_gendir, "get_%d.mid" % sidx), word2event)
        else:
            np.save(os.path.join(path_gendir, "get_%d.npy" % sidx), res)
        song_time_list.append(time.time() - start)
        words_len_list.append(len(res))
        log("song %d: %d tokens in %.3f s" % (sidx, len(res), song_time_list[-1]))
    result = {"song_time": song_time_list, "words_len_list": words_len_list,
              "ave token time:": sum(words_len_list) / sum(song_time_list),
              "ave song time": float(np.mean(song_time_list))}
    if stats_path:
        with open(stats_path, "w") as f:
            json.dump(result, f)
    return result
