"""Token-by-token generation on the recurrent form of the encoder.

Surface of the reference's generation scripts (dqn_policy/testing-no-type-cp.py:126-223,
dqn_policy/agent_pretrain.py:636-706, ppo_policy/inference.py:78-160):
    res = inference_from_scratch(model, word2event, bar_cond)      # (n_tokens, 6) int64 numpy
    generate(model, word2event, ...)                               # songs + runtime_stats.json
`model` is a `LinearTransformer` / `Actor_Transformer` built with `is_training=False`.

`DecodeSession` is the device side of one song: the 12 x [S (1,H,64,64), Zs (1,H,64)] state lives in HBM
and is updated in place, the CW token is written into a static (1,1,6) buffer, and the whole per-token
step (embedding gather -> in_linear -> 12 recurrent layers -> final LN -> fused 6-head GEMV) is ONE
hipGraph replay of `cwlt_decode_step` (csrc/decode.hip: GEMVs with LayerNorm prologues and bias / GELU /
residual epilogues, 63 launches per token); the only host traffic per token is 48 B of ids in and
sum(n_token) f32 logits out.
Sampling stays on the host with numpy, in the reference's order of draws, so that a seeded
`np.random` reproduces the reference's token stream.
"""
import ctypes
import json
import os
import time

import numpy as np
import torch

from . import ops
from .sampling import sample_cw

INIT_CW = np.array([[0, 0, 1, 0, 0, 0]])          # "Bar" token, testing-no-type-cp.py:135-137


class _FusedPlan:
    """Host description of the model for `cwlt_decode_step` (include/cwlt.h: cwlt_decode_model): stacked
    QKV / head weights, device pointers of every parameter, the per-song state and workspace.  Holds references
    to every tensor whose pointer it hands out.  Built once per song (`DecodeSession.reset` rebuilds it, so
    weights loaded between songs are picked up)."""

    def __init__(self, model, memory, n_songs):
        from . import _lib
        lib = _lib.load()
        enc = model.transformer_encoder
        f32 = lambda t: t.detach().float().contiguous()
        self.keep = []
        self.packed = []           # (buffer, source tensors): row-stacked copies, re-filled in place by refresh()

        def P(t):
            t = f32(t)             # an f32 contiguous parameter is used in place: optimizer steps are seen directly
            self.keep.append(t)
            return _lib.dev(t).value

        def PACK(ts):
            buf = torch.cat([f32(t) for t in ts], 0)
            self.packed.append((buf, list(ts)))
            return _lib.dev(buf).value

        layers = (_lib.DecodeLayer * len(enc.layers))()
        for i, (L, (S, Z)) in enumerate(zip(enc.layers, memory)):
            at = L.attention
            d = layers[i]
            d.wqkv = PACK([at.query_projection.weight, at.key_projection.weight, at.value_projection.weight])
            d.bqkv = PACK([at.query_projection.bias, at.key_projection.bias, at.value_projection.bias])
            d.wo, d.bo = P(at.out_projection.weight), P(at.out_projection.bias)
            d.ln1_w, d.ln1_b = P(L.norm1.weight), P(L.norm1.bias)
            d.w1, d.b1 = P(L.linear1.weight), P(L.linear1.bias)
            d.w2, d.b2 = P(L.linear2.weight), P(L.linear2.bias)
            d.ln2_w, d.ln2_b = P(L.norm2.weight), P(L.norm2.bias)
            d.S, d.Z = _lib.dev(S).value, _lib.dev(Z).value
            for nrm in (L.norm1, L.norm2):
                if abs(nrm.eps - enc.layers[0].norm1.eps) > 0:
                    raise RuntimeError("decode step needs one LayerNorm eps for the whole encoder")
        tables = model._tables()
        heads = model._heads()
        m = _lib.DecodeModel()
        m.n_layer, m.n_head = len(enc.layers), enc.layers[0].attention.n_heads
        m.d_model, m.d_ff = model.d_model, enc.layers[0].linear1.out_features
        m.n_attr, m.emb_width = len(tables), sum(t.shape[1] for t in tables)
        m.n_logits = sum(h.out_features for h in heads)
        m.eps_ln, m.eps_attn = enc.layers[0].norm1.eps, ops.CLA_EPS
        self._tables = (ctypes.c_void_p * len(tables))(*[P(t) for t in tables])
        self._widths = _lib.int_array([t.shape[1] for t in tables])
        self._nrows = _lib.int_array([t.shape[0] for t in tables])
        m.tables = ctypes.cast(self._tables, ctypes.POINTER(ctypes.c_void_p))
        m.widths = ctypes.cast(self._widths, ctypes.POINTER(ctypes.c_int))
        m.nrows = ctypes.cast(self._nrows, ctypes.POINTER(ctypes.c_int))
        m.w_in, m.b_in = P(model.in_linear.weight), P(model.in_linear.bias)
        m.pe0 = P(model.pos_emb.pe[0, 0])
        self._layers = layers
        m.layers = ctypes.cast(layers, ctypes.POINTER(_lib.DecodeLayer))
        if enc.norm is not None:
            if enc.norm.eps != enc.layers[0].norm1.eps:
                raise RuntimeError("decode step needs one LayerNorm eps for the whole encoder")
            m.lnf_w, m.lnf_b = P(enc.norm.weight), P(enc.norm.bias)
        m.w_heads = PACK([h.weight for h in heads])
        m.b_heads = PACK([h.bias for h in heads])
        self.model = m
        per_song = lib.cwlt_decode_workspace_floats(ctypes.byref(m))
        if per_song <= 0:
            raise RuntimeError("cwlt_decode_step does not support this model shape (d_model %d, d_ff %d)"
                               % (m.d_model, m.d_ff))
        dev = memory[0][0].device
        self.work = torch.zeros(n_songs * per_song, dtype=torch.float32, device=dev)
        self.hidden = torch.zeros((n_songs, m.d_model), dtype=torch.float32, device=dev)
        self.logits = torch.zeros((n_songs, m.n_logits), dtype=torch.float32, device=dev)
        self.n_songs = n_songs

    def refresh(self):
        """Re-fill the row-stacked copies from the parameters, in place (pointers, and a captured graph, stay valid)."""
        with torch.no_grad():
            for buf, srcs in self.packed:
                torch.cat([t.detach().float() for t in srcs], 0, out=buf)

    def step(self, tok):
        from . import _lib
        st = _lib.load().cwlt_decode_step(ctypes.byref(self.model), _lib.dev(tok), _lib.dev(self.work),
                                          _lib.dev(self.hidden), _lib.dev(self.logits), self.n_songs,
                                          _lib.stream_ptr())
        _lib.check(st, "cwlt_decode_step")
        return self.logits


class DecodeSession:
    """Decode state of `n_songs` songs + the captured one-token step.
    `step(ids) -> (sum n_token,) f32 numpy logits` (or (n_songs, sum n_token) when n_songs > 1).

    fused=True (default for f32 models): the step is `cwlt_decode_step` (csrc/decode.hip, 5 launches per layer);
    fused=False: the layer-by-layer module path (recurrent.py), the only one for bf16 activations."""

    def __init__(self, model, graph=None, fused=None, n_songs=1):
        if not getattr(model, "_recurrent", False):
            raise RuntimeError("generation needs a model built with is_training=False (recurrent encoder)")
        p = next(model.parameters())
        if not p.is_cuda:
            raise RuntimeError("rlmg_amd models run on the GPU only (no CPU fallback): call .cuda() first")
        self.model, self.dev = model, p.device
        self.n_token = list(model.n_token)
        self.width = sum(self.n_token)
        self.n_songs = int(n_songs)
        if fused is None:
            fused = model.compute_dtype == torch.float32
        if fused and model.compute_dtype != torch.float32:
            raise RuntimeError("the fused decode step computes in f32; use fused=False for bf16 activations")
        if not fused and self.n_songs != 1:
            raise RuntimeError("the module-by-module decode path generates one song at a time (as the reference)")
        self.fused = bool(fused)
        enc = model.transformer_encoder
        H = enc.layers[0].attention.n_heads
        d = model.d_model // H
        n, A = self.n_songs, len(self.n_token)
        self.tok = torch.zeros((n, 1, A), dtype=torch.int64, device=self.dev)
        per = n * H * d * (d + 1)                           # all layers' [S, Zs] in ONE buffer: reset = one memset
        self._state = torch.zeros(per * len(enc.layers), dtype=torch.float32, device=self.dev)
        self.memory = [[self._state[i * per:i * per + n * H * d * d].view(n, H, d, d),
                        self._state[i * per + n * H * d * d:(i + 1) * per].view(n, H, d)]
                       for i in range(len(enc.layers))]
        self._host_tok = torch.zeros((n, 1, A), dtype=torch.int64).pin_memory()
        self._host_logits = torch.zeros((n, self.width), dtype=torch.float32).pin_memory()
        self.use_graph = ops.GRAPHS_ENABLED if graph is None else bool(graph)
        self._graph, self._out, self._plan = None, None, None
        self.census = None                                    # node kinds of the captured step (ops.capture_hip_graph)
        self.hidden = None                                  # (n_songs, d_model) device tensor after a step
        self.n_steps = 0

    def _weights_tag(self):
        return tuple(p.data_ptr() for p in self.model.parameters())

    def reset(self):
        """Start new songs: zero the state and bring the step's weights up to date.  Unstacked f32 parameters are
        read in place; the row-stacked copies (Q/K/V, heads) are re-filled in place -- unconditionally: version
        counters do not see fused optimizers -- so the captured graph stays valid.  Only parameters that moved to
        other storage (.to(), load_state_dict(assign=True)) force a rebuild."""
        self._state.zero_()
        if self.fused and self._plan is not None:
            if self._plan.tag != self._weights_tag():
                self._plan, self._graph = None, None
            else:
                self._plan.refresh()
        self.n_steps = 0

    def _device_step(self):
        """testing-no-type-cp.py:150 / :166 (`forward_hidden(input_, memory, is_training=False)`) followed by the six
        head projections of forward_output_sampling (dqn_policy/model.py:273-278) as one fused GEMV."""
        if self.fused:
            if self._plan is None:
                with torch.no_grad():
                    self._plan = _FusedPlan(self.model, self.memory, self.n_songs)
                self._plan.tag = self._weights_tag()
            out = self._plan.step(self.tok)
            self.hidden = self._plan.hidden
            return out
        h, mem = self.model.forward_hidden(self.tok, self.memory, is_training=False)
        for (S, Z), (S2, Z2) in zip(self.memory, mem):
            if S2.data_ptr() != S.data_ptr() or Z2.data_ptr() != Z.data_ptr():
                raise RuntimeError("recurrent state must be updated in place")
        self.hidden = h
        return self.model.fused_logits(h).float()[:, :self.width].contiguous()

    def _capture(self):
        saved = self._state.clone()
        side = torch.cuda.Stream(device=self.dev)
        side.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(2):
                self._device_step()
        torch.cuda.current_stream(self.dev).wait_stream(side)
        torch.cuda.synchronize(self.dev)
        # ops.capture_hip_graph: memset nodes (none in cwlt_decode_step; the module path's torch ops may add some) are
        # rewritten as kernels before the graph is instantiated; a capture that cannot be made safe is not replayed
        g, out, self.census, _ = ops.capture_hip_graph(self._device_step, torch.no_grad, "decode step")
        self._state.copy_(saved)                              # warm-up ran the step: restore
        if g is None:
            self.use_graph = False
        self._graph, self._out = g, out

    def step(self, ids):
        """Feed one CW token per song (6 ids each), advance the state, return the next-token logits (host numpy)."""
        if self.model.training:
            raise RuntimeError("generation runs in eval() mode (agent_pretrain.py:657)")
        self._host_tok.view(-1).copy_(torch.as_tensor(np.asarray(ids, dtype=np.int64).reshape(-1)))
        self.tok.copy_(self._host_tok, non_blocking=True)
        if self.use_graph:
            if self._graph is None:
                self._capture()
        if self.use_graph:
            self._graph.replay()
            out = self._out
        else:
            with torch.no_grad():
                out = self._device_step()
        self._host_logits.copy_(out, non_blocking=True)
        torch.cuda.current_stream(self.dev).synchronize()
        self.n_steps += 1
        res = self._host_logits.numpy()
        return res[0] if self.n_songs == 1 else res

    def split(self, logits):
        outs, o = [], 0
        for n in self.n_token:
            outs.append(logits[..., o:o + n])
            o += n
        return outs


class _DeviceLoop:
    """Generation loop that never returns to the host: per token the decode step, ONE sampling kernel
    (csrc/sample.hip) that writes the drawn ids into the step's token buffer and into row `count` of `song`, and
    the counter increment -- eager for the first two tokens, then one captured hipGraph replayed per token."""

    def __init__(self, sess, capacity, temperature=None, top_p=None, carry_memory=True, graph=None):
        if sess.n_songs != 1:
            raise RuntimeError("device-side generation loops run one song per session")
        self.sess, self.capacity, self.carry = sess, int(capacity), carry_memory
        self.A = len(sess.n_token)
        self.song = torch.zeros((self.capacity, 1, self.A), dtype=torch.int64, device=sess.dev)
        self.count = torch.zeros(1, dtype=torch.int64, device=sess.dev)
        self.temperature, self.top_p = temperature, top_p
        self.seed = ops.next_seed()                       # keyed from torch.manual_seed, like the dropout seeds
        self.use_graph = ops.GRAPHS_ENABLED if graph is None else bool(graph)
        self._graph, self.enqueued = None, 0

    def _one(self):
        s = self.sess
        if not self.carry:
            s._state.zero_()
        logits = s._device_step()
        ops.sample_categorical(logits, s.n_token, s.tok.view(1, self.A), self.seed, counter=self.count, song=self.song,
                               temperature=self.temperature, top_p=self.top_p)
        self.count.add_(1)

    def run(self, n):
        """Enqueue n more tokens (no host sync)."""
        n = min(n, self.capacity - self.enqueued)
        with torch.no_grad():
            for _ in range(n):
                if not self.use_graph or self.enqueued < 2:
                    self._one()
                else:
                    if self._graph is None:
                        torch.cuda.synchronize(self.sess.dev)
                        self._graph = ops.capture_hip_graph(self._one, torch.no_grad, "decode loop")[0]   # recorded,
                        if self._graph is None:                                                      # not executed
                            self.use_graph = False
                            self._one()
                            self.enqueued += 1
                            continue
                    self._graph.replay()
                self.enqueued += 1
        return n

    def tokens(self, start, stop):
        """Rows [start, stop) of the song as host numpy (syncs)."""
        if int(self.count.item()) < stop:
            raise RuntimeError("device generation loop produced %d of %d tokens" % (int(self.count.item()), stop))
        return self.song[start:stop, 0].cpu().numpy()


def categorical_rollout(model, token_count, init=None, carry_memory=False, graph=None):
    """ppo_policy/inference.py:78-160 (`testing()`): start from the all-zero token, per step run the recurrent-form
    actor on the PREVIOUS token only -- the reference passes `memory=None` on every call (:106), so no state is
    carried; `carry_memory=True` is the evident intent -- and draw each attribute from Categorical(softmax(logits))
    (:121-133).  Everything stays on the device (`_DeviceLoop`); ONE host sync at the end.  Same distribution as the
    reference's torch.distributions draws, not the same random stream.  -> (token_count, 6) int64 numpy."""
    sess = DecodeSession(model, graph=False)
    if sess.model.training:
        raise RuntimeError("generation runs in eval() mode (ppo_policy/inference.py:96)")
    A = len(sess.n_token)
    sess.tok.copy_(torch.as_tensor(np.zeros(A) if init is None else np.asarray(init), dtype=torch.int64)
                   .view(1, 1, A).to(sess.dev))
    loop = _DeviceLoop(sess, token_count, carry_memory=carry_memory, graph=graph)
    loop.run(token_count)
    return loop.tokens(0, token_count)


# per-attribute sampler settings of forward_output_sampling (dqn_policy/model.py:281-286), attribute order
DQN_TEMPERATURE = (1.2, 1.0, 1.2, 1.0, 2.0, 5.0)
DQN_TOP_P = (0.9, 0.99, None, 0.9, 0.9, None)


def inference_from_scratch(model, word2event, bar_cond, max_tokens=None, log=None, session=None,
                           device_sampling=False, chunk=128):
    """testing-no-type-cp.py:126-179: start from the Bar token, sample until `bar_cond` bars have begun.
    `max_tokens` (not in the reference, whose loop is unbounded) caps the song length.

    device_sampling=False: the reference's numpy samplers on the host (a seeded np.random reproduces its stream).
    device_sampling=True: the same per-attribute temperature / nucleus settings drawn on the device
    (`cwlt_sample_categorical`); the host only looks at the song every `chunk` tokens to count bars, and cuts it
    where the reference's loop would have stopped.  Same distribution, different random stream, ~1.2x faster (no host round trip per token)."""
    classes = list(word2event.keys())
    sess = session or DecodeSession(model)
    sess.reset()

    def show(cp, prefix=""):
        if log is not None:
            log(prefix + " | ".join("{:15s}".format(str(word2event[k][int(cp[i])])) for i, k in enumerate(classes)))

    final_res = []
    cnt_bar = 1
    if device_sampling:
        if len(INIT_CW) != 1:
            raise RuntimeError("device-side sampling starts from a single initial token")
        cap = max_tokens - 1 if max_tokens is not None else 16384
        show(INIT_CW[0])
        final_res.append(INIT_CW[0][None, ...])
        sess.tok.copy_(torch.as_tensor(INIT_CW[0], dtype=torch.int64).view(1, 1, -1).to(sess.dev))
        loop = _DeviceLoop(sess, cap, temperature=DQN_TEMPERATURE, top_p=DQN_TOP_P, carry_memory=True,
                           graph=sess.use_graph)
        done = 0
        while done < cap:
            n = loop.run(chunk)
            for next_arr in loop.tokens(done, done + n):
                final_res.append(next_arr[None, ...])
                show(next_arr, "bar: %d  ==" % cnt_bar)
                if word2event["bar-beat"][int(next_arr[2])] == "Bar":
                    cnt_bar += 1
                if cnt_bar == bar_cond:
                    return np.concatenate(final_res)
            done += n
        return np.concatenate(final_res)
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
             max_tokens=None, stats_path="runtime_stats.json", log=print, device_sampling=False):
    """testing-no-type-cp.py:182-223 / agent_pretrain.py:663-706: generate `n_songs`, time them, write
    runtime_stats.json with the reference's keys.  `write_midi(res, path, word2event)` is the caller's MIDI writer
    (miditoolkit-based in the reference; out of scope here) -- when None the token array is saved as .npy."""
    os.makedirs(path_gendir, exist_ok=True)
    sess = DecodeSession(model)
    song_time_list, words_len_list = [], []
    for sidx in range(n_songs):
        start = time.time()
        res = inference_from_scratch(model, word2event, bar_cond, max_tokens=max_tokens, session=sess,
                                     device_sampling=device_sampling)
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
