"""GPU: token-by-token generation (generation.DecodeSession / inference_from_scratch / generate) against the
fixture recorded from the reference's own recurrent-form model and samplers."""
import json
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
from fill import fill_params  # noqa: E402

import rlmg_amd  # noqa: E402,F401
from rlmg_amd import generation  # noqa: E402
from rlmg_amd.sampling import sample_cw  # noqa: E402

pytestmark = pytest.mark.gpu
FIX = np.load(os.path.join(HERE, "golden", "dqn_generation_small.npz"))
N_CLASS = [int(v) for v in FIX["n_class"]]


def _small_model(cuda):
    from rlmg_amd.dqn_policy import config, model
    old = dict(config.AgentConfig)
    config.AgentConfig.update({"D_MODEL": 128, "N_LAYER": 2, "N_HEAD": 2})
    try:
        net = model.LinearTransformer(N_CLASS, is_training=False)
    finally:
        config.AgentConfig.update(old)
    return fill_params(net, seed=int(FIX["fill_seed"])).to(cuda).eval()


def _word2event():
    keys = ["tempo", "chord", "bar-beat", "pitch", "duration", "velocity"]
    w2e = {k: {i: "%s_%d" % (k, i) for i in range(n)} for k, n in zip(keys, N_CLASS)}
    w2e["bar-beat"][1] = "Bar"
    w2e["bar-beat"][9] = "Bar"
    return w2e


def test_decode_gemv_building_block(cuda):
    """csrc/decode.hip GEMV with LayerNorm prologue(s) and bias / GELU / residual epilogue vs torch f32."""
    import torch.nn.functional as F
    from rlmg_amd import ops
    g = torch.Generator().manual_seed(5)
    for n, K, n_out in [(1, 512, 1536), (1, 512, 2048), (1, 2048, 512), (1, 1216, 512), (1, 512, 339), (3, 128, 384),
                        (2, 128, 339), (1, 512, 1)]:
        x = torch.randn(n, K, generator=g).to(cuda)
        w = (torch.randn(n_out, K, generator=g) / K ** 0.5).to(cuda)
        b = torch.randn(n_out, generator=g).to(cuda)
        res = torch.randn(n, n_out, generator=g).to(cuda)
        ln = (1 + 0.1 * torch.randn(K, generator=g)).to(cuda), (0.1 * torch.randn(K, generator=g)).to(cuda)
        ln2 = (1 + 0.1 * torch.randn(K, generator=g)).to(cuda), (0.1 * torch.randn(K, generator=g)).to(cuda)
        assert (ops.decode_gemv(w, b, x) - F.linear(x, w, b)).abs().max().item() < 2e-5
        assert (ops.decode_gemv(w, None, x, res=res) - (F.linear(x, w) + res)).abs().max().item() < 2e-5
        x1 = F.layer_norm(x, (K,), ln[0], ln[1], 1e-5)
        got, xn = ops.decode_gemv(w, b, x, ln=ln, act="gelu", want_normed=True)
        assert (xn - x1).abs().max().item() < 2e-5
        assert (got - F.gelu(F.linear(x1, w, b))).abs().max().item() < 2e-5
        x2 = F.layer_norm(x1, (K,), ln2[0], ln2[1], 1e-5)
        got, xn = ops.decode_gemv(w, b, x, ln=ln, ln2=ln2, res=res, want_normed=True)
        assert (xn - x2).abs().max().item() < 2e-5
        assert (got - (F.linear(x2, w, b) + res)).abs().max().item() < 3e-5


@pytest.mark.parametrize("graph,fused", [(False, False), (True, False), (False, True), (True, True)])
def test_decode_session_matches_reference_stream(cuda, graph, fused):
    net = _small_model(cuda)
    sess = generation.DecodeSession(net, graph=graph, fused=fused)
    # teacher forced: the recorded tokens in, the recorded logits out
    for t in range(len(FIX["logits"])):
        got = sess.step(FIX["tokens"][t])
        assert np.abs(got - FIX["logits"][t]).max() < 1e-4, t
    if graph:       # captured through ops.capture_hip_graph: counted, and free of memset nodes when it is replayed
        assert sess.use_graph and sess.census.get("kernel", 0) > 0, sess.census
        if fused:
            assert sess.census.get("memset", 0) == 0, sess.census      # cwlt_decode_step issues kernels only
    # free running under the recorded np seed: the same token stream
    sess.reset()
    np.random.seed(int(FIX["np_seed"]))
    tok = FIX["tokens"][0]
    for t in range(len(FIX["logits"])):
        tok = sample_cw(sess.split(sess.step(tok)))
        assert tok.tolist() == FIX["tokens"][t + 1].tolist(), t


def test_inference_from_scratch_and_generate(cuda, tmp_path):
    net = _small_model(cuda)
    w2e = _word2event()
    np.random.seed(int(FIX["np_seed"]))
    res = generation.inference_from_scratch(net, w2e, bar_cond=3)
    # same seed, same model => the fixture's stream, cut where the third bar begins
    bars = np.cumsum([w2e["bar-beat"][int(r[2])] == "Bar" for r in FIX["tokens"]])
    stop = int(np.argmax(bars == 3)) + 1
    assert res.shape == (stop, 6) and res.tolist() == FIX["tokens"][:stop].tolist()
    # generate(): songs + the reference's runtime_stats.json keys (testing-no-type-cp.py:211-219)
    stats = generation.generate(net, w2e, n_songs=2, bar_cond=3, path_gendir=str(tmp_path / "gen"),
                                stats_path=str(tmp_path / "runtime_stats.json"), log=lambda *a: None)
    saved = json.load(open(tmp_path / "runtime_stats.json"))
    assert set(saved) == {"song_time", "words_len_list", "ave token time:", "ave song time"}
    assert len(stats["song_time"]) == 2 and os.path.exists(tmp_path / "gen" / "get_1.npy")
    # max_tokens caps an otherwise unbounded song
    res = generation.inference_from_scratch(net, w2e, bar_cond=10 ** 6, max_tokens=20)
    assert len(res) == 20
    # device-side sampling: same loop semantics (starts with the Bar token, stops WITH the token that opens bar 4),
    # reproducible under torch.manual_seed, graph replay == eager launches
    songs = []
    for graph in (True, False, True):
        torch.manual_seed(11)
        sess = generation.DecodeSession(net, graph=graph)
        songs.append(generation.inference_from_scratch(net, w2e, bar_cond=4, session=sess, device_sampling=True,
                                                       chunk=16))
    assert songs[0].tolist() == songs[1].tolist() == songs[2].tolist()
    song = songs[0]
    assert song[0].tolist() == generation.INIT_CW[0].tolist()
    bars = [w2e["bar-beat"][int(r[2])] == "Bar" for r in song]
    assert sum(bars) == 4 and bars[-1] and all((song[:, i] < n).all() for i, n in enumerate(N_CLASS))
    res = generation.inference_from_scratch(net, w2e, bar_cond=10 ** 6, max_tokens=40, device_sampling=True, chunk=16)
    assert len(res) == 40


def test_fused_decode_batch_of_songs_and_weight_reload(cuda):
    """n_songs > 1: each song's state is its own (song i == a single-song session fed song i's tokens); the packed
    weights follow the model's parameters across reset()."""
    net = _small_model(cuda)
    g = torch.Generator().manual_seed(9)
    toks = torch.stack([torch.randint(0, n, (3, 12), generator=g) for n in N_CLASS], -1).numpy()    # (3, 12, 6)
    batch = generation.DecodeSession(net, n_songs=3)
    got = np.stack([batch.step(toks[:, t]).copy() for t in range(12)], 1)                            # (3, 12, W)
    assert batch.hidden.shape == (3, 128)
    single = generation.DecodeSession(net)
    for i in range(3):
        single.reset()
        for t in range(12):
            assert np.abs(single.step(toks[i, t]) - got[i, t]).max() < 1e-5, (i, t)
    # the hidden row is what the module path's forward_hidden returns
    ref = generation.DecodeSession(net, fused=False, graph=False)
    single.reset()
    for t in range(12):
        a, b = single.step(toks[0, t]).copy(), ref.step(toks[0, t]).copy()
        assert np.abs(a - b).max() < 1e-4
        assert (single.hidden.view(-1) - ref.hidden.view(-1)).abs().max().item() < 1e-4
    # new weights are picked up at the next reset()
    with torch.no_grad():
        net.proj_pitch.bias.add_(1.0)
    single.reset()
    o = sum(N_CLASS[:3])
    after = single.step(toks[0, 0]).copy()
    assert np.abs((after - got[0, 0])[o:o + N_CLASS[3]] - 1.0).max() < 1e-5
    assert np.abs(np.delete(after - got[0, 0], np.s_[o:o + N_CLASS[3]])).max() < 1e-5


def test_generation_refuses_training_form_and_train_mode(cuda):
    from rlmg_amd.dqn_policy import config, model
    old = dict(config.AgentConfig)
    config.AgentConfig.update({"D_MODEL": 128, "N_LAYER": 2, "N_HEAD": 2})
    try:
        par = model.LinearTransformer(N_CLASS, is_training=True).to(cuda)
    finally:
        config.AgentConfig.update(old)
    with pytest.raises(RuntimeError):
        generation.DecodeSession(par)
    net = _small_model(cuda).train()
    with pytest.raises(RuntimeError):
        generation.DecodeSession(net).step(FIX["tokens"][0])


def test_agent_pretrain_generate_writes_midis(cuda, tmp_path, monkeypatch):
    """agent_pretrain.generate() (MODE='inference'): recurrent-form net -> sampled songs -> get_<i>.mid that parse
    back, + runtime_stats.json."""
    from rlmg_amd import midi
    from rlmg_amd.dqn_policy import agent_pretrain, config
    monkeypatch.chdir(tmp_path)
    old = dict(config.AgentConfig)
    config.AgentConfig.update({"D_MODEL": 128, "N_LAYER": 2, "N_HEAD": 2})
    try:
        torch.manual_seed(0)
        np.random.seed(0)
        stats = agent_pretrain.generate(n_songs=2, bar_cond=4, max_tokens=300, log=lambda *a: None)
    finally:
        config.AgentConfig.update(old)
    assert len(stats["words_len_list"]) == 2 and all(2 <= n <= 300 for n in stats["words_len_list"])
    for i in range(2):
        song = midi.read_smf(str(tmp_path / "gen_midis" / ("get_%d.mid" % i)))
        assert song["ticks_per_beat"] == 480
    assert os.path.exists(tmp_path / "runtime_stats.json")


def test_device_categorical_sampler_distribution_and_determinism(cuda):
    """csrc/sample.hip: empirical frequencies of 40 000 draws per attribute match softmax(logits / t) within 5 sigma;
    the draw is a pure function of (seed, counter, row, attribute); ids land in the token buffer and the song."""
    from rlmg_amd import ops
    n_class = [49, 19, 19, 89, 135, 25]
    g = torch.Generator().manual_seed(2)
    logits = (2.0 * torch.randn(1, sum(n_class), generator=g)).to(cuda)
    temp = [1.2, 1.0, 1.0, 0.8, 2.0, 5.0]
    T = 40000
    rows = logits.expand(T, -1).contiguous()               # T independent rows = T draws in one launch
    toks = torch.zeros((T, 6), dtype=torch.int64, device=cuda)
    ops.sample_categorical(rows, n_class, toks, seed=1234, temperature=temp)
    again = torch.zeros_like(toks)
    ops.sample_categorical(rows, n_class, again, seed=1234, temperature=temp)
    assert (toks == again).all()
    other = torch.zeros_like(toks)
    ops.sample_categorical(rows, n_class, other, seed=1235, temperature=temp)
    assert (toks != other).any()
    o = 0
    for a, n in enumerate(n_class):
        p = torch.softmax(logits[0, o:o + n].double() / temp[a], -1).cpu().numpy()
        ids = toks[:, a].cpu().numpy()
        assert ids.min() >= 0 and ids.max() < n
        freq = np.bincount(ids, minlength=n) / T
        sigma = np.sqrt(p * (1 - p) / T)
        assert (np.abs(freq - p) < 5 * sigma + 1e-4).all(), a
        o += n
    # nucleus: classes outside the reference's nucleus (dqn_policy/model.py:33-47) are never drawn, the rest follow
    # the renormalised probabilities
    top_p = [0.9, 0.99, None, 0.5, 0.9, None]
    ops.sample_categorical(rows, n_class, toks, seed=99, temperature=temp, top_p=top_p)
    o = 0
    for a, n in enumerate(n_class):
        z = logits[0, o:o + n].cpu().numpy() / temp[a]
        p = np.exp(z) / np.sum(np.exp(z))
        if top_p[a] is not None:
            q = p / (sum(p) + 1e-5)
            order = np.argsort(q)[::-1]
            after = np.cumsum(np.sort(q)[::-1]) > top_p[a]
            keep = order[:np.where(after)[0][0] + 1] if after.sum() > 0 else order
            mask = np.zeros(n, dtype=bool)
            mask[keep] = True
            p = np.where(mask, p, 0.0)
            p = p / p.sum()
        ids = toks[:, a].cpu().numpy()
        freq = np.bincount(ids, minlength=n) / T
        assert (freq[p == 0] == 0).all(), a
        sigma = np.sqrt(p * (1 - p) / T)
        assert (np.abs(freq - p) < 5 * sigma + 1e-4).all(), a
        o += n
    # counter keys the draw and indexes the song
    count = torch.zeros(1, dtype=torch.int64, device=cuda)
    song = torch.full((3, 2, 6), -1, dtype=torch.int64, device=cuda)
    tok = torch.zeros((2, 6), dtype=torch.int64, device=cuda)
    two = logits.expand(2, -1).contiguous()
    seen = []
    for t in range(3):
        ops.sample_categorical(two, n_class, tok, seed=7, counter=count, song=song)
        seen.append(tok.clone())
        count.add_(1)
    assert (song == torch.stack(seen)).all() and not (seen[0] == seen[1]).all()


def test_ppo_categorical_rollout(cuda, tmp_path, monkeypatch):
    """ppo_policy/inference.py::testing: device-side Categorical sampling; with memory=None per call (the
    reference's quirk) token t+1 depends on token t only, through the fresh-state logits."""
    from rlmg_amd.ppo_policy import config, inference, model
    monkeypatch.chdir(tmp_path)
    old = dict(config.ActorConfig)
    config.ActorConfig.update({"D_MODEL": 128, "N_LAYER": 2, "N_HEAD": 2})
    try:
        n_token = [49, 19, 19, 89, 67, 25]
        net = fill_params(model.Actor_Transformer(n_token, is_training=False), seed=3).to(cuda).eval()
        torch.manual_seed(7)
        a = generation.categorical_rollout(net, 40, graph=True)
        torch.manual_seed(7)
        b = generation.categorical_rollout(net, 40, graph=False)
        assert a.shape == (40, 6) and all((a[:, i] < n).all() and (a[:, i] >= 0).all() for i, n in enumerate(n_token))
        assert (a == b).all()                              # same seed => same song, graph replay or eager launches
        # every drawn id has non-negligible probability under the fresh-state logits of the previous token
        sess = generation.DecodeSession(net, graph=False)
        prev = np.zeros(6, dtype=np.int64)
        for t in range(40):
            sess.reset()
            ys = sess.split(sess.step(prev).copy())
            for i, y in enumerate(ys):
                p = np.exp(y - y.max())
                assert p[b[t, i]] / p.sum() > 1e-6
            prev = b[t]
        # carry_memory=True carries the state (differs from the quirk path after the first tokens)
        torch.manual_seed(7)
        c = generation.categorical_rollout(net, 40, carry_memory=True, graph=False)
        assert (c[0] == b[0]).all() and c.shape == (40, 6)
        song = inference.testing(token_count=12, log=lambda *a: None)
        assert song.shape == (12, 6) and os.path.exists(tmp_path / "gen_midi" / "ppo_song.npy")
    finally:
        config.ActorConfig.update(old)
