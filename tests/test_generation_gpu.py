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


@pytest.mark.parametrize("graph", [False, True])
def test_decode_session_matches_reference_stream(cuda, graph):
    net = _small_model(cuda)
    sess = generation.DecodeSession(net, graph=graph)
    # teacher forced: the recorded tokens in, the recorded logits out
    for t in range(len(FIX["logits"])):
        got = sess.step(FIX["tokens"][t])
        assert np.abs(got - FIX["logits"][t]).max() < 1e-4, t
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
