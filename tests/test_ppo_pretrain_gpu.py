"""GPU: section 8f leftovers on the PPO side --
  * the reward model (ppo_policy/model.py::LongFormer.token_forward) is differentiable: every parameter gradient
    against the fixture recorded from the reference's own class + HF Longformer backward (ppo_reward_grads_small.npz);
  * the drop-in ppo_policy/my_pretrain.py (`pretrain()` / `main()`, `--reward_pretrain`) trains both models;
  * ppo_train.main() runs from the dataset FILES written by the reference's own writers (tests/golden/ppo_dataset)."""
import os
import shutil
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
from fill import fill_params  # noqa: E402

import rlmg_amd  # noqa: E402,F401

pytestmark = pytest.mark.gpu
SMALL = {"D_MODEL": 128, "N_LAYER": 2, "N_HEAD": 2}


@pytest.fixture()
def small_ppo(monkeypatch, tmp_path):
    from rlmg_amd.ppo_policy import config
    monkeypatch.chdir(tmp_path)
    for c in (config.ActorConfig, config.CriticConfig, config.DiscriConfig):
        for k, v in SMALL.items():
            monkeypatch.setitem(c, k, v)
    return config


def test_reward_model_gradients_match_reference_fixture(cuda, small_ppo):
    from rlmg_amd.ppo_policy import model as pmodel
    fx = np.load(os.path.join(HERE, "golden", "ppo_reward_grads_small.npz"), allow_pickle=False)
    net = fill_params(pmodel.LongFormer(fx["n_token"].tolist()), seed=33).to(cuda).eval()
    net.compute_dtype = torch.float32
    x, mask, w = (torch.from_numpy(fx[k]).to(cuda) for k in ("x", "mask", "w"))
    score = net.token_forward(x, None, mask)
    assert score.requires_grad
    assert (score.detach().cpu() - torch.from_numpy(fx["score"])).abs().max().item() < 1e-4
    (score * w).sum().backward()
    params = dict(net.named_parameters())
    for k, want_norm in zip(fx["names"].tolist(), fx["norms"]):
        g = params[k].grad
        if g is None:
            assert want_norm == 0.0, k              # word_embeddings: only HF's window padding touches it
            continue
        assert abs(g.double().norm().item() - want_norm) < 1e-5 + 1e-3 * want_norm, k
        want = torch.from_numpy(fx["grad." + k])
        got = (g[:8] if g.numel() > 4096 else g).cpu()
        assert (got - want).abs().max().item() < 1e-5 + 1e-4 * want.abs().max().item(), k
    with torch.no_grad():                           # the frozen use inside ppo_train.py gives the same score
        assert (net.token_forward(x, None, mask) - score.detach()).abs().max().item() < 1e-5


@pytest.mark.parametrize("reward", [False, True])
def test_my_pretrain_main_trains(cuda, small_ppo, monkeypatch, reward, capsys):
    """`python my_pretrain.py [--reward_pretrain]` on synthetic data of the dataset's schema: batches of 4 (the
    reference's 12 needs a real dataset), 3 epochs; the loss falls, the checkpoint and config log are written, the
    LR schedule is stepped per batch."""
    from rlmg_amd.ppo_policy import my_pretrain as M
    monkeypatch.setattr(M, "BATCH_SIZE", 4)
    monkeypatch.setattr(M, "NUM_EPOCH", 3)
    monkeypatch.setattr(M, "Init_lr", 1e-3)          # 1e-2 with Adam diverges on 8 random sequences
    from rlmg_amd import data as cwdata
    real = cwdata.load_ppo
    monkeypatch.setattr(cwdata, "load_ppo", lambda a, b, **kw: real(a, b, n_seq=8, T=64))
    torch.manual_seed(0)
    rec = M.main(["--reward_pretrain"] if reward else [])
    assert len(rec) == 3 and all(np.isfinite(rec)) and rec[-1] < rec[0]
    exp = os.path.join("Exp-Pretrain", os.listdir("Exp-Pretrain")[0])
    sd = torch.load(os.path.join(exp, "model", "pretrain_best.pth"))
    assert any(k.startswith("longformer." if reward else "transformer_encoder.") for k in sd)
    log = open(os.path.join(exp, "log", "config_log.txt")).read().splitlines()
    assert log[0] == "Model Type      = " + ("Pretrain with Longformer for reward model" if reward
                                             else "Pretrain with Linearformer for agent.")
    assert log[2:] == ["Num epochs      = 3", "Batch size      = 4", "Learning rate   = 0.001"]
    out = capsys.readouterr().out
    assert ("Reward Model Pretraining..." if reward else "Agent Pretraining...") in out
    assert "Num of token class >> [50, 20, 20, 90, 68, 26]" in out           # len + 1 pad word (my_pretrain.py:176-178)


def test_ppo_train_main_runs_from_the_references_dataset_files(cuda, small_ppo, monkeypatch):
    from rlmg_amd.ppo_policy import ppo_train as P
    os.makedirs("dataset")
    for name in ("dictionary.pickle", "our_dataset.pickle"):
        shutil.copy(os.path.join(HERE, "golden", "ppo_dataset", name), os.path.join("dataset", name))
    monkeypatch.setattr(P, "NUM_SONGS", 1)          # train_y holds ONE song (preprocess.py's split of 4 sequences)
    monkeypatch.setattr(P, "PPO_STEPS", 1)
    monkeypatch.setenv("CWLT_NO_PRETRAIN", "1")
    P.main()
    assert os.path.exists("ckpt/ppo_best.pt") and os.path.exists("ckpt/policy_loss.pickle")
