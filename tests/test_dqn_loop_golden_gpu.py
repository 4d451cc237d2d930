"""GPU: the product's DQN + AIRL rollout loop (rlmg_amd/dqn_policy/IRL_dqn_train.py::main with its DQN, buffers and
RewardDiscri) against tests/golden/dqn_loop_small.npz -- the REFERENCE's own `__main__` block
(/root/reference/dqn_policy/IRL_dqn_train.py:386-497) run for two songs with BUFFER_SIZE 60
(tests/golden/make_golden.py::dqn_loop_small).  Greedy actions bit-exact while no weight has moved (61 steps), the
first re-scoring of both buffers and the first update within 1e-4, the update batches themselves (two `sampling` calls
under one np.random stream; CE target = the agent's own next state) exact, and the expert side of the final buffers
exact."""
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
from fill import fill_params  # noqa: E402

import rlmg_amd  # noqa: E402,F401

pytestmark = pytest.mark.gpu
FX = np.load(os.path.join(HERE, "golden", "dqn_loop_small.npz"), allow_pickle=False)
N_CLASS = FX["n_class"].tolist()
DISK = [56, 135, 18, 3, 87, 18, 25]
KEYS = ["tempo", "chord", "bar-beat", "type", "pitch", "duration", "velocity"]


def test_main_loop_reproduces_the_reference_run(cuda, tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    monkeypatch.delenv("CWLT_NO_PRETRAIN", raising=False)
    from rlmg_amd.dqn_policy import AIRL, AIRL_model, IRL_dqn_train as T, config, model
    old_cfg = dict(config.AgentConfig)
    old_am = (AIRL_model.D_MODEL, AIRL_model.N_LAYER, AIRL_model.N_HEAD)
    config.AgentConfig.update({"D_MODEL": 128, "N_LAYER": 2, "N_HEAD": 2})
    AIRL_model.D_MODEL, AIRL_model.N_LAYER, AIRL_model.N_HEAD = 128, 2, 2
    try:
        # the two checkpoints the loop reads: the pretrained agent (Pretrain=True path) and ./ckpt/disc_IRL.pt
        pre = fill_params(model.LinearTransformer(N_CLASS), seed=61)
        torch.save({"epoch": 0, "model_state_dict": pre.state_dict()}, str(tmp_path / "pretrain.pt"))
        disc = fill_params(AIRL_model.LongFormer(N_CLASS), seed=41)
        with torch.no_grad():
            disc.score_classifier[1].running_mean.copy_(torch.linspace(-0.2, 0.2, 128))
            disc.score_classifier[1].running_var.copy_(torch.linspace(0.5, 1.5, 128))
        os.makedirs("ckpt")
        torch.save({"epoch": 0, "model_state_dict": disc.state_dict()}, "./ckpt/disc_IRL.pt")
        monkeypatch.setattr(T, "Pretrain_ckpt", str(tmp_path / "pretrain.pt"))
        monkeypatch.setattr(T, "NUM_SONGS", 2)
        monkeypatch.setattr(T, "BUFFER_SIZE", int(FX["buffer_size"]))
        e2w = {k: {"%s_%d" % (k, i): i for i in range(n)} for k, n in zip(KEYS, DISK)}
        data = {"x": FX["x"].astype(np.int64), "y": FX["y"].astype(np.int64), "mask": FX["mask"]}
        monkeypatch.setattr(T.cwdata, "load_dqn", lambda *a, **k: ((e2w, None), data))
        rec = {"actions": [], "upd": [], "losses": [], "rewards": [], "buffers": []}

        class Agent(T.DQN):
            def __init__(self, n_class, Pretrain=True):
                assert Pretrain is True                       # the checkpoint exists: main() takes the reference's path
                super().__init__(n_class, Pretrain)
                self.eval_net.eval()
                self.target_net.eval()

            def choose_action(self, x, target=None):
                a = super().choose_action(x, target)
                rec["actions"].append(a.cpu().numpy().copy())
                return a

            def update(self, agent_transition, expert_transition, mask_next_states, update_flag, epoch):
                rec["upd"].append({k: v.detach().cpu().clone() for k, v in agent_transition.items()} |
                                  {"e_" + k: v.detach().cpu().clone() for k, v in expert_transition.items()} |
                                  {"mask": mask_next_states.detach().cpu().clone(), "flag": update_flag, "epoch": epoch,
                                   "lr": float(self.optim.param_groups[0]["lr"])})
                out = super().update(agent_transition, expert_transition, mask_next_states, update_flag, epoch)
                rec["losses"].append(out)
                return out

        class Rewarder(AIRL.RewardDiscri):
            def __init__(self, n_class, Pretrain=False):
                super().__init__(n_class, Pretrain)
                for mod in self.disc_model.modules():         # the record was taken with dropout off
                    if isinstance(mod, torch.nn.Dropout):
                        mod.p = 0.0
                    if hasattr(mod, "p_hidden"):
                        mod.p_hidden = mod.p_attn = 0.0
                self.batch_size = int(FX["score_batch"])

            def update_disc(self, agent_traj, expert_traj, train=True):
                r = super().update_disc(agent_traj, expert_traj, train=train)
                rec["rewards"].append((r[0].cpu().numpy().copy(), r[1].cpu().numpy().copy(), train))
                return r

        made = []
        real_agent, real_expert = T.AgentMemory, T.ExpertMemory

        class AB(real_agent):
            def __init__(self):
                super().__init__()
                made.append(self)

        class EB(real_expert):
            def __init__(self):
                super().__init__()
                made.append(self)

        monkeypatch.setattr(T, "DQN", Agent)
        monkeypatch.setattr(T, "RewardDiscri", Rewarder)
        monkeypatch.setattr(T, "AgentMemory", AB)
        monkeypatch.setattr(T, "ExpertMemory", EB)
        np.random.seed(int(FX["np_seed"]))
        T.main()
    finally:
        config.AgentConfig.update(old_cfg)
        AIRL_model.D_MODEL, AIRL_model.N_LAYER, AIRL_model.N_HEAD = old_am

    BUF = int(FX["buffer_size"])
    got = np.stack(rec["actions"]).astype(np.int64)
    want = FX["actions"].astype(np.int64)
    assert got.shape == want.shape == (100, 25, 6)
    assert np.array_equal(got[:BUF + 1], want[:BUF + 1])                      # no weight has moved yet: bit-exact
    assert (got == want).mean() > 0.99, (got != want).sum()                   # 40 Adam steps at lr 1e-2 later
    assert len(rec["upd"]) == 40 and all(r[2] is False for r in rec["rewards"])
    # first re-scoring of both buffers (3 batches of 16 scored, the 12-window tail keeps 1.0) and first update
    assert np.abs(rec["rewards"][0][0][:, 0] - FX["traj_reward"][0]).max() <= 1e-4
    assert np.abs(rec["rewards"][0][1][:, 0] - FX["answer_reward"][0]).max() <= 1e-4
    assert (rec["rewards"][0][0][48:, 0] == 1.0).all()
    u0 = rec["upd"][0]
    for k in ("state", "action", "nextstate", "e_nextstate"):
        assert np.array_equal(u0[k].numpy(), FX["update3." + k][0].astype(np.int64)), k
    assert np.array_equal(u0["mask"].numpy(), FX["update3.mask"][0])
    assert torch.equal(u0["e_nextstate"], u0["nextstate"])                    # CE target: the agent's own next state
    assert np.allclose(u0["reward"].numpy(), FX["update.reward"][0], atol=1e-4)
    assert np.array_equal(u0["e_done"].numpy(), FX["update.e_done"][0]) and u0["flag"] is True
    assert np.allclose(np.array(rec["losses"][0]), FX["update.losses"][0], rtol=1e-4, atol=1e-4)
    assert np.allclose(np.array([u["lr"] for u in rec["upd"]]), FX["update.lr_before"], rtol=1e-9)
    assert [u["epoch"] for u in rec["upd"]] == FX["update.epoch"].tolist()
    # while the trajectories coincide, so do the batches and the losses (loosening with every optimizer step)
    n_same = 0
    while n_same < 40 and np.array_equal(got[:BUF + 1 + n_same], want[:BUF + 1 + n_same]):
        n_same += 1
    for i in range(min(n_same, 40)):
        for k in ("state", "action", "nextstate"):
            assert abs(float(rec["upd"][i][k].double().sum()) - FX["update.sum." + k][i]) < 1e-6, (i, k)
        assert np.allclose(np.array(rec["losses"][i]), FX["update.losses"][i], rtol=5e-3, atol=5e-3), i
    assert n_same >= 3
    ab, eb = made[0], made[1]
    assert [ab.memory_counter, eb.memory_counter] == FX["final.counters"].tolist()
    assert np.array_equal(eb.states_exp.cpu().numpy().astype(np.int64), FX["final.expert_states"].astype(np.int64))
    assert np.array_equal(eb.next_states_exp.cpu().numpy().astype(np.int64), FX["final.expert_next"].astype(np.int64))
    assert np.array_equal(eb.mask_state.cpu().numpy(), FX["final.mask_state"])
    assert np.array_equal(eb.mask_next_state.cpu().numpy(), FX["final.mask_next_state"])
    assert (ab.rewards_agent.cpu().numpy()[48:, 0] == 1.0).all()
