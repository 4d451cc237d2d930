"""Drop-in for /root/reference/dqn_policy/AIRL.py: `RewardDiscri`, the AIRL reward module around the
Longformer discriminator, on the libcwlt kernels.  Scoring (`all_forward`, `calculate_reward`,
`update_disc(train=False)`) is the mode the RL loop uses (IRL_dqn_train.py:477) and runs without autograd; the
training branch (`train=True`, AIRL.py:135-212) runs the same kernels as autograd Functions (band-attention,
LN, GELU, embedding and CE backward kernels).  wandb / tqdm / loss plots are not reproduced.
"""
import os
import pickle
import sys

import torch
import torch.nn as nn
import torch.optim as optim

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)
import rlmg_amd  # noqa: E402,F401
from rlmg_amd import dist as rdist  # noqa: E402

try:
    from AIRL_model import LongFormer
except ImportError:
    from .AIRL_model import LongFormer

MAX_SEQ_LEN = 1024
D_MODEL = 512
N_LAYER = 10
N_HEAD = 8
path_exp = "exp"
N_STATES = 50
Pretrain_ckpt = "/data/Der_CODES/DQN-cp/ckpt/trainloss_22.pt"


class RewardDiscri(nn.Module):
    def __init__(self, n_token, Pretrain=True):
        super().__init__()
        self.disc_model = LongFormer(n_token).cuda()
        if Pretrain:
            checkpoint = torch.load(Pretrain_ckpt)
            self.disc_model.load_state_dict(checkpoint["model_state_dict"])
        self.BCE_criterion = nn.BCELoss()
        self.CrossEntropy = nn.CrossEntropyLoss()
        self.avg_score = nn.Sequential(nn.Linear(6, 1).cuda(), nn.Sigmoid())     # declared, unused (AIRL.py:45-49)
        self.last_unit = nn.Linear(6, 1).cuda()
        self.init_lr = 0.001
        self.epoch_disc = 5
        self.batch_size = 100
        # data parallel: flat gradient buckets, reduced once per batch in finish() (the batch runs the discriminator
        # three times -- expert score, token CE, agent score -- before its one backward: overlap=False)
        self.sync_disc = rdist.GradSync(self.disc_model.parameters(), overlap=False)
        self.optim_disc = optim.Adam(self.disc_model.parameters(), lr=self.init_lr)
        self.sched_disc = torch.optim.lr_scheduler.StepLR(self.optim_disc, step_size=10, gamma=0.1)
        self.reward_path = "./exp/IRL_reward.pickle"
        self.IRL_ckpt_path = "./ckpt/disc_IRL.pt"
        self._ckpt_mtime = None

    def all_forward(self, states_batch, dones_batch, next_states_batch, mask_states_batch, mask_next_states_batch):
        """AIRL.py:61-65 -- forces train() (dropout + BatchNorm batch statistics live even while scoring)."""
        self.disc_model.train()
        return self.disc_model(states_batch, mask_states_batch)

    def _maybe_reload(self):
        """The reference re-reads ./ckpt/disc_IRL.pt from disk on EVERY call (AIRL.py:73); here the file is
        re-read only when it changed.  A missing file keeps the current weights where the reference's torch.load
        would raise FileNotFoundError (listed under "differences" in INTEGRATION.md): IRL_dqn_train.py only ever
        calls update_disc(train=False), which never writes that file, so the reference's loop cannot get past its
        first scoring call without a checkpoint prepared by hand."""
        p = self.IRL_ckpt_path
        if os.path.exists(p):
            m = os.path.getmtime(p)
            if m != self._ckpt_mtime:
                self.disc_model.load_state_dict(torch.load(p)["model_state_dict"])
                self._ckpt_mtime = m

    def calculate_reward(self, states, dones, next_states, mask_states, mask_next_states):
        """AIRL.py:69-91: score the whole buffer in batches of 100 (`all_forward`, which forces train(): dropout
        live, BatchNorm on batch statistics); a tail shorter than 100 keeps 1.0.
        The RL loop calls this on 2 x BUFFER_SIZE windows per environment step, so the 100-window batches are not
        launched one by one (each is ~150 launches on 5 000 tokens): the Longformer body -- per-window
        arithmetic, nothing crosses windows -- runs over thousands of windows per pass, and only the score
        classifier, whose BatchNorm couples the windows of a batch, is applied per consecutive group of 100
        (`LongFormer.score_in_groups`).  Scores are copied to the host once."""
        n = states.shape[0]
        bs = self.batch_size
        full = (n // bs) * bs
        self._maybe_reload()
        pred_dev = torch.ones((n, 1), device="cuda")
        if full:
            self.disc_model.train()                      # what all_forward does on every batch (AIRL.py:62)
            with torch.no_grad():
                pred_dev[:full] = self.disc_model.score_in_groups(states[:full].long().cuda(),
                                                                  mask_states[:full].long().cuda(), bs).float()
        return pred_dev.cpu()

    def update_disc(self, agent_episode, expert_episode, train=True):
        """AIRL.py:121-236: optional discriminator training, then the rewards of the agent and expert buffers +
        the reward pickle.  Training (AIRL.py:135-212), per batch of 100: BCE(expert score, 1) + BCE(agent score, 0)
        + token CE of the agent windows against the expert windows, one Adam step and one StepLR step; tails
        shorter than a batch are dropped.  The checkpoint is written during epoch 0 only (`epoch % 5 == 0` with
        5 epochs), and `calculate_reward` below reloads that file -- so, as in the reference, the rewards come
        from the weights as they stood at the end of epoch 0."""
        agent_state_action, _, _, agent_nextstate_action, agent_done = agent_episode
        exp_state_action, _, _, exp_nextstate_action, exp_done, mask_states, mask_next_states = expert_episode
        self.last_losses = []
        if train:
            bs = self.batch_size
            agent_label = torch.zeros((bs, 1)).float().cuda()
            exp_label = torch.ones((bs, 1)).float().cuda()
            n_batch = agent_state_action.shape[0] // bs
            for epoch in range(self.epoch_disc):
                sums = torch.zeros(4, device="cuda")
                for idx in range(n_batch):
                    s, e = idx * bs, (idx + 1) * bs
                    self.sync_disc.zero_grad()
                    st_exp = exp_state_action[s:e].long().cuda()
                    m_st = mask_states[s:e].long().cuda()
                    m_nx = mask_next_states[s:e].long().cuda()
                    exp_logits = self.all_forward(st_exp, exp_done[s:e].long().cuda(),
                                                  exp_nextstate_action[s:e].long().cuda(), m_st, m_nx)
                    exp_bce = self.BCE_criterion(exp_logits, exp_label)
                    st_ag = agent_state_action[s:e].long().cuda()
                    ce = self.disc_model.token_forward(st_ag, st_exp, m_st)
                    agent_logits = self.all_forward(st_ag, agent_done[s:e].long().cuda(),
                                                    agent_nextstate_action[s:e].long().cuda(), m_st, m_nx)
                    agent_bce = self.BCE_criterion(agent_logits, agent_label)
                    global_loss = exp_bce + (agent_bce + ce)
                    global_loss.backward()
                    self.sync_disc.finish()                  # world > 1: mean gradient over ranks (RCCL)
                    self.optim_disc.step()
                    self.sched_disc.step()
                    sums += torch.stack([exp_bce.detach(), agent_bce.detach(), ce.detach(), global_loss.detach()])
                world = self.sync_disc.world
                if world > 1:
                    # replicas saw different batches: average the BatchNorm running statistics so that every rank
                    # holds (and rank 0 saves) the same discriminator
                    for buf in self.disc_model.buffers():
                        if buf.dtype.is_floating_point:
                            torch.distributed.all_reduce(buf)
                            buf.div_(world)
                if epoch % 5 == 0 and n_batch and (world == 1 or torch.distributed.get_rank() == 0):
                    os.makedirs(os.path.dirname(self.IRL_ckpt_path) or ".", exist_ok=True)
                    torch.save({"epoch": self.epoch_disc, "model_state_dict": self.disc_model.state_dict(),
                                "optimizer_state_dict": self.optim_disc.state_dict()}, self.IRL_ckpt_path)
                if world > 1:
                    torch.distributed.barrier()              # the checkpoint is on disk before any rank reloads it
                if n_batch:
                    e_l, a_l, c_l, g_l = (sums / n_batch).tolist()
                    self.last_losses.append({"expert": e_l, "agent": a_l, "ce": c_l, "global": g_l})
                    print("Epoch:{}/{}| Exp_L:{}| Gene_L:{}| CE_L:{}| Global_L:{}".format(
                        epoch, self.epoch_disc, e_l, a_l, c_l, g_l))
        traj_reward = self.calculate_reward(agent_state_action, agent_done, agent_nextstate_action, mask_states,
                                            mask_next_states)
        answer_reward = self.calculate_reward(exp_state_action, exp_done, exp_nextstate_action, mask_states,
                                              mask_next_states)
        os.makedirs(os.path.dirname(self.reward_path), exist_ok=True)
        with open(self.reward_path, "wb") as f:
            pickle.dump({"Agent": traj_reward, "Expert": answer_reward}, f)
        return traj_reward, answer_reward
