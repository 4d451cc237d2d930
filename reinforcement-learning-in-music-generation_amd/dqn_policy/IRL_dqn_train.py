"""Drop-in for /root/reference/dqn_policy/IRL_dqn_train.py: DQN + AIRL fine-tuning of the CW Linear
Transformer.  Same constants, classes (`AgentMemory`, `ExpertMemory`, `DQN`) and rollout loop; the
network passes and the RL arithmetic run on the libcwlt kernels, the buffers live in HBM.

    python IRL_dqn_train.py          (from this directory; data / checkpoints as in the reference, synthetic
                                      CW tokens when the dataset files are absent)
Environment knobs for short runs: CWLT_NUM_SONGS, CWLT_BUFFER_SIZE, CWLT_NO_PRETRAIN=1.
"""
import os
import sys
import time

import torch
import torch.optim as optim

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)
import rlmg_amd  # noqa: E402,F401
from rlmg_amd import data as cwdata, dist as rdist, ops, replay, rl_ops  # noqa: E402

try:
    from model import LinearTransformer
    from AIRL import RewardDiscri
except ImportError:
    from .model import LinearTransformer
    from .AIRL import RewardDiscri

################################################################################
# config (IRL_dqn_train.py:33-65)
################################################################################
path_data_root = "/data/dataset_Pop1K7/representations/uncond/cp/ailab17k_from-scratch_cp"
path_train_data = os.path.join(path_data_root, "train_data_linear.npz")
path_dictionary = os.path.join(path_data_root, "dictionary.pkl")
Pretrain_ckpt = "/data/Der_CODES/DQN-cp/ckpt/trainloss_13.pt"
save_ckpt_path = "./ckpt/dqn_best.pt"

Target_update = 50
EPSILON = 0.9
GAMMA = 0.95

NUM_SONGS = int(os.environ.get("CWLT_NUM_SONGS", 1500))
EPISODES = 50
SEQ_LEN = 1000
N_STATES = 50
N_FEATURES = 6
N_ACTIONS = 25
WINDOW_SIZE = 50
BUFFER_SIZE = int(os.environ.get("CWLT_BUFFER_SIZE", 20000))
ACTION_DIM = 6
NUM_ACTION = 25
batch_size = 30
init_lr = 0.01


def _device():
    return torch.device("cuda", torch.cuda.current_device())


class AgentMemory(replay.AgentMemory):
    def __init__(self):
        super().__init__(BUFFER_SIZE, N_STATES, N_ACTIONS, N_FEATURES, _device())


class ExpertMemory(replay.ExpertMemory):
    def __init__(self):
        super().__init__(BUFFER_SIZE, N_STATES, N_ACTIONS, N_FEATURES, _device())


class DQN(object):
    def __init__(self, n_class, Pretrain=True):
        self.n_class = list(n_class)
        self.eval_net = LinearTransformer(n_class).cuda()
        self.target_net = LinearTransformer(n_class).cuda()
        if Pretrain:
            print(f"Load Pretrain from: {Pretrain_ckpt}")
            checkpoint = torch.load(Pretrain_ckpt)
            self.eval_net.load_state_dict(checkpoint["model_state_dict"])
        self.eval_net.train()
        self.target_net.train()
        self.agent_buffer = AgentMemory()
        self.expert_buffer = ExpertMemory()
        # flat gradient buckets; RCCL all-reduce if world > 1.  overlap=False: update() runs eval_net twice before one
        # backward (TD pass + train_step), so every encoder gradient is delivered twice -- reduce once, in finish()
        self.sync = rdist.GradSync(self.eval_net.parameters(), overlap=False)
        self.optim = ops.graph_adam(self.eval_net.parameters(), lr=init_lr)
        self.scheduler = optim.lr_scheduler.MultiStepLR(self.optim, milestones=[20, 40], gamma=0.1)
        self.target_count = 0
        self.cnt_update = 0
        self.mse_val = self.ce_val = self.total_val = 0.0
        self.record_fore_epoch = 0
        self.update_flag = False
        self.history = {"mse": [], "ce": [], "total": []}

    def _fused(self, net, x):
        return net.fused_logits(net.forward_hidden(x)).view(x.shape[0], x.shape[1], -1)

    def _choose_action_device(self, x):
        with torch.no_grad():
            logits = self._fused(self.eval_net, x)
            B, T, W = logits.shape
            ids = ops.heads_forward(logits.view(B * T, W), self.n_class, want_argmax=True)["argmax"].view(B, T, -1)
            action, _ = rl_ops.rollout_gather(ids, None, self.n_class, N_ACTIONS, mode=0)
        return action

    def choose_action(self, x, target=None):
        """(R, 50, 6) -> (25, 6) greedy CW tokens at positions [0, -1, ..., -24] (the reference's `-idx`
        indexing, IRL_dqn_train.py:256-258); (R, 25, 6) for a batch of R states.  The ~300 launches of the
        50-token forward are replayed as one hipGraph (ops.GraphedCall; CWLT_GRAPHS=0 launches them eagerly)."""
        if ops.GRAPHS_ENABLED:
            if getattr(self, "_graph_choose", None) is None:
                self._graph_choose = ops.GraphedCall(self._choose_action_device)
            action = self._graph_choose(x).clone()
        else:
            action = self._choose_action_device(x)
        return action[0] if x.shape[0] == 1 else action

    def _update_device(self, agent_state, agent_next_state, agent_action, agent_reward, agent_done,
                       expert_next_state, mask_next_states):
        """Device part of one update (IRL_dqn_train.py:285-345): losses, backward, Adam step."""
        y = self._fused(self.eval_net, agent_state)
        with torch.no_grad():        # the reference leaves this under autograd; the target net is never stepped
            yt = self._fused(self.target_net, agent_next_state)
        mse = rl_ops.dqn_td_mse(y, yt, agent_action, agent_reward, agent_done, self.n_class, GAMMA)
        MSEloss = mse.sum() / 6
        ce = self.eval_net.train_step(agent_state, expert_next_state, mask_next_states)
        CEloss = (ce[0] + ce[1] + ce[2] + ce[3] + ce[4] + ce[5]) / 6
        alpha = 0.3
        total_loss = alpha * MSEloss + (1 - alpha) * CEloss
        self.sync.zero_grad()
        total_loss.backward()
        self.sync.finish()
        self.optim.step()
        return MSEloss.detach(), CEloss.detach(), total_loss.detach()

    def update(self, agent_transition, expert_transition, mask_next_states, update_flag, epoch):
        if self.target_count % Target_update == 0:
            self.target_net.load_state_dict(self.eval_net.state_dict())
        self.target_count += 1
        expert_next_state = expert_transition["nextstate"]
        agent_state = agent_transition["state"].long().cuda()
        agent_next_state = agent_transition["nextstate"].long().cuda()
        agent_action = agent_transition["action"].long().cuda()
        agent_reward = agent_transition["reward"].float().cuda()
        agent_done = agent_transition["done"].long().cuda()

        args = (agent_state, agent_next_state, agent_action, agent_reward, agent_done,
                expert_next_state.long().cuda(), mask_next_states.float().cuda())
        if ops.train_graphs_enabled():
            # batch 30 x 50 tokens: ~2 000 small launches per update -- with CWLT_TRAIN_GRAPHS=1 forward, backward and
            # Adam step are captured once and replayed as one hipGraph (opt-in: ops.train_graphs_enabled)
            if getattr(self, "_graph_update", None) is None:
                self._graph_update = ops.GraphedCall(self._update_device, grad=True,
                                                          params=list(self.eval_net.parameters()))
            MSEloss, CEloss, total_loss = self._graph_update(*args)
        else:
            MSEloss, CEloss, total_loss = self._update_device(*args)
        self.scheduler.step()
        self.cnt_update += 1
        m, c, t = MSEloss.item(), CEloss.item(), total_loss.item()
        self.mse_val += m
        self.ce_val += c
        self.total_val += t
        print("Epoch: {}/{}| MSE_Loss: {:03f}| CE_Loss: {:03f}| TD_Loss: {:03f}".format(epoch, NUM_SONGS, m, c, t))
        if self.record_fore_epoch < epoch:
            self.record_fore_epoch += 1
            for k, v in (("mse", self.mse_val), ("ce", self.ce_val), ("total", self.total_val)):
                self.history[k].append(v / self.cnt_update)
        if update_flag and epoch >= 410:
            os.makedirs("./ckpt", exist_ok=True)
            torch.save({"epoch": epoch, "model_state_dict": self.eval_net.state_dict(),
                        "optimizer_state_dict": self.optim.state_dict()}, save_ckpt_path)
        return m, c, t


def main():
    rank, local, world = rdist.init_from_env()
    torch.cuda.set_device(local)
    dictionary, train_data = cwdata.load_dqn(path_train_data, path_dictionary)
    event2word, word2event = dictionary
    n_class = [len(event2word[k]) for k in event2word.keys() if k != "type"]     # [56, 135, 18, 87, 18, 25]

    AgentBuffer, ExpertBuffer = AgentMemory(), ExpertMemory()
    pre = os.environ.get("CWLT_NO_PRETRAIN") != "1" and os.path.exists(Pretrain_ckpt)
    Agent = DQN(n_class, Pretrain=pre)
    Rewarder = RewardDiscri(n_class, Pretrain=False)

    train_x, train_y = torch.from_numpy(train_data["x"]), torch.from_numpy(train_data["y"])
    train_mask = torch.from_numpy(train_data["mask"])
    train_x = torch.cat((train_x[:, :, :3], train_x[:, :, 4:]), dim=-1)         # drop `type` (index 3)
    train_y = torch.cat((train_y[:, :, :3], train_y[:, :, 4:]), dim=-1)
    data_x = train_x[:, :SEQ_LEN, :].long().cuda()
    data_y = train_y[:, :SEQ_LEN * 2, :].long().cuda()
    train_mask = train_mask.float().cuda()

    gene_reward = []
    t0 = time.time()
    for epoch in range(NUM_SONGS):
        song = (epoch * world + rank) % data_x.shape[0]          # data-parallel: rank r plays songs r, r+world, ...
        state_x, expert_x = data_x[song, :WINDOW_SIZE, :], data_y[song]
        for num in range(EPISODES):
            Expert_state = expert_x[num: num + WINDOW_SIZE]
            Expert_next_state = expert_x[num + 50: num + 50 + WINDOW_SIZE]
            Expert_reward = torch.tensor(1.0).float().cuda()
            Expert_done = torch.tensor(0).long().cuda()
            Expert_mask_state = train_mask[song, num: num + WINDOW_SIZE]
            Expert_mask_nextstate = train_mask[song, num + 1: num + 1 + WINDOW_SIZE]
            done = torch.tensor(0).long().cuda()
            action = Agent.choose_action(state_x.unsqueeze(0), Expert_state.unsqueeze(0))
            next_state = torch.cat((state_x[:N_ACTIONS, :], action), dim=0)
            agent_reward = torch.tensor(0.5).float().cuda()
            AgentBuffer.store_transition(state_x, action, agent_reward, next_state, done)
            ExpertBuffer.store_transition(Expert_state, action, Expert_reward, Expert_next_state, Expert_done,
                                          Expert_mask_state, Expert_mask_nextstate)
            state_x = next_state
            if AgentBuffer.memory_counter > BUFFER_SIZE:
                traj_reward, _ = Rewarder.update_disc(AgentBuffer.get(), ExpertBuffer.get(), train=False)
                AgentBuffer.rewards_agent[:, :] = traj_reward.to(AgentBuffer.rewards_agent)
                gene_reward.append(float(AgentBuffer.rewards_agent.sum().item()) / 300)
                state, action_b, reward, next_state_b, done_b = AgentBuffer.sampling(batch_size)
                agent_transition = {"state": state, "action": action_b, "reward": reward, "nextstate": next_state_b,
                                    "done": done_b}
                _, _, _, _, expert_done, _, mask_next_states = ExpertBuffer.sampling(batch_size)
                expert_transition = {"state": state, "action": action_b, "reward": reward,
                                     "nextstate": next_state_b, "done": expert_done}      # reference :486-487
                Agent.update(agent_transition, expert_transition, mask_next_states.cuda(), True, epoch)
            elif num == EPISODES - 1:
                print("Epoch: {}/{} | Buffer_Size:{} | {:.1f} env-steps/s".format(
                    epoch, NUM_SONGS, AgentBuffer.memory_counter, AgentBuffer.memory_counter / (time.time() - t0)))


if __name__ == "__main__":
    main()
