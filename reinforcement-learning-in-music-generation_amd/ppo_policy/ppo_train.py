"""Drop-in for /root/reference/ppo_policy/ppo_train.py: PPO + IRL fine-tuning of the CW Linear Transformer.
Same constants, classes (`AgentMemory`, `ExpertMemory`, `PPO`) and rollout loop; network passes and the PPO
arithmetic run on the libcwlt kernels, the buffers live in HBM.

    python ppo_train.py        (from this directory; ./dataset and ./ckpt as in the reference, synthetic CW
                                tokens when the dataset files are absent)
Environment knobs for short runs: CWLT_NUM_SONGS, CWLT_NO_PRETRAIN=1.
"""
import os
import pickle
import sys
import time

import torch
import torch.nn.functional as F

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)
import rlmg_amd  # noqa: E402,F401
from rlmg_amd import data as cwdata, dist as rdist, ops, replay, rl_ops  # noqa: E402

try:
    from model import Actor_Transformer, Critic_Transformer, LongFormer
    from config import device, datapath, Load_Pretrain
except ImportError:
    from .model import Actor_Transformer, Critic_Transformer, LongFormer
    from .config import device, datapath, Load_Pretrain

Pretrain_agent_ckpt = "./ckpt/pretrain_actor.pth"
Pretrain_eval_model_ckpt = "./ckpt/pretrain_eval.pth"
save_ckpt_path = "./ckpt/ppo_best.pt"

Target_update = 50
EPSILON = 0.9
GAMMA = 0.95
PPO_STEPS = 10
PPO_CLIP = 0.2
# CWLT_PPO_SELECT_ALL=1: `select_udpate` runs the actor on every state of the batch, as the reference does, although
# only the last one's rows are used (see PPO.select_udpate)
SELECT_ALL_STATES = os.environ.get("CWLT_PPO_SELECT_ALL", "0") == "1"
DISCOUNT_FACTOR = 0.99

NUM_SONGS = int(os.environ.get("CWLT_NUM_SONGS", 1000))
EPISODES = 30
SEQ_LEN = 1000
N_STATES = 50
N_FEATURES = 6
N_ACTIONS = 25
WINDOW_SIZE = 50
BUFFER_SIZE = EPISODES
ACTION_DIM = 6
NUM_ACTION = 25
batch_size = 6
init_lr = 0.01

AgentBuffer = None      # module-level, as in the reference: PPO.update_policy reads these (ppo_train.py:370-371)
ExpertBuffer = None


class AgentMemory(replay.AgentMemory):
    def __init__(self):
        super().__init__(BUFFER_SIZE, N_STATES, N_ACTIONS, N_FEATURES, device, with_ppo_fields=True)


class ExpertMemory(replay.ExpertMemory):
    def __init__(self):
        super().__init__(BUFFER_SIZE, N_STATES, N_ACTIONS, N_FEATURES, device, as_dict=True)


class PPO(object):
    def __init__(self, n_class, Pretrain=Load_Pretrain):
        self.n_class = list(n_class)
        self.actor_net = Actor_Transformer(n_class).to(device)
        self.critic_net = Critic_Transformer(n_class).to(device)
        self.eval_net = LongFormer(n_class).to(device)
        if Pretrain:
            print(f"Load pretrain From: {Pretrain_agent_ckpt}")
            self.actor_net.load_state_dict(torch.load(Pretrain_agent_ckpt, map_location=device), strict=False)
            print(f"Reward Model From: {Pretrain_eval_model_ckpt}")
            self.eval_net.load_state_dict(torch.load(Pretrain_eval_model_ckpt, map_location=device), strict=False)
        self.actor_net.train()
        self.critic_net.train()          # eval_net is left in its default train() mode, as in the reference (:235)
        self.agent_buffer = AgentMemory()
        self.expert_buffer = ExpertMemory()
        # RCCL all-reduce when world > 1.  overlap=False: every PPO step runs the actor twice before one backward
        # (select_udpate + train_step) or accumulates several backward passes (update_rollouts), so the buckets
        # are reduced once per step, in finish()
        self.actor_sync = rdist.GradSync(self.actor_net.parameters(), overlap=False)
        self.critic_sync = rdist.GradSync(self.critic_net.parameters(), overlap=False)
        self.actor_optim = ops.graph_adam(self.actor_net.parameters(), lr=init_lr)
        self.critic_optim = ops.graph_adam(self.critic_net.parameters(), lr=init_lr)
        self.target_count = self.cnt_update = 0
        self.mse_val = self.ce_val = self.total_val = 0.0
        self.record_for_epoch = 0

    def _heads(self, net, state_x, want_probs):
        logits = net.fused_logits(net.forward_hidden(state_x))
        B, T = state_x.shape[0], state_x.shape[1]
        res = ops.heads_forward(logits.float(), self.n_class, want_argmax=True, want_probs=want_probs)
        return logits, res["argmax"].view(B, T, -1), (res["probs"].view(B, T, -1) if want_probs else None)

    def choose_action(self, state_x):
        """(1, 50, 6) -> action (25, 6) at positions -1..-25, log-probs (25, 6) -- tempo / chord probabilities
        looked up with the class chosen at position +idx, as the reference does (ppo_train.py:273-274).
        A batch (R, 50, 6) gives (R, 25, 6)."""
        with torch.no_grad():
            _, ids, probs = self._heads(self.actor_net, state_x, True)
            action, logp = rl_ops.rollout_gather(ids, probs, self.n_class, N_ACTIONS, mode=1)
        if state_x.shape[0] == 1:
            return action[0], logp[0]
        return action, logp

    def _rollout_step_device(self, state_x, reward_mask):
        with torch.no_grad():
            action, logp = self.choose_action(state_x)
            if state_x.shape[0] == 1:
                action, logp = action.unsqueeze(0), logp.unsqueeze(0)
            next_state = torch.cat((state_x[:, :N_ACTIONS], action), dim=1)
            value = self.critic_net.value_produce(next_state)
            reward = self.eval_net.token_forward(next_state, None, reward_mask)
        return action, logp, next_state, value, reward

    def rollout_step(self, state_x, reward_mask):
        """One environment step of the reference's rollout loop (ppo_train.py:475-489) for R rollouts in lock-step:
        actor greedy action + log-probs, next state = first half of the window + the action rows, critic value of
        the next state, reward model score.  state_x (R, W, 6) int64, reward_mask (R, W).
        -> action (R, NA, 6), logp (R, NA, 6), next_state (R, W, 6), value, reward.  The whole step (three
        network forwards, ~900 launches at W = 50) is replayed as one hipGraph unless CWLT_GRAPHS=0."""
        if not ops.GRAPHS_ENABLED:
            return self._rollout_step_device(state_x, reward_mask)
        if getattr(self, "_graph_step", None) is None:
            self._graph_step = ops.GraphedCall(self._rollout_step_device)
        return tuple(o.clone() for o in self._graph_step(state_x, reward_mask.float()))

    def select_udpate(self, state_x):
        """ppo_train.py:293-346: the greedy rows / log-probs of the LAST batch element + critic values.
        The reference runs the actor on all B states, builds rows and log-probs for every one of them in a Python loop
        and returns the loop variables of the last iteration (:346): the other B - 1 elements' actor outputs reach
        nothing -- no return value, no loss, no gradient.  Sequences do not interact in the network (attention and
        LayerNorm are per sequence / per row), so the actor is run on the last state only: same action rows, same
        log-probs, same gradients, 1 / B of the pass (SELECT_ALL_STATES / CWLT_PPO_SELECT_ALL=1 runs all B, the
        literal form; tests/test_ppo_configs_gpu.py compares the two).  The critic's values are needed for all B."""
        net = self.actor_net
        B, T = state_x.shape[0], state_x.shape[1]
        back = torch.arange(N_ACTIONS, device=state_x.device)
        if SELECT_ALL_STATES:
            logits = net.fused_logits(net.forward_hidden(state_x))              # (B*T, W)
            rows = (B - 1) * T + (T - 1 - back)
        else:
            logits = net.fused_logits(net.forward_hidden(state_x[B - 1:]))      # (T, W): the state whose rows are used
            rows = T - 1 - back
        logp, action = rl_ops.logp_argmax(logits.float().index_select(0, rows), self.n_class)
        value_state = self.critic_net.value_produce(state_x)
        return action, logp, value_state

    def update_rollouts(self, states, old_logp_int, advs, rets, expert, mask, group=8, clip=None):
        """`_update_rollouts_device`, replayed as one hipGraph when the step is launch-bound (few tokens) and graphs
        are enabled; eager otherwise (at 64 x 1024 the GPU is the bottleneck and a capture would only hold memory)."""
        clip = PPO_CLIP if clip is None else clip
        E, R, W = states.shape[0], states.shape[1], states.shape[2]
        if ops.train_graphs_enabled() and E * R * W <= 32768:
            key = (int(group), float(clip))
            graphs = self.__dict__.setdefault("_graph_rollouts", {})
            if key not in graphs:
                graphs[key] = ops.GraphedCall(
                    lambda st, ol, ad, rt, ex, mk: self._update_rollouts_device(st, ol, list(ad), list(rt), ex, mk,
                                                                                group, clip) or st.new_zeros(()),
                    grad=True, params=list(self.actor_net.parameters()) + list(self.critic_net.parameters()))
            graphs[key](states, old_logp_int, torch.stack(list(advs)), torch.stack(list(rets)), expert, mask.float())
            return
        self._update_rollouts_device(states, old_logp_int, advs, rets, expert, mask, group, clip)

    def _update_rollouts_device(self, states, old_logp_int, advs, rets, expert, mask, group=8, clip=None):
        """One PPO inner step (the body of update_policy's loop, ppo_train.py:360-420) over R rollouts run in
        lock-step -- the many-rollout generalisation bench_ppo.py measures (BASELINE configs[2]).
        states (E, R, W, 6) int64, old_logp_int (E, R, NA, 6) int64 (the buffer's `.long()` log-probs),
        advs / rets: R tensors (E,), expert (R, >= E + W, 6), mask (R, >= E + W).
        Per rollout, as the reference does for its single one: `select_udpate` on the (E, W, 6) states, the
        ratio-clip surrogate on the last batch element's rows, the mean of 6 CE losses vs the expert windows,
        the (E,) vs (E, 1) broadcast MSE for the critic; losses averaged over the R rollouts, one Adam step per
        net.  `group` rollouts are stacked to (group*E, W, 6) per network pass; summing their losses before one
        backward equals accumulating their separate backwards."""
        clip = PPO_CLIP if clip is None else clip
        E, R, W = states.shape[0], states.shape[1], states.shape[2]
        NA = old_logp_int.shape[2]
        dev = states.device
        actor, critic = self.actor_net, self.critic_net
        back = torch.arange(NA, device=dev)
        self.actor_sync.zero_grad()
        self.critic_sync.zero_grad()
        # several backward passes accumulate into the buckets: reduced across ranks once, in finish() (overlap=False)
        for r0 in range(0, R, group):
            k = min(group, R - r0)
            st = states[:, r0:r0 + k].transpose(0, 1).reshape(k * E, W, 6)              # rollout-major
            if SELECT_ALL_STATES:
                logits = actor.fused_logits(actor.forward_hidden(st))                     # select_udpate pass, literal
                rows = (((torch.arange(k, device=dev) + 1) * E - 1) * W + (W - 1))[:, None] - back[None, :]
            else:
                # select_udpate pass on each rollout's LAST state only: the only one whose rows it returns (see there)
                logits = actor.fused_logits(actor.forward_hidden(states[E - 1, r0:r0 + k]))
                rows = (torch.arange(k, device=dev) * W + (W - 1))[:, None] - back[None, :]
            value_pred = critic.value_produce(st).view(k, E, 1)
            logp_all, _ = rl_ops.logp_argmax(logits.index_select(0, rows.reshape(-1)).float(), self.n_class)
            logp_all = logp_all.view(k, NA, 6)
            tgt = expert[r0:r0 + k, :E + W].unfold(1, W, 1)[:, :E].permute(0, 1, 3, 2).reshape(k * E, W, 6)
            m = mask[r0:r0 + k, :E + W].unfold(1, W, 1)[:, :E].float()                    # (k, E, W)
            # train_step returns sum(mask * nll) / sum(mask) over its batch; weighting each rollout's mask by
            # 1 / its own mask sum makes that the mean of the k per-rollout masked means, exactly
            m = (m / m.sum(dim=(1, 2), keepdim=True)).reshape(k * E, W)
            ce = actor.train_step(st, tgt, m)                                             # second actor pass
            actor_loss = (ce[0] + ce[1] + ce[2] + ce[3] + ce[4] + ce[5]) / 6 * k
            critic_loss = 0
            for j in range(k):
                actor_loss = actor_loss + rl_ops.ppo_policy_loss(logp_all[j], old_logp_int[:, r0 + j],
                                                                 advs[r0 + j], clip)
                critic_loss = critic_loss + torch.nn.functional.mse_loss(rets[r0 + j], value_pred[j]).sum()
            (actor_loss / R).backward()
            (critic_loss / R).backward()
        self.actor_sync.finish()
        self.critic_sync.finish()
        self.actor_optim.step()
        self.critic_optim.step()

    def calculate_returns(self, rewards, discount_factor, normalize=True):
        ret, _ = rl_ops.ppo_returns_adv(rewards.to(device), torch.zeros_like(rewards, device=device), discount_factor,
                                        normalize)
        return ret

    def calculate_advantages(self, returns, values, normalize=True):
        adv = returns - values
        if normalize:
            adv = (adv - adv.mean()) / adv.std()
        return adv

    def _ppo_step_device(self, input_state, log_actions, advantages, returns, expert_states, expert_mask):
        """Device part of one inner step of update_policy (ppo_train.py:360-420)."""
        new_action, new_log_prob_action, value_pred = self.select_udpate(input_state)
        policy_loss = rl_ops.ppo_policy_loss(new_log_prob_action, log_actions, advantages, self._ppo_clip)
        ce = self.actor_net.train_step(input_state, expert_states, expert_mask)
        ce_loss = (ce[0] + ce[1] + ce[2] + ce[3] + ce[4] + ce[5]) / 6
        actor_loss = policy_loss + ce_loss
        value_loss = F.mse_loss(returns, value_pred).sum()
        self.actor_sync.zero_grad()
        actor_loss.backward()
        self.actor_sync.finish()
        self.actor_optim.step()
        self.critic_sync.zero_grad()
        value_loss.backward()
        self.critic_sync.finish()
        self.critic_optim.step()
        return actor_loss.detach(), value_loss.detach()

    def update_policy(self, ppo_steps, ppo_clip, advantages, returns):
        agent_all = AgentBuffer.get()
        expert_all = ExpertBuffer.get()
        log_actions = agent_all["log_actions"].detach()
        advantages = advantages.to(device).detach()
        returns = returns.to(device).detach()
        total_policy_loss = total_value_loss = 0.0
        args = (agent_all["states"], log_actions, advantages.float(), returns.float(), expert_all["states"],
                expert_all["mask_state"].float())
        self._ppo_clip = ppo_clip
        for epoch in range(ppo_steps):
            if ops.train_graphs_enabled():
                # 30 x 50-token states: both networks' forward/backward and Adam steps (~4 000 small launches)
                # are captured once and replayed as one hipGraph per inner step (CWLT_GRAPHS=0: eager)
                if getattr(self, "_graph_ppo_step", None) is None:
                    self._graph_ppo_step = ops.GraphedCall(
                        self._ppo_step_device, grad=True,
                        params=list(self.actor_net.parameters()) + list(self.critic_net.parameters()))
                actor_loss, value_loss = self._graph_ppo_step(*args)
            else:
                actor_loss, value_loss = self._ppo_step_device(*args)
            a, c = actor_loss.item(), value_loss.item()
            total_policy_loss += a
            total_value_loss += c
            print("Update_PPO:{}/{}| Actor_loss:{:03f}| Critic_loss:{:.03f}".format(epoch, ppo_steps, a, c))
        return total_policy_loss / ppo_steps


def main():
    global AgentBuffer, ExpertBuffer
    rank, local, world = rdist.init_from_env()
    if torch.cuda.is_available():
        torch.cuda.set_device(local)
    dictionary, my_dataset = cwdata.load_ppo(datapath["path_dictionary"], datapath["path_train_data"])
    event2word, word2event = dictionary
    token_class = [len(event2word[k]) for k in event2word.keys()]
    pre = Load_Pretrain and os.environ.get("CWLT_NO_PRETRAIN") != "1" and os.path.exists(Pretrain_agent_ckpt)
    Agent = PPO(token_class, Pretrain=pre)
    train_x = torch.from_numpy(my_dataset["train_x"]).long().to(device)
    train_y = torch.from_numpy(my_dataset["train_y"]).long().to(device)
    train_mask = torch.from_numpy(my_dataset["mask"]).to(device)
    policy_loss_list = []
    t0, steps = time.time(), 0
    for epoch in range(NUM_SONGS):
        song = (epoch * world + rank) % train_x.shape[0]
        state_x, expert_x = train_x[song, :WINDOW_SIZE, :], train_y[song]
        AgentBuffer, ExpertBuffer = AgentMemory(), ExpertMemory()
        for num in range(EPISODES):
            Expert_state = expert_x[num: num + WINDOW_SIZE]
            Expert_next_state = expert_x[num + 50: num + 50 + WINDOW_SIZE]
            Expert_reward = torch.tensor(1.0).float().to(device)
            Expert_done = torch.tensor(0).long().to(device)
            Expert_mask_state = train_mask[song, num: num + WINDOW_SIZE]
            Expert_mask_nextstate = train_mask[song, num + 1: num + 1 + WINDOW_SIZE]
            done = torch.tensor(0).long().to(device)
            action, log_prob_res, next_state, value_state, agent_reward = (
                t[0] for t in Agent.rollout_step(state_x.unsqueeze(0), Expert_mask_state.unsqueeze(0)))
            value_state, agent_reward = value_state.reshape(1, 1), agent_reward.reshape(1, 1)
            state_x = next_state
            AgentBuffer.store_transition(state_x, action, log_prob_res, value_state, agent_reward, next_state, done)
            ExpertBuffer.store_transition(Expert_state, action, Expert_reward, Expert_next_state, Expert_done,
                                          Expert_mask_state, Expert_mask_nextstate)
            steps += 1
        agent_all = AgentBuffer.get()
        returns_result = Agent.calculate_returns(agent_all["rewards"], DISCOUNT_FACTOR)
        advantages = Agent.calculate_advantages(returns_result, agent_all["values"])
        policy_loss_list.append(Agent.update_policy(PPO_STEPS, PPO_CLIP, advantages, returns_result))
        print("Overall Progress...Epoch:{}/{} | {:.1f} env-steps/s".format(epoch, NUM_SONGS, steps / (time.time() - t0)))
        if rank == 0 and epoch % 5 == 0:
            os.makedirs("./ckpt", exist_ok=True)
            torch.save(Agent.actor_net.state_dict(), save_ckpt_path)
        if rank == 0 and epoch % 20 == 0:
            with open("./ckpt/policy_loss.pickle", "wb") as f:
                pickle.dump({"policy_loss": policy_loss_list}, f)


if __name__ == "__main__":
    main()
