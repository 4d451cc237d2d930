"""The DQN + AIRL rollout loop of the reference, restated on the CPU (TEST INFRASTRUCTURE; see oracle/__init__.py).

Follows /root/reference/dqn_policy/IRL_dqn_train.py literally with the device moves stripped:
  RefAgentMemory / RefExpertMemory   :78-204   numpy float64 rings, `np.random.choice(BUFFER_SIZE, batch)` sampling
  rollout(...)                       :436-497  the loop body: expert windows and masks, `next_state = cat(state[:25],
                                               action)`, both `store_transition`s, and -- once the counter exceeds
                                               BUFFER_SIZE -- re-scoring of the whole buffer, the overwrite of EVERY
                                               stored reward, the two `sampling` calls and `Agent.update` with the
                                               expert transition built from the AGENT's sampled state / action / next
                                               state (:486-487)
The agent, the rewarder and the update are callables supplied by the test (a scripted agent), so the loop's own
composition is what is pinned.  The reference's DQN class calls `.cuda()` in its constructor and cannot be run in the
build container: this restatement is checked by reading, not against recorded vectors.
"""
import numpy as np
import torch


class RefAgentMemory(object):
    def __init__(self, buffer_size, n_states=50, n_actions=25, n_features=6):
        self.BUFFER_SIZE = buffer_size
        self.states_agent = np.zeros((buffer_size, n_states, n_features))
        self.actions_agent = np.zeros((buffer_size, n_actions, n_features))
        self.rewards_agent = np.zeros((buffer_size, 1))
        self.next_states_agent = np.zeros((buffer_size, n_states, n_features))
        self.dones_agent = np.zeros((buffer_size, 1))
        self.memory_counter = 0

    def store_transition(self, state, action, reward, next_state, done):
        index = self.memory_counter % self.BUFFER_SIZE
        self.states_agent[index, :, :] = state.detach().cpu().numpy()
        self.actions_agent[index, :, :] = action.detach().cpu().numpy()
        self.rewards_agent[index, :] = reward.detach().cpu().numpy()
        self.next_states_agent[index, :, :] = next_state.detach().cpu().numpy()
        self.dones_agent[index, :] = done.detach().cpu().numpy()
        self.memory_counter += 1

    def sampling(self, batch_size):
        sample_idx = np.random.choice(self.BUFFER_SIZE, batch_size)
        return (torch.from_numpy(self.states_agent[sample_idx, :, :]).long(),
                torch.from_numpy(self.actions_agent[sample_idx, :, :]).long(),
                torch.from_numpy(self.rewards_agent[sample_idx, :]).float(),
                torch.from_numpy(self.next_states_agent[sample_idx, :, :]).long(),
                torch.from_numpy(self.dones_agent[sample_idx, :]).long())

    def get(self):
        return (torch.from_numpy(self.states_agent).long(), torch.from_numpy(self.actions_agent).long(),
                torch.from_numpy(self.rewards_agent).float(), torch.from_numpy(self.next_states_agent).long(),
                torch.from_numpy(self.dones_agent).long())


class RefExpertMemory(object):
    def __init__(self, buffer_size, n_states=50, n_actions=25, n_features=6):
        self.BUFFER_SIZE = buffer_size
        self.states_exp = np.zeros((buffer_size, n_states, n_features))
        self.actions_exp = np.zeros((buffer_size, n_actions, n_features))
        self.rewards_exp = np.zeros((buffer_size, 1))
        self.next_states_exp = np.zeros((buffer_size, n_states, n_features))
        self.dones_exp = np.zeros((buffer_size, 1))
        self.mask_state = torch.zeros((buffer_size, n_states))
        self.mask_next_state = torch.zeros((buffer_size, n_states))
        self.memory_counter = 0

    def store_transition(self, state, action, reward, next_state, done, mask_state, mask_next_state):
        index = self.memory_counter % self.BUFFER_SIZE
        self.states_exp[index, :, :] = state.detach().cpu().numpy()
        self.actions_exp[index, :, :] = action.detach().cpu().numpy()
        self.rewards_exp[index, :] = reward.detach().cpu().numpy()
        self.next_states_exp[index, :, :] = next_state.detach().cpu().numpy()
        self.dones_exp[index, :] = done.detach().cpu().numpy()
        self.mask_state[index, :] = mask_state
        self.mask_next_state[index, :] = mask_next_state
        self.memory_counter += 1

    def sampling(self, batch_size):
        sample_idx = np.random.choice(self.BUFFER_SIZE, batch_size)
        return (torch.from_numpy(self.states_exp[sample_idx, :, :]).long(),
                torch.from_numpy(self.actions_exp[sample_idx, :, :]).long(),
                torch.from_numpy(self.rewards_exp[sample_idx, :]).float(),
                torch.from_numpy(self.next_states_exp[sample_idx, :, :]).long(),
                torch.from_numpy(self.dones_exp[sample_idx, :]).long(),
                self.mask_state[sample_idx, :], self.mask_next_state[sample_idx, :])

    def get(self):
        return (torch.from_numpy(self.states_exp).long(), torch.from_numpy(self.actions_exp).long(),
                torch.from_numpy(self.rewards_exp).float(), torch.from_numpy(self.next_states_exp).long(),
                torch.from_numpy(self.dones_exp).long(), self.mask_state.long(), self.mask_next_state.long())


def rollout(data_x, data_y, train_mask, choose_action, update_disc, update, num_songs, buffer_size, batch_size=30,
            episodes=50, window=50, n_actions=25):
    """IRL_dqn_train.py:436-497.  data_x (songs, >= window, 6), data_y (songs, >= episodes + 2 * window, 6) int64,
    train_mask (songs, T) float.  -> (AgentBuffer, ExpertBuffer, gene_reward)."""
    AgentBuffer, ExpertBuffer = RefAgentMemory(buffer_size), RefExpertMemory(buffer_size)
    gene_reward = []
    for epoch in range(num_songs):
        state_x = data_x[epoch, :window, :]
        expert_x = data_y[epoch, :, :]
        for num in range(0, episodes):
            Expert_state = expert_x[num: num + window]
            Expert_next_state = expert_x[num + 50: num + 50 + window]
            Expert_reward = torch.tensor(1.0).float()
            Expert_done = torch.tensor(0).long()
            Expert_mask_state = train_mask[epoch, num: num + window]
            Expert_mask_nextstate = train_mask[epoch, num + 1: num + 1 + window]
            done = torch.tensor(0).long()
            action = choose_action(state_x.unsqueeze(0), Expert_state.unsqueeze(0))
            next_state = torch.cat((state_x[:n_actions, :], action), dim=0)
            agent_reward = torch.tensor(0.5).float()
            AgentBuffer.store_transition(state_x, action, agent_reward, next_state, done)
            ExpertBuffer.store_transition(Expert_state, action, Expert_reward, Expert_next_state, Expert_done,
                                          Expert_mask_state, Expert_mask_nextstate)
            state_x = next_state
            if AgentBuffer.memory_counter > buffer_size:
                agent_traj = AgentBuffer.get()
                expert_traj = ExpertBuffer.get()
                traj_reward, answer_reward = update_disc(agent_traj, expert_traj, train=False)
                AgentBuffer.rewards_agent[:, :] = traj_reward.detach().cpu().numpy()
                gene_reward.append(np.sum(AgentBuffer.rewards_agent[:]) / 300)
                state, action, reward, next_state, done = AgentBuffer.sampling(batch_size)
                agent_transition = {'state': state, 'action': action, 'reward': reward, 'nextstate': next_state,
                                    'done': done}
                (expert_state, expert_action, expert_reward, expert_next_state, expert_done, _,
                 mask_next_states) = ExpertBuffer.sampling(batch_size)
                expert_transition = {'state': state, 'action': action, 'reward': reward, 'nextstate': next_state,
                                     'done': expert_done}
                update(agent_transition, expert_transition, mask_next_states, True, epoch)
    return AgentBuffer, ExpertBuffer, gene_reward


class RefDQN(object):
    """`DQN` of IRL_dqn_train.py:210-345 on the oracle network, devices and logging stripped: two nets, Adam(1e-2),
    MultiStepLR([20, 40], 0.1) stepped once per update, target sync when target_count % 50 == 0 (before the step),
    loss = 0.3 * TD-MSE + 0.7 * CE of `train_step(agent_state, expert_next_state, mask)`.  Pinned by
    tests/golden/dqn_rl_small.npz and dqn_loop_small.npz, both recorded from the reference's own class."""
    Target_update, GAMMA, N_ACTIONS, init_lr = 50, 0.95, 25, 0.01

    def __init__(self, eval_net, target_net):
        from . import rl_math
        self.rl_math = rl_math
        self.eval_net, self.target_net = eval_net, target_net
        self.optim = torch.optim.Adam(self.eval_net.parameters(), lr=self.init_lr)
        self.scheduler = torch.optim.lr_scheduler.MultiStepLR(self.optim, milestones=[20, 40], gamma=0.1)
        self.target_count = 0
        self.cnt_update = 0
        self.losses = []

    def choose_action(self, x, target=None):
        with torch.no_grad():
            y = self.eval_net.forward_output(self.eval_net.forward_hidden(x))
        return self.rl_math.dqn_choose_action(y, self.N_ACTIONS)

    def update(self, agent_transition, expert_transition, mask_next_states, update_flag, epoch):
        if self.target_count % self.Target_update == 0:
            self.target_net.load_state_dict(self.eval_net.state_dict())
        self.target_count += 1
        expert_next_state = expert_transition['nextstate']
        agent_state = agent_transition['state'].long()
        agent_next_state = agent_transition['nextstate'].long()
        agent_action = agent_transition['action'].long()
        agent_reward = agent_transition['reward'].float()
        agent_done = agent_transition['done'].long()
        y = self.eval_net.forward_output(self.eval_net.forward_hidden(agent_state))
        yt = self.target_net.forward_output(self.target_net.forward_hidden(agent_next_state))
        MSEloss, _ = self.rl_math.dqn_td_loss(y, yt, agent_action, agent_reward, agent_done, self.GAMMA, self.N_ACTIONS)
        CEloss = sum(self.eval_net.train_step(agent_state, expert_next_state, mask_next_states)) / 6
        alpha = 0.3
        total_loss = alpha * MSEloss + (1 - alpha) * CEloss
        self.optim.zero_grad()
        total_loss.backward()
        self.optim.step()
        self.scheduler.step()
        self.cnt_update += 1
        self.losses.append((MSEloss.item(), CEloss.item(), total_loss.item()))
        return self.losses[-1]
