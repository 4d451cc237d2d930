"""GPU-resident replay buffers with the reference's interface and sampling semantics.

Reference: numpy float64 ring buffers, one D2H copy per field per env step and a whole-buffer H2D
(2 x 48 MB) on every `get()` (dqn_policy/IRL_dqn_train.py:78-204, ppo_policy/ppo_train.py:69-212).
Here the rings are int64 / f32 tensors in HBM: `store_transition` is a device-side row write, `get` returns
views, `sampling` keeps `np.random.choice(BUFFER_SIZE, batch_size)` (with replacement, over ALL slots even
if unfilled) so seeds reproduce the reference's indices.  PPO's stored log-probs come back through
`.long()` (truncated toward zero), as the reference returns them (ppo_train.py:122,135).
"""
import numpy as np
import torch


class AgentMemory(object):
    def __init__(self, buffer_size, n_states, n_actions, n_features, device, with_ppo_fields=False):
        B = self.BUFFER_SIZE = buffer_size
        i64, f32 = dict(dtype=torch.int64, device=device), dict(dtype=torch.float32, device=device)
        self.states_agent = torch.zeros((B, n_states, n_features), **i64)
        self.actions_agent = torch.zeros((B, n_actions, n_features), **i64)
        self.rewards_agent = torch.zeros((B, 1), **f32)
        self.next_states_agent = torch.zeros((B, n_states, n_features), **i64)
        self.dones_agent = torch.zeros((B, 1), **i64)
        self.ppo = with_ppo_fields
        if with_ppo_fields:
            self.value_agent = torch.zeros((B, 1), **f32)
            self.log_actions_agent = torch.zeros((B, n_actions, n_features), **f32)
        self.memory_counter = 0

    def store_transition(self, state, action, *rest):
        i = self.memory_counter % self.BUFFER_SIZE
        if self.ppo:
            log_action, value_state, reward, next_state, done = rest
            self.log_actions_agent[i] = log_action.detach()
            self.value_agent[i] = value_state.detach().reshape(1)
        else:
            reward, next_state, done = rest
        self.states_agent[i] = state.detach()
        self.actions_agent[i] = action.detach()
        self.rewards_agent[i] = reward.detach().reshape(1)
        self.next_states_agent[i] = next_state.detach()
        self.dones_agent[i] = done.detach().reshape(1)
        self.memory_counter += 1

    def sampling(self, batch_size):
        idx = torch.from_numpy(np.random.choice(self.BUFFER_SIZE, batch_size)).to(self.states_agent.device)
        out = [self.states_agent[idx], self.actions_agent[idx]]
        if self.ppo:
            out += [self.log_actions_agent[idx].long(), self.value_agent[idx].cpu(), self.rewards_agent[idx].cpu()]
        else:
            out += [self.rewards_agent[idx].cpu()]            # the reference leaves rewards on the host (:118)
        out += [self.next_states_agent[idx], self.dones_agent[idx]]
        return tuple(out)

    def get(self):
        if self.ppo:
            return {"states": self.states_agent, "actions": self.actions_agent,
                    "log_actions": self.log_actions_agent.long(), "values": self.value_agent,
                    "rewards": self.rewards_agent, "next_states": self.next_states_agent, "dones": self.dones_agent}
        return (self.states_agent, self.actions_agent, self.rewards_agent, self.next_states_agent, self.dones_agent)


class ExpertMemory(object):
    def __init__(self, buffer_size, n_states, n_actions, n_features, device, as_dict=False):
        B = self.BUFFER_SIZE = buffer_size
        i64, f32 = dict(dtype=torch.int64, device=device), dict(dtype=torch.float32, device=device)
        self.states_exp = torch.zeros((B, n_states, n_features), **i64)
        self.actions_exp = torch.zeros((B, n_actions, n_features), **i64)
        self.rewards_exp = torch.zeros((B, 1), **f32)
        self.next_states_exp = torch.zeros((B, n_states, n_features), **i64)
        self.dones_exp = torch.zeros((B, 1), **i64)
        self.mask_state = torch.zeros((B, n_states), **f32)
        self.mask_next_state = torch.zeros((B, n_states), **f32)
        self.memory_counter = 0
        self.as_dict = as_dict

    def store_transition(self, state, action, reward, next_state, done, mask_state, mask_next_state):
        i = self.memory_counter % self.BUFFER_SIZE
        self.states_exp[i] = state.detach()
        self.actions_exp[i] = action.detach()
        self.rewards_exp[i] = reward.detach().reshape(1)
        self.next_states_exp[i] = next_state.detach()
        self.dones_exp[i] = done.detach().reshape(1)
        self.mask_state[i] = mask_state.to(self.mask_state)
        self.mask_next_state[i] = mask_next_state.to(self.mask_state)
        self.memory_counter += 1

    def sampling(self, batch_size):
        idx = torch.from_numpy(np.random.choice(self.BUFFER_SIZE, batch_size)).to(self.states_exp.device)
        return (self.states_exp[idx], self.actions_exp[idx], self.rewards_exp[idx], self.next_states_exp[idx],
                self.dones_exp[idx], self.mask_state[idx], self.mask_next_state[idx])

    def get(self):
        if self.as_dict:
            return {"states": self.states_exp, "actions": self.actions_exp, "rewards": self.rewards_exp,
                    "next_states": self.next_states_exp, "mask_state": self.mask_state.long(),
                    "mask_next_state": self.mask_next_state.long()}
        return (self.states_exp, self.actions_exp, self.rewards_exp, self.next_states_exp, self.dones_exp,
                self.mask_state.long(), self.mask_next_state.long())
