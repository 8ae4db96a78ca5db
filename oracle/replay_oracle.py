"""ORACLE — TEST INFRASTRUCTURE ONLY.  Never imported by the product.

CPU restatement of the reference's ReplayBuffer and PERBuffer (src/buffer.py:8-89) on numpy rows, and of an agent's
update() around a PERBuffer (src/agent.py:1380-1387 and its copies: prioritised draw, importance-sampling weights in the
critic losses, priorities from the per-sample td_error).  Pinned by tests/golden/per_*.npz, captured from the
reference's own classes.
"""
from __future__ import annotations

import random
from collections import deque

import numpy as np
import torch


class ReplayBufferOracle:
    def __init__(self, max_len, rng=random):                       # :9-11
        self.buffer = deque(maxlen=max_len)
        self.rng = rng

    def push(self, state, action, reward, next_state, done):        # :13-14
        self.buffer.append((np.asarray(state, np.float32), np.asarray(action, np.float32), np.float32(reward),
                            np.asarray(next_state, np.float32), np.float32(done)))

    def _collate(self, batch):
        s, a, r, ns, d = zip(*batch)
        return (np.stack(s), np.stack(a), np.array(r, np.float32)[:, None], np.stack(ns), np.array(d, np.float32)[:, None])

    def sample(self, batch_size):                                   # :16-32
        assert len(self.buffer) >= batch_size, "Not enough in buffer to sample"
        return self._collate(self.rng.sample(self.buffer, batch_size))

    def __len__(self):
        return len(self.buffer)


class PERBufferOracle(ReplayBufferOracle):
    def __init__(self, max_len, alpha):                             # :39-44
        super().__init__(max_len)
        self.priorities = deque(maxlen=max_len)
        self.alpha, self.epsilon = alpha, 1e-6

    def push(self, *row):                                           # :46-48
        super().push(*row)
        self.priorities.append(1.0)

    def sample(self, batch_size, beta):                             # :50-83
        N = len(self)
        P = np.array(self.priorities, dtype=np.float32)
        P_sum = P.sum()
        if P_sum > 0:
            P /= P_sum
        else:
            P[:] = 1.0 / N
        indices = np.random.choice(N, batch_size, p=P)
        weights = (N * P[indices]) ** (-beta)
        weights /= weights.max()
        return self._collate([self.buffer[i] for i in indices]) + (weights.astype(np.float32)[:, None], indices)

    def update_priorities(self, indices, priorities):               # :88-91
        priorities = priorities.squeeze(-1)
        for index, priority in zip(indices, priorities):
            self.priorities[index] = (abs(priority) + self.epsilon) ** self.alpha


def per_update(agent, buf: PERBufferOracle, step, beta_state, eps_next=None, eps_cur=None):
    """One agent.update(step) of an OracleAgent (oracle/agent_oracle.py) around a PERBuffer; returns
    (tuple with td_error as the [B,1] array, indices, weights)."""
    cfg = agent.cfg
    s, a, r, ns, d, w, indices = buf.sample(cfg.batch_size, beta_state["beta"])
    batch = tuple(torch.from_numpy(x) for x in (s, a, r, ns, d))
    info = agent.update(step, batch=batch, weights=torch.from_numpy(w), noise=torch.zeros(cfg.batch_size, agent.ac_dim),
                        eps_next=eps_next, eps_cur=eps_cur)
    td = agent.last["td_per_sample"]
    buf.update_priorities(indices, td)
    ratio = step / cfg.beta_end                                      # beta_scheduler, src/agent.py:134-138
    beta_state["beta"] = min(1.0, beta_state["beta0"] + ratio * (1.0 - beta_state["beta0"]))
    return info, td, indices, w
