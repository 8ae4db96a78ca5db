"""Synthetic inputs for benchmarks, examples and profiling runs (SURVEY.md §8d shapes): panda-gym is not
installable here, so the workloads are made of random transitions with the reference's structure — state =
[observation (last entry a t/50 time feature, src/utils.py:137-174) | desired goal], sparse goal-distance
reward, a goal that is fixed per episode and an achieved goal that wanders, so that HER relabelling produces
both reached and unreached rows.  Also the hyper-parameter container with the reference YAMLs' field names.
"""
from __future__ import annotations

from types import SimpleNamespace

import numpy as np


def agent_config(kind: str = "DDPG", **over):
    """Agent hyper-parameters under the field names of the reference's config files
    (src/config/<AGENT>/*.yaml, validated by src/utils.py:10-39); defaults = config_ddpg_reach.yaml."""
    cfg = dict(hidden_dim=64, layer_count=3, actor_lr=1e-3, actor_lr_min=1e-3, ac_scheduler_steps=1,
               critic_lr=1e-3, critic_lr_min=1e-3, cr_scheduler_steps=1, buffer_type="HER", max_len=100000,
               alpha=1.0, batch_size=256, gamma=0.98, ac_update_freq=1, noise_std=0.2, noise_clamp=0.5,
               policy_noise=0.0, grad_clip=10.0, beta=1.0, beta_end=1, k_future=4, max_eps_len=50, tau=0.05)
    if kind in ("SAC", "TQC"):
        cfg.update(alpha_lr=3e-4, alpha_min=0.05, alpha_min_steps=0.0)
    cfg.update(over)
    return SimpleNamespace(**cfg)


def sparse_goal_reward(achieved_goal, desired_goal, info=None, threshold: float = 0.05):
    """panda-gym's sparse task reward: -(distance > threshold), float32 distance."""
    d = np.linalg.norm(np.asarray(achieved_goal, dtype=np.float32) - np.asarray(desired_goal, dtype=np.float32), axis=-1)
    return -np.array(d > threshold, dtype=np.float32)


def synthetic_episode(rng: np.random.Generator, T: int, S: int, A: int, G: int = 3):
    """T transitions (s, a, ns, r, done, desired_goal, achieved_goal) of one synthetic episode."""
    goal = rng.uniform(-0.15, 0.15, size=G).astype(np.float32)
    achieved = rng.uniform(-0.15, 0.15, size=G).astype(np.float32)
    n_obs = S - G
    obs = rng.standard_normal(n_obs).astype(np.float32)
    out = []
    for t in range(T):
        obs[-1] = t / 50.0
        s = np.concatenate([obs, goal]).astype(np.float32)
        a = rng.uniform(-1, 1, size=A).astype(np.float32)
        achieved = (achieved + rng.normal(0, 0.02, size=G)).astype(np.float32)
        nxt = rng.standard_normal(n_obs).astype(np.float32)
        nxt[-1] = (t + 1) / 50.0
        ns = np.concatenate([nxt, goal]).astype(np.float32)
        r = float(sparse_goal_reward(achieved, goal))
        out.append((s, a, ns, np.float64(r), False, goal.copy(), achieved.copy()))
        obs = nxt
    return out
