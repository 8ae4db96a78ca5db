#!/usr/bin/env python3
"""Where one vector-env step of the stand-in trainer spends its time (host wall clock per call, 8 envs)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "examples"))
import gcrl_amd
from gcrl_amd.src.utils import DeviceRunningNormalizer
from gcrl_amd.src.synthetic import agent_config
from trainer_standin import PointReachVecEnv

n = 8
env = PointReachVecEnv(n, seed=0)
cfg = agent_config("DDPG", hidden_dim=64, layer_count=3, batch_size=256, max_len=200_000)
ag = gcrl_amd.DDPG(env.obs_dim + env.goal_dim, env.ac_dim, cfg, None, nenvs=n, gradient_step=40, rng="engine", seed=0)
ag.buffer.obs_normalizer = DeviceRunningNormalizer(env.obs_dim)
ag.buffer.dg_normalizer = DeviceRunningNormalizer(env.goal_dim)
ag.buffer.compute_reward = env.compute_reward
state, _ = env.reset()
t = dict(act=0.0, env=0.0, proc=0.0)
N = 3000
for i in range(N + 200):
    if i == 200:
        t = dict(act=0.0, env=0.0, proc=0.0)
    t0 = time.perf_counter()
    a = np.asarray(ag.observe_act(state["observation"], state["desired_goal"]), dtype=np.float32)
    t1 = time.perf_counter()
    nxt, r, term, trunc, _ = env.step(a)
    t2 = time.perf_counter()
    ag.process_step(state, a, nxt, r, term)
    t3 = time.perf_counter()
    t["act"] += t1 - t0; t["env"] += t2 - t1; t["proc"] += t3 - t2
    if trunc.any():
        env._reset(np.nonzero(trunc)[0]); nxt = env._obs()
    state = nxt
print({k: round(1e6 * v / N, 1) for k, v in t.items()}, "us per vector step of", n, "envs")
