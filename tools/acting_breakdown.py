#!/usr/bin/env python3
"""Where one vector-env step of the stand-in trainer spends its HOST time (wall clock per call, 8 envs), split into the Python
wrapper (array marshalling, RNG draws, ctypes argument conversion) and the native call (staging, launches, the wait for the
actions).  GCRL_ACT_STAGED=1 / GCRL_PROC_STAGED=1 select the round-3 forms (staged copies + stream synchronisation) for an A/B.
Writes gpurun_out/acting_breakdown.json."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "examples"))
import gcrl_amd  # noqa: E402
from gcrl_amd import _ffi  # noqa: E402
from gcrl_amd.src.utils import DeviceRunningNormalizer  # noqa: E402
from gcrl_amd.src.synthetic import agent_config  # noqa: E402
from trainer_standin import PointReachVecEnv  # noqa: E402


class Timed:
    """a ctypes function with a clock around it"""

    def __init__(self, fn):
        self.fn, self.t, self.n = fn, 0.0, 0

    def __call__(self, *a):
        t0 = time.perf_counter()
        r = self.fn(*a)
        self.t += time.perf_counter() - t0
        self.n += 1
        return r


def run(kind, n=8, N=3000):
    env = PointReachVecEnv(n, seed=0)
    cfg = agent_config(kind, hidden_dim=64, layer_count=3, batch_size=256, max_len=200_000)
    cls = dict(DDPG=gcrl_amd.DDPG, TD3=gcrl_amd.TD3Agent)[kind]
    ag = cls(env.obs_dim + env.goal_dim, env.ac_dim, cfg, None, nenvs=n, gradient_step=40, rng="engine", seed=0)
    ag.buffer.obs_normalizer = DeviceRunningNormalizer(env.obs_dim)
    ag.buffer.dg_normalizer = DeviceRunningNormalizer(env.goal_dim)
    ag.buffer.compute_reward = env.compute_reward
    lib = _ffi.lib
    act_c, proc_c = Timed(lib.gcrl_agent_observe_act), Timed(lib.gcrl_her_process_step_g)
    import gcrl_amd.src.agent as agent_mod
    state, _ = env.reset()
    t = dict(act=0.0, env=0.0, proc=0.0)
    for i in range(N + 200):
        if i == 200:
            t = dict(act=0.0, env=0.0, proc=0.0)
            act_c.t = proc_c.t = 0.0
            lib.gcrl_agent_observe_act, lib.gcrl_her_process_step_g = act_c, proc_c
            agent_mod.lib = lib
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
    us = lambda x: round(1e6 * x / N, 2)
    return dict(agent=kind, envs=n, vector_steps=N,
                observe_act=dict(total_us=us(t["act"]), native_call_us=us(act_c.t), python_wrapper_us=us(t["act"] - act_c.t),
                                 note="native call = argument staging + ONE launch (rows and noise inside the kernel arguments) + the wait for the "
                                      "per-workgroup flags in host-visible memory; the wrapper draws np.random noise and converts arrays"),
                process_step=dict(total_us=us(t["proc"]), native_call_us=us(proc_c.t), python_wrapper_us=us(t["proc"] - proc_c.t),
                                  note="native call = payload packing + ONE launch (data inside the kernel arguments), nothing waited for; "
                                       "includes the flush launch on the steps that end an episode"),
                env_step_us=us(t["env"]),
                env_steps_per_s_acting_only=round(n * N / (t["act"] + t["env"] + t["proc"]), 1),
                forms=dict(act="staged" if os.environ.get("GCRL_ACT_STAGED") else "inline", proc="staged" if os.environ.get("GCRL_PROC_STAGED") else "inline"))


if __name__ == "__main__":
    out = [run("DDPG")]
    print(json.dumps(out, indent=1))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    tag = "_staged" if os.environ.get("GCRL_ACT_STAGED") else ""
    with open(os.path.join(ROOT, "gpurun_out", f"acting_breakdown{tag}.json"), "w") as f:
        json.dump(out, f, indent=1)
