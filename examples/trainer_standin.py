#!/usr/bin/env python3
"""Stand-in for the reference trainer's HER loop: the calls `GoalEnvHER._train_her` makes
(reference src/env.py:334-406, `_process_step` :163-201, `_push_to_buffer` :203-232), in the same
order, against a synthetic vectorised goal environment — panda-gym is not installable here, and the
trainer itself is outside the hot path.  What it exercises end to end:

    normalize_state_batch -> select_action -> env.step -> update_normalizers
        -> HERBuffer.push_batch (all envs of a step in one call; episode flush + HER relabel on device)
        -> every `max_episode` episodes: agent.update_many(gradient_step)

and what it reports: success rate per cycle (does it learn?), env steps/s, gradient steps/s.

    python examples/trainer_standin.py --agent DDPG --cycles 60
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class PointReachVecEnv:
    """`num_envs` point masses in a box; dict observations shaped like panda-gym's Reach task with the
    reference's time feature appended (src/utils.py:137-174): observation = [pos, vel, t/T]."""

    def __init__(self, num_envs, dim=3, max_steps=50, threshold=0.05, seed=0):
        self.n, self.dim, self.T, self.thr = num_envs, dim, max_steps, threshold
        self.gen = np.random.default_rng(seed)
        self.obs_dim, self.goal_dim, self.ac_dim = 2 * dim + 1, dim, dim
        self.pos = np.zeros((num_envs, dim)); self.vel = np.zeros((num_envs, dim))
        self.goal = np.zeros((num_envs, dim)); self.t = np.zeros(num_envs, dtype=np.int64)

    def compute_reward(self, achieved_goal, desired_goal, info):
        d = np.linalg.norm(np.asarray(achieved_goal) - np.asarray(desired_goal), axis=-1)
        return -(d > self.thr).astype(np.float32)

    def _reset(self, idx):
        k = len(idx)
        self.pos[idx] = self.gen.uniform(-0.15, 0.15, (k, self.dim))
        self.vel[idx] = 0.0
        self.goal[idx] = self.gen.uniform(-0.15, 0.15, (k, self.dim))
        self.t[idx] = 0

    def _obs(self):
        o = np.concatenate([self.pos, self.vel, (self.t / self.T)[:, None]], axis=1).astype(np.float32)
        return {"observation": o, "achieved_goal": self.pos.astype(np.float32).copy(),
                "desired_goal": self.goal.astype(np.float32).copy()}

    def reset(self):
        self._reset(np.arange(self.n))
        return self._obs(), {}

    def step(self, actions):
        a = np.clip(np.asarray(actions, dtype=np.float64), -1, 1)
        self.vel = 0.04 * a                      # position control: the displacement of a step is the action
        self.pos = np.clip(self.pos + self.vel, -0.3, 0.3)
        self.t += 1
        obs = self._obs()
        rewards = self.compute_reward(obs["achieved_goal"], obs["desired_goal"], None)
        terminated = np.zeros(self.n, dtype=bool)
        truncated = self.t >= self.T
        return obs, rewards, terminated, truncated, {}


def train(agent_name="DDPG", num_envs=8, cycles=60, max_episode=8, gradient_step=40, hidden=64, layers=3, batch=256,
          seed=0, verbose=True, per_env_push=False, fused=False):
    import gcrl_amd
    from gcrl_amd.src.utils import DeviceRunningNormalizer, RunningNormalizer
    if fused:   # the normalisers live on the device; acting and _process_step are one native call each per vector step
        RunningNormalizer = DeviceRunningNormalizer
    from gcrl_amd.src.synthetic import agent_config as make_config   # hyper-parameter container with the YAML field names

    np.random.seed(seed)
    env = PointReachVecEnv(num_envs, seed=seed)
    cfg = make_config(agent_name, hidden_dim=hidden, layer_count=layers, batch_size=batch, max_len=200_000, k_future=4,
                      gamma=0.95, tau=0.05, grad_clip=10.0, ac_update_freq=2 if agent_name == "TD3" else 1,
                      policy_noise=0.2 if agent_name == "TD3" else 0.0)
    cls = dict(DDPG=gcrl_amd.DDPG, TD3=gcrl_amd.TD3Agent, SAC=gcrl_amd.SACAgent, TQC=gcrl_amd.TQCAgent)[agent_name]
    agent = cls(env.obs_dim + env.goal_dim, env.ac_dim, cfg, None, nenvs=num_envs, gradient_step=gradient_step,
                rng="engine", seed=seed)
    # what GoalEnvHER.__init__ injects (src/env.py:93-105)
    agent.buffer.obs_normalizer = RunningNormalizer(env.obs_dim)
    agent.buffer.dg_normalizer = RunningNormalizer(env.goal_dim)
    agent.buffer.compute_reward = env.compute_reward

    state, _ = env.reset()
    grad_counter, env_steps = 1, 0
    success_per_cycle, final_success = [], []
    t_env = t_upd = 0.0
    t0 = time.perf_counter()
    for cycle in range(1, cycles + 1):
        episode_count = 0
        tc = time.perf_counter()
        while episode_count < max_episode:
            if fused:
                actions = np.asarray(agent.observe_act(state["observation"], state["desired_goal"]), dtype=np.float32)
            else:
                state_input = agent.normalize_state_batch(state["observation"], state["desired_goal"], True, False)
                actions = np.asarray(agent.select_action(state_input), dtype=np.float32)
            next_obs, rewards, terminateds, truncateds, _ = env.step(actions)
            dones = np.logical_or(terminateds, truncateds)
            if fused:
                agent.process_step(state, actions, next_obs, rewards, terminateds)
                env_steps += num_envs
                if dones.any():
                    idx = np.nonzero(dones)[0]
                    d = np.linalg.norm(next_obs["achieved_goal"][idx] - next_obs["desired_goal"][idx], axis=1)
                    final_success.extend((d < env.thr).tolist())
                    episode_count += len(idx)
                    env._reset(idx)
                    next_obs = env._obs()
                state = next_obs
                continue
            # _process_step: normaliser update, normalised [obs | goal] rows, one push for all envs
            agent.update_normalizers([state["observation"], next_obs["observation"]],
                                     [state["desired_goal"], next_obs["desired_goal"], state["achieved_goal"],
                                      next_obs["achieved_goal"]], True, False)
            obs_b = torch.from_numpy(agent.normalize_state_batch(state["observation"], state["desired_goal"], True, False)).float().cuda()
            nxt_b = torch.from_numpy(agent.normalize_state_batch(next_obs["observation"], next_obs["desired_goal"], True, False)).float().cuda()
            if per_env_push:   # the reference's own loop (src/env.py:192-201): one push_her per env
                for i in range(num_envs):
                    agent.push_her(i, obs_b[i].detach(), actions[i], nxt_b[i].detach(), rewards[i], terminateds[i],
                                   agent.normalize_goal(next_obs["desired_goal"][i], False),
                                   agent.normalize_goal(next_obs["achieved_goal"][i], False))
            else:
                agent.buffer.push_batch(obs_b, actions, nxt_b, rewards, terminateds, next_obs["achieved_goal"])
            env_steps += num_envs
            if dones.any():
                idx = np.nonzero(dones)[0]
                d = np.linalg.norm(next_obs["achieved_goal"][idx] - next_obs["desired_goal"][idx], axis=1)
                final_success.extend((d < env.thr).tolist())
                episode_count += len(idx)
                env._reset(idx)
                next_obs = env._obs()
            state = next_obs
        t_env += time.perf_counter() - tc
        tu = time.perf_counter()
        if agent.is_buffer_filled():
            infos = agent.update_many(grad_counter, gradient_step)
            grad_counter += gradient_step
            float(infos[-1][0])   # reads one metric: waits for the cycle's updates
        t_upd += time.perf_counter() - tu
        recent = final_success[-max_episode:]
        success_per_cycle.append(float(np.mean(recent)))
        if verbose and cycle % 10 == 0:
            print(f"cycle {cycle:4d}  success(last {len(recent)} episodes) {success_per_cycle[-1]:.2f}  buffer {len(agent.buffer)}")
    wall = time.perf_counter() - t0
    return dict(success_per_cycle=success_per_cycle, env_steps=env_steps, gradient_steps=grad_counter - 1, wall_s=wall,
                env_steps_per_s=env_steps / max(t_env, 1e-9), gradient_steps_per_s=(grad_counter - 1) / max(t_upd, 1e-9))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--agent", default="DDPG", choices=["DDPG", "TD3", "SAC", "TQC"])
    ap.add_argument("--cycles", type=int, default=60)
    ap.add_argument("--nenv", type=int, default=8)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--per-env-push", action="store_true", help="push one env at a time, exactly as the reference's trainer does")
    ap.add_argument("--fused", action="store_true", help="device normalisers + one native call for acting and one for _process_step")
    args = ap.parse_args()
    out = train(args.agent, num_envs=args.nenv, cycles=args.cycles, seed=args.seed, per_env_push=args.per_env_push, fused=args.fused)
    tail = out["success_per_cycle"][-10:]
    print(f"{args.agent}: success over the last 10 cycles {np.mean(tail):.2f}; {out['env_steps']} env steps "
          f"({out['env_steps_per_s']:.0f}/s in the acting phase), {out['gradient_steps']} gradient steps "
          f"({out['gradient_steps_per_s']:.0f}/s in the update phase), {out['wall_s']:.1f} s")
