"""Generate tests/golden/*.npz from the REAL reference (run only in the build container, where
/root/reference exists; the reference never travels to the GPU box — only these vectors do).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

What is captured (SURVEY.md §8c fixture set G1-G8):
  her_index_streams.npz  G1/G2  random.randint future-index streams and random.sample index
                                streams (pool and set paths) + MT state words after each
  her_rows.npz           G3/G4  rows the reference's HERBuffer stores for given episodes
                                (T=50, early done, T=1, ring wrap) and a sampled batch
  update_<agent>.npz     G5-G8  per-step: injected batch/noise, returned tuple, pre-/post-clip
                                gradients, parameters / targets / BN stats / log_alpha after
The reference's modules import gymnasium at top level (src/utils.py:5), which is not installed:
an in-memory stub with the three names touched at import time stands in.
compute_reward is panda-gym's (absent): the harness injects oracle.her_oracle.sparse_reward.
"""
import os
import random
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

gym = types.ModuleType("gymnasium")


class _W:
    def __init__(self, env=None):
        self.env = env


gym.Wrapper = _W
gym.ObservationWrapper = _W
gym.vector = types.SimpleNamespace(AsyncVectorEnv=object)
gym.spaces = types.SimpleNamespace(Dict=dict, Box=object)
sys.modules["gymnasium"] = gym
sys.path.insert(0, "/root/reference")

from src.agent import DDPG, SACAgent, TD3Agent, TQCAgent  # noqa: E402  (the reference)
from src.buffer import HERBuffer  # noqa: E402
from src.utils import load_her_config  # noqa: E402

from oracle.her_oracle import sparse_reward, synthetic_episode  # noqa: E402

torch.set_num_threads(1)
CFG_DIR = "/root/reference/src/config"
META = dict(torch=torch.__version__, numpy=np.__version__, python=sys.version.split()[0])


def mt_words():
    return np.array(random.getstate()[1], dtype=np.uint32)


# --------------------------------------------------------------------------- G1 / G2
def gen_index_streams():
    out = {}
    for seed, T, k in [(0, 2, 4), (1898, 17, 4), (1898, 50, 4), (7, 50, 8), (3, 1, 4)]:
        random.seed(seed)
        fut = [random.randint(i + 1, T - 1) for i in range(T - 1) for _ in range(k)]
        out[f"future_s{seed}_T{T}_k{k}"] = np.array(fut, dtype=np.int32)
        out[f"future_s{seed}_T{T}_k{k}_state"] = mt_words()
    for seed, n, b in [(1898, 300, 256), (1898, 1045, 256), (1898, 1046, 256), (5, 100000, 256),
                       (11, 16405, 2048), (11, 20000, 2048), (2, 256, 256), (9, 1000000, 1024), (4, 7, 5)]:
        random.seed(seed)
        idx = random.sample(range(n), b)
        out[f"sample_s{seed}_n{n}_b{b}"] = np.array(idx, dtype=np.int64)
        out[f"sample_s{seed}_n{n}_b{b}_state"] = mt_words()
    # consecutive draws share the stream: two batches back to back, then a randint, then random()
    random.seed(1898)
    a = random.sample(range(5000), 64)
    b = random.sample(range(5000), 64)
    c = random.randint(3, 40)
    d = random.random()
    out["mixed_stream"] = np.array(a + b + [c], dtype=np.int64)
    out["mixed_stream_random"] = np.array([d], dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, "her_index_streams.npz"), **out)


# --------------------------------------------------------------------------- G3 / G4
def episode_arrays(steps):
    s, a, ns, r, d, dg, ag = zip(*steps)
    return dict(s=np.array(s, np.float32), a=np.array(a, np.float32), ns=np.array(ns, np.float32),
                r=np.array(r, np.float64), d=np.array(d, np.bool_), dg=np.array(dg, np.float32),
                ag=np.array(ag, np.float32))


def rows_arrays(buf):
    s, a, ns, r, d, dg, ag = zip(*buf.buffer)
    return dict(s=np.array(s, np.float32), a=np.array(a, np.float32), ns=np.array(ns, np.float32),
                r=np.array([np.float32(x) for x in r], np.float32), d=np.array(d, np.float32))


def gen_her_rows():
    out = {}
    S, A = 10, 3
    cases = {
        # name: (capacity, k, [(env, T, done_at_end)])
        "full50": (100000, 4, [(0, 50, False)]),
        "done12": (100000, 4, [(0, 12, True)]),
        "single": (100000, 4, [(0, 1, True)]),
        "wrap300": (300, 4, [(0, 50, False), (1, 50, False)]),
        "k8_two_envs": (100000, 8, [(1, 50, False), (0, 20, True)]),
        "tiny_cap100": (100, 4, [(0, 50, False)]),
    }
    for name, (cap, k, eps) in cases.items():
        gen = np.random.default_rng(abs(hash(name)) % 2**31 if False else sum(map(ord, name)))
        random.seed(1898)
        buf = HERBuffer(cap, 50, 2, k_future=k)
        buf.device = "cpu"
        buf.compute_reward = sparse_reward
        for e, (env, T, done_end) in enumerate(eps):
            steps = synthetic_episode(gen, T, S, A)
            arr = episode_arrays(steps)
            for key, v in arr.items():
                out[f"{name}_ep{e}_{key}"] = v
            out[f"{name}_ep{e}_env"] = np.array([env])
            for t, (s, a, ns, r, d, dg, ag) in enumerate(steps):
                done = done_end and t == T - 1
                out_done = np.bool_(done)
                buf.push(env, torch.from_numpy(s), a, torch.from_numpy(ns), r, out_done, dg, ag)
            out[f"{name}_ep{e}_done_last"] = np.array([done_end])
        for key, v in rows_arrays(buf).items():
            out[f"{name}_rows_{key}"] = v
        out[f"{name}_cap_k"] = np.array([cap, k])
        out[f"{name}_state_after_push"] = mt_words()
        if len(buf) >= 32:
            batch = buf.sample(32)
            for key, t in zip(("s", "a", "r", "ns", "d"), batch):
                out[f"{name}_batch_{key}"] = t.numpy()
            out[f"{name}_state_after_sample"] = mt_words()
    np.savez_compressed(os.path.join(HERE, "her_rows.npz"), **out)


# --------------------------------------------------------------------------- G5 - G8
def flat(tensors):
    return np.concatenate([t.detach().cpu().numpy().reshape(-1) for t in tensors]).astype(np.float32)


def synthetic_batch(gen, B, S, A):
    s = gen.standard_normal((B, S)).astype(np.float32)
    a = gen.uniform(-1, 1, (B, A)).astype(np.float32)
    ns = (s + 0.1 * gen.standard_normal((B, S))).astype(np.float32)
    r = -(gen.uniform(size=(B, 1)) > 0.3).astype(np.float32)
    d = (gen.uniform(size=(B, 1)) > 0.9).astype(np.float32)
    return s, a, r, ns, d


class Recorder:
    """Snapshots gradients at clip time (pre-clip) and at optimiser-step time (post-clip)."""

    def __init__(self):
        self.pre, self.post = {}, {}
        self.names = {}

    def register(self, name, module, opt):
        self.names[id(next(iter(module.parameters())))] = name
        orig = opt.step
        rec = self

        def step(*a, **k):
            rec.post[name] = flat(p.grad for p in module.parameters())
            rec.pre.setdefault(name, rec.post[name])  # unclipped nets: pre == post
            return orig(*a, **k)

        opt.step = step

    def clip_hook(self, orig):
        rec = self

        def clip(parameters, max_norm, *a, **k):
            params = list(parameters)
            name = rec.names.get(id(params[0]))
            if name is not None:
                rec.pre[name] = flat(p.grad for p in params)
            return orig(params, max_norm, *a, **k)

        return clip

    def reset(self):
        self.pre, self.post = {}, {}


def gen_update(kind, tag, yaml_name, overrides, dims, steps, B, gradient_step=40, store_params=True):
    S, A = dims
    agent_type = kind
    cfg = load_her_config(os.path.join(CFG_DIR, kind, yaml_name), agent_type)
    acfg = cfg.agent.model_copy(update=dict(batch_size=B, **overrides))
    torch.manual_seed(1898); np.random.seed(1898); random.seed(1898)
    cls = dict(DDPG=DDPG, TD3=TD3Agent, SAC=SACAgent, TQC=TQCAgent)[kind]
    agent = cls(obs_dim=S, ac_dim=A, config=acfg, weights=None, nenvs=1, gradient_step=gradient_step)
    if kind == "DDPG":
        nets = dict(actor=agent.actor, target_actor=agent.target_actor, critic_0=agent.critic,
                    target_critic_0=agent.target_critic)
        opts = dict(actor=agent.actor_opt, critic_0=agent.critic_opt)
    elif kind in ("TD3", "SAC"):
        nets = dict(actor=agent.actor, critic_0=agent.critic_1, critic_1=agent.critic_2,
                    target_critic_0=agent.target_critic_1, target_critic_1=agent.target_critic_2)
        if kind == "TD3":
            nets["target_actor"] = agent.target_actor
        opts = dict(actor=agent.actor_opt, critic_0=agent.critic_1_opt, critic_1=agent.critic_2_opt)
    else:
        nets = dict(actor=agent.actor)
        opts = dict(actor=agent.actor_opt)
        for i, (c, t, o) in enumerate(zip(agent.critics, agent.target_critics, agent.critic_opts)):
            nets[f"critic_{i}"] = c; nets[f"target_critic_{i}"] = t; opts[f"critic_{i}"] = o
    # make the weights less trivially symmetric than fresh xavier + 0.01 biases
    gen = np.random.default_rng(1234 + len(tag))
    with torch.no_grad():
        for name, net in nets.items():
            if name.startswith("target"):
                continue
            for p in net.parameters():
                p.add_(torch.from_numpy((0.02 * gen.standard_normal(tuple(p.shape))).astype(np.float32)))
    agent.update_target_network()

    out = dict(meta=np.array([str(META)]), kind=np.array([kind]), dims=np.array([S, A, B, gradient_step]),
               steps=np.array(steps))
    hp = acfg.model_dump()
    out["hparams_keys"] = np.array(list(hp.keys()))
    out["hparams_vals"] = np.array([str(v) for v in hp.values()])
    for name, net in nets.items():
        if store_params or not name.startswith("target"):   # targets start as hard copies
            out[f"init_{name}"] = flat(net.parameters())

    rec = Recorder()
    for name, opt in opts.items():
        rec.register(name, nets[name], opt)
    orig_clip = torch.nn.utils.clip_grad_norm_
    torch.nn.utils.clip_grad_norm_ = rec.clip_hook(orig_clip)
    orig_randn_like = torch.randn_like
    orig_rsample = torch.distributions.Normal.rsample
    queue = []
    torch.randn_like = lambda t, *a, **k: queue.pop(0)
    torch.distributions.Normal.rsample = lambda self, sample_shape=torch.Size(): self.loc + queue.pop(0) * self.scale
    try:
        for i, step in enumerate(steps):
            batch = synthetic_batch(gen, B, S, A)
            tb = tuple(torch.from_numpy(x) for x in batch)
            agent.buffer.sample = lambda bs, tb=tb: tb
            for key, v in zip(("s", "a", "r", "ns", "d"), batch):
                out[f"step{i}_{key}"] = v
            queue.clear()
            if kind == "TD3":
                nz = gen.standard_normal((B, A)).astype(np.float32)
                out[f"step{i}_noise"] = nz
                queue.append(torch.from_numpy(nz))
            if kind in ("SAC", "TQC"):
                e1 = gen.standard_normal((B, A)).astype(np.float32)
                e2 = gen.standard_normal((B, A)).astype(np.float32)
                out[f"step{i}_eps_next"] = e1
                out[f"step{i}_eps_cur"] = e2
                queue += [torch.from_numpy(e1), torch.from_numpy(e2)]
            rec.reset()
            info = agent.update(step=step)
            out[f"step{i}_tuple"] = np.array([float(np.asarray(x)) for x in info], dtype=np.float64)
            for name in opts:
                if name in rec.post:
                    out[f"step{i}_gradpre_{name}"] = rec.pre[name]
                    if store_params:
                        out[f"step{i}_gradpost_{name}"] = rec.post[name]
            if store_params:
                for name, net in nets.items():
                    out[f"step{i}_param_{name}"] = flat(net.parameters())
            if kind in ("SAC", "TQC"):
                out[f"step{i}_log_alpha"] = agent.log_alpha.detach().numpy().copy()
                out[f"step{i}_alpha"] = agent.alpha.detach().numpy().copy()
                bns = [m for m in agent.actor.base_net if isinstance(m, torch.nn.BatchNorm1d)]
                out[f"step{i}_bn_mean"] = np.concatenate([m.running_mean.numpy() for m in bns])
                out[f"step{i}_bn_var"] = np.concatenate([m.running_var.numpy() for m in bns])
        if store_params:
            for name, opt in opts.items():
                st = [opt.state[p] for p in nets[name].parameters()]
                out[f"final_adam_m_{name}"] = flat(s["exp_avg"] for s in st)
                out[f"final_adam_v_{name}"] = flat(s["exp_avg_sq"] for s in st)
    finally:
        torch.nn.utils.clip_grad_norm_ = orig_clip
        torch.randn_like = orig_randn_like
        torch.distributions.Normal.rsample = orig_rsample
    np.savez_compressed(os.path.join(HERE, f"update_{tag}.npz"), **out)
    print(tag, "ok", {k: v.shape for k, v in out.items() if k.endswith("_tuple")})


# --------------------------------------------------------------------------- G9: RunningNormalizer
def gen_normalizer():
    """The reference's RunningNormalizer (src/utils.py:68-98) over a sequence of float32 batches the way the trainer
    feeds it (src/env.py:164-175: [obs ; next_obs] of a vector step), statistics after every update and the normalised
    probe rows (float64, and as the float32 the trainer casts them to)."""
    from src.utils import RunningNormalizer
    gen = np.random.default_rng(77)
    D = 7
    nz = RunningNormalizer(D)
    out = dict(D=np.array([D]))
    sizes = [16, 16, 2, 128, 33, 16, 1000, 5]
    out["sizes"] = np.array(sizes)
    probe = (gen.standard_normal((9, D)) * np.array([1, 10, 0.1, 3, 1, 50, 1e-3]) + np.array([0, 5, -2, 0, 100, 0, 0])).astype(np.float32)
    out["probe"] = probe
    for i, n in enumerate(sizes):
        x = (gen.standard_normal((n, D)) * np.array([1, 10, 0.1, 3, 1, 50, 1e-3]) + np.array([0, 5, -2, 0, 100, 0, 0])).astype(np.float32)
        out[f"x{i}"] = x
        nz.update(x)
        out[f"mean{i}"], out[f"var{i}"], out[f"count{i}"] = nz.mean.copy(), nz.var.copy(), np.array([nz.count])
        z = nz.normalize(probe)
        out[f"norm64_{i}"] = z
        out[f"norm32_{i}"] = torch.from_numpy(z).float().numpy()
    np.savez_compressed(os.path.join(HERE, "normalizer.npz"), **out)
    print("normalizer ok")


# --------------------------------------------------------------------------- G10: PER / Replay buffers (src/buffer.py:8-89)
def gen_per(kind, tag, yaml_name):
    """The reference agent with buffer_type="PER": transitions pushed through agent.push, three update() steps drawing
    through np.random.choice over the priorities (global numpy stream, seeded), importance-sampling weights in the critic
    loss, priorities updated from the per-sample td_error.  Captured: the pushed rows, drawn indices, weights, returned
    tuples (td_error is a [B,1] array on this path), pre-clip gradients, priorities after every step."""
    S, A, B, N = 10, 3, 32, 200
    cfg = load_her_config(os.path.join(CFG_DIR, kind, yaml_name), kind)
    acfg = cfg.agent.model_copy(update=dict(batch_size=B, hidden_dim=32, layer_count=2, buffer_type="PER", max_len=150, alpha=0.6,
                                            beta=0.4, beta_end=1000, ac_update_freq=1))
    torch.manual_seed(1898); np.random.seed(1898); random.seed(1898)
    cls = dict(DDPG=DDPG, TD3=TD3Agent)[kind]
    agent = cls(obs_dim=S, ac_dim=A, config=acfg, weights=None, nenvs=1, gradient_step=40)
    agent.buffer.device = "cpu"
    if kind == "DDPG":
        nets = dict(actor=agent.actor, critic_0=agent.critic)
        opts = dict(actor=agent.actor_opt, critic_0=agent.critic_opt)
    else:
        nets = dict(actor=agent.actor, critic_0=agent.critic_1, critic_1=agent.critic_2)
        opts = dict(actor=agent.actor_opt, critic_0=agent.critic_1_opt, critic_1=agent.critic_2_opt)
    gen = np.random.default_rng(55)
    with torch.no_grad():
        for net in nets.values():
            for p in net.parameters():
                p.add_(torch.from_numpy((0.02 * gen.standard_normal(tuple(p.shape))).astype(np.float32)))
    agent.update_target_network()
    out = dict(kind=np.array([kind]), dims=np.array([S, A, B, N]))
    hp = acfg.model_dump()
    out["hparams_keys"] = np.array(list(hp.keys()))
    out["hparams_vals"] = np.array([str(v) for v in hp.values()])
    for name, net in nets.items():
        out[f"init_{name}"] = flat(net.parameters())
    rows = synthetic_batch(gen, N, S, A)            # (s, a, r, ns, d); 200 pushes into a 150-row deque: evicts
    for key, v in zip(("s", "a", "r", "ns", "d"), rows):
        out[f"rows_{key}"] = v
    for i in range(N):
        agent.push(torch.from_numpy(rows[0][i]), rows[1][i], float(rows[2][i, 0]), torch.from_numpy(rows[3][i]), bool(rows[4][i, 0]))
    rec = Recorder()
    for name, opt in opts.items():
        rec.register(name, nets[name], opt)
    orig_clip, orig_choice, orig_rl = torch.nn.utils.clip_grad_norm_, np.random.choice, torch.randn_like
    torch.nn.utils.clip_grad_norm_ = rec.clip_hook(orig_clip)
    drawn = []
    np.random.choice = lambda *a, **k: drawn.append(orig_choice(*a, **k)) or drawn[-1]
    torch.randn_like = lambda t, *a, **k: torch.zeros_like(t)       # TD3 smoothing noise off (policy_noise scales it anyway)
    np.random.seed(4242)
    try:
        for i, step in enumerate((1, 2, 3)):
            rec.reset()
            info = agent.update(step=step)
            out[f"step{i}_indices"] = np.asarray(drawn[-1], dtype=np.int64)
            td_pos = {6: 2, 8: 3}[len(info)]
            out[f"step{i}_td"] = np.asarray(info[td_pos], dtype=np.float32)
            out[f"step{i}_tuple"] = np.array([float(np.asarray(x)) if j != td_pos else float(np.mean(info[td_pos])) for j, x in enumerate(info)])
            out[f"step{i}_priorities"] = np.array(agent.buffer.priorities, dtype=np.float64)
            out[f"step{i}_beta"] = np.array([agent.beta])
            for name in opts:
                if name in rec.pre:
                    out[f"step{i}_gradpre_{name}"] = rec.pre[name]
            for name, net in nets.items():
                out[f"step{i}_param_{name}"] = flat(net.parameters())
    finally:
        torch.nn.utils.clip_grad_norm_, np.random.choice, torch.randn_like = orig_clip, orig_choice, orig_rl
    np.savez_compressed(os.path.join(HERE, f"per_{tag}.npz"), **out)
    print("per", tag, "ok", out["step0_tuple"])


def main():
    gen_per("DDPG", "ddpg", "config_ddpg_reach.yaml")
    gen_per("TD3", "td3", "config_td3_reach.yaml")
    gen_normalizer()
    gen_index_streams()
    gen_her_rows()
    # DDPG with the reference's Reach config (H=64, L=3); step 40 exercises the Polyak cadence
    gen_update("DDPG", "ddpg_reach", "config_ddpg_reach.yaml", {}, (10, 3), [1, 2, 3, 40], B=64)
    # cosine schedule actually moving: lr_min != lr, T_max = 3
    gen_update("DDPG", "ddpg_cosine", "config_ddpg_reach.yaml",
               dict(hidden_dim=32, layer_count=2, actor_lr_min=1e-4, critic_lr_min=2e-4, ac_scheduler_steps=3,
                    cr_scheduler_steps=3, grad_clip=0.05), (10, 3), [1, 2, 3, 4, 5, 6, 7], B=32)
    # headline shape: PickAndPlace dims, H=256, B=256 (one step, gradients only)
    gen_update("DDPG", "ddpg_pickplace_h256", "config_ddpg_pickplace.yaml", {}, (23, 4), [1], B=256,
               store_params=False)
    gen_update("TD3", "td3", "config_td3_pickplace.yaml", dict(hidden_dim=32, layer_count=2), (23, 4), [1, 2, 3, 4], B=64)
    gen_update("SAC", "sac", "config_sac_slide.yaml", dict(hidden_dim=32, layer_count=2, alpha_min_steps=1.0),
               (22, 3), [1, 2, 3, 4], B=64, gradient_step=2)
    gen_update("TQC", "tqc", "config_tqc_push.yaml", dict(hidden_dim=32, layer_count=2), (22, 3), [1, 2, 3], B=64)


if __name__ == "__main__":
    main()
