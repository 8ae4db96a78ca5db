"""Generate tests/golden/select_action.npz from the REAL reference's select_action methods (run only in the build
container, where /root/reference exists):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_act.py

For DDPG, TD3Agent, SACAgent, TQCAgent (src/agent.py:1345-1366, :253-270, :641-647, :1044-1050): the three host generators
(`random`, `np.random`, torch's) are seeded, then a sequence of calls with varying row counts is made — exploring and
evaluating, DDPG's epsilon-random branch included (it draws `random.random()` from the stream HER shares).  Captured: the
actor's parameters (and BatchNorm running statistics, moved away from 0 / 1), every call's input rows, flag and returned
array (with its dtype), and the state of all three generators after the sequence — so a replacement is held to the
reference's VALUES and to its CONSUMPTION of every stream.
"""
import os
import random
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (stubs gymnasium, imports the reference, pins torch to one thread)

SEED = 777
N_CALLS = 28


def np_state_arrays():
    name, keys, pos, has_gauss, cached = np.random.get_state()
    assert name == "MT19937"
    return np.asarray(keys, np.uint32), np.array([pos, has_gauss], np.int64), np.array([cached], np.float64)


def gen_select_action():
    out = dict(meta=np.array([str(mg.META)]), seed=np.array([SEED]), n_calls=np.array([N_CALLS]))
    H, L = 32, 2
    for kind, yaml_name, (S, A) in [("DDPG", "config_ddpg_reach.yaml", (10, 3)), ("TD3", "config_td3_pickplace.yaml", (23, 4)),
                                    ("SAC", "config_sac_slide.yaml", (22, 3)), ("TQC", "config_tqc_push.yaml", (22, 3))]:
        cfg = mg.load_her_config(os.path.join(mg.CFG_DIR, kind, yaml_name), kind)
        acfg = cfg.agent.model_copy(update=dict(batch_size=64, hidden_dim=H, layer_count=L))
        torch.manual_seed(1898); np.random.seed(1898); random.seed(1898)
        cls = dict(DDPG=mg.DDPG, TD3=mg.TD3Agent, SAC=mg.SACAgent, TQC=mg.TQCAgent)[kind]
        agent = cls(obs_dim=S, ac_dim=A, config=acfg, weights=None, nenvs=1, gradient_step=40)
        gen = np.random.default_rng(31 + len(kind) + S)
        with torch.no_grad():
            for p in agent.actor.parameters():
                p.add_(torch.from_numpy((0.05 * gen.standard_normal(tuple(p.shape))).astype(np.float32)))
            if kind in ("SAC", "TQC"):
                bns = [m for m in agent.actor.base_net if isinstance(m, torch.nn.BatchNorm1d)]
                for m in bns:
                    m.running_mean.copy_(torch.from_numpy((0.3 * gen.standard_normal(H)).astype(np.float32)))
                    m.running_var.copy_(torch.from_numpy(gen.uniform(0.5, 2.0, H).astype(np.float32)))
                out[f"{kind}_bn_mean"] = np.concatenate([m.running_mean.numpy() for m in bns])
                out[f"{kind}_bn_var"] = np.concatenate([m.running_var.numpy() for m in bns])
        out[f"{kind}_dims"] = np.array([S, A, H, L])
        out[f"{kind}_noise_std"] = np.array([acfg.noise_std], np.float64)
        out[f"{kind}_actor"] = mg.flat(agent.actor.parameters())
        random.seed(SEED); np.random.seed(SEED); torch.manual_seed(SEED)
        n_eps_branch = 0
        for i in range(N_CALLS):
            n = (1, 4, 8, 3)[i % 4]
            ev = (i % 5 == 4)
            obs = gen.standard_normal((n, S)).astype(np.float32)
            before = random.getstate()[1]
            act = agent.select_action(obs, eval_action=ev)
            if kind == "DDPG" and not ev and random.getstate()[1] != before:
                pass
            assert isinstance(act, np.ndarray) and act.shape == (n, A)
            out[f"{kind}_obs{i}"] = obs
            out[f"{kind}_eval{i}"] = np.array([ev])
            out[f"{kind}_act{i}"] = act
            out[f"{kind}_dtype{i}"] = np.array([str(act.dtype)])
        out[f"{kind}_py_state"] = mg.mt_words()
        k, p, c = np_state_arrays()
        out[f"{kind}_np_keys"], out[f"{kind}_np_pos"], out[f"{kind}_np_cached"] = k, p, c
        out[f"{kind}_torch_state"] = torch.get_rng_state().numpy().copy()
        # how many of DDPG's exploring calls took the epsilon branch (a replay of the stream: src/agent.py:1348)
        if kind == "DDPG":
            random.seed(SEED)
            n_eps_branch = sum(1 for i in range(N_CALLS) if i % 5 != 4 and random.random() < 0.2)
            out["DDPG_eps_branches"] = np.array([n_eps_branch])
        print(kind, "ok", out[f"{kind}_act0"].dtype, out[f"{kind}_act4"].dtype, n_eps_branch)
    # the property the engine's host-side eps draw leans on: rsample's eps on a CPU module == torch.randn of the same shape
    torch.manual_seed(5)
    a = torch.distributions.Normal(torch.zeros(7, 3), torch.ones(7, 3)).rsample()
    torch.manual_seed(5)
    assert torch.equal(a, torch.randn(7, 3))
    np.savez_compressed(os.path.join(HERE, "select_action.npz"), **out)


def gen_per_stochastic(kind, tag, yaml_name):
    """The reference's SACAgent / TQCAgent around a PERBuffer (importance-sampling branches src/agent.py:577-581, :993-997):
    like make_golden.gen_per, with the reparameterisation noise of actor.sample recorded (Normal.rsample patched to
    loc + eps * scale with eps from a seeded generator), BatchNorm running statistics and log_alpha after every step."""
    S, A, B, N = 10, 3, 32, 200
    cfg = mg.load_her_config(os.path.join(mg.CFG_DIR, kind, yaml_name), kind)
    acfg = cfg.agent.model_copy(update=dict(batch_size=B, hidden_dim=32, layer_count=2, buffer_type="PER", max_len=150, alpha=0.6,
                                            beta=0.4, beta_end=1000, ac_update_freq=1, alpha_min_steps=1.0))
    torch.manual_seed(1898); np.random.seed(1898); random.seed(1898)
    cls = dict(SAC=mg.SACAgent, TQC=mg.TQCAgent)[kind]
    agent = cls(obs_dim=S, ac_dim=A, config=acfg, weights=None, nenvs=1, gradient_step=2)
    agent.buffer.device = "cpu"
    if kind == "SAC":
        crit, opts_c = [agent.critic_1, agent.critic_2], [agent.critic_1_opt, agent.critic_2_opt]
    else:
        crit, opts_c = list(agent.critics), list(agent.critic_opts)
    nets = dict(actor=agent.actor, **{f"critic_{i}": c for i, c in enumerate(crit)})
    opts = dict(actor=agent.actor_opt, **{f"critic_{i}": o for i, o in enumerate(opts_c)})
    gen = np.random.default_rng(91 + len(kind))
    with torch.no_grad():
        for net in nets.values():
            for p in net.parameters():
                p.add_(torch.from_numpy((0.02 * gen.standard_normal(tuple(p.shape))).astype(np.float32)))
        # trained-policy-like heads (std ~ 0.4): fresh Xavier heads put |pre-tanh| past 4, where log(1 - tanh^2 + 1e-8) has no
        # fp32 precision left in the reference itself
        agent.actor.mean_head.weight.mul_(0.2); agent.actor.log_std_head.weight.mul_(0.1); agent.actor.log_std_head.bias.fill_(-0.9)
    agent.update_target_network() if hasattr(agent, "update_target_network") else None
    out = dict(kind=np.array([kind]), dims=np.array([S, A, B, N]))
    hp = acfg.model_dump()
    out["hparams_keys"] = np.array(list(hp.keys()))
    out["hparams_vals"] = np.array([str(v) for v in hp.values()])
    for name, net in nets.items():
        out[f"init_{name}"] = mg.flat(net.parameters())
    rows = mg.synthetic_batch(gen, N, S, A)
    for key, v in zip(("s", "a", "r", "ns", "d"), rows):
        out[f"rows_{key}"] = v
    for i in range(N):
        agent.push(torch.from_numpy(rows[0][i]), rows[1][i], float(rows[2][i, 0]), torch.from_numpy(rows[3][i]), bool(rows[4][i, 0]))
    rec = mg.Recorder()
    for name, opt in opts.items():
        rec.register(name, nets[name], opt)
    orig_clip, orig_choice, orig_rs = torch.nn.utils.clip_grad_norm_, np.random.choice, torch.distributions.Normal.rsample
    torch.nn.utils.clip_grad_norm_ = rec.clip_hook(orig_clip)
    drawn, queue = [], []
    np.random.choice = lambda *a, **k: drawn.append(orig_choice(*a, **k)) or drawn[-1]
    torch.distributions.Normal.rsample = lambda self, sample_shape=torch.Size(): self.loc + queue.pop(0) * self.scale
    np.random.seed(4242)
    try:
        for i, step in enumerate((1, 2, 3, 4)):
            e1 = gen.standard_normal((B, A)).astype(np.float32)
            e2 = gen.standard_normal((B, A)).astype(np.float32)
            out[f"step{i}_eps_next"], out[f"step{i}_eps_cur"] = e1, e2
            queue[:] = [torch.from_numpy(e1), torch.from_numpy(e2)]
            rec.reset()
            info = agent.update(step=step)
            assert len(info) == 9 and not queue
            out[f"step{i}_indices"] = np.asarray(drawn[-1], dtype=np.int64)
            out[f"step{i}_td"] = np.asarray(info[3], dtype=np.float32)
            out[f"step{i}_tuple"] = np.array([float(np.asarray(x)) if j != 3 else float(np.mean(info[3])) for j, x in enumerate(info)])
            out[f"step{i}_priorities"] = np.array(agent.buffer.priorities, dtype=np.float64)
            out[f"step{i}_beta"] = np.array([agent.beta])
            for name in opts:
                if name in rec.pre:
                    out[f"step{i}_gradpre_{name}"] = rec.pre[name]
            for name, net in nets.items():
                out[f"step{i}_param_{name}"] = mg.flat(net.parameters())
            out[f"step{i}_log_alpha"] = agent.log_alpha.detach().numpy().copy()
            bns = [m for m in agent.actor.base_net if isinstance(m, torch.nn.BatchNorm1d)]
            out[f"step{i}_bn_mean"] = np.concatenate([m.running_mean.numpy() for m in bns])
            out[f"step{i}_bn_var"] = np.concatenate([m.running_var.numpy() for m in bns])
    finally:
        torch.nn.utils.clip_grad_norm_, np.random.choice, torch.distributions.Normal.rsample = orig_clip, orig_choice, orig_rs
    np.savez_compressed(os.path.join(HERE, f"per_{tag}.npz"), **out)
    print("per", tag, "ok", out["step0_tuple"], out["step3_tuple"][-1])


def gen_update_fp64(tag):
    """tests/golden/update_<tag>.npz replayed through the reference in FLOAT64 (`.double()` networks, the same inputs, the same
    recorded noise; after every step the state is put back to the fp32 run's, as the parity tests do): returned tuple and
    pre-clip gradients per step.  Lets the GPU test hold SAC / TQC to |hip - ref64| <= max(3 |ref32 - ref64|, 1e-5 scale) — the
    criterion of the full-size fixtures (tests/fullsize.py) — instead of a flat relaxed tolerance: on the fresh-Xavier SAC
    fixture the reference's own fp32 run is what sits 1e-5..1e-3 away from its fp64 run (log(1 - tanh^2 + 1e-8) at |x| > 4)."""
    from src.utils import BaseAgentConfig, SACAgentConfig
    g = np.load(os.path.join(HERE, f"update_{tag}.npz"))
    kind = str(g["kind"][0])
    S, A, B, gstep = (int(x) for x in g["dims"])
    Model = SACAgentConfig if kind in ("SAC", "TQC") else BaseAgentConfig
    hp = {}
    for k, v in zip(g["hparams_keys"], g["hparams_vals"]):
        k, v = str(k), str(v)
        ann = Model.model_fields[k].annotation
        hp[k] = v if ann is str else (int(float(v)) if ann is int else float(v))
    acfg = Model(**hp)
    torch.manual_seed(1898); np.random.seed(1898); random.seed(1898)
    cls = dict(DDPG=mg.DDPG, TD3=mg.TD3Agent, SAC=mg.SACAgent, TQC=mg.TQCAgent)[kind]
    agent = cls(obs_dim=S, ac_dim=A, config=acfg, weights=None, nenvs=1, gradient_step=gstep)
    if kind == "SAC":
        nets = dict(actor=agent.actor, critic_0=agent.critic_1, critic_1=agent.critic_2, target_critic_0=agent.target_critic_1,
                    target_critic_1=agent.target_critic_2)
        opts = dict(actor=agent.actor_opt, critic_0=agent.critic_1_opt, critic_1=agent.critic_2_opt)
    else:
        nets, opts = dict(actor=agent.actor), dict(actor=agent.actor_opt)
        for i, (c, t, o) in enumerate(zip(agent.critics, agent.target_critics, agent.critic_opts)):
            nets[f"critic_{i}"] = c; nets[f"target_critic_{i}"] = t; opts[f"critic_{i}"] = o

    def set_flat(net, vec):
        off = 0
        with torch.no_grad():
            for p in net.parameters():
                n = p.numel()
                p.copy_(torch.from_numpy(np.asarray(vec[off:off + n], np.float64)).view_as(p))
                off += n

    for net in nets.values():
        net.double()
    agent.log_alpha.data = agent.log_alpha.data.double()
    agent.alpha = agent.log_alpha.exp()
    for name, net in nets.items():
        key = f"init_{name}"
        set_flat(net, g[key] if key in g.files else g[f"init_{name.replace('target_', '')}"])
    bns = [m for m in agent.actor.base_net if isinstance(m, torch.nn.BatchNorm1d)]
    pre = {}
    orig_clip, orig_rs = torch.nn.utils.clip_grad_norm_, torch.distributions.Normal.rsample
    ids = {id(next(iter(nets[n].parameters()))): n for n in opts}

    def clip(parameters, max_norm, *a, **k):
        params = list(parameters)
        name = ids.get(id(params[0]))
        if name is not None:
            pre[name] = np.concatenate([p.grad.detach().numpy().reshape(-1).astype(np.float64) for p in params])
        return orig_clip(params, max_norm, *a, **k)

    queue = []
    torch.nn.utils.clip_grad_norm_ = clip
    torch.distributions.Normal.rsample = lambda self, sample_shape=torch.Size(): self.loc + queue.pop(0) * self.scale
    out = dict(kind=np.array([kind]), steps=g["steps"])
    try:
        for i, step in enumerate(g["steps"]):
            tb = tuple(torch.from_numpy(g[f"step{i}_{k}"].astype(np.float64)) for k in ("s", "a", "r", "ns", "d"))
            agent.buffer.sample = lambda bs, tb=tb: tb
            queue[:] = [torch.from_numpy(g[f"step{i}_eps_next"].astype(np.float64)), torch.from_numpy(g[f"step{i}_eps_cur"].astype(np.float64))]
            pre.clear()
            info = agent.update(step=int(step))
            out[f"step{i}_tuple"] = np.array([float(np.asarray(x)) for x in info], dtype=np.float64)
            for name in opts:
                out[f"step{i}_gradpre_{name}"] = pre[name]
            for name, net in nets.items():                      # back to the fp32 run's state
                set_flat(net, g[f"step{i}_param_{name}"])
            with torch.no_grad():
                agent.log_alpha.copy_(torch.from_numpy(g[f"step{i}_log_alpha"].astype(np.float64)))
                agent.alpha = agent.log_alpha.exp()
                H = bns[0].running_mean.numel()
                for l, m in enumerate(bns):
                    m.running_mean.copy_(torch.from_numpy(g[f"step{i}_bn_mean"][l * H:(l + 1) * H].astype(np.float64)))
                    m.running_var.copy_(torch.from_numpy(g[f"step{i}_bn_var"][l * H:(l + 1) * H].astype(np.float64)))
    finally:
        torch.nn.utils.clip_grad_norm_, torch.distributions.Normal.rsample = orig_clip, orig_rs
    np.savez_compressed(os.path.join(HERE, f"update_{tag}_fp64.npz"), **out)
    d32 = max(float(np.max(np.abs(out[f"step{i}_tuple"] - g[f"step{i}_tuple"]) / np.maximum(np.abs(out[f"step{i}_tuple"]), 1e-6))) for i in range(len(g["steps"])))
    print(f"update_{tag}_fp64 ok; the reference's own fp32 tuple error vs fp64: {d32:.2e}")


if __name__ == "__main__":
    gen_select_action()
    gen_per_stochastic("SAC", "sac", "config_sac_slide.yaml")
    gen_per_stochastic("TQC", "tqc", "config_tqc_push.yaml")
    gen_update_fp64("sac")
    gen_update_fp64("tqc")
