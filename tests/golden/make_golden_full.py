"""Full-size golden fixtures: one update() of the REAL reference at BASELINE.json's cfg 2-5 shapes
(and the headline shape), in fp32 AND in fp64 (`.double()` networks, same inputs).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_full.py

Run only in the build container (needs /root/reference).  The small fixtures of make_golden.py pin
every quantity of every agent at H=32/64; these pin the code paths that only the benchmarked sizes
select (LDS-tiled GEMM at H=512/B=2048, the k-split dW rule at K >= 1024, 8-/16-row blocks of the
row-chain kernel) to the reference itself.

Inputs are NOT stored (3.2 M parameters for the 5-critic TQC): parameters and batches are rebuilt
from seeds by tests/detdata.py on both sides; the fixture carries their checksums.  Stored per case:
  tuple32 / tuple64          the returned tuple of the fp32 / fp64 reference run
  gnorm32_<net> / gnorm64_   pre-clip global gradient norm per network
  gidx_<net>                 strided sample positions (~20000 / n_nets per network)
  g32_<net> / g64_<net>      pre-clip gradient at those positions
  p32_<net> / p64_<net>      parameters after the optimiser (+ Polyak for targets) step, sampled
  bn_mean32/64, bn_var32/64, log_alpha32/64, alpha32/64   (SAC / TQC)
  kink_units, kinku_<net>, kinkF_<net>, kinkdot_<net>, kinkgram_<net>   see "activation kinks" below
The fp64 run is what lets the GPU test say "the HIP error is no larger than the reference's own
fp32 error": |hip - f64| is compared with |ref32 - f64| per quantity.
Every case starts from a FRESH agent (optimiser step count 0), so cases are independent.

Activation kinks.  LeakyReLU / ReLU are discontinuous in their derivative at 0.  Among the ~16 M hidden
pre-activations of a B=2048, H=512 step a few lie within fp32 rounding error of 0, and two correct fp32
implementations with different summation orders may put such a unit on opposite sides: the unit's
derivative flips (1 <-> 0.01), one batch row's contribution to every gradient below changes — a discrete,
exactly computable change, not an accuracy defect (the reference itself flips between BLAS kernels / thread
counts).  The generator therefore lists every unit of a gradient-carrying forward pass whose fp64
pre-activation is within KINK_TAU x rms of zero, re-runs the fp64 step once per unit with that unit's
derivative flipped, and stores the resulting gradient change ("flip vector") at the sampled positions plus
its inner products with the gradient and the other flip vectors.  The tests explain a deviation as an
integer combination of flip vectors (least squares) and hold the REMAINDER to the strict bound.
"""
import os
import random
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

import make_golden as mg  # noqa: E402  (installs the gymnasium stub, imports the reference)
import detdata  # noqa: E402

CASES = {
    # headline: PickAndPlace DDPG, batch 256
    "ddpg_pickplace_b256": dict(kind="DDPG", yaml="config_ddpg_pickplace.yaml", over={}, dims=(23, 4), B=256, step=1, gstep=40),
    # cfg 1 (BASELINE configs[0], the reference's own CPU-runnable case): Reach, the YAML's H = 64 / L = 3, batch 256
    "cfg1_ddpg_reach_b256": dict(kind="DDPG", yaml="config_ddpg_reach.yaml", over={}, dims=(10, 3), B=256, step=1, gstep=40),
    # cfg 2: Reach, B=1024, MLP(256,256)
    "cfg2_ddpg_reach_b1024": dict(kind="DDPG", yaml="config_ddpg_reach.yaml", over=dict(hidden_dim=256, layer_count=2),
                                  dims=(10, 3), B=1024, step=1, gstep=40),
    # cfg 3: TD3 PickAndPlace B=2048: a critic-only step and an actor step
    "cfg3_td3_pickplace_b2048_s1": dict(kind="TD3", yaml="config_td3_pickplace.yaml", over={}, dims=(23, 4), B=2048, step=1, gstep=40),
    "cfg3_td3_pickplace_b2048_s2": dict(kind="TD3", yaml="config_td3_pickplace.yaml", over={}, dims=(23, 4), B=2048, step=2, gstep=40),
    # cfg 4: TQC Push B=2048 H=512, 5 critics (reference semantics); alpha step active
    "cfg4_tqc_push_b2048": dict(kind="TQC", yaml="config_tqc_push.yaml", over=dict(alpha_min_steps=0.0), dims=(22, 3), B=2048,
                                step=1, gstep=40),
    # cfg 5: SAC Slide B=512 (per GPU); step 2 with gradient_step 2: Polyak branch + alpha step
    "cfg5_sac_slide_b512": dict(kind="SAC", yaml="config_sac_slide.yaml", over=dict(alpha_min_steps=1.0), dims=(22, 3), B=512,
                                step=2, gstep=2),
}


def set_flat(net, flat):
    off = 0
    with torch.no_grad():
        for p in net.parameters():
            n = p.numel()
            p.copy_(torch.from_numpy(flat[off:off + n]).view_as(p))
            off += n
    assert off == flat.size


def net_table(kind, agent):
    """name -> (module, optimiser or None), names as the engine's ABI uses them."""
    if kind == "DDPG":
        return dict(actor=(agent.actor, agent.actor_opt), target_actor=(agent.target_actor, None),
                    critic_0=(agent.critic, agent.critic_opt), target_critic_0=(agent.target_critic, None))
    if kind in ("TD3", "SAC"):
        t = dict(actor=(agent.actor, agent.actor_opt), critic_0=(agent.critic_1, agent.critic_1_opt),
                 critic_1=(agent.critic_2, agent.critic_2_opt), target_critic_0=(agent.target_critic_1, None),
                 target_critic_1=(agent.target_critic_2, None))
        if kind == "TD3":
            t["target_actor"] = (agent.target_actor, None)
        return t
    t = dict(actor=(agent.actor, agent.actor_opt))
    for i, (c, tc, o) in enumerate(zip(agent.critics, agent.target_critics, agent.critic_opts)):
        t[f"critic_{i}"] = (c, o)
        t[f"target_critic_{i}"] = (tc, None)
    return t


def init_vector(case, name, kind, S, A, H, L):
    if "actor" in name:
        return detdata.net_params(f"{case}/{name}", "sac_actor" if kind in ("SAC", "TQC") else "mlp", S, H, L, A)
    return detdata.net_params(f"{case}/{name}", "mlp", S + A, H, L, 1)


KINK_TAU = 2e-6   # |z| < KINK_TAU * rms(z of the layer): ~10x the fp32 rounding error of a 512-term dot product


class Kinks:
    """Forward hooks on the activation modules of the gradient-carrying (online) networks: in detection mode
    lists near-zero pre-activations of grad-enabled passes; in flip mode overrides the slope of given units."""

    def __init__(self, nets, flips=()):
        self.found, self.calls, self.flips = [], {}, set(flips)
        self.handles = []
        for name, (net, opt) in nets.items():
            if opt is None:
                continue
            acts = [m for m in net.modules() if isinstance(m, (torch.nn.LeakyReLU, torch.nn.ReLU))]
            for ai, m in enumerate(acts):
                self.handles.append(m.register_forward_hook(self._hook(name, ai, m)))

    def _hook(self, name, ai, mod):
        slope = mod.negative_slope if isinstance(mod, torch.nn.LeakyReLU) else 0.0

        def hook(module, inputs, output):
            if not torch.is_grad_enabled():
                return None
            z = inputs[0]
            call = self.calls.get((name, ai), 0)
            self.calls[(name, ai)] = call + 1
            if not self.flips:
                thr = KINK_TAU * float(z.detach().pow(2).mean().sqrt())
                for b, j in (z.detach().abs() < thr).nonzero().tolist():
                    self.found.append((name, ai, call, b, j, float(z[b, j])))
                return None
            mine = [(b, j) for (n, a, c, b, j) in self.flips if (n, a, c) == (name, ai, call)]
            if not mine:
                return None
            out = output.clone()
            for b, j in mine:
                other = slope if float(z[b, j]) > 0 else 1.0     # the slope of the OTHER side of the kink
                out[b, j] = z[b, j] * other
            return out

        return hook

    def remove(self):
        for h in self.handles:
            h.remove()


def run_case(case, spec, dtype, flips=(), detect=False):
    kind, (S, A), B, step = spec["kind"], spec["dims"], spec["B"], spec["step"]
    cfg = mg.load_her_config(os.path.join(mg.CFG_DIR, kind, spec["yaml"]), kind)
    acfg = cfg.agent.model_copy(update=dict(batch_size=B, **spec["over"]))
    H, L = acfg.hidden_dim, acfg.layer_count
    torch.manual_seed(1898); np.random.seed(1898); random.seed(1898)
    cls = dict(DDPG=mg.DDPG, TD3=mg.TD3Agent, SAC=mg.SACAgent, TQC=mg.TQCAgent)[kind]
    agent = cls(obs_dim=S, ac_dim=A, config=acfg, weights=None, nenvs=1, gradient_step=spec["gstep"])
    nets = net_table(kind, agent)
    sums = {}
    for name, (net, _) in nets.items():
        vec = init_vector(case, name, kind, S, A, H, L)
        set_flat(net, vec)
        sums[name] = detdata.checksum(vec)
        if dtype == torch.float64:
            net.double()
    if dtype == torch.float64 and kind in ("SAC", "TQC"):
        agent.log_alpha.data = agent.log_alpha.data.double()
        agent.alpha = agent.log_alpha.exp()
    batch = detdata.batch(case, B, S, A)
    tb = tuple(torch.from_numpy(x).to(dtype) for x in batch)
    agent.buffer.sample = lambda bs: tb
    queue = []
    if kind == "TD3":
        queue.append(torch.from_numpy(detdata.normalish(detdata.seed_of(case, "noise"), (B, A))).to(dtype))
    if kind in ("SAC", "TQC"):
        queue.append(torch.from_numpy(detdata.normalish(detdata.seed_of(case, "eps_next"), (B, A))).to(dtype))
        queue.append(torch.from_numpy(detdata.normalish(detdata.seed_of(case, "eps_cur"), (B, A))).to(dtype))

    kinks = Kinks(nets, flips) if (detect or flips) else None
    rec = mg.Recorder()
    for name, (net, opt) in nets.items():
        if opt is not None:
            rec.register(name, net, opt)
    orig_clip, orig_rl, orig_rs = torch.nn.utils.clip_grad_norm_, torch.randn_like, torch.distributions.Normal.rsample
    torch.nn.utils.clip_grad_norm_ = rec.clip_hook(orig_clip)
    torch.randn_like = lambda t, *a, **k: queue.pop(0)
    torch.distributions.Normal.rsample = lambda self, sample_shape=torch.Size(): self.loc + queue.pop(0) * self.scale
    try:
        info = agent.update(step=step)
    finally:
        torch.nn.utils.clip_grad_norm_, torch.randn_like, torch.distributions.Normal.rsample = orig_clip, orig_rl, orig_rs
        if kinks is not None:
            kinks.remove()
    assert not queue
    out = dict(tuple=np.array([float(np.asarray(x)) for x in info], dtype=np.float64))
    flat64 = lambda ts: np.concatenate([t.detach().numpy().reshape(-1).astype(np.float64) for t in ts])
    for name, (net, opt) in nets.items():
        out[f"p_{name}"] = flat64(net.parameters())
    for name in rec.pre64:   # pre-clip gradients, not cast to fp32 (the fp64 run needs them as they are)
        out[f"g_{name}"] = rec.pre64[name]
    if kind in ("SAC", "TQC"):
        bns = [m for m in agent.actor.base_net if isinstance(m, torch.nn.BatchNorm1d)]
        out["bn_mean"] = np.concatenate([m.running_mean.numpy().astype(np.float64) for m in bns])
        out["bn_var"] = np.concatenate([m.running_var.numpy().astype(np.float64) for m in bns])
        out["log_alpha"] = agent.log_alpha.detach().numpy().astype(np.float64).copy()
        out["alpha"] = agent.alpha.detach().numpy().astype(np.float64).copy()
    if detect:
        out["kinks"] = kinks.found
    return out, acfg, sums, batch


class Recorder64(mg.Recorder):
    """Also keeps the pre-clip gradients in float64 (the base class casts to fp32)."""

    def __init__(self):
        super().__init__()
        self.pre64 = {}

    def register(self, name, module, opt):
        super().register(name, module, opt)
        orig = opt.step
        rec = self

        def step(*a, **k):
            rec.pre64.setdefault(name, np.concatenate([p.grad.detach().numpy().reshape(-1).astype(np.float64)
                                                       for p in module.parameters()]))
            return orig(*a, **k)

        opt.step = step

    def clip_hook(self, orig):
        base = super().clip_hook(orig)
        rec = self

        def clip(parameters, max_norm, *a, **k):
            params = list(parameters)
            name = rec.names.get(id(params[0]))
            if name is not None:
                rec.pre64[name] = np.concatenate([p.grad.detach().numpy().reshape(-1).astype(np.float64) for p in params])
            return base(params, max_norm, *a, **k)

        return clip


mg.Recorder = Recorder64


def main():
    only = sys.argv[1:]
    for case, spec in CASES.items():
        if only and case not in only:
            continue
        r32, acfg, sums, batch = run_case(case, spec, torch.float32)
        r64, _, _, _ = run_case(case, spec, torch.float64, detect=True)
        kind, (S, A), B = spec["kind"], spec["dims"], spec["B"]
        out = dict(meta=np.array([str(mg.META)]), kind=np.array([kind]),
                   dims=np.array([S, A, B, spec["gstep"], acfg.hidden_dim, acfg.layer_count]), step=np.array([spec["step"]]),
                   tuple32=r32["tuple"], tuple64=r64["tuple"])
        hp = acfg.model_dump()
        out["hparams_keys"] = np.array(list(hp.keys()))
        out["hparams_vals"] = np.array([str(v) for v in hp.values()])
        for name, cs in sums.items():
            out[f"initsum_{name}"] = cs
        for key, arr in zip(("s", "a", "r", "ns", "d"), batch):
            out[f"batchsum_{key}"] = detdata.checksum(arr)
        names = [k[2:] for k in r32 if k.startswith("p_")]
        per_net = max(512, 20000 // len(names))
        for name in names:
            n = r32[f"p_{name}"].size
            idx = np.unique(np.linspace(0, n - 1, min(n, per_net)).astype(np.int64))
            out[f"gidx_{name}"] = idx
            out[f"p32_{name}"] = r32[f"p_{name}"][idx].astype(np.float32)
            out[f"p64_{name}"] = r64[f"p_{name}"][idx]
            if f"g_{name}" in r32:
                g32, g64 = r32[f"g_{name}"], r64[f"g_{name}"]
                out[f"gnorm32_{name}"] = np.array([np.sqrt(np.square(g32).sum())])
                out[f"gnorm64_{name}"] = np.array([np.sqrt(np.square(g64).sum())])
                out[f"g32_{name}"] = g32[idx].astype(np.float32)
                out[f"g64_{name}"] = g64[idx]
        for k in ("bn_mean", "bn_var", "log_alpha", "alpha"):
            if k in r32:
                out[k + "32"] = r32[k].astype(np.float32)
                out[k + "64"] = r64[k]
        # activation kinks: one fp64 re-run per near-zero unit with its derivative flipped
        gnets = [n for n in names if f"g_{n}" in r64]
        units, effects = [], []
        for (net, ai, call, b, j, z) in r64["kinks"]:
            rf, _, _, _ = run_case(case, spec, torch.float64, flips=[(net, ai, call, b, j)])
            eff = {n: rf[f"g_{n}"] - r64[f"g_{n}"] for n in gnets}
            if max(float(np.abs(e).max()) for e in eff.values()) == 0.0:
                continue      # a pass that is never back-propagated (TQC's post-update metric forward)
            units.append((gnets.index(net) if net in gnets else -1, ai, call, b, j, z))
            effects.append(eff)
        out["kink_units"] = np.array(units, dtype=np.float64).reshape(len(units), 6)
        out["kink_nets"] = np.array(gnets)
        for n in gnets:   # per network: the units that move its gradient, their flip vectors (sampled), inner products
            rows = [u for u, e in enumerate(effects) if float(np.abs(e[n]).max()) > 0.0]
            F = np.array([effects[u][n] for u in rows]).reshape(len(rows), r64[f"g_{n}"].size)
            out[f"kinku_{n}"] = np.array(rows, dtype=np.int64)
            out[f"kinkF_{n}"] = F[:, out[f"gidx_{n}"]].astype(np.float32)
            out[f"kinkdot_{n}"] = F @ r64[f"g_{n}"]
            out[f"kinkgram_{n}"] = F @ F.T
        path = os.path.join(HERE, f"full_{case}.npz")
        np.savez_compressed(path, **out)
        rel = np.abs(r32["tuple"] - r64["tuple"]) / np.maximum(1e-12, np.abs(r64["tuple"]))
        print(case, "ok", os.path.getsize(path) // 1024, "KiB; ref32 vs ref64 tuple rel err max", float(rel.max()),
              "; near-kink units", len(r64["kinks"]), "with effect", len(units))


if __name__ == "__main__":
    main()
