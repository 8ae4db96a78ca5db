"""ORACLE — TEST INFRASTRUCTURE ONLY.  Never imported by the product (gcrl_amd / libgcrl_hip.so).

CPU (eager torch, fp32) restatement of the reference's four agents' update step, as ONE class
parameterised by the agent kind instead of the reference's four near-duplicate classes.  It is
the checker in tests/ and smoke(), and the timed `cpu_baseline` ("port") of bench.py: same op
sequence and per-step `.item()` host syncs as the reference, so its speed is representative.

Pinned against the real reference: tests/golden/update_*.npz are captured from the reference's
own DDPG / TD3Agent / SACAgent / TQCAgent classes (tests/golden/make_golden.py);
tests/test_oracle_golden.py replays them through this file (losses, pre-/post-clip gradients,
parameters after the optimiser step, targets, BN statistics, log_alpha).

Reference lines (under /root/reference/src) are cited per method.
"""
from __future__ import annotations

import math
import random
from types import SimpleNamespace

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.optim import Adam, AdamW
from torch.optim.lr_scheduler import CosineAnnealingLR

from .her_oracle import HERBufferOracle

KINDS = ("DDPG", "TD3", "SAC", "TQC")


def _init_linear(m):
    """model.py:39-42 — xavier-uniform weights, bias 0.01."""
    if isinstance(m, nn.Linear):
        nn.init.xavier_uniform_(m.weight)
        m.bias.data.fill_(0.01)


def _hidden_stack(in_dim, H, L, bn):
    layers, k = [], in_dim
    for _ in range(L):
        layers.append(nn.Linear(k, H))
        if bn:
            layers += [nn.BatchNorm1d(H), nn.ReLU()]
        else:
            layers.append(nn.LeakyReLU())
        k = H
    return layers


class DetActor(nn.Module):
    """model.py:7-30 — state_dict keys base_net.{0,2,..}."""

    def __init__(self, obs_dim, H, ac_dim, L):
        super().__init__()
        self.base_net = nn.Sequential(*_hidden_stack(obs_dim, H, L, False), nn.Linear(H, ac_dim), nn.Tanh())
        self.apply(_init_linear)

    def forward(self, x):
        return self.base_net(x)


class QNet(nn.Module):
    """model.py:48-68 — keys net.{0,2,..}."""

    def __init__(self, in_dim, H, L):
        super().__init__()
        self.net = nn.Sequential(*_hidden_stack(in_dim, H, L, False), nn.Linear(H, 1))
        self.apply(_init_linear)

    def forward(self, x):
        return self.net(x)


class GaussActor(nn.Module):
    """model.py:86-141 — BN trunk, mean / log_std heads, tanh-squashed Gaussian."""

    def __init__(self, obs_dim, H, ac_dim, L):
        super().__init__()
        self.base_net = nn.Sequential(*_hidden_stack(obs_dim, H, L, True))
        self.mean_head = nn.Linear(H, ac_dim)
        self.log_std_head = nn.Linear(H, ac_dim)
        self.apply(_init_linear)

    def forward(self, x):
        f = self.base_net(x)
        return self.mean_head(f), torch.clamp(self.log_std_head(f), -20.0, 2.0)

    def sample(self, x, deterministic=False, eps=None):
        mean, log_std = self.forward(x)
        std = log_std.exp()
        if deterministic:
            return torch.tanh(mean), None
        dist = torch.distributions.Normal(mean, std)
        if eps is None:
            pre = dist.rsample()                      # model.py:134
        else:
            pre = mean + eps * std                    # what rsample computes, with recorded eps
        act = torch.tanh(pre)
        logp = dist.log_prob(pre)
        logp = logp - torch.log(1 - act.pow(2) + 1e-8)
        return act, logp.sum(dim=-1, keepdim=True)


def _grad_norm(model):
    """agent.py:1279-1286 — python-float accumulation of per-tensor fp32 norms."""
    tot = 0.0
    for p in model.parameters():
        if p.grad is not None:
            tot += p.grad.data.norm(2).item() ** 2
    return tot ** 0.5


def _flat(tensors):
    return np.concatenate([t.detach().cpu().numpy().reshape(-1) for t in tensors]).astype(np.float32)


class OracleAgent:
    def __init__(self, kind, obs_dim, ac_dim, config, nenvs=1, gradient_step=40, num_critics=5, top_drop=2,
                 rng=random):
        assert kind in KINDS
        self.kind, self.cfg, self.gradient_step = kind, config, gradient_step
        self.obs_dim, self.ac_dim = obs_dim, ac_dim
        H, L = config.hidden_dim, config.layer_count
        stochastic = kind in ("SAC", "TQC")
        self.stochastic = stochastic
        self.actor = (GaussActor if stochastic else DetActor)(obs_dim, H, ac_dim, L)
        self.target_actor = None if stochastic else DetActor(obs_dim, H, ac_dim, L)
        C = {"DDPG": 1, "TD3": 2, "SAC": 2, "TQC": num_critics}[kind]
        self.top_drop = top_drop if kind == "TQC" else 0
        self.critics = [QNet(obs_dim + ac_dim, H, L) for _ in range(C)]
        self.target_critics = [QNet(obs_dim + ac_dim, H, L) for _ in range(C)]
        Opt = Adam if kind == "DDPG" else AdamW     # agent.py:1201 vs :47,:420,:815
        self.actor_opt = Opt(self.actor.parameters(), config.actor_lr)
        self.critic_opts = [Opt(c.parameters(), config.critic_lr) for c in self.critics]
        self.actor_sched = CosineAnnealingLR(self.actor_opt, T_max=config.ac_scheduler_steps, eta_min=config.actor_lr_min)
        self.critic_scheds = [CosineAnnealingLR(o, T_max=config.cr_scheduler_steps, eta_min=config.critic_lr_min)
                              for o in self.critic_opts]
        if stochastic:
            self.target_entropy = -ac_dim * 0.5 if kind == "SAC" else -ac_dim   # :424 / :820
            self.log_alpha = torch.zeros(1, requires_grad=True)
            self.alpha = self.log_alpha.exp()
            self.alpha_opt = AdamW([self.log_alpha], lr=config.alpha_lr)
            self.alpha_min_steps = getattr(config, "alpha_min_steps", 10000)
        self.buffer = HERBufferOracle(config.max_len, config.max_eps_len, nenvs, k_future=config.k_future, rng=rng)
        self.hard_update()
        self.last = {}
        # data-parallel emulation hook (oracle/dp_oracle.py): called with (name, parameters) right after a
        # network's backward pass, before clipping — where a DP run exchanges that network's gradients
        self.grad_sync = None

    # ------------------------------------------------------------------ plumbing
    def hard_update(self):
        if self.target_actor is not None:
            self.target_actor.load_state_dict(self.actor.state_dict())
        for c, t in zip(self.critics, self.target_critics):
            t.load_state_dict(c.state_dict())

    @staticmethod
    def _polyak(net, target, tau):
        """agent.py:1260-1271 and copies."""
        for tp, p in zip(target.parameters(), net.parameters()):
            tp.data.copy_(tau * p.data + (1 - tau) * tp.data)

    def flat_params(self, net):
        return _flat(net.parameters())

    def set_flat_params(self, net, flat):
        off = 0
        with torch.no_grad():
            for p in net.parameters():
                n = p.numel()
                p.copy_(torch.from_numpy(np.asarray(flat[off:off + n], dtype=np.float32)).view_as(p))
                off += n

    def _train_mode(self):
        self.actor.train()
        for c in self.critics:
            c.train()
        if self.target_actor is not None:
            self.target_actor.eval()
        for t in self.target_critics:
            t.eval()

    def _target_q(self, x):
        qs = torch.stack([t(x) for t in self.target_critics])
        if self.kind == "DDPG":
            return qs[0]
        if self.kind in ("TD3", "SAC"):
            return torch.min(qs[0], qs[1])
        srt, _ = torch.sort(qs, dim=0)                       # :972-974
        return srt[: -self.top_drop].mean(dim=0) if self.top_drop > 0 else qs.mean(dim=0)

    # ------------------------------------------------------------------ critic update
    def critic_update(self, s, a, r, ns, d, noise=None, eps_next=None, weights=None):
        cfg, kind = self.cfg, self.kind
        with torch.no_grad():
            if kind == "DDPG":                                # :1311-1317
                na = self.target_actor(ns)
                y = r + cfg.gamma * (1.0 - d) * self._target_q(torch.cat([ns, na], dim=-1))
                y = torch.clamp(y, min=-1.0 / (1.0 - cfg.gamma), max=0.0)
            elif kind == "TD3":                               # :173-186
                z = torch.randn_like(a) if noise is None else noise
                nz = torch.clamp(z * cfg.policy_noise, -cfg.noise_clamp, cfg.noise_clamp)
                na = torch.clamp(self.target_actor(ns) + nz, -1, 1)
                y = r + cfg.gamma * (1 - d) * self._target_q(torch.cat([ns, na], dim=-1))
            else:                                             # :557-570 / :960-979
                na, nlp = self.actor.sample(ns, eps=eps_next)
                tq = self._target_q(torch.cat([ns, na], dim=-1))
                ent = 0.2 if kind == "SAC" else self.alpha    # SAC: literal 0.2 (:569)
                y = r + cfg.gamma * (1 - d) * (tq - ent * nlp)
        x = torch.cat([s, a], dim=-1)
        loss_fn = F.smooth_l1_loss if kind == "TD3" else F.mse_loss
        losses, gnorms, tds, qs = [], [], [], []
        pre, post = [], []
        if kind != "TQC":
            cur = [c(x) for c in self.critics]                # both current Qs first (:189-190, :573-574)
        for i, (c, opt) in enumerate(zip(self.critics, self.critic_opts)):
            q = cur[i] if kind != "TQC" else c(x)             # TQC: forward inside the loop (:990)
            opt.zero_grad()
            if weights is not None and weights.numel() > 0:       # PER: (weights * loss).mean()  (:193-197, :577-581, :993-997, :1320-1325)
                loss = (weights * loss_fn(q, y, reduction="none")).mean()
            else:
                loss = loss_fn(q, y)
            loss.backward()
            if self.grad_sync is not None:
                self.grad_sync(f"critic_{i}", list(c.parameters()))
            pre.append(_flat(p.grad for p in c.parameters()))
            clip_this = cfg.grad_clip is not None and not (kind == "TD3" and i == 0)   # :201 commented out
            if clip_this:
                torch.nn.utils.clip_grad_norm_(c.parameters(), cfg.grad_clip)
            post.append(_flat(p.grad for p in c.parameters()))
            gnorms.append(_grad_norm(c))
            opt.step()
            if kind in ("DDPG", "TQC"):
                self.critic_scheds[i].step()                  # :1334 / :1005
            losses.append(loss.item())
            tds.append(torch.abs(q - y).detach())
            qs.append(q.detach())
        if kind in ("TD3", "SAC"):
            for sc in self.critic_scheds:                     # :218-219 / :606-607
                sc.step()
        self.last.update(critic_grads_pre=pre, critic_grads_post=post, target=y.detach().numpy().copy(),
                         td_per_sample=torch.stack(tds).max(dim=0)[0].numpy().copy())
        if kind == "DDPG":                                    # :1336-1343
            return losses[0], torch.mean(tds[0]).cpu().numpy(), qs[0].mean().cpu().item(), gnorms[0]
        if kind == "TQC":                                     # :1013-1042
            q_value = torch.stack([c(x) for c in self.critics]).mean().detach().cpu().item()
            td = torch.mean(torch.stack(tds).max(dim=0)[0]).cpu().numpy()
            return np.mean(losses), np.mean(losses), td, q_value, np.mean(gnorms), np.mean(gnorms)
        q_value = torch.cat(qs, dim=-1).mean().cpu().item()   # :224-230 / :612-618
        td = torch.mean(torch.maximum(tds[0], tds[1])).cpu().numpy()
        return losses[0], losses[1], td, q_value, gnorms[0], gnorms[1]

    # ------------------------------------------------------------------ actor update
    def actor_update(self, s, eps_cur=None):
        cfg, kind = self.cfg, self.kind
        logp = None
        if not self.stochastic:                               # :1288-1300 / :149-162
            a = self.actor(s)
            loss = -self.critics[0](torch.cat([s, a], dim=-1)).mean()
        else:
            a, logp = self.actor.sample(s, eps=eps_cur)
            x = torch.cat([s, a], dim=-1)
            if kind == "SAC":                                 # :513-521
                mq = torch.min(self.critics[0](x), self.critics[1](x))
                loss = (0.2 * logp - mq).mean()
            else:                                             # :912-925
                qv = torch.stack([c(x) for c in self.critics])
                if self.top_drop > 0:
                    srt, _ = torch.sort(qv, dim=0)
                    mq = srt[: -self.top_drop].mean(dim=0)
                else:
                    mq = qv.mean(dim=0)
                loss = (self.alpha.detach() * logp - mq).mean()
        self.actor_opt.zero_grad()
        loss.backward()
        if self.grad_sync is not None:
            self.grad_sync("actor", list(self.actor.parameters()))
        self.last["actor_grads_pre"] = _flat(p.grad for p in self.actor.parameters())
        if cfg.grad_clip is not None:
            torch.nn.utils.clip_grad_norm_(self.actor.parameters(), cfg.grad_clip)
        self.last["actor_grads_post"] = _flat(p.grad for p in self.actor.parameters())
        gn = _grad_norm(self.actor)
        self.actor_opt.step()
        self.actor_sched.step()
        return loss.item(), gn, (None if logp is None else logp.detach())

    def alpha_update(self, logp, step):
        """:532-546 / :936-949."""
        if step <= self.alpha_min_steps:
            return 0.0
        loss = -(self.log_alpha * (logp + self.target_entropy).detach()).mean()
        self.alpha_opt.zero_grad()
        loss.backward()
        if self.grad_sync is not None:
            self.grad_sync("log_alpha", [self.log_alpha])
        self.last["alpha_grad"] = float(self.log_alpha.grad.item())
        self.alpha_opt.step()
        self.alpha = self.log_alpha.exp()
        return loss.item()

    # ------------------------------------------------------------------ update
    def update(self, step, batch=None, noise=None, eps_next=None, eps_cur=None, weights=None):
        """DDPG :1378-1404, TD3 :281-317, SAC :659-699, TQC :1062-1100."""
        self._train_mode()
        if batch is None:
            batch = tuple(torch.from_numpy(x) for x in self.buffer.sample(self.cfg.batch_size))
        s, a, r, ns, d = batch
        cfg, kind = self.cfg, self.kind
        cinfo = self.critic_update(s, a, r, ns, d, noise=noise, eps_next=eps_next, weights=weights)
        do_actor = step % cfg.ac_update_freq == 0
        if kind == "DDPG":
            if step % 40 == 0:                                # literal 40 (:1397)
                self._polyak(self.actor, self.target_actor, cfg.tau)
                self._polyak(self.critics[0], self.target_critics[0], cfg.tau)
            if do_actor:
                al, ag, _ = self.actor_update(s)
                return cinfo[0], al, cinfo[1], cinfo[2], cinfo[3], ag
            return cinfo
        if kind == "TD3":
            for c, t in zip(self.critics, self.target_critics):
                self._polyak(c, t, cfg.tau)
            if do_actor:
                al, ag, _ = self.actor_update(s)
                self._polyak(self.actor, self.target_actor, cfg.tau)
                return cinfo[0], cinfo[1], al, cinfo[2], cinfo[3], cinfo[4], cinfo[5], ag
            return cinfo
        if kind == "TQC" or step % self.gradient_step == 0:   # SAC cadence :681; TQC every step :1083
            for c, t in zip(self.critics, self.target_critics):
                self._polyak(c, t, cfg.tau)
        if do_actor:
            al, ag, logp = self.actor_update(s, eps_cur=eps_cur)
            alpha_loss = self.alpha_update(logp, step)
            return cinfo[0], cinfo[1], al, cinfo[2], cinfo[3], cinfo[4], cinfo[5], ag, alpha_loss
        return cinfo

    def push_her(self, *args):
        self.buffer.push(*args)


def make_config(kind="DDPG", **over):
    """Hyper-parameters with the reference YAMLs' field names (src/utils.py:10-39)."""
    base = dict(hidden_dim=64, layer_count=3, actor_lr=1e-3, actor_lr_min=1e-3, ac_scheduler_steps=1,
                critic_lr=1e-3, critic_lr_min=1e-3, cr_scheduler_steps=1, buffer_type="HER", max_len=100000,
                alpha=1.0, batch_size=256, gamma=0.98, ac_update_freq=1, noise_std=0.2, noise_clamp=0.5,
                policy_noise=0.0, grad_clip=10.0, beta=1.0, beta_end=1, k_future=4, max_eps_len=50, tau=0.05)
    if kind in ("SAC", "TQC"):
        base.update(alpha_lr=3e-4, alpha_min=0.05, alpha_min_steps=0.0)
    base.update(over)
    return SimpleNamespace(**base)
