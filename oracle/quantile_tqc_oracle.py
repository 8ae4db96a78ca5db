"""ORACLE — TEST INFRASTRUCTURE ONLY.  Never imported by the product.

CPU (eager torch) specification of the DISTRIBUTIONAL TQC variant that BASELINE.json configs[3] describes
("25 quantiles x 2 critics, top-2 truncate") — Truncated Quantile Critics, Kuznetsov et al. 2020.

**No reference parity.**  The reference's TQCAgent is an ensemble of five SCALAR critics whose outputs are sorted per
sample (src/agent.py:916-923, :964-976; SURVEY.md headline facts): a quantile critic has no counterpart in
/root/reference.  This file is therefore the specification the HIP path (n_quantiles > 1) is tested against, built
from the reference's own pieces wherever they exist: its SACActorModel / Critic architectures (src/model.py), AdamW,
gradient clipping, Polyak of every critic each step, learned alpha with target_entropy = -A, the post-update q_value
metric (src/agent.py:774-1100) — with the critic head widened to Q atoms and the loss replaced by:

  target    z = the C*Q pooled atoms of the target critics at (s', a'), sorted ascending per row; the top `drop`*C dropped;
            y_j = r + gamma*(1-d)*(z_(j) - alpha*logp(a'|s'))                                 j < K = C*(Q - drop)
  critic c  loss_c = mean over (B, Q, K) of |tau_i - 1(u<0)| * huber_1(u),  u = y_j - z_c,i,  tau_i = (2i+1)/(2Q)
  actor     loss = mean_b(alpha*logp_b - mean over all C*Q atoms of z(s_b, pi(s_b)))
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn
from torch.optim import AdamW
from torch.optim.lr_scheduler import CosineAnnealingLR

from .agent_oracle import GaussActor, _flat, _grad_norm, _hidden_stack, _init_linear


class QuantileNet(nn.Module):
    """The reference's Critic (src/model.py:48-68) with a Q-wide head."""

    def __init__(self, in_dim, H, L, Q):
        super().__init__()
        self.net = nn.Sequential(*_hidden_stack(in_dim, H, L, False), nn.Linear(H, Q))
        self.apply(_init_linear)

    def forward(self, x):
        return self.net(x)


class QuantileTQCOracle:
    def __init__(self, obs_dim, ac_dim, config, n_quantiles=25, num_critics=2, top_drop=2, gradient_step=40):
        self.cfg, self.Q, self.C, self.drop = config, n_quantiles, num_critics, top_drop
        H, L = config.hidden_dim, config.layer_count
        self.actor = GaussActor(obs_dim, H, ac_dim, L)
        self.critics = [QuantileNet(obs_dim + ac_dim, H, L, n_quantiles) for _ in range(num_critics)]
        self.target_critics = [QuantileNet(obs_dim + ac_dim, H, L, n_quantiles) for _ in range(num_critics)]
        self.actor_opt = AdamW(self.actor.parameters(), config.actor_lr)
        self.critic_opts = [AdamW(c.parameters(), config.critic_lr) for c in self.critics]
        self.actor_sched = CosineAnnealingLR(self.actor_opt, T_max=config.ac_scheduler_steps, eta_min=config.actor_lr_min)
        self.critic_scheds = [CosineAnnealingLR(o, T_max=config.cr_scheduler_steps, eta_min=config.critic_lr_min) for o in self.critic_opts]
        self.target_entropy = -ac_dim
        self.log_alpha = torch.zeros(1, requires_grad=True)
        self.alpha = self.log_alpha.exp()
        self.alpha_opt = AdamW([self.log_alpha], lr=config.alpha_lr)
        self.alpha_min_steps = getattr(config, "alpha_min_steps", 10000)
        self.hard_update()
        self.last = {}

    def hard_update(self):
        for c, t in zip(self.critics, self.target_critics):
            t.load_state_dict(c.state_dict())

    flat_params = staticmethod(lambda net: _flat(net.parameters()))

    def set_flat_params(self, net, flat):
        off = 0
        with torch.no_grad():
            for p in net.parameters():
                n = p.numel()
                p.copy_(torch.from_numpy(np.asarray(flat[off:off + n], dtype=np.float32)).view_as(p))
                off += n

    def update(self, step, batch, eps_next, eps_cur):
        cfg, Q, C = self.cfg, self.Q, self.C
        s, a, r, ns, d = batch
        self.actor.train()
        K = C * (Q - self.drop)
        with torch.no_grad():
            na, nlp = self.actor.sample(ns, eps=eps_next)
            z = torch.cat([t(torch.cat([ns, na], -1)) for t in self.target_critics], dim=1)        # [B, C*Q]
            z, _ = torch.sort(z, dim=1)
            y = r + cfg.gamma * (1 - d) * (z[:, :K] - self.alpha * nlp)                             # [B, K]
        x = torch.cat([s, a], -1)
        tau = ((2 * torch.arange(Q, dtype=torch.float32) + 1) / (2 * Q)).view(1, Q, 1)
        losses, gnorms, pre, tds = [], [], [], []
        for c, opt, sch in zip(self.critics, self.critic_opts, self.critic_scheds):
            q = c(x)                                                                                # [B, Q]
            u = y[:, None, :] - q[:, :, None]                                                       # [B, Q, K]
            hub = torch.where(u.abs() <= 1.0, 0.5 * u * u, u.abs() - 0.5)
            loss = (torch.abs(tau - (u.detach() < 0).float()) * hub).mean()
            opt.zero_grad()
            loss.backward()
            pre.append(_flat(p.grad for p in c.parameters()))
            if cfg.grad_clip is not None:
                torch.nn.utils.clip_grad_norm_(c.parameters(), cfg.grad_clip)
            gnorms.append(_grad_norm(c))
            opt.step()
            sch.step()
            losses.append(loss.item())
            tds.append((y.mean(dim=1) - q.detach().mean(dim=1)).abs())
        q_value = torch.cat([c(x) for c in self.critics], dim=1).mean().item()                      # updated critics (:1016-1019)
        td = torch.stack(tds).max(dim=0)[0].mean().item()
        for c, t in zip(self.critics, self.target_critics):                                         # Polyak every step (:1083)
            for tp, p in zip(t.parameters(), c.parameters()):
                tp.data.copy_(cfg.tau * p.data + (1 - cfg.tau) * tp.data)
        self.last = dict(critic_grads_pre=pre)
        cinfo = (float(np.mean(losses)), float(np.mean(losses)), td, q_value, float(np.mean(gnorms)), float(np.mean(gnorms)))
        if step % cfg.ac_update_freq != 0:
            return cinfo
        act, logp = self.actor.sample(s, eps=eps_cur)
        zz = torch.cat([c(torch.cat([s, act], -1)) for c in self.critics], dim=1)
        aloss = (self.alpha.detach() * logp - zz.mean(dim=1, keepdim=True)).mean()
        self.actor_opt.zero_grad()
        aloss.backward()
        self.last["actor_grads_pre"] = _flat(p.grad for p in self.actor.parameters())
        if cfg.grad_clip is not None:
            torch.nn.utils.clip_grad_norm_(self.actor.parameters(), cfg.grad_clip)
        agn = _grad_norm(self.actor)
        self.actor_opt.step()
        self.actor_sched.step()
        alpha_loss = 0.0
        if step > self.alpha_min_steps:
            al = -(self.log_alpha * (logp.detach() + self.target_entropy)).mean()
            self.alpha_opt.zero_grad()
            al.backward()
            self.alpha_opt.step()
            self.alpha = self.log_alpha.exp()
            alpha_loss = al.item()
        return cinfo[0], cinfo[1], aloss.item(), cinfo[2], cinfo[3], cinfo[4], cinfo[5], agn, alpha_loss
