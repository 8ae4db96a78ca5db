"""GPU (-m gpu): the BENCHMARKED multi-step path of TD3 / SAC / TQC — `update_many` = the trainer's
`for _ in range(gradient_step): agent.update(step)` loop (reference src/env.py:384-385; update bodies src/agent.py:281-317,
:659-699, :1062-1100) — held to

  (1) N x update(): one gather launch for all batches, runs of identical steps replayed as one hipGraph of up to 8 steps,
      control-block advances riding on the optimiser launches — none of which may change a bit: tuples, parameters, targets,
      BatchNorm statistics and log_alpha after `update_many` are BITWISE those of repeated `update()` (the device noise is
      keyed by a per-step counter, so both sides draw the same values), across 8-step run boundaries, SAC's
      `step % gradient_step` Polyak cadence and TD3's `ac_update_freq`;
  (2) the oracle: sampled batches (same MT stream) and the noise the engine itself drew — restated by
      oracle/device_rng_oracle.py from the documented counter scheme — fed to OracleAgent.update step by step.
"""
import random

import numpy as np
import pytest
import torch

from oracle import her_oracle
from oracle.agent_oracle import OracleAgent, make_config
from oracle.device_rng_oracle import hash_normal

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, scope="module")
def _oracle_on_one_thread():
    """The oracle is eager torch on the CPU: its fp32 sums depend on the thread count (the reference itself moves 9.8e-4 between
    thread counts over a few Adam steps), and a trajectory test must not depend on which test ran before it."""
    n = torch.get_num_threads()
    torch.set_num_threads(1)
    yield
    torch.set_num_threads(n)
S, A = 10, 3


def _cls(gcrl, kind):
    return dict(DDPG=gcrl.DDPG, TD3=gcrl.TD3Agent, SAC=gcrl.SACAgent, TQC=gcrl.TQCAgent)[kind]


def _cfg(kind, H, L, B, **over):
    base = dict(hidden_dim=H, layer_count=L, batch_size=B, max_len=4000, grad_clip=1.0, policy_noise=0.2, noise_clamp=0.5,
                ac_update_freq=2 if kind == "TD3" else 1, tau=0.05, actor_lr_min=2e-4, ac_scheduler_steps=30,
                critic_lr_min=3e-4, cr_scheduler_steps=25)
    if kind in ("SAC", "TQC"):
        base.update(alpha_min_steps=6.0, alpha_lr=1e-2)       # the alpha branch switches on inside the run
    base.update(over)
    return make_config(kind, **base)


def _build(gcrl, kind, cfg, gstep, seed=21, **kw):
    ag = _cls(gcrl, kind)(S, A, cfg, None, nenvs=2, gradient_step=gstep, rng="engine", seed=seed, **kw)
    gen = np.random.default_rng(3)
    for ep in range(4):
        for st in her_oracle.synthetic_episode(gen, 50, S, A):
            ag.push_her(ep % 2, *st)
    gen2 = np.random.default_rng(8)
    for v in [ag.actor] + ag.critics:
        v.set_flat((v.flat() + 0.05 * gen2.standard_normal(v.numel())).astype(np.float32))
    ag.update_target_network()
    return ag


def _state(ag):
    out = [ag.actor.flat()] + [c.flat() for c in ag.critics] + [t.flat() for t in ag.target_critics]
    if hasattr(ag, "target_actor"):
        out.append(ag.target_actor.flat())
    if ag._sac:
        out += [ag.actor._get("bn_running_mean"), ag.actor._get("bn_running_var"), np.array([ag.alpha.item()], np.float32)]
    return out


@pytest.mark.parametrize("use_graph", [True, 2])
@pytest.mark.parametrize("kind,H,L,B", [("TD3", 32, 2, 32), ("SAC", 32, 2, 32), ("TQC", 32, 2, 32), ("TD3", 64, 3, 300), ("SAC", 64, 3, 130)])
def test_update_many_is_bitwise_repeated_update(gcrl, kind, H, L, B, use_graph):
    gstep = 5                      # SAC: Polyak on steps 5, 10, ... (src/agent.py:681); chunks below cross it and the 8-step graph runs
    cfg = _cfg(kind, H, L, B)
    one, many = _build(gcrl, kind, cfg, gstep, use_graph=use_graph), _build(gcrl, kind, cfg, gstep, use_graph=use_graph)
    chunks = [(1, 19), (20, 1), (21, 8), (29, 11)]          # 19 = 8 + 8 + 3 steps of graph runs; 39 steps in all
    t_many, t_one = [], []
    for s0, n in chunks:
        t_many += [tuple(float(x) for x in t) for t in many.update_many(s0, n)]
    for step in range(1, 40):
        t_one.append(tuple(float(x) for x in one.update(step)))
    assert [len(t) for t in t_one] == [len(t) for t in t_many]
    if kind == "TD3":
        assert {len(t) for t in t_one} == {6, 8}            # critic-only and actor steps both occurred
    for step, (a, b) in enumerate(zip(t_one, t_many), start=1):
        assert a == b, (kind, step, a, b)
    for x, y in zip(_state(one), _state(many)):
        assert np.array_equal(x, y)
    if kind in ("SAC", "TQC"):
        assert any(t[-1] != 0.0 for t in t_one) and t_one[0][-1] == 0.0       # alpha_loss: off for step <= alpha_min_steps, live after


@pytest.mark.parametrize("kind,H,L,B", [("TD3", 256, 3, 2048), ("TQC", 512, 3, 2048), ("SAC", 256, 3, 512)])
def test_update_many_is_bitwise_at_the_benchmarked_sizes(gcrl, kind, H, L, B):
    """The same identity at BASELINE's shapes, where the round-3 forms run: dW reductions split over workgroups with a ticket
    fix-up (TD3 / TQC at batch 2048: the result must not depend on which workgroup arrives last), the BatchNorm slab launches
    (SAC at batch 512).  update_many (multi-step graphs) vs one update() per step: tuples and every parameter bitwise."""
    gstep = 4
    cfg = _cfg(kind, H, L, B, max_len=20000)

    def build():
        ag = _cls(gcrl, kind)(S, A, cfg, None, nenvs=2, gradient_step=gstep, rng="engine", seed=33)
        gen = np.random.default_rng(5)
        ep = 0
        while len(ag.buffer) < B + 500:
            for st in her_oracle.synthetic_episode(gen, 50, S, A):
                ag.push_her(ep % 2, *st)
            ep += 1
        return ag

    one, many = build(), build()
    t_many = [tuple(float(x) for x in t) for t in many.update_many(1, 5)] + [tuple(float(x) for x in t) for t in many.update_many(6, 4)]
    t_one = [tuple(float(x) for x in one.update(step)) for step in range(1, 10)]
    for step, (a, b) in enumerate(zip(t_one, t_many), start=1):
        assert a == b, (kind, step, a, b)
    for x, y in zip(_state(one), _state(many)):
        assert np.array_equal(x, y)
    assert all(np.isfinite(v) for t in t_one for v in t)


def _headline_ddpg(gcrl, orc=None, seed=1898, **kw):
    """bench.py's headline call: DDPG, PickAndPlace dims (S 23, A 4), H 256, L 3, B 256, k_future 8, gradient_step 40."""
    Sh, Ah, B = 23, 4, 256
    cfg = make_config("DDPG", hidden_dim=256, layer_count=3, batch_size=B, max_len=20000, grad_clip=10.0, tau=0.05, k_future=8)
    ag = gcrl.DDPG(Sh, Ah, cfg, None, nenvs=2, gradient_step=40, rng="engine", seed=seed, **kw)
    gen = np.random.default_rng(7)
    for ep in range(6):
        for st in her_oracle.synthetic_episode(gen, 50, Sh, Ah):
            ag.push_her(ep % 2, *st)
            if orc is not None:
                orc.push_her(ep % 2, *st)
    return ag, cfg


def test_the_benchmarked_ddpg_call_is_bitwise_repeated_update(gcrl):
    """the call bench.py times — pipelined `update_many` at S 23 / A 4 / H 256 / B 256 / gradient_step 40 — was pinned only
    by transitivity (a sequential `update()` at these dims against the reference; pipelined == sequential at S 10 / A 3).  Here:
    update_many(1, 40) + update_many(41, 40) (a trainer cycle each: K-only, 39 merged launches — the fused dW + optimiser launch
    among them — P-only, the step-40 / step-80 Polyak updates inside the overlapped schedule, the deferred draw) against 80 x
    update(): every tuple, parameter, target and Adam moment bitwise (reference: src/agent.py:1378-1404, trainer loop
    src/env.py:384-385)."""
    one, _ = _headline_ddpg(gcrl)
    many, _ = _headline_ddpg(gcrl)
    t_one = [tuple(float(x) for x in one.update(step)) for step in range(1, 81)]
    t_many = [tuple(float(x) for x in t) for t in many.update_many(1, 40)] + [tuple(float(x) for x in t) for t in many.update_many(41, 40)]
    for step, (a, b) in enumerate(zip(t_one, t_many), start=1):
        assert a == b, (step, a, b)
    for x, y in zip(_state(one), _state(many)):
        assert np.array_equal(x.view(np.uint32), y.view(np.uint32))
    for name in ("adam_m:actor", "adam_v:actor", "adam_m:critic_0", "adam_v:critic_0"):
        assert np.array_equal(one.actor._get(name), many.actor._get(name)), name
    assert all(np.isfinite(v) for t in t_one for v in t)


def test_the_benchmarked_ddpg_call_tracks_the_oracle(gcrl):
    """... and five sampled steps of that call against OracleAgent fed the same pushes and the same MT index stream: the first
    step (same parameters on both sides) at the north star's 1e-5, the following ones as a trajectory (Adam amplifies an fp32
    rounding of a ~0 gradient into a full lr-sized step of that weight)."""
    Sh, Ah, seed = 23, 4, 1898
    cfg = make_config("DDPG", hidden_dim=256, layer_count=3, batch_size=256, max_len=20000, grad_clip=10.0, tau=0.05, k_future=8)
    orc = OracleAgent("DDPG", Sh, Ah, cfg, nenvs=2, gradient_step=40, rng=random.Random(seed))
    ag, _ = _headline_ddpg(gcrl, orc=orc, seed=seed)
    ag.actor.set_flat(orc.flat_params(orc.actor))
    ag.critic.set_flat(orc.flat_params(orc.critics[0]))
    ag.update_target_network()
    orc.hard_update()
    outs = ag.update_many(1, 5)
    for k, info in enumerate(outs):
        got = np.array([float(x) for x in info])
        want = np.array([float(np.asarray(x)) for x in orc.update(k + 1)])
        tol = 1e-5 if k == 0 else 2e-3
        assert np.allclose(got, want, rtol=tol, atol=tol * 1e-1), (k + 1, got, want)


def test_round4_layer_path_forms_change_nothing_but_metric_roundings(gcrl, monkeypatch):
    """TQC at BASELINE cfg 4's shape with round 4's layer-per-launch forms (BatchNorm partials out of the tiled GEMM's epilogue,
    multi-workgroup td_loss / actor_select_alpha, control advance riding the actor's optimiser launch) against the same agent
    with the three knobs that restore round 3's launches: every parameter, running statistic and alpha bitwise equal — the
    fused partials are the sums bn_stats_kernel forms, in its order; the reductions feed logged numbers only — and the logged
    numbers equal to rounding."""
    kind, H, L, B, gstep = "TQC", 512, 3, 2048, 4
    cfg = _cfg(kind, H, L, B, max_len=20000)

    def build():
        ag = _cls(gcrl, kind)(S, A, cfg, None, nenvs=2, gradient_step=gstep, rng="engine", seed=33)
        gen = np.random.default_rng(5)
        ep = 0
        while len(ag.buffer) < B + 500:
            for st in her_oracle.synthetic_episode(gen, 50, S, A):
                ag.push_her(ep % 2, *st)
            ep += 1
        return ag

    new = build()
    for k in ("GCRL_NO_BN_TILED_STATS", "GCRL_NO_LAYER_ADV", "GCRL_NO_MB_REDUCE"):
        monkeypatch.setenv(k, "1")
    old = build()
    t_new = [tuple(float(x) for x in t) for t in new.update_many(1, 9)]
    t_old = [tuple(float(x) for x in t) for t in old.update_many(1, 9)]
    for x, y in zip(_state(new), _state(old)):
        assert np.array_equal(x, y)
    for a, b in zip(t_new, t_old):
        assert len(a) == len(b) and np.allclose(a, b, rtol=2e-6, atol=1e-7), (a, b)
    assert all(np.isfinite(v) for t in t_new for v in t)


def _engine_noise(seed, stream, step_index, B, n_cols):
    """What the engine draws for the step planned `step_index`-th since the agent was created (csrc/agent.hip plan_step:
    counter base += 16 * B per step; csrc/ops.h hash_normal(seed + stream, base + b * A + j)): TD3 smoothing noise is stream 0,
    SAC / TQC eps of actor.sample(next_state) stream 1, of actor.sample(state) stream 2."""
    base = np.uint64(step_index) * np.uint64(16 * B)
    return hash_normal(seed + stream, base + np.arange(B * n_cols, dtype=np.uint64)).reshape(B, n_cols)


@pytest.mark.parametrize("kind", ["TD3", "SAC", "TQC"])
def test_sampled_noisy_updates_track_oracle(gcrl, kind):
    """push -> flush -> sample -> update with DEVICE noise: 4 steps through update_many against OracleAgent.update fed the same
    MT-sampled batches and the noise values the engine drew."""
    B, H, L, seed = 64, 32, 2, 5
    cfg = _cfg(kind, H, L, B, max_len=5000, grad_clip=5.0, alpha_min_steps=0.0, alpha_lr=3e-4)
    ag = _cls(gcrl, kind)(S, A, cfg, None, nenvs=2, gradient_step=4, rng="engine", seed=seed)
    orc = OracleAgent(kind, S, A, cfg, nenvs=2, gradient_step=4, rng=random.Random(seed))
    gen = np.random.default_rng(1)
    for ep in range(4):
        for st in her_oracle.synthetic_episode(gen, 50, S, A):
            ag.push_her(ep % 2, torch.from_numpy(st[0]).cuda(), *st[1:])
            orc.push_her(ep % 2, *st)
    if kind in ("SAC", "TQC"):
        # trained-policy-like heads (std ~ 0.4, small means): with fresh Xavier heads |pre-tanh| runs past 4, where
        # log(1 - tanh^2 + 1e-8) has no fp32 precision left in the reference itself
        with torch.no_grad():
            orc.actor.mean_head.weight.mul_(0.2); orc.actor.log_std_head.weight.mul_(0.1); orc.actor.log_std_head.bias.fill_(-0.9)
    ag.actor.set_flat(orc.flat_params(orc.actor))
    for i, c in enumerate(orc.critics):
        ag.critics[i].set_flat(orc.flat_params(c))
    ag.update_target_network()
    orc.hard_update()
    outs = ag.update_many(1, 4)
    worst = 0.0
    for k, (step, info) in enumerate(zip((1, 2, 3, 4), outs)):
        kw = {}
        if kind == "TD3":
            kw["noise"] = torch.from_numpy(_engine_noise(seed, 0, k, B, A))
        else:
            kw["eps_next"] = torch.from_numpy(_engine_noise(seed, 1, k, B, A))
            kw["eps_cur"] = torch.from_numpy(_engine_noise(seed, 2, k, B, A))
        ref = orc.update(step, **kw)
        got = np.array([float(x) for x in info])
        want = np.array([float(np.asarray(x)) for x in ref])
        assert got.shape == want.shape
        worst = max(worst, float(np.max(np.abs(got - want) / np.maximum(np.abs(want), 1e-2))))
        assert np.allclose(got, want, rtol=5e-5, atol=5e-6), (kind, step, got, want)
    print(f"sampled noisy updates [{kind}]: worst relative tuple error {worst:.2e}")


@pytest.mark.parametrize("kind,H,L,B", [("SAC", 256, 3, 512), ("TD3", 256, 3, 2048), ("DDPG", 256, 3, 256), ("TQC", 64, 2, 200)])
def test_failed_meeting_inside_a_launch_is_reported_and_the_handle_recovers(gcrl, tmp_path, kind, H, L, B):
    """TQC at batch 200 (layer-per-launch critics, round 5): the only waits are those of the BatchNorm slab launches, whose row groups
    exchange their column partials as words that are their own flags (csrc/bn_slab.hip slab_exchange_df) — the hook makes row group 1 of
    slab 0 withhold its words once.  SAC at batch 512: the BatchNorm slab launches split their rows over four workgroups that wait for each other inside the
    launch, and so do the role workgroups of the twin-critic row chains (csrc/meet.h).  TD3 at batch 2048: the critic phase's
    online-critic workgroups wait for the target roles of their rows (producers / consumers: 1 024 workgroups, not all resident);
    DDPG at batch 256 (round 4): the same two roles inside the fused launch (k_split).  A wait that times out used to leave NaN
    statistics / gradients and nothing else: now the status word makes the next synchronising call fail ONCE with
    GCRL_ERR_STATE — the reference raises on any failed step (src/agent.py:659-699) — and the handle works again afterwards.
    The fault is injected by knocking one meeting counter off its multiple-of-arrivals state (gcrl_agent_debug_meet_fault)."""
    from gcrl_amd import _ffi
    cfg = _cfg(kind, H, L, B, max_len=20000)
    ag = _cls(gcrl, kind)(S, A, cfg, None, nenvs=2, gradient_step=4, rng="engine", seed=33)
    gen = np.random.default_rng(5)
    ep = 0
    while len(ag.buffer) < B + 500:
        for st in her_oracle.synthetic_episode(gen, 50, S, A):
            ag.push_her(ep % 2, *st)
        ep += 1
    active = ag.set_meetings(True)
    if not active:
        pytest.skip("no launch form with an in-kernel wait is admissible on this device (shared GPU?)")
    good = [float(x) for x in ag.update(1)]
    assert all(np.isfinite(good))
    ag.save_state(str(tmp_path / "ckpt"))
    _ffi.check(_ffi.lib.gcrl_agent_debug_meet_fault(ag._h))
    t = ag.update(2)
    with pytest.raises(_ffi.GcrlError, match="timed out"):
        [float(x) for x in t]
    # the error was consumed and the counters were reset: the next steps run and are finite again (the parameters took one
    # poisoned step: back to the last checkpoint first, what a trainer would do)
    ag.load_state(str(tmp_path / "ckpt"))
    after = [float(x) for x in ag.update(3)]
    assert all(np.isfinite(after)), after
    # ... and with the meetings switched off the same step sequence never waits
    assert ag.set_meetings(False) == 0
    off = [float(x) for x in ag.update(4)]
    assert all(np.isfinite(off)), off
