"""GPU (-m gpu): the weight-gradient GEMMs, the global-norm clip and the optimiser step of the row-chain DDPG step as ONE launch
(csrc/dw_adam.hip, round 5) against the two launches it replaces (GCRL_NO_OPT_FUSE=1: batched dW | db GEMMs, then Adam).

The reference's sequence is backward -> clip_grad_norm_ -> optimizer.step() (src/agent.py:1326-1333 critic, :1288-1300 actor):
only the global norm stands between a gradient element and its parameter's step.  A workgroup of the fused launch keeps its
16 x 16 gradient tile in registers, publishes the tile's sum of squares into its own slot (the data is the flag), waits until every
slot of its net is there, adds them up in the order the optimiser launch adds the GEMM launch's partials, and steps its elements.
What has to hold: the same BITS as the two-launch form — tuples, parameters, targets, Adam moments, the stored gradients — over
K-only, merged and P-only launches, Polyak steps and graph replays; the reference's full-size fixtures; and a slot that never
arrives is an error at the next synchronising call (src/agent.py:659-699: the reference raises on any failed step), after which
the handle works again."""
import os

import numpy as np
import pytest

from fullsize import Case, Report, compare
from test_gpu_full_size import build, run
from test_gpu_parity import _ddpg_for_schedules, _run_many

pytestmark = pytest.mark.gpu


def _agent(gcrl, monkeypatch, fused, H, L, B, **kw):
    monkeypatch.delenv("GCRL_ROWTILE", raising=False)
    if fused:
        monkeypatch.delenv("GCRL_NO_OPT_FUSE", raising=False)
    else:
        monkeypatch.setenv("GCRL_NO_OPT_FUSE", "1")
    ag = _ddpg_for_schedules(gcrl, H, L, 2, B=B, **kw)
    active = ag.meetings()
    if fused and not (active & 8):
        pytest.skip("the fused optimiser launch is not admissible on this device (shared GPU, or too few CUs for its workgroups)")
    assert fused or not (active & 8)
    return ag


def _everything(ag):
    out = [v.flat() for v in (ag.actor, ag.critic, ag.target_actor, ag.target_critic)]
    for name in ("grad:actor", "grad:critic_0", "adam_m:actor", "adam_v:actor", "adam_m:critic_0", "adam_v:critic_0"):
        out.append(ag.actor._get(name))
    return out


@pytest.mark.parametrize("H,L,B,S,A", [(256, 3, 256, 23, 4), (64, 3, 64, 10, 3), (128, 2, 40, 9, 2), (72, 4, 50, 12, 5)])
def test_fused_optimiser_launch_is_bitwise_the_two_launch_form(gcrl, monkeypatch, H, L, B, S, A):
    """90 pipelined steps at the headline's shape (S 23, A 4, H 256, L 3, B 256) and three others (ragged tiles: 40 rows, 72 and
    9 + 2 columns): every tuple, parameter, target, Adam moment and stored gradient bitwise equal to the two-launch form."""
    two = _agent(gcrl, monkeypatch, False, H, L, B, S=S, A=A)
    ref = _run_many(two)
    one = _agent(gcrl, monkeypatch, True, H, L, B, S=S, A=A)
    got = _run_many(one)
    assert len(got) == len(ref) == 90
    for i, (x, y) in enumerate(zip(ref, got)):
        assert x == y, (i + 1, x, y)
    for i, (x, y) in enumerate(zip(_everything(two), _everything(one))):
        assert np.array_equal(x.view(np.uint32), y.view(np.uint32)), i
    assert np.all(np.isfinite(np.array(got)))


def test_fused_optimiser_update_equals_update_many(gcrl, monkeypatch):
    """update() per step (K-only then P-only fused launches) and the pipelined update_many (the paired launch) are the same
    arithmetic: bitwise equal tuples and parameters over 90 steps, eager and replayed from graphs."""
    a_seq, a_pipe = _agent(gcrl, monkeypatch, True, 64, 3, 64), _agent(gcrl, monkeypatch, True, 64, 3, 64)
    a_eager = _agent(gcrl, monkeypatch, True, 64, 3, 64, use_graph=False)
    seq = [tuple(float(x) for x in a_seq.update(s)) for s in range(1, 91)]
    assert seq == _run_many(a_pipe) == _run_many(a_eager)
    for x, y, z in zip(_everything(a_seq), _everything(a_pipe), _everything(a_eager)):
        assert np.array_equal(x, y) and np.array_equal(x, z)


@pytest.mark.parametrize("name", ["ddpg_pickplace_b256", "cfg1_ddpg_reach_b256", "cfg2_ddpg_reach_b1024"])
def test_fused_optimiser_launch_matches_the_reference_at_full_size(gcrl, monkeypatch, name):
    """The headline shape, BASELINE cfg 1 and cfg 2 against the fixtures captured from the reference
    (tests/golden/make_golden_full.py): the criterion of tests/test_gpu_full_size.py, and the flat 1e-5."""
    monkeypatch.delenv("GCRL_NO_OPT_FUSE", raising=False)
    c = Case(name)
    ag, views = build(gcrl, c)
    if not (ag.meetings() & 8):
        pytest.skip("the fused optimiser launch is not admissible on this device")
    rep = Report("hip/fused_optimiser", name)
    compare(c, rep, *run(c, ag, views))
    print(rep.summary())
    assert not rep.bad, rep.bad[:8]
    assert not rep.beyond_flat, rep.beyond_flat[:8]


def test_norm_slot_that_never_arrives_is_reported(gcrl, monkeypatch, tmp_path):
    """gcrl_agent_debug_meet_fault makes one workgroup of the next fused launch keep its norm slot to itself — what a workgroup
    held off the chip would cause: every workgroup of the critic's net waits (bounded), the step is poisoned, the status word
    turns the next synchronising call into GCRL_ERR_STATE once; the slots are re-initialised and the next steps are finite."""
    from gcrl_amd import _ffi
    monkeypatch.setenv("GCRL_NO_DDPG_KSPLIT", "1")   # (so that the fused optimiser launch is the only launch with a wait)
    ag = _agent(gcrl, monkeypatch, True, 256, 3, 256, S=23, A=4)
    assert ag.meetings() == 8
    good = [float(x) for x in ag.update(1)]
    assert all(np.isfinite(good))
    ag.save_state(str(tmp_path / "ckpt"))
    _ffi.check(_ffi.lib.gcrl_agent_debug_meet_fault(ag._h))
    t = ag.update(2)
    with pytest.raises(_ffi.GcrlError, match="timed out"):
        [float(x) for x in t]
    ag.load_state(str(tmp_path / "ckpt"))   # (back to the last checkpoint, as a trainer would)
    after = [float(x) for x in ag.update(3)]
    assert all(np.isfinite(after)), after
    more = [tuple(float(x) for x in tt) for tt in ag.update_many(4, 40)]
    assert np.all(np.isfinite(np.array(more)))


def test_meetings_switch_selects_the_two_launch_form(gcrl, monkeypatch):
    """gcrl_agent_set_meetings(0) — what DataParallelUpdater does for ranks that share a device — takes the fused launch out (it
    contains a wait); back on, the agent continues on the fused form: one trajectory, bitwise, whichever form ran each step."""
    a, b = _agent(gcrl, monkeypatch, True, 64, 3, 64), _agent(gcrl, monkeypatch, True, 64, 3, 64)
    ref = [tuple(float(x) for x in t) for t in a.update_many(1, 30)]
    got = [tuple(float(x) for x in t) for t in b.update_many(1, 10)]
    assert not (b.set_meetings(False) & 8)
    got += [tuple(float(x) for x in t) for t in b.update_many(11, 10)]
    assert b.set_meetings(True) & 8
    got += [tuple(float(x) for x in t) for t in b.update_many(21, 10)]
    assert got == ref
    for x, y in zip(_everything(a), _everything(b)):
        assert np.array_equal(x, y)


@pytest.mark.parametrize("kind,H,L,B", [("TD3", 64, 3, 300), ("SAC", 64, 3, 130), ("SAC", 256, 3, 512), ("TD3", 128, 2, 40)])
def test_fused_optimiser_launch_for_twin_critics_is_bitwise_the_two_launch_form(gcrl, monkeypatch, kind, H, L, B):
    """TD3 / SAC on the row-chain path: both critics' dW | db + clip + AdamW (+ Polyak; TD3's critic_1 unclipped, src/agent.py:201)
    as one launch at the end of the critic phase, TD3's actor likewise — 39 steps of update_many (multi-step graphs, SAC's
    step % gradient_step Polyak, TD3's delayed actor, the alpha branch switching on) against GCRL_NO_OPT_FUSE=1: tuples and every
    parameter, target, BatchNorm statistic bitwise."""
    import test_gpu_multistep as ms
    cfg = ms._cfg(kind, H, L, B, max_len=20000 if B > 200 else 4000)

    def build():
        ag = ms._cls(gcrl, kind)(ms.S, ms.A, cfg, None, nenvs=2, gradient_step=5, rng="engine", seed=21)
        gen = np.random.default_rng(3)
        ep = 0
        while len(ag.buffer) < B + 300:
            for st in ms.her_oracle.synthetic_episode(gen, 50, ms.S, ms.A):
                ag.push_her(ep % 2, *st)
            ep += 1
        return ag

    monkeypatch.setenv("GCRL_NO_OPT_FUSE", "1")
    two = build()
    assert not (two.meetings() & 8)
    monkeypatch.delenv("GCRL_NO_OPT_FUSE")
    one = build()
    if not (one.meetings() & 8):
        pytest.skip("the fused optimiser launch is not admissible on this device")
    chunks = [(1, 19), (20, 1), (21, 8), (29, 11)]
    t_two, t_one = [], []
    for s0, n in chunks:
        t_two += [tuple(float(x) for x in t) for t in two.update_many(s0, n)]
        t_one += [tuple(float(x) for x in t) for t in one.update_many(s0, n)]
    for step, (x, y) in enumerate(zip(t_two, t_one), start=1):
        assert x == y, (kind, step, x, y)
    for x, y in zip(ms._state(two), ms._state(one)):
        assert np.array_equal(x.view(np.uint32), y.view(np.uint32))
    assert all(np.isfinite(v) for t in t_one for v in t)


def test_a_second_process_on_the_device_switches_the_waiting_forms_off(gcrl, monkeypatch, tmp_path):
    """launch forms whose workgroups wait for each other assume the process has the GPU to itself; that used to
    be detected under DataParallelUpdater only.  Now every process holds a shared lock on a per-device presence file
    (csrc/abi_misc.hip meet_probe_device): a second process that creates a handle on the same device gets the forms without waits by
    itself, and the process that was there first notices at its next probe (get_meetings, every 32nd update call)."""
    import subprocess
    import sys
    import textwrap
    first = _agent(gcrl, monkeypatch, True, 64, 3, 64)
    if not first.meetings():
        pytest.skip("no launch form with an in-kernel wait is admissible on this device")
    child = textwrap.dedent("""
        import os, sys
        sys.path.insert(0, %r)
        import numpy as np
        import gcrl_amd
        from oracle.agent_oracle import make_config
        cfg = make_config("DDPG", hidden_dim=64, layer_count=3, batch_size=64, max_len=3000)
        ag = gcrl_amd.DDPG(10, 3, cfg, None, nenvs=1, gradient_step=4, rng="engine", seed=1)
        print("child meetings", ag.meetings(), flush=True)
        sys.stdin.readline()
    """) % (str(__import__("pathlib").Path(__file__).resolve().parents[1]),)
    env = dict(os.environ)      # (same TMPDIR: the presence file is found there)
    p = subprocess.Popen([sys.executable, "-c", child], stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True, env=env)
    try:
        line = p.stdout.readline()
        assert line.strip() == "child meetings 0", line          # the newcomer saw the presence lock of this process
        assert first.meetings() == 0                              # ... and this process sees the newcomer's
        out = [tuple(float(x) for x in t) for t in first.update_many(1, 8)]
        assert np.all(np.isfinite(np.array(out)))
    finally:
        p.stdin.write("\n"); p.stdin.flush()
        p.wait(timeout=60)
    gcrl._ffi.lib.gcrl_set_shared_device(0)                       # (process-wide switch: back for the tests that follow)
    assert first.set_meetings(True) != 0


@pytest.mark.parametrize("kind,H,L,B", [("SAC", 256, 3, 512), ("SAC", 64, 3, 130), ("TQC", 32, 2, 40)])
def test_heads_and_sampling_as_one_launch_are_bitwise_the_two_launches(gcrl, monkeypatch, kind, H, L, B):
    """SACActorModel's mean / log_std heads (src/model.py:114-115) and sample() (:125-141) as one launch (csrc/sac_heads.h: the
    heads' tiles by the batched GEMM's own tile body, one thread per (row, action) for the sampling arithmetic, the log-prob summed
    in action order) against the GEMM launch + sampling launch of rounds 1-4 (GCRL_NO_HEADS_FUSED=1): 24 steps with device noise,
    every tuple, parameter, BatchNorm statistic and alpha bitwise."""
    import test_gpu_multistep as ms
    cfg = ms._cfg(kind, H, L, B, max_len=20000 if B > 200 else 4000)

    def build():
        ag = ms._cls(gcrl, kind)(ms.S, ms.A, cfg, None, nenvs=2, gradient_step=5, rng="engine", seed=21)
        gen = np.random.default_rng(3)
        ep = 0
        while len(ag.buffer) < B + 300:
            for st in ms.her_oracle.synthetic_episode(gen, 50, ms.S, ms.A):
                ag.push_her(ep % 2, *st)
            ep += 1
        return ag

    monkeypatch.setenv("GCRL_NO_HEADS_FUSED", "1")
    two = build()
    monkeypatch.delenv("GCRL_NO_HEADS_FUSED")
    one = build()
    t_two = [tuple(float(x) for x in t) for t in two.update_many(1, 24)]
    t_one = [tuple(float(x) for x in t) for t in one.update_many(1, 24)]
    for step, (x, y) in enumerate(zip(t_two, t_one), start=1):
        assert x == y, (kind, step, x, y)
    for x, y in zip(ms._state(two), ms._state(one)):
        assert np.array_equal(x.view(np.uint32), y.view(np.uint32))
    assert all(np.isfinite(v) for t in t_one for v in t)


def test_two_agents_queued_back_to_back_do_not_wait_on_each_other(gcrl, monkeypatch):
    """Two handles of one process, their update calls queued WITHOUT a synchronisation in between (each agent has its own stream):
    launches whose workgroups wait for each other must not end up side by side on the device.  The library orders a handle's call
    after the other handle's last one (an event wait on the device, csrc/agent.hip order_after_other_handles): 12 interleaved
    40-step calls finish without a timed-out wait and give the trajectories the agents produce when run one after the other."""
    a1, a2 = _agent(gcrl, monkeypatch, True, 256, 3, 256, S=23, A=4), _agent(gcrl, monkeypatch, True, 256, 3, 256, S=23, A=4)
    b1, b2 = _agent(gcrl, monkeypatch, True, 256, 3, 256, S=23, A=4), _agent(gcrl, monkeypatch, True, 256, 3, 256, S=23, A=4)
    pend = []
    for c in range(6):                       # interleaved, nothing fetched until the end
        pend.append((0, a1.update_many(1 + 40 * c, 40)))
        pend.append((1, a2.update_many(1 + 40 * c, 40)))
    got = [[], []]
    for who, tickets in pend:
        got[who] += [tuple(float(x) for x in t) for t in tickets]
    ref = [[], []]
    for who, ag in ((0, b1), (1, b2)):       # one after the other, fetched call by call
        for c in range(6):
            ref[who] += [tuple(float(x) for x in t) for t in ag.update_many(1 + 40 * c, 40)]
    assert got == ref
    assert np.all(np.isfinite(np.array(got)))
