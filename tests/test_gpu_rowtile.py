"""GPU (-m gpu): the weight-slice form of the DDPG launch (csrc/rowtile.hip, opt-in: GCRL_ROWTILE=1) against the row-chain
launch it replaces and against the reference's full-size fixture of the headline shape.

A workgroup owns the 16 x 16 tile of every layer of its role's chain and the workgroups of a row block hand the layers to
each other INSIDE the launch (the data is the flag: 0xFFFFFFFF = not written yet).  What has to hold: the same numbers as the
row-chain launch up to the summation order of a dot product; the same BITS from run to run (a hand-off must never change what
is summed, or in which order); hipGraph replays and the K-only / P-only / merged launch shapes of the pipeline leave every
hand-off word ready for the next launch; a hand-off that never arrives is an error at the next synchronising call
(src/agent.py:659-699: the reference raises on any failed step), after which the handle works again."""
import numpy as np
import pytest
import torch

from fullsize import Case, Report, compare
from test_gpu_full_size import build, run
from test_gpu_parity import _ddpg_for_schedules, _run_many

pytestmark = pytest.mark.gpu


def _tile_agent(gcrl, monkeypatch, on, H, L, B, **kw):
    if on:
        monkeypatch.setenv("GCRL_ROWTILE", "1")
    else:
        monkeypatch.delenv("GCRL_ROWTILE", raising=False)
    ag = _ddpg_for_schedules(gcrl, H, L, 2, B=B, **kw)
    active = ag.set_meetings(True)
    if on and not (active & 4):
        pytest.skip("the weight-slice launch is not admissible on this device (shared GPU, or too few CUs for its workgroups)")
    assert on or not (active & 4)
    return ag


@pytest.mark.parametrize("H,L,B", [(64, 3, 64), (256, 3, 256), (128, 2, 32), (192, 4, 48)])
def test_weight_slice_launch_tracks_the_row_chain_launch(gcrl, monkeypatch, H, L, B):
    """90 pipelined steps (K-only, merged and P-only launches, two Polyak boundaries, graph replays) with the weight-slice
    launch against the row-chain launch: the same trajectory while fp32 reordering allows, finite throughout; and twice the
    weight-slice run: bitwise the same."""
    ref = np.array(_run_many(_tile_agent(gcrl, monkeypatch, False, H, L, B)))
    a1 = _tile_agent(gcrl, monkeypatch, True, H, L, B)
    got = np.array(_run_many(a1))
    assert np.allclose(ref[:5], got[:5], rtol=2e-5, atol=1e-6), np.abs(ref[:5] - got[:5]).max()
    assert np.all(np.isfinite(got))
    a2 = _tile_agent(gcrl, monkeypatch, True, H, L, B)
    again = np.array(_run_many(a2))
    assert np.array_equal(got, again)
    for v1, v2 in [(a1.actor, a2.actor), (a1.critic, a2.critic), (a1.target_actor, a2.target_actor), (a1.target_critic, a2.target_critic)]:
        assert np.array_equal(v1.flat(), v2.flat())


def test_weight_slice_update_equals_update_many(gcrl, monkeypatch):
    """update() per step (K-only then P-only launches) and the pipelined update_many (merged launches) are the same arithmetic:
    bitwise equal tuples and parameters over 90 steps."""
    a_seq, a_pipe = _tile_agent(gcrl, monkeypatch, True, 64, 3, 64), _tile_agent(gcrl, monkeypatch, True, 64, 3, 64)
    seq = [tuple(float(x) for x in a_seq.update(s)) for s in range(1, 91)]
    assert seq == _run_many(a_pipe)
    assert np.array_equal(a_seq.actor.flat(), a_pipe.actor.flat()) and np.array_equal(a_seq.critic.flat(), a_pipe.critic.flat())


@pytest.mark.parametrize("name", ["ddpg_pickplace_b256", "cfg1_ddpg_reach_b256"])
def test_weight_slice_launch_matches_the_reference_at_full_size(gcrl, monkeypatch, name):
    """The headline shape (PickAndPlace, H = 256, L = 3, B = 256) and BASELINE cfg 1 against the fixtures captured from the
    reference (tests/golden/make_golden_full.py): the criterion of tests/test_gpu_full_size.py, and the flat 1e-5."""
    monkeypatch.setenv("GCRL_ROWTILE", "1")
    c = Case(name)
    ag, views = build(gcrl, c)
    if not (ag.set_meetings(True) & 4):
        pytest.skip("the weight-slice launch is not admissible on this device")
    rep = Report("hip/weight_slice", name)
    compare(c, rep, *run(c, ag, views))
    print(rep.summary())
    assert not rep.bad, rep.bad[:8]
    assert not rep.beyond_flat, rep.beyond_flat[:8]


def test_weight_slice_hand_off_that_never_arrives_is_reported(gcrl, monkeypatch, tmp_path):
    """gcrl_agent_debug_meet_fault knocks a row block's first-arrival counter off its multiple-of-arrivals state: seven of its
    workgroups wait (bounded) for arrivals that never come, their row block's hand-offs run late and time out too.  The status
    word turns the next synchronising call into GCRL_ERR_STATE once; the hand-off words are re-initialised and the next steps
    are finite again."""
    from gcrl_amd import _ffi
    ag = _tile_agent(gcrl, monkeypatch, True, 256, 3, 256)
    good = [float(x) for x in ag.update(1)]
    assert all(np.isfinite(good))
    ag.save_state(str(tmp_path / "ckpt"))
    _ffi.check(_ffi.lib.gcrl_agent_debug_meet_fault(ag._h))
    t = ag.update(2)
    with pytest.raises(_ffi.GcrlError, match="timed out"):
        [float(x) for x in t]
    ag.load_state(str(tmp_path / "ckpt"))   # (the step that timed out may have taken garbage: back to the last checkpoint, as a trainer would)
    after = [float(x) for x in ag.update(3)]
    assert all(np.isfinite(after)), after
    more = [tuple(float(x) for x in tt) for tt in ag.update_many(4, 40)]
    assert np.all(np.isfinite(np.array(more)))


@pytest.mark.parametrize("H,L,B", [(64, 3, 64), (256, 3, 256)])
def test_two_role_critic_phase_is_bitwise_the_one_role_launch(gcrl, monkeypatch, H, L, B):
    """DDPG's critic phase as two roles of the fused row-chain launch (target chain | online critic, producers / consumers:
    csrc/rowchain.hip k_split, the default) against the launch that walks a row block through both (GCRL_NO_DDPG_KSPLIT=1): the
    same per-row arithmetic in the same order — 90 pipelined steps, every tuple and every parameter bitwise equal."""
    monkeypatch.delenv("GCRL_ROWTILE", raising=False)
    monkeypatch.setenv("GCRL_NO_DDPG_KSPLIT", "1")
    a_one = _ddpg_for_schedules(gcrl, H, L, 2, B=B)
    assert not (a_one.meetings() & 2)
    one = _run_many(a_one)
    monkeypatch.delenv("GCRL_NO_DDPG_KSPLIT")
    a_two = _ddpg_for_schedules(gcrl, H, L, 2, B=B)
    if not (a_two.meetings() & 2):
        pytest.skip("launch forms with in-kernel waits are not admissible on this device")
    assert _run_many(a_two) == one
    for v1, v2 in [(a_one.actor, a_two.actor), (a_one.critic, a_two.critic), (a_one.target_critic, a_two.target_critic)]:
        assert np.array_equal(v1.flat(), v2.flat())
