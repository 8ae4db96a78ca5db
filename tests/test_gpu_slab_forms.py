"""GPU (-m gpu): the BatchNorm actor's slab launches (csrc/bn_slab.hip) in round 5's forms against round 4's.

[Linear -> BatchNorm1d(train) -> ReLU] of SACActorModel (src/model.py:100-123) and its backward run as one launch per layer and
direction.  Round 5 changed HOW, not what: 64-row workgroups (8 row groups per slab) with 8 load stages, the narrow layers split
as well, the row groups' column partials exchanged as words that are their own flags, W's tile of the backward pass through an LDS
image, and the sampling backward (src/model.py:125-141 differentiated; src/agent.py:516-521) folded into the top layer's backward
launch with the selection + log-alpha block riding along; and (SAC on the role-parallel row-chain launches) the two heads and the
sampling itself (src/model.py:114-115, :125-141) formed by the chain launches that consume the actions.  The knobs that select round 4's forms are read once per process, so each
form runs in a child process on the same seeds; the summation orders differ (merge of 8 partials instead of 4, multiply-adds
instead of MFMAs for the heads' K = 2 x action_dim contraction), so the first step is compared at 1e-5 and the trajectory with
a bound, as the oracle comparisons do."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import json, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
sys.path.insert(0, sys.argv[1] + "/tests")
import gcrl_amd as gcrl
import test_gpu_multistep as ms
kind, H, L, B, steps = sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
cfg = ms._cfg(kind, H, L, B, max_len=20000 if B > 200 else 4000)
ag = ms._cls(gcrl, kind)(ms.S, ms.A, cfg, None, nenvs=2, gradient_step=5, rng="engine", seed=21)
gen = np.random.default_rng(3)
ep = 0
while len(ag.buffer) < B + 300:
    for st in ms.her_oracle.synthetic_episode(gen, 50, ms.S, ms.A):
        ag.push_her(ep % 2, *st)
    ep += 1
tuples = [[float(x) for x in t] for t in ag.update_many(1, steps)]
state = [np.asarray(v, np.float32).ravel().tolist() for v in ms._state(ag)]
print("RESULT " + json.dumps({"tuples": tuples, "state": state, "meetings": int(ag.meetings())}))
"""

ROUND4 = {"GCRL_SLAB_WAVES": "8", "GCRL_SLAB_MEET": "1", "GCRL_NO_SLAB_WTILE": "1", "GCRL_NO_TG_FOLD": "1", "GCRL_NO_SLAB_SPLIT_ALL": "1",
          "GCRL_NO_HEADS_FOLD": "1"}


def _child(kind, H, L, B, steps, extra):
    env = dict(os.environ)
    for k in ROUND4:
        env.pop(k, None)
    env.update(extra)
    r = subprocess.run([sys.executable, "-c", CHILD, ROOT, kind, str(H), str(L), str(B), str(steps)], capture_output=True, text=True,
                       timeout=300, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1]
    return json.loads(line[len("RESULT "):])


@pytest.mark.parametrize("kind,H,L,B", [("SAC", 256, 3, 512), ("SAC", 64, 3, 130), ("TQC", 32, 2, 200)])
def test_round5_slab_forms_track_round4s(kind, H, L, B):
    steps = 12
    new = _child(kind, H, L, B, steps, {})
    old = _child(kind, H, L, B, steps, ROUND4)
    single = [_child(kind, H, L, B, steps, {k: v}) for k, v in (("GCRL_SLAB_MEET", "1"), ("GCRL_NO_TG_FOLD", "1"), ("GCRL_NO_HEADS_FOLD", "1"),
                                                                 ("GCRL_HEADS_CUR_IN_P", "1"))]
    a, b = np.array(new["tuples"]), np.array(old["tuples"])
    assert a.shape == b.shape and a.shape[0] == steps and np.all(np.isfinite(a))
    np.testing.assert_allclose(a[0], b[0], rtol=1e-5, atol=1e-6)                      # one step: summation-order noise only
    np.testing.assert_allclose(a, b, rtol=1e-2, atol=1e-3)                            # the trajectory stays together
    # parameters: Adam divides a gradient by its own running magnitude, so an element whose gradient is summation-order noise moves by up to
    # a learning rate per step either way — a few per mille of the elements; everything else stays within the trajectory bound
    for x, y in zip(new["state"], old["state"]):
        x, y = np.array(x), np.array(y)
        off = ~np.isclose(x, y, rtol=2e-3, atol=2e-4)
        assert np.abs(x - y).max() < steps * 1.5e-3, np.abs(x - y).max()
        assert x.size < 10000 or off.mean() < 0.01, off.mean()      # (the small arrays are BatchNorm's running statistics and alpha: they follow the weights)
    # the exchange through flag words moves the SAME partials as the counter meeting: bitwise the same run
    meet = single[0]
    assert meet["tuples"] == new["tuples"]
    for x, y in zip(meet["state"], new["state"]):
        assert x == y
    # the folded sampling backward changes one contraction's summation order only
    nofold = np.array(single[1]["tuples"])
    np.testing.assert_allclose(a[0], nofold[0], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(a, nofold, rtol=1e-2, atol=1e-3)
    # ... and so do the heads formed inside the chain launches (a row's 256-term dot products in another order)
    # pi(s) formed by the critic phase's online roles (default) or by the actor phase's critic roles: the same arithmetic on the same rows
    assert single[3]["tuples"] == new["tuples"]
    for x, y in zip(single[3]["state"], new["state"]):
        assert x == y
    noheads = np.array(single[2]["tuples"])
    np.testing.assert_allclose(a[0], noheads[0], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(a, noheads, rtol=1e-2, atol=1e-3)
