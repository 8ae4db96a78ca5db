"""GPU (-m gpu): the hot path driven the way the reference's trainer drives it (examples/trainer_standin.py
reproduces the call pattern of src/env.py:334-406 / :163-232 on a synthetic goal environment): vectorised
acting, one push per env step for all envs, episode flush + HER relabelling on the device, update_many per
cycle.  The check is end to end and behavioural: the agents must actually learn the task."""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(ROOT, "examples"))


@pytest.mark.parametrize("agent,cycles,floor", [("TD3", 200, 0.7), ("DDPG", 400, 0.5), ("SAC", 200, 0.6), ("TQC", 120, 0.6)])
def test_agents_learn_point_reach_through_the_engine(gcrl, agent, cycles, floor):
    import trainer_standin
    out = trainer_standin.train(agent, cycles=cycles, seed=0, verbose=False)
    early = float(np.mean(out["success_per_cycle"][:20]))
    late = float(np.mean(out["success_per_cycle"][-10:]))
    assert early < 0.2                      # nothing is solved before learning
    assert late >= floor, (agent, late)     # (20 % of acting steps are random by construction: ~0.8-1.0 is solved)
    assert out["gradient_steps"] == cycles * 40
