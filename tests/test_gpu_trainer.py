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


def test_fused_acting_side_learns_and_is_faster(gcrl):
    """The same loop with the acting side on the device (DeviceRunningNormalizer + observe_act + process_step: SURVEY.md
    §8f-3): it must learn just the same, and the acting phase must run at >= 1.5x the env steps/s of the separate calls."""
    import json
    import trainer_standin
    slow = trainer_standin.train("DDPG", cycles=150, seed=0, verbose=False)
    fast = trainer_standin.train("DDPG", cycles=400, seed=0, verbose=False, fused=True)
    assert float(np.mean(fast["success_per_cycle"][-10:])) >= 0.5
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "acting_rates.json"), "w") as f:
        json.dump(dict(env_steps_per_s_separate_calls=slow["env_steps_per_s"], env_steps_per_s_fused=fast["env_steps_per_s"],
                       gradient_steps_per_s=fast["gradient_steps_per_s"], envs=8), f)
    assert fast["env_steps_per_s"] >= 1.5 * slow["env_steps_per_s"], (fast["env_steps_per_s"], slow["env_steps_per_s"])
