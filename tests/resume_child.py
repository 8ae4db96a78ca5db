"""Child process of tests/test_gpu_surface.py::test_full_state_resume_in_a_new_process: a FRESH agent in a FRESH process
loads a saved state, continues with the given update_many calls and writes what it got.
    python tests/resume_child.py <kind> <state_dir> <out.npz> <step0> <n> [<step0> <n> ...]"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
import gcrl_amd  # noqa: E402
from test_gpu_surface import resume_agent  # noqa: E402

kind, state_dir, out = sys.argv[1:4]
calls = [int(x) for x in sys.argv[4:]]
ag = resume_agent(gcrl_amd, kind)
ag.load_state(state_dir)
tuples = []
for step0, n in zip(calls[0::2], calls[1::2]):
    tuples += [[float(x) for x in t] for t in ag.update_many(step0, n)]
width = max(len(t) for t in tuples)
np.savez(out, tuples=np.array([t + [0.0] * (width - len(t)) for t in tuples]), actor=ag.actor.flat(), critic=ag.critics[-1].flat(),
         target=ag.target_critics[0].flat(), rows=ag.buffer.rows()[0], n=len(ag.buffer))
