"""Child process of tests/test_gpu_surface.py::test_full_state_resume_in_a_new_process: a FRESH agent in a FRESH process
loads a saved state, continues for the given number of steps and writes what it got.
    python tests/resume_child.py <kind> <state_dir> <step0> <n> <out.npz>"""
import sys

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
sys.path.insert(0, __file__.rsplit("/", 1)[0])
import gcrl_amd  # noqa: E402
from test_gpu_surface import resume_agent  # noqa: E402

kind, state_dir, step0, n, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
ag = resume_agent(gcrl_amd, kind)
ag.load_state(state_dir)
tuples = [[float(x) for x in t] for t in ag.update_many(step0, n)]
width = max(len(t) for t in tuples)
np.savez(out, tuples=np.array([t + [0.0] * (width - len(t)) for t in tuples]), actor=ag.actor.flat(), critic=ag.critics[-1].flat(),
         target=ag.target_critics[0].flat(), rows=ag.buffer.rows()[0], n=len(ag.buffer))
