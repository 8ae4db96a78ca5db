import sys, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import gcrl_amd as gcrl
from test_gpu_parity import _ddpg_for_schedules
H, L, B = 256, 3, 256
a = _ddpg_for_schedules(gcrl, H, L, 0, B=B)
b = _ddpg_for_schedules(gcrl, H, L, 2, B=B)
c = _ddpg_for_schedules(gcrl, H, L, 2, B=B)
ra = np.array([[float(x) for x in a.update(s)] for s in range(1, 13)])
rb = np.array([[float(x) for x in t] for t in b.update_many(1, 12)])
rc = []
for s in range(1, 13):
    rc.append([float(x) for x in c.update(s)])
    c.actor.set_flat(c.actor.flat())   # forces a rebuild of the [in][out] copies
rc = np.array(rc)
np.set_printoptions(linewidth=200, precision=3)
print("old - rowchain(many):\n", np.abs(ra - rb).max(axis=1))
print("rowchain(many) - rowchain(rebuild each step):\n", np.abs(rb - rc).max(axis=1))
print("param diff b vs c", np.abs(b.actor.flat() - c.actor.flat()).max(), np.abs(b.critic.flat() - c.critic.flat()).max())
print("param diff a vs b", np.abs(a.actor.flat() - b.actor.flat()).max(), np.abs(a.critic.flat() - b.critic.flat()).max())
