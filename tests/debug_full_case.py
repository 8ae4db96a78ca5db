"""Debug helper (not a test): per-layer gradient error of the HIP engine vs the oracle for one full-size case.
    python tests/debug_full_case.py cfg4_tqc_push_b2048 [pipeline]"""
import sys

import numpy as np
import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
sys.path.insert(0, __file__.rsplit("/", 1)[0])
import gcrl_amd  # noqa: E402
from fullsize import Case  # noqa: E402
from oracle.agent_oracle import OracleAgent  # noqa: E402
from test_gpu_full_size import build, run  # noqa: E402

name = sys.argv[1]
kw = dict(pipeline=int(sys.argv[2])) if len(sys.argv) > 2 else {}
c = Case(name)
ag, views = build(gcrl_amd, c, **kw)
tup, grads, params, extras = run(c, ag, views)
orc = OracleAgent(c.kind, c.S, c.A, c.cfg, nenvs=1, gradient_step=c.gstep)
nets = {"actor": orc.actor}
if orc.target_actor is not None:
    nets["target_actor"] = orc.target_actor
for i, (q, t) in enumerate(zip(orc.critics, orc.target_critics)):
    nets[f"critic_{i}"], nets[f"target_critic_{i}"] = q, t
for n, net in nets.items():
    orc.set_flat_params(net, c.init_vector(n))
okw = {}
if c.noise is not None:
    okw["noise"] = torch.from_numpy(c.noise)
if c.eps_next is not None:
    okw["eps_next"], okw["eps_cur"] = torch.from_numpy(c.eps_next), torch.from_numpy(c.eps_cur)
ot = [float(np.asarray(x)) for x in orc.update(c.step, batch=tuple(torch.from_numpy(x) for x in c.batch), **okw)]
print("hip   ", tup)
print("oracle", ot)
og = {f"critic_{i}": g for i, g in enumerate(orc.last["critic_grads_pre"])}
if "actor_grads_pre" in orc.last:
    og["actor"] = orc.last["actor_grads_pre"]
for n, g in grads.items():
    w = og[n]
    lay = views[n]._param_layout()
    off = 0
    print(n, "max|g|", float(np.abs(w).max()))
    for key, shape in lay:
        k = int(np.prod(shape))
        d = np.abs(g[off:off + k].astype(np.float64) - w[off:off + k])
        j = int(np.argmax(d))
        print(f"   {key:28s} {str(shape):12s} max|diff| {d.max():.3e} at {j} (hip {g[off + j]:.6e} ref {w[off + j]:.6e})  layer max|g| {np.abs(w[off:off + k]).max():.3e}  n_bad(>1e-5*max) {(d > 1e-5 * np.abs(w).max()).sum()}")
        off += k
