"""CPU: the oracle at the BASELINE.json cfg 2-5 shapes against the full-size fixtures captured from
the REAL reference (tests/golden/make_golden_full.py; inputs rebuilt from seeds, tests/detdata.py)."""
import numpy as np
import pytest
import torch

import detdata
from fullsize import CASES, Case, Report, compare
from oracle.agent_oracle import OracleAgent


def test_detdata_is_bit_stable():
    """Known answers of the integer generator (a change here silently invalidates every full-size fixture)."""
    assert detdata.seed_of("a", 1) == detdata.seed_of("a", "1") == 4006276348668905061
    assert detdata.uniform01(12345, 4).tolist() == [0.1330796686614273, 0.20481663336165912, 0.11954258300911547,
                                                    0.17611780724496118]
    x = detdata.normalish(7, (3, 2))
    assert x.dtype == np.float32 and x.view(np.uint32).tolist() == [[1060052062, 3174817498], [1065244044, 3206199683],
                                                                    [3200453606, 3180323020]]
    v = detdata.net_params("t/actor", "sac_actor", 5, 8, 2, 3)
    assert v.size == (5 * 8 + 8 + 16) + (8 * 8 + 8 + 16) + 2 * (8 * 3 + 3)
    assert detdata.checksum(v).tolist() == [11.444417976541445, 36.22116031637189, 206.0]


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_reference_at_full_size(name):
    torch.set_num_threads(1)
    c = Case(name)
    orc = OracleAgent(c.kind, c.S, c.A, c.cfg, nenvs=1, gradient_step=c.gstep)
    nets = {"actor": orc.actor}
    if orc.target_actor is not None:
        nets["target_actor"] = orc.target_actor
    for i, (q, t) in enumerate(zip(orc.critics, orc.target_critics)):
        nets[f"critic_{i}"], nets[f"target_critic_{i}"] = q, t
    assert sorted(nets) == sorted(c.net_names)
    for n, net in nets.items():
        orc.set_flat_params(net, c.init_vector(n))
    tb = tuple(torch.from_numpy(x) for x in c.batch)
    kw = {}
    if c.noise is not None:
        kw["noise"] = torch.from_numpy(c.noise)
    if c.eps_next is not None:
        kw["eps_next"], kw["eps_cur"] = torch.from_numpy(c.eps_next), torch.from_numpy(c.eps_cur)
    tup = [float(np.asarray(x)) for x in orc.update(c.step, batch=tb, **kw)]
    grads = {f"critic_{i}": g for i, g in enumerate(orc.last["critic_grads_pre"])}
    if "actor_grads_pre" in orc.last:   # set only when this step updated the actor
        grads["actor"] = orc.last["actor_grads_pre"]
    params = {n: orc.flat_params(net) for n, net in nets.items()}
    extras = {}
    if c.kind in ("SAC", "TQC"):
        bns = [m for m in orc.actor.base_net if isinstance(m, torch.nn.BatchNorm1d)]
        extras = dict(bn_mean=np.concatenate([m.running_mean.numpy() for m in bns]),
                      bn_var=np.concatenate([m.running_var.numpy() for m in bns]),
                      log_alpha=orc.log_alpha.detach().numpy(), alpha=orc.alpha.detach().numpy())
    rep = Report("oracle", name)
    compare(c, rep, tup, grads, params, extras)
    print(rep.summary())
    assert not rep.bad, rep.bad[:8]
