"""ORACLE — TEST INFRASTRUCTURE ONLY.  Never imported by the product.

Data-parallel emulation of the reference's update step on the CPU: G replicas of the oracle agent
(oracle/agent_oracle.py) run `update(step)` in lock-step threads, each on ITS rank's batch; after
every network's backward pass the replicas' gradients are replaced by their mean — what an
all-reduce(sum) scaled by 1/G does in the engine — and each replica then clips and
steps identically.  BatchNorm statistics stay LOCAL to a replica (batch statistics of its own rows,
its own running statistics): the semantics DESIGN.md §6 declares for SAC / TQC under data parallelism.
The reference itself is single-process; this is the specification of the DP extension, built from the
reference's per-step arithmetic.
"""
from __future__ import annotations

import threading

import torch


class DPOracle:
    def __init__(self, agents):
        self.agents = list(agents)
        self.G = len(self.agents)
        self._barrier = threading.Barrier(self.G, timeout=120)
        self._grads = [None] * self.G
        self._mean = None
        for r, ag in enumerate(self.agents):
            ag.grad_sync = (lambda name, params, r=r: self._sync(r, params))

    def _sync(self, rank, params):
        self._grads[rank] = [p.grad.detach().clone() for p in params]
        self._barrier.wait()
        if rank == 0:
            # sum in rank order, then one multiply by 1/G: the engine's all-reduce(sum) + grad_scale
            scale = 1.0 / self.G
            mean = []
            for i in range(len(params)):
                s = self._grads[0][i].clone()
                for g in self._grads[1:]:
                    s += g[i]
                mean.append(s * scale)
            self._mean = mean
        self._barrier.wait()
        with torch.no_grad():
            for p, m in zip(params, self._mean):
                p.grad.copy_(m)
        self._barrier.wait()

    def update(self, step, batches, kwargs=None):
        """batches[r]: rank r's (s, a, r, ns, d) tensors; kwargs[r]: its injected noise / eps.  Returns the G tuples."""
        kwargs = kwargs or [{} for _ in range(self.G)]
        out, err = [None] * self.G, []

        def run(r):
            try:
                out[r] = self.agents[r].update(step, batch=batches[r], **kwargs[r])
            except BaseException as e:   # noqa: BLE001  (a failed replica must not leave the others at the barrier)
                err.append(e)
                self._barrier.abort()

        threads = [threading.Thread(target=run, args=(r,)) for r in range(self.G)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        if err:
            self._barrier.reset()
            raise err[0]
        return out
