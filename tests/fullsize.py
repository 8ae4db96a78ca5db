"""Shared by the CPU (oracle) and GPU (HIP) full-size parity tests: rebuild a fixture's inputs from
seeds (tests/detdata.py), check them against the fixture's checksums, and compare results with the
reference's fp32 run THROUGH its fp64 run:

    |x - ref64|  <=  max(K * |ref32 - ref64|,  1e-5 * scale)

i.e. an implementation passes when its error is no larger than (K times) the reference's own fp32
error, or within the north star's 1e-5 relative — whichever is looser for that quantity.
"""
from __future__ import annotations

import numpy as np

import detdata
from conftest import hparams_from_golden, load_golden

CASES = ["ddpg_pickplace_b256", "cfg2_ddpg_reach_b1024", "cfg3_td3_pickplace_b2048_s1", "cfg3_td3_pickplace_b2048_s2",
         "cfg4_tqc_push_b2048", "cfg5_sac_slide_b512"]
K_REF = 3.0       # fp32 rounding error is one random draw per summation order: allow 3x the reference's own
RTOL = 1e-5       # north-star tolerance (relative to the quantity's scale)


class Case:
    def __init__(self, name: str):
        self.name = name
        g = self.g = load_golden(f"full_{name}.npz")
        self.kind = str(g["kind"][0])
        self.S, self.A, self.B, self.gstep, self.H, self.L = (int(x) for x in g["dims"])
        self.step = int(g["step"][0])
        self.cfg = hparams_from_golden(g)
        self.net_names = [k[len("initsum_"):] for k in g.files if k.startswith("initsum_")]
        self.batch = detdata.batch(name, self.B, self.S, self.A)
        for key, arr in zip(("s", "a", "r", "ns", "d"), self.batch):
            assert np.array_equal(detdata.checksum(arr), g[f"batchsum_{key}"]), f"batch {key} regenerated differently"
        self.noise = detdata.normalish(detdata.seed_of(name, "noise"), (self.B, self.A)) if self.kind == "TD3" else None
        self.eps_next = self.eps_cur = None
        if self.kind in ("SAC", "TQC"):
            self.eps_next = detdata.normalish(detdata.seed_of(name, "eps_next"), (self.B, self.A))
            self.eps_cur = detdata.normalish(detdata.seed_of(name, "eps_cur"), (self.B, self.A))

    def init_vector(self, net: str) -> np.ndarray:
        if "actor" in net:
            kind = "sac_actor" if self.kind in ("SAC", "TQC") else "mlp"
            vec = detdata.net_params(f"{self.name}/{net}", kind, self.S, self.H, self.L, self.A)
        else:
            vec = detdata.net_params(f"{self.name}/{net}", "mlp", self.S + self.A, self.H, self.L, 1)
        assert np.array_equal(detdata.checksum(vec), self.g[f"initsum_{net}"]), f"{net} regenerated differently"
        return vec


class Report:
    """Collects (quantity, error of the implementation vs fp64, the reference's own fp32 error vs fp64, scale)."""

    def __init__(self, who: str, case: str):
        self.who, self.case, self.rows, self.bad = who, case, [], []

    def check(self, what: str, got, ref32, ref64, mask=None):
        got, ref32, ref64 = (np.asarray(x, np.float64).reshape(-1) for x in (got, ref32, ref64))
        if mask is not None:
            got, ref32, ref64 = got[mask], ref32[mask], ref64[mask]
        if got.size == 0:
            return
        scale = float(np.max(np.abs(ref64)))
        e_got = float(np.max(np.abs(got - ref64)))
        e_ref = float(np.max(np.abs(ref32 - ref64)))
        ok = e_got <= max(K_REF * e_ref, RTOL * scale)
        self.rows.append(dict(case=self.case, who=self.who, quantity=what, scale=scale, err_vs_f64=e_got, ref32_err_vs_f64=e_ref,
                              rel_err=e_got / scale if scale > 0 else 0.0, rel_ref=e_ref / scale if scale > 0 else 0.0,
                              rel_vs_ref32=float(np.max(np.abs(got - ref32))) / scale if scale > 0 else 0.0, ok=bool(ok)))
        if not ok:
            self.bad.append((what, e_got, e_ref, scale))

    def summary(self) -> str:
        w = max(len(r["quantity"]) for r in self.rows)
        lines = [f"{self.case} [{self.who}]  (relative to each quantity's scale; K={K_REF}, floor {RTOL})"]
        for r in self.rows:
            lines.append(f"  {r['quantity']:<{w}}  err/f64 {r['rel_err']:.2e}   ref32/f64 {r['rel_ref']:.2e}   vs ref32 {r['rel_vs_ref32']:.2e}"
                         + ("" if r["ok"] else "   <-- FAIL"))
        return "\n".join(lines)


def compare(case: Case, rep: Report, tup, grads: dict, params: dict, extras: dict):
    """tup: returned tuple; grads[net] / params[net]: full flat vectors (pre-clip gradient, parameters after the
    step); extras: bn_mean / bn_var / log_alpha / alpha for SAC / TQC."""
    g = case.g
    t32, t64 = g["tuple32"], g["tuple64"]
    assert len(tup) == len(t32), (len(tup), len(t32))
    for j, v in enumerate(tup):
        rep.check(f"tuple[{j}]", [v], [t32[j]], [t64[j]])
    for net in case.net_names:
        idx = g[f"gidx_{net}"]
        mask = None
        src = net.replace("target_", "")
        if f"g64_{src}" in g.files:
            g64 = np.abs(g[f"g64_{src}"])
            mask = g64 > 1e-3 * float(g64.max())     # Adam turns noise on ~0 gradients into +-lr: compare where it is signal
        if f"g64_{net}" in g.files and net in grads:
            full = np.asarray(grads[net], np.float64)
            rep.check(f"gnorm {net}", [np.sqrt(np.square(full).sum())], g[f"gnorm32_{net}"], g[f"gnorm64_{net}"])
            rep.check(f"grad {net}", full[idx], g[f"g32_{net}"], g[f"g64_{net}"])
        if net in params:
            rep.check(f"param {net}", np.asarray(params[net])[idx], g[f"p32_{net}"], g[f"p64_{net}"], mask)
    for k, v in extras.items():
        rep.check(k, v, g[k + "32"], g[k + "64"])
