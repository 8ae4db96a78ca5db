"""Shared by the CPU (oracle) and GPU (HIP) full-size parity tests: rebuild a fixture's inputs from
seeds (tests/detdata.py), check them against the fixture's checksums, and compare results with the
reference's fp32 run THROUGH its fp64 run:

    |x - ref64|  <=  max(K * |ref32 - ref64|,  1e-5 * scale)

i.e. an implementation passes when its error is no larger than (K times) the reference's own fp32
error, or within the north star's 1e-5 relative — whichever is looser for that quantity.

Activation kinks (see tests/golden/make_golden_full.py): a hidden unit whose pre-activation is within
fp32 rounding error of 0 may land on the other side of the LeakyReLU / ReLU kink under a different
summation order; its derivative flips and one batch row's contribution to the gradients below changes
by a discrete, exactly known amount.  The fixture lists those units with their fp64 "flip vectors";
`explain_flips` fits a deviation as a 0/1 combination of them (joint least squares over all networks)
and the strict bound is then applied to what remains.  The number of flips found is reported.
"""
from __future__ import annotations

import numpy as np

import detdata
from conftest import hparams_from_golden, load_golden

CASES = ["ddpg_pickplace_b256", "cfg1_ddpg_reach_b256", "cfg2_ddpg_reach_b1024", "cfg3_td3_pickplace_b2048_s1", "cfg3_td3_pickplace_b2048_s2",
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
        self.who, self.case, self.rows, self.bad, self.beyond_flat = who, case, [], [], []
        self.flips = self.flips_ref32 = self.kink_units = 0

    def check(self, what: str, got, ref32, ref64, mask=None, extra_abs: float = 0.0):
        got, ref32, ref64 = (np.asarray(x, np.float64).reshape(-1) for x in (got, ref32, ref64))
        if mask is not None:
            got, ref32, ref64 = got[mask], ref32[mask], ref64[mask]
        if got.size == 0:
            return
        scale = float(np.max(np.abs(ref64)))
        e_got = float(np.max(np.abs(got - ref64)))
        e_ref = float(np.max(np.abs(ref32 - ref64)))
        ok = e_got <= max(K_REF * e_ref, RTOL * scale) + extra_abs
        # the north star's flat bound on its own: every quantity of every case meets it today (round 3: worst 7.2e-6), so a
        # regression may not hide behind the K_REF term.  Kept apart from `ok` so that a report still says which
        # of the two a quantity missed.
        if not e_got <= RTOL * scale + extra_abs:
            self.beyond_flat.append((what, e_got / scale if scale > 0 else e_got, e_ref / scale if scale > 0 else e_ref))
        self.rows.append(dict(case=self.case, who=self.who, quantity=what, scale=scale, err_vs_f64=e_got, ref32_err_vs_f64=e_ref,
                              rel_err=e_got / scale if scale > 0 else 0.0, rel_ref=e_ref / scale if scale > 0 else 0.0,
                              rel_vs_ref32=float(np.max(np.abs(got - ref32))) / scale if scale > 0 else 0.0, ok=bool(ok)))
        if not ok:
            self.bad.append((what, e_got, e_ref, scale))

    def summary(self) -> str:
        w = max(len(r["quantity"]) for r in self.rows)
        lines = [f"{self.case} [{self.who}]  (relative to each quantity's scale; K={K_REF}, floor {RTOL}); activation-kink flips "
                 f"explained: {self.flips} (reference's own fp32 run: {self.flips_ref32}) of {self.kink_units} near-zero units listed"]
        for r in self.rows:
            lines.append(f"  {r['quantity']:<{w}}  err/f64 {r['rel_err']:.2e}   ref32/f64 {r['rel_ref']:.2e}   vs ref32 {r['rel_vs_ref32']:.2e}"
                         + ("" if r["ok"] else "   <-- FAIL"))
        return "\n".join(lines)


def explain_flips(case: Case, sampled: dict):
    """sampled[net] = an implementation's gradient of `net` at the fixture's sample positions.
    -> (c, shift, dnorm): c[u] in {0, 1} per listed kink unit; shift[net] = sum_u c[u] * flip vector (sampled);
    dnorm[net] = the gradient norm of the fp64 reference with those flips applied."""
    g = case.g
    nets = [n for n in sampled if f"kinku_{n}" in g.files]
    shift = {n: np.zeros(g[f"g64_{n}"].size) for n in sampled}
    dnorm = {n: float(g[f"gnorm64_{n}"][0]) for n in sampled}
    nU = int(g["kink_units"].shape[0]) if "kink_units" in g.files else 0
    c = np.zeros(nU)
    if nU and nets:
        A, r = [], []
        for n in nets:
            g64 = g[f"g64_{n}"]
            scale = float(np.max(np.abs(g64)))
            M = np.zeros((g64.size, nU))
            M[:, g[f"kinku_{n}"]] = g[f"kinkF_{n}"].astype(np.float64).T
            A.append(M / scale)
            r.append((np.asarray(sampled[n], np.float64) - g64) / scale)
        sol = np.linalg.lstsq(np.vstack(A), np.concatenate(r), rcond=None)[0]
        c = (sol > 0.5).astype(np.float64)
    for n in nets:
        rows = g[f"kinku_{n}"]
        cu = c[rows]
        shift[n] = cu @ g[f"kinkF_{n}"].astype(np.float64) if rows.size else np.zeros(g[f"g64_{n}"].size)
        n2 = float(g[f"gnorm64_{n}"][0]) ** 2 + 2.0 * float(cu @ g[f"kinkdot_{n}"]) + float(cu @ g[f"kinkgram_{n}"] @ cu) if rows.size \
            else float(g[f"gnorm64_{n}"][0]) ** 2
        dnorm[n] = np.sqrt(max(n2, 0.0))
    return c, shift, dnorm


def compare(case: Case, rep: Report, tup, grads: dict, params: dict, extras: dict):
    """tup: returned tuple; grads[net] / params[net]: full flat vectors (pre-clip gradient, parameters after the
    step); extras: bn_mean / bn_var / log_alpha / alpha for SAC / TQC."""
    g = case.g
    t32, t64 = g["tuple32"], g["tuple64"]
    assert len(tup) == len(t32), (len(tup), len(t32))
    gnets = [n for n in case.net_names if f"g64_{n}" in g.files and n in grads]
    full = {n: np.asarray(grads[n], np.float64) for n in gnets}
    # kink flips of the implementation and of the reference's own fp32 run, each against the fp64 run
    c_got, sh_got, nm_got = explain_flips(case, {n: full[n][g[f"gidx_{n}"]] for n in gnets})
    c_ref, sh_ref, nm_ref = explain_flips(case, {n: g[f"g32_{n}"] for n in gnets})
    rep.flips, rep.flips_ref32 = int(c_got.sum()), int(c_ref.sum())
    rep.kink_units = int(c_got.size)
    # a flip moves the gradient norms (and with them the un-clipped norm metrics) by a known amount
    slack = max([abs(nm_got[n] - float(g[f"gnorm64_{n}"][0])) / float(g[f"gnorm64_{n}"][0]) for n in gnets] + [0.0])
    for j, v in enumerate(tup):
        rep.check(f"tuple[{j}]", [v], [t32[j]], [t64[j]], extra_abs=slack * abs(t64[j]))
    for net in case.net_names:
        idx = g[f"gidx_{net}"]
        mask = None
        src = net.replace("target_", "")
        if f"g64_{src}" in g.files:
            g64 = np.abs(g[f"g64_{src}"])
            mask = g64 > 1e-3 * float(g64.max())     # Adam turns noise on ~0 gradients into +-lr: compare where it is signal
            if src in sh_got:
                mask &= g64 > 2.0 * np.maximum(np.abs(sh_got[src]), np.abs(sh_ref[src]))   # ... and where no flip can change the sign
        if net in gnets:
            rep.check(f"gnorm {net}", [np.sqrt(np.square(full[net]).sum())], g[f"gnorm32_{net}"] - (nm_ref[net] - g[f"gnorm64_{net}"]),
                      [nm_got[net]])
            rep.check(f"grad {net}", full[net][idx] - sh_got[net], g[f"g32_{net}"] - sh_ref[net], g[f"g64_{net}"])
        if net in params:
            rep.check(f"param {net}", np.asarray(params[net])[idx], g[f"p32_{net}"], g[f"p64_{net}"], mask)
    for k, v in extras.items():
        rep.check(k, v, g[k + "32"], g[k + "64"])
