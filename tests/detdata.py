"""Deterministic test data from integer arithmetic only (splitmix64 in numpy uint64, then exact
IEEE adds / multiplies): the same bits on every host and numpy version.

The full-size golden fixtures (tests/golden/full_*.npz, BASELINE.json cfg 2-5 shapes) would be tens
of megabytes if they stored their inputs (a 5-critic H=512 TQC has 3.2 M parameters), so the
generator (tests/golden/make_golden_full.py) and the tests both REBUILD the inputs from seeds with
these functions; the fixture stores only checksums of them, next to the reference's outputs.
"""
from __future__ import annotations

import zlib

import numpy as np

_M = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix(seed: int, n: int) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = np.uint64(seed & 0xFFFFFFFFFFFFFFFF) + np.arange(1, n + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def seed_of(*parts) -> int:
    """Stable 63-bit seed from strings / ints (crc32 chain, not Python's salted hash)."""
    h = 0x1234ABCD
    for p in parts:
        h = (h * 0x100000001B3 + zlib.crc32(str(p).encode())) & 0x7FFFFFFFFFFFFFFF
    return h


def uniform01(seed: int, n: int) -> np.ndarray:
    """[0, 1) float64, 53 random bits each."""
    return (_splitmix(seed, n) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def uniform(seed: int, shape, lo=-1.0, hi=1.0) -> np.ndarray:
    n = int(np.prod(shape))
    return (lo + (hi - lo) * uniform01(seed, n)).astype(np.float32).reshape(shape)


def normalish(seed: int, shape) -> np.ndarray:
    """Unit-variance, bell-shaped (Irwin-Hall of 4 uniforms): sums and one multiply, bit-portable."""
    n = int(np.prod(shape))
    u = uniform01(seed, 4 * n).reshape(4, n)
    return (((u[0] + u[1]) + (u[2] + u[3]) - 2.0) * 1.7320508075688772).astype(np.float32).reshape(shape)


# ---------------------------------------------------------------------------- network parameters
def net_layout(kind: str, in_dim: int, H: int, L: int, out_dim: int):
    """[(role, shape)] in torch module.parameters() order for the reference's networks
    (src/model.py): kind "mlp" = Actor / Critic, "sac_actor" = SACActorModel."""
    out = []
    for l in range(L):
        k = in_dim if l == 0 else H
        out += [("w", (H, k)), ("b", (H,))]
        if kind == "sac_actor":
            out += [("bn_g", (H,)), ("bn_b", (H,))]
    if kind == "sac_actor":   # mean_head, log_std_head (src/model.py:114-115)
        out += [("w_mean", (out_dim, H)), ("b", (out_dim,)), ("w_ls", (out_dim, H)), ("b_ls", (out_dim,))]
    else:
        out += [("w", (out_dim, H)), ("b", (out_dim,))]
    return out


def net_params(tag: str, kind: str, in_dim: int, H: int, L: int, out_dim: int) -> np.ndarray:
    """Flat fp32 parameter vector: Xavier-uniform-scaled weights, small random biases, BatchNorm
    affine near (1, 0) — away from the symmetric fresh initialisation."""
    parts = []
    for i, (role, shape) in enumerate(net_layout(kind, in_dim, H, L, out_dim)):
        s = seed_of(tag, i)
        if role in ("w", "w_mean", "w_ls"):
            # the tanh-Gaussian heads are kept moderate (|mean| ~ 0.5, std ~ 0.4), as in a trained policy: the
            # reference's log(1 - tanh(x)^2 + 1e-8) loses all fp32 precision for |x| > 4 (its own fp32 run is
            # then 2 % off its fp64 run), which would make the fixture a pin on rounding noise
            bound = float(np.sqrt(6.0 / (shape[0] + shape[1]))) * {"w": 1.0, "w_mean": 0.5, "w_ls": 0.25}[role]
            parts.append(uniform(s, shape, -bound, bound).reshape(-1))
        elif role == "b_ls":
            parts.append(uniform(s, shape, -1.05, -0.95))
        elif role == "b":
            parts.append(uniform(s, shape, -0.05, 0.05))
        elif role == "bn_g":
            parts.append(uniform(s, shape, 0.9, 1.1))
        else:
            parts.append(uniform(s, shape, -0.1, 0.1))
    return np.concatenate(parts).astype(np.float32)


def batch(tag: str, B: int, S: int, A: int):
    """(s, a, r, ns, d) with the structure the replay ring hands out: rewards in {-1, 0}, ~10 % dones."""
    s = normalish(seed_of(tag, "s"), (B, S))
    a = uniform(seed_of(tag, "a"), (B, A))
    ns = (s + np.float32(0.1) * normalish(seed_of(tag, "ns"), (B, S))).astype(np.float32)
    r = -(uniform01(seed_of(tag, "r"), B) > 0.3).astype(np.float32).reshape(B, 1)
    d = (uniform01(seed_of(tag, "d"), B) > 0.9).astype(np.float32).reshape(B, 1)
    return s, a, r, ns, d


def checksum(x: np.ndarray) -> np.ndarray:
    x = np.asarray(x, np.float64).reshape(-1)
    return np.array([x.sum(), np.square(x).sum(), float(x.size)])
