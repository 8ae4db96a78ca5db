"""ORACLE (test infrastructure only — imported by tests/, never by the product path).

Restatement of the device-RNG mode's Gaussian `gcrl::hash_normal` (csrc/ops.h): the counter-hash Box-Muller normal the
engine draws when an update is not given injected noise — its stand-in for `torch.randn_like(actions)` in TD3's target
smoothing (reference src/agent.py:175) and for `Normal.rsample`'s eps (src/model.py:134) in SAC / TQC.  The reference
draws from torch's generator; no replacement can reproduce that stream on a GPU, so what is pinned here is (a) this
definition, value by value, against the device (the transcendental functions differ by ulps between libm and the
device library) and (b) its distribution: moments, tails, serial correlation.
"""
import numpy as np

_M = np.uint64(0xFFFFFFFFFFFFFFFF)


def mix64(z):
    """splitmix64 finaliser (csrc/ops.h mix64d, her_ring.h mix64) on uint64 arrays (wrap-around arithmetic)."""
    z = np.asarray(z, np.uint64)
    with np.errstate(over="ignore"):
        z = z + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def hash_uniforms(seed, ctr):
    """(u1, u2) float32: u1 in (0, 1], u2 in [0, 1) — the two 24-bit fields of h = mix64(mix64(seed) + ctr).
    (The C literal 16777217.0f is not a float: it rounds to 2^24, so u1's scale is exactly 2^-24 and u1 = 1.0 occurs.)"""
    with np.errstate(over="ignore"):
        h = mix64(mix64(np.uint64(seed)) + np.asarray(ctr, np.uint64))
    u1 = ((h >> np.uint64(40)) + np.uint64(1)).astype(np.float32) * (np.float32(1.0) / np.float32(16777217.0))
    u2 = ((h >> np.uint64(8)) & np.uint64(0xFFFFFF)).astype(np.float32) * (np.float32(1.0) / np.float32(16777216.0))
    return u1, u2


def hash_normal(seed, ctr):
    """sqrtf(-2 logf(u1)) * cosf(2 pi u2) in float32, operation by operation."""
    u1, u2 = hash_uniforms(seed, ctr)
    r = np.sqrt(np.float32(-2.0) * np.log(u1), dtype=np.float32)
    return (r * np.cos(np.float32(6.2831853071795864) * u2, dtype=np.float32)).astype(np.float32)
