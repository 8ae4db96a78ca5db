"""The device-RNG mode's Gaussian (csrc/ops.h hash_normal — what every benchmarked TD3 / SAC / TQC step draws in place of
torch.randn_like, reference src/agent.py:175, and Normal.rsample's eps, src/model.py:134): its restatement
oracle/device_rng_oracle.py is checked for the distribution it must have (CPU), and the device function is checked value by
value against the restatement and for the same distribution (GPU, through gcrl_hash_normal_fill)."""
import ctypes as C
import math

import numpy as np
import pytest

from oracle.device_rng_oracle import hash_normal, hash_uniforms

N = 1 << 20


def check_standard_normal(x, tag):
    """Moments, tails and serial structure of N(0, 1) at n = 2^20 (each bound ~5 standard errors)."""
    x = np.asarray(x, np.float64)
    n = x.size
    se = 1.0 / math.sqrt(n)
    assert abs(x.mean()) < 5 * se, (tag, "mean", x.mean())
    assert abs(x.var() - 1.0) < 5 * math.sqrt(2.0) * se, (tag, "var", x.var())
    z = (x - x.mean()) / x.std()
    assert abs((z ** 3).mean()) < 5 * math.sqrt(6.0) * se, (tag, "skew", (z ** 3).mean())
    assert abs((z ** 4).mean() - 3.0) < 5 * math.sqrt(24.0) * se, (tag, "kurtosis", (z ** 4).mean())
    for lag in (1, 2, 3, 4, 7, 64):     # consecutive counters are consecutive elements of a [B, A] noise matrix
        r = float(np.mean(z[:-lag] * z[lag:]))
        assert abs(r) < 5 * se, (tag, "autocorrelation", lag, r)
    # tails: P(|x| > t) for t = 1, 2, 3, 4 against the normal law
    for t in (1.0, 2.0, 3.0, 4.0):
        p = math.erfc(t / math.sqrt(2.0))
        got = float(np.mean(np.abs(x) > t))
        assert abs(got - p) < 5 * math.sqrt(p * (1 - p) / n) + 1e-7, (tag, "tail", t, got, p)
    assert np.all(np.isfinite(x)) and np.abs(x).max() < 5.8     # sqrt(-2 ln 2^-24) = 5.77: the method's hard tail limit


def test_oracle_hash_normal_is_standard_normal():
    for seed, ctr0 in [(1898, 0), (7, 1 << 40), (2 ** 63 + 11, 12345)]:
        x = hash_normal(seed, np.uint64(ctr0) + np.arange(N, dtype=np.uint64))
        assert x.dtype == np.float32
        check_standard_normal(x, (seed, ctr0))
    # different seeds / disjoint counter ranges give unrelated streams
    a = hash_normal(1, np.arange(N, dtype=np.uint64)).astype(np.float64)
    b = hash_normal(2, np.arange(N, dtype=np.uint64)).astype(np.float64)
    c = hash_normal(1, np.uint64(N) + np.arange(N, dtype=np.uint64)).astype(np.float64)
    assert abs(np.mean(a * b)) < 5 / math.sqrt(N) and abs(np.mean(a * c)) < 5 / math.sqrt(N)
    u1, u2 = hash_uniforms(3, np.arange(N, dtype=np.uint64))
    assert u1.min() > 0.0 and u1.max() <= 1.0 and u2.min() >= 0.0 and u2.max() < 1.0
    assert abs(u1.mean() - 0.5) < 5 / math.sqrt(12 * N) and abs(u2.mean() - 0.5) < 5 / math.sqrt(12 * N)


@pytest.mark.gpu
def test_device_hash_normal_matches_the_restatement(gcrl):
    import torch
    lib, check = gcrl._ffi.lib, gcrl._ffi.check
    worst = 0.0
    for seed, ctr0 in [(1898, 0), (7, 1 << 40), (2 ** 63 + 11, 12345)]:
        out = torch.empty(N, dtype=torch.float32, device="cuda")
        check(lib.gcrl_hash_normal_fill(C.c_uint64(seed), C.c_uint64(ctr0), N, out.data_ptr(), gcrl._ffi.stream_handle()))
        torch.cuda.synchronize()
        got = out.cpu().numpy()
        want = hash_normal(seed, np.uint64(ctr0) + np.arange(N, dtype=np.uint64))
        # same hash bits, same float32 formula; logf / cosf / sqrtf of the device library and of libm agree to a few ulp of
        # the factors (|r| <= 5.8, |cos| <= 1): absolute 4e-6
        err = float(np.max(np.abs(got.astype(np.float64) - want.astype(np.float64))))
        worst = max(worst, err)
        assert err <= 4e-6, (seed, ctr0, err)
        check_standard_normal(got, ("device", seed, ctr0))
    print(f"hash_normal device vs restatement: worst |diff| {worst:.2e}")
