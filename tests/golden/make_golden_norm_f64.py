"""Generate tests/golden/normalizer_f64.npz from the REAL reference (build container only; see make_golden.py).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_norm_f64.py

The reference's RunningNormalizer (src/utils.py:68-117) fed FLOAT64 rows, as its trainer feeds the observation normaliser: the
vector env allocates observation batches with the observation space's dtype, and TimeFeatureWrapper declares that space float64
(src/utils.py:156) although the values inside are float32-valued (panda-gym's float32 observation + a float32 time feature).
numpy's type rules then put batch moments, merge and normalize in float64 — also for a LOADED normaliser, whose float32
statistics turn float64 with the first update (only `self.var * self.count` and `sqrt(self.var) + 1e-8` are float32 values until
then).  Captured (inputs are float32 VALUES stored as float64 arrays):
  part A  a created normaliser: updates with float64 batches, statistics and normalised float64 probes after each;
  part B  save -> load -> normalize float64 probes (float32 statistics, float64 rows) -> updates with float64 batches
          (statistics dtype after each: float64 from the first one) -> probes;
  part C  the mixed case a resumed run can see: load -> update with FLOAT32 rows (stays float32) -> update with float64 rows."""
import os
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
gym = types.ModuleType("gymnasium")


class _W:
    def __init__(self, env=None):
        self.env = env


gym.Wrapper = _W
gym.ObservationWrapper = _W
gym.vector = types.SimpleNamespace(AsyncVectorEnv=object)
gym.spaces = types.SimpleNamespace(Dict=dict, Box=object)
sys.modules["gymnasium"] = gym
sys.path.insert(0, "/root/reference")

from src.utils import RunningNormalizer  # noqa: E402  (the reference)


def main():
    gen = np.random.default_rng(417)
    D = 7
    scale, shift = np.array([1, 10, 0.1, 3, 1, 50, 1e-3]), np.array([0, 5, -2, 0, 100, 0, 0])
    draw32 = lambda n: (gen.standard_normal((n, D)) * scale + shift).astype(np.float32)
    draw = lambda n: draw32(n).astype(np.float64)            # float32 VALUES in float64 arrays: what the vector env hands over
    out = dict(D=np.array([D]), numpy=np.array(np.__version__))
    probe = draw(9)
    out["probe"] = probe

    # ---- part A: created normaliser, float64 rows
    nz = RunningNormalizer(D)
    sizes_a = [16, 2, 128, 33, 1000]
    out["a_sizes"] = np.array(sizes_a)
    for i, n in enumerate(sizes_a):
        x = draw(n)
        out[f"a_x{i}"] = x
        nz.update(x)
        z = nz.normalize(probe)
        out[f"a_mean{i}"], out[f"a_var{i}"], out[f"a_count{i}"], out[f"a_norm{i}"] = nz.mean.copy(), nz.var.copy(), np.array([nz.count]), z
        assert nz.mean.dtype == np.float64 and z.dtype == np.float64

    # ---- part B: save -> load -> float64 rows
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "sub", "obs.yaml")
        nz.save(path)
        with open(path) as fh:
            out["yaml_text"] = np.array(fh.read())
        ld = RunningNormalizer(D)
        ld.load(path)
        ld2 = RunningNormalizer(D)
        ld2.load(path)
    assert ld.mean.dtype == np.float32 and ld.var.dtype == np.float32
    out["load_mean"], out["load_var"], out["load_count"], out["load_clip"] = ld.mean.copy(), ld.var.copy(), np.array([ld.count]), np.array([ld.clip_range])
    z = ld.normalize(probe)
    out["b_load_norm"], out["b_load_norm_dtype"] = z, np.array(str(z.dtype))
    sizes_b = [16, 2, 64, 1000]
    out["b_sizes"] = np.array(sizes_b)
    for i, n in enumerate(sizes_b):
        x = draw(n)
        out[f"b_x{i}"] = x
        ld.update(x)
        out[f"b_mean{i}"], out[f"b_var{i}"], out[f"b_count{i}"] = ld.mean.copy(), ld.var.copy(), np.array([ld.count])
        out[f"b_mean{i}_dtype"] = np.array(str(ld.mean.dtype))
        out[f"b_norm{i}"] = ld.normalize(probe)

    # ---- part C: load -> float32 rows (float32 regime) -> float64 rows (back to float64)
    x32 = draw32(24)
    out["c_x32"] = x32
    ld2.update(x32)
    out["c_mean0"], out["c_var0"], out["c_count0"], out["c_mean0_dtype"] = ld2.mean.copy(), ld2.var.copy(), np.array([ld2.count]), np.array(str(ld2.mean.dtype))
    z32 = ld2.normalize(probe.astype(np.float32))
    out["c_norm0_f32rows"], out["c_norm0_f32rows_dtype"] = z32, np.array(str(z32.dtype))
    out["c_norm0_f64rows"] = ld2.normalize(probe)
    x64 = draw(40)
    out["c_x64"] = x64
    ld2.update(x64)
    out["c_mean1"], out["c_var1"], out["c_count1"], out["c_mean1_dtype"] = ld2.mean.copy(), ld2.var.copy(), np.array([ld2.count]), np.array(str(ld2.mean.dtype))
    out["c_norm1"] = ld2.normalize(probe)
    np.savez_compressed(os.path.join(HERE, "normalizer_f64.npz"), **out)
    print("normalizer_f64 ok", out["b_load_norm_dtype"], out["b_mean0_dtype"], out["c_mean0_dtype"], out["c_mean1_dtype"], out["c_norm0_f32rows_dtype"])


if __name__ == "__main__":
    main()
