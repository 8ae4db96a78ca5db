"""Generate tests/golden/normalizer_loaded.npz from the REAL reference (build container only; see make_golden.py).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_norm_load.py

The reference's RunningNormalizer AFTER `load()` (src/utils.py:108-117): mean / var come back as float32 arrays and every later
operation — normalize (:95-97) and the parallel-variance merge of update (:83-93) — runs in float32 from then on (the count stays a
Python float).  Captured: the statistics right after load, normalised probe rows (float32 in, as the trainer passes them), then
updates with float32 batches and the statistics / probes after each.  What evaluation runs (`env.py` test mode: load obs.yaml, then
normalize only) and what a resumed training run does."""
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
    gen = np.random.default_rng(91)
    D = 7
    scale, shift = np.array([1, 10, 0.1, 3, 1, 50, 1e-3]), np.array([0, 5, -2, 0, 100, 0, 0])
    draw = lambda n: (gen.standard_normal((n, D)) * scale + shift).astype(np.float32)
    nz = RunningNormalizer(D)
    out = dict(D=np.array([D]), numpy=np.array(np.__version__))
    pre = [16, 128, 33]
    for i, n in enumerate(pre):
        x = draw(n)
        out[f"pre_x{i}"] = x
        nz.update(x)
    out["n_pre"] = np.array([len(pre)])
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "sub", "obs.yaml")
        nz.save(path)
        with open(path) as fh:
            out["yaml_text"] = np.array(fh.read())
        ld = RunningNormalizer(D)
        ld.load(path)
    assert ld.mean.dtype == np.float32 and ld.var.dtype == np.float32
    out["load_mean"], out["load_var"], out["load_count"] = ld.mean.copy(), ld.var.copy(), np.array([ld.count])
    probe = draw(9)
    out["probe"] = probe
    z = ld.normalize(probe)
    out["load_norm"], out["load_norm_dtype"] = z, np.array(str(z.dtype))
    sizes = [16, 2, 64, 1000]
    out["sizes"] = np.array(sizes)
    for i, n in enumerate(sizes):
        x = draw(n)
        out[f"x{i}"] = x
        ld.update(x)
        out[f"mean{i}"], out[f"var{i}"], out[f"count{i}"] = ld.mean.copy(), ld.var.copy(), np.array([ld.count])
        out[f"mean{i}_dtype"] = np.array(str(ld.mean.dtype))
        out[f"norm{i}"] = ld.normalize(probe)
    np.savez_compressed(os.path.join(HERE, "normalizer_loaded.npz"), **out)
    print("normalizer_loaded ok", out["load_norm_dtype"], out["mean0_dtype"], out["mean3_dtype"])


if __name__ == "__main__":
    main()
