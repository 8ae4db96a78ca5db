"""ORACLE — TEST INFRASTRUCTURE ONLY.  Never imported by the product.

CPU restatement of the reference's RunningNormalizer (src/utils.py:68-98), with the batch moments written out as
the explicit float32 sequential sums numpy's axis-0 reductions perform — the order the device kernel
(csrc/normalizer.hip) follows.  Pinned by tests/golden/normalizer.npz, normalizer_loaded.npz (the float32 regime after
`load`) and normalizer_f64.npz (float64 rows, created and loaded), all captured from the reference's own class.
"""
from __future__ import annotations

import numpy as np


class RunningNormalizerOracle:
    def __init__(self, size, clip_range=5.0, eps=1e-8):            # :69-73
        self.mean = np.zeros(size)
        self.var = np.ones(size)
        self.count = eps
        self.clip_range = clip_range

    @staticmethod
    def batch_moments(x):
        """np.mean(x, axis=0), np.var(x, axis=0): numpy reduces the row axis row after row, every operation rounded to the
        rows' own type — float32 for float32 rows, float64 for the float64 batches the trainer's vector env hands over for
        observations (TimeFeatureWrapper declares the space float64, src/utils.py:156).  The order csrc/norm_math.h follows."""
        x = np.asarray(x)
        ft = np.float64 if x.dtype == np.float64 else np.float32
        x = np.asarray(x, ft)
        n, D = x.shape
        s = np.zeros(D, ft)
        for i in range(n):
            s = s + x[i]
        mean = s / ft(n)
        q = np.zeros(D, ft)
        for i in range(n):
            d = x[i] - mean
            q = q + d * d
        return mean, q / ft(n), n

    def update(self, x):                                            # :75-81
        self._update_from_moments(*self.batch_moments(x))

    def _update_from_moments(self, mean, var, count):              # :83-93
        total_count = self.count + count
        delta = mean - self.mean
        new_mean = self.mean + delta * count / total_count
        m_a = self.var * self.count
        m_b = var * count
        M2 = m_a + m_b + np.square(delta) * self.count * count / total_count
        self.mean, self.var, self.count = new_mean, M2 / total_count, total_count

    def load_state(self, mean, var, count, clip_range):             # load(), :108-117, after the yaml has been read
        """float32 arrays from here on: numpy's type rules then keep every later operation of update / normalize in float32
        (Python scalars are weak), exactly as in the reference — the expressions below are the reference's own."""
        self.mean = np.array(mean, dtype=np.float32)
        self.var = np.array(var, dtype=np.float32)
        self.count = float(count)
        self.clip_range = float(clip_range)

    def normalize(self, x):                                         # :95-97
        z = (x - self.mean) / (np.sqrt(self.var) + 1e-8)
        return np.clip(z, -self.clip_range, self.clip_range)
