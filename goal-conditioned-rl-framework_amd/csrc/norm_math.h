// norm_math.h — the arithmetic of the reference's RunningNormalizer (src/utils.py:68-117), shared by every kernel that
// updates or applies one (normalizer.hip, the fused process_step of her_ring.hip, the row-chain act prologue).
//
// Two regimes, as in the reference:
//   * a normaliser that was CREATED keeps float64 mean / var (np.zeros / np.ones): batch moments of float32 rows in float32
//     (numpy's sequential axis-0 sums), the parallel-variance merge in float64, normalize in float64;
//   * a normaliser that was LOADED holds float32 arrays (load(): np.array(..., dtype=np.float32), :113-114) and EVERYTHING
//     after that runs in float32 — Python scalars (count, 1e-8, the clip range) are weak and take the array's type; only the
//     count itself stays a Python float (a double).  Pinned by tests/golden/normalizer_loaded.npz.
// The statistics are stored as doubles either way (float32 values are exact in them); `f32` selects the regime.
#pragma once
#include <hip/hip_runtime.h>

namespace gcrl {

// _update_from_moments (:83-93) for one feature: (m, v) <- merge with a batch of `rows` rows whose float32 moments are bm, bv;
// c0 = the count before this batch
__device__ inline void norm_merge(double& m, double& v, float bm, float bv, int rows, double c0, bool f32) {
  const double total = c0 + (double)rows;                       // total_count = self.count + count   (Python floats)
  if (!f32) {
    const double cb = (double)rows;
    const double delta = (double)bm - m;
    const double nm = m + delta * cb / total;                   // (delta * count) / total_count
    // `var * count`: a float32 array times a Python int stays float32 in numpy
    const double M2 = v * c0 + (double)__fmul_rn(bv, (float)rows) + delta * delta * c0 * cb / total;
    m = nm; v = M2 / total;
  } else {
    const float mf = (float)m, vf = (float)v, c0f = (float)c0, cbf = (float)rows, tf = (float)total;
    const float delta = __fsub_rn(bm, mf);
    const float nm = __fadd_rn(mf, __fdiv_rn(__fmul_rn(delta, cbf), tf));
    const float m_a = __fmul_rn(vf, c0f), m_b = __fmul_rn(bv, cbf);
    const float term = __fdiv_rn(__fmul_rn(__fmul_rn(__fmul_rn(delta, delta), c0f), cbf), tf);   // ((square(delta) * self.count) * count) / total_count
    const float M2 = __fadd_rn(__fadd_rn(m_a, m_b), term);
    m = (double)nm; v = (double)__fdiv_rn(M2, tf);
  }
}

// the divisor of normalize (:96): sqrt(var) + 1e-8
__device__ inline double norm_den(double v, bool f32) {
  // (sqrtf, not __fsqrt_rn: without OCML_BASIC_ROUNDED_OPERATIONS the latter is the NATIVE square root — 1 ulp off numpy's)
  return f32 ? (double)__fadd_rn(sqrtf((float)v), 1e-8f) : sqrt(v) + 1e-8;
}

// clip((x - mean) / den, -clip, clip), rounded to the float32 the trainer casts to (src/env.py:189-190)
__device__ inline float norm_apply(float x, double m, double den, double clip, bool f32) {
  if (!f32) return (float)fmin(fmax(((double)x - m) / den, -clip), clip);
  const float z = __fdiv_rn(__fsub_rn(x, (float)m), (float)den);
  return fminf(fmaxf(z, -(float)clip), (float)clip);
}

}  // namespace gcrl
