// norm_math.h — the arithmetic of the reference's RunningNormalizer (src/utils.py:68-117), shared by every kernel that
// updates or applies one (normalizer.hip, the fused process_step of her_ring.hip, the row-chain act prologue).
//
// numpy's type rules decide the arithmetic, so two things select it (`mode` bits):
//   NORM_F32     the STATISTICS are float32 arrays: a normaliser that was LOADED (load(): np.array(..., dtype=np.float32),
//                :113-114); a created one keeps float64 mean / var (np.zeros / np.ones);
//   NORM_ROWS64  the ROWS the trainer passes are float64 arrays.  They are for observations: the vector env allocates its
//                batches with the observation space's dtype, which TimeFeatureWrapper declares float64 (src/utils.py:156),
//                while the goal spaces stay float32.  (The values are float32-valued either way — panda-gym produces float32
//                — so float32 rows carry them exactly across the ABI.)
// float32 rows (round 2-3, goldens normalizer.npz / normalizer_loaded.npz): batch moments in float32 (numpy's sequential axis-0
//   sums); a created normaliser merges and normalises in float64, a loaded one does EVERYTHING in float32 from then on (Python
//   scalars are weak; only the count stays a double).
// float64 rows (round 4, golden normalizer_f64.npz): batch moments and the merge in float64 whatever the statistics' type — only
//   `self.var * self.count` of a loaded normaliser is still a float32 product — so the first update after a load puts the
//   statistics back into float64 (the float32 regime must not be sticky there); normalize subtracts and divides in
//   float64, with the divisor sqrt(var) + 1e-8 a float32 value while the statistics are float32.
// The statistics are stored as doubles either way (float32 values are exact in them).
#pragma once
#include <hip/hip_runtime.h>

namespace gcrl {

enum { NORM_F32 = 1, NORM_ROWS64 = 2 };

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
// the same with float64 batch moments (float64 rows): everything in float64 except a float32 `self.var * self.count`
__device__ inline void norm_merge64(double& m, double& v, double bm, double bv, int rows, double c0, bool f32) {
  const double total = c0 + (double)rows, cb = (double)rows;
  const double delta = bm - m;
  const double nm = m + delta * cb / total;
  const double m_a = f32 ? (double)__fmul_rn((float)v, (float)c0) : v * c0;
  const double M2 = m_a + bv * cb + delta * delta * c0 * cb / total;
  m = nm; v = M2 / total;
}

// RunningNormalizer.update (:75-93) for one feature of rows x(0) .. x(rows - 1); `mode` comes back as the statistics' type
// AFTER the update (float64 rows turn float32 statistics into float64 ones)
template <typename Row>
__device__ inline void norm_update_col(double& m, double& v, int rows, double c0, int& mode, Row x) {
  if (mode & NORM_ROWS64) {
    double s = 0.0;
    for (int i = 0; i < rows; ++i) s = __dadd_rn(s, (double)x(i));
    const double bm = __ddiv_rn(s, (double)rows);
    double q = 0.0;
    for (int i = 0; i < rows; ++i) { const double d = __dsub_rn((double)x(i), bm); q = __dadd_rn(q, __dmul_rn(d, d)); }
    norm_merge64(m, v, bm, __ddiv_rn(q, (double)rows), rows, c0, (mode & NORM_F32) != 0);
    mode &= ~NORM_F32;
  } else {
    float s = 0.f;
    for (int i = 0; i < rows; ++i) s = __fadd_rn(s, x(i));
    const float bm = __fdiv_rn(s, (float)rows);
    float q = 0.f;
    for (int i = 0; i < rows; ++i) { const float d = __fsub_rn(x(i), bm); q = __fadd_rn(q, __fmul_rn(d, d)); }
    norm_merge(m, v, bm, __fdiv_rn(q, (float)rows), rows, c0, (mode & NORM_F32) != 0);
  }
}

// the divisor of normalize (:96): sqrt(var) + 1e-8 — a float32 value while the statistics are float32
__device__ inline double norm_den(double v, bool f32) {
  // (sqrtf, not __fsqrt_rn: without OCML_BASIC_ROUNDED_OPERATIONS the latter is the NATIVE square root — 1 ulp off numpy's)
  return f32 ? (double)__fadd_rn(sqrtf((float)v), 1e-8f) : sqrt(v) + 1e-8;
}

// clip((x - mean) / den, -clip, clip), rounded to the float32 the trainer casts to (src/env.py:189-190).
// f32 = float32 statistics AND float32 rows: the whole expression in float32; otherwise float64 (den from norm_den above)
__device__ inline float norm_apply(float x, double m, double den, double clip, bool f32) {
  if (!f32) return (float)fmin(fmax(((double)x - m) / den, -clip), clip);
  const float z = __fdiv_rn(__fsub_rn(x, (float)m), (float)den);
  return fminf(fmaxf(z, -(float)clip), (float)clip);
}
__device__ inline bool norm_apply_f32(int mode) { return (mode & NORM_F32) && !(mode & NORM_ROWS64); }

}  // namespace gcrl
