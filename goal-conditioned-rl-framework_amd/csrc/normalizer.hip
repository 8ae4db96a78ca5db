// normalizer.hip — RunningNormalizer on the device (reference src/utils.py:68-98) and the acting-side entry
// points that use it: the env-facing half of the trainer's step — observation statistics,
// normalisation, actor inference, exploration noise — as one native call per vector-env step instead of
// ~180 us of host numpy + two device round trips.
//
// Arithmetic restated exactly (bit-for-bit against numpy, tests/golden/normalizer.npz):
//   update(x)        batch_mean = np.mean(x, axis=0), batch_var = np.var(x, axis=0) on float32 rows: numpy
//                    reduces the row axis SEQUENTIALLY in float32 (sum, / n; (x - mean)^2 sum, / n);
//                    then the parallel-variance merge in float64, operation by operation as written in
//                    _update_from_moments (src/utils.py:83-93)
//   normalize(x)     (x - mean) / (sqrt(var) + 1e-8) in float64, clipped to +-clip_range; the trainer casts the
//                    result to float32 before it reaches the networks (src/env.py:189-190), so does this
#include <hip/hip_runtime.h>

#include <cmath>
#include <vector>

#include "common.h"
#include "her_ring.h"
#include "norm_math.h"

struct gcrl_normalizer {
  int size = 0, device = 0;
  double clip = 5.0;
  double* mean = nullptr;    // device [size]
  double* var = nullptr;     // device [size]
  double* count = nullptr;   // device [1]
  float* xdev = nullptr;     // staging of host rows
  float* xpin = nullptr;
  size_t xcap = 0;           // floats
  bool f32 = false;          // loaded statistics: float32 arrays (norm_math.h NORM_F32)
  bool rows64 = false;       // the trainer's rows are float64 arrays (NORM_ROWS64): moments / merge / normalize in float64
  int mode() const { return (f32 ? gcrl::NORM_F32 : 0) | (rows64 ? gcrl::NORM_ROWS64 : 0); }
};

namespace {

// one thread per feature: the batch moments and the merge in the types numpy would use (norm_math.h)
__global__ void norm_update_kernel(const float* __restrict__ x, int n, int ld, int D, double* mean, double* var, double* count, int mode) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= D) return;
  double m = mean[j], v = var[j];
  gcrl::norm_update_col(m, v, n, *count, mode, [&](int i) { return x[(long long)i * ld + j]; });
  mean[j] = m;
  var[j] = v;
}
__global__ void norm_count_kernel(double* count, int n) { *count = *count + (double)n; }

// out[i][col0 + j] = float32(clip((x[i][j] - mean[j]) / (sqrt(var[j]) + 1e-8)))   (mean == null: plain copy)
__global__ void norm_apply_kernel(const float* __restrict__ x, int n, int ld, int D, const double* mean, const double* var,
                                  double clip, float* out, int ld_out, int col0, int mode) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * D) return;
  const int i = t / D, j = t - i * D;
  const float v = x[(long long)i * ld + j];
  float r = v;
  if (mean) r = gcrl::norm_apply(v, mean[j], gcrl::norm_den(var[j], (mode & gcrl::NORM_F32) != 0), clip, gcrl::norm_apply_f32(mode));
  out[(long long)i * ld_out + col0 + j] = r;
}

int ensure_staging(gcrl_normalizer* z, size_t floats) {
  if (floats <= z->xcap) return GCRL_OK;
  GCRL_HIP(hipDeviceSynchronize());
  if (z->xdev) GCRL_HIP(hipFree(z->xdev));
  if (z->xpin) GCRL_HIP(hipHostFree(z->xpin));
  const size_t want = std::max<size_t>(floats, 4096);
  GCRL_HIP(hipMalloc((void**)&z->xdev, want * sizeof(float)));
  GCRL_HIP(hipHostMalloc((void**)&z->xpin, want * sizeof(float), hipHostMallocDefault));
  z->xcap = want;
  return GCRL_OK;
}

hipStream_t pick_stream(void* s) { return s == GCRL_STREAM_LEGACY ? (hipStream_t) nullptr : (hipStream_t)s; }

}  // namespace

namespace gcrl {
void normalizer_view(const gcrl_normalizer* z, const double** mean, const double** var, double** count, double* clip, int* mode) {
  *mean = z ? z->mean : nullptr; *var = z ? z->var : nullptr; if (count) *count = z ? z->count : nullptr; *clip = z ? z->clip : 0.0;
  if (mode) *mode = z ? z->mode() : 0;
}
// a launch that updates `z` has been enqueued: float64 rows leave float64 statistics behind (later launches see them in stream order)
void normalizer_updated(gcrl_normalizer* z) { if (z && z->rows64) z->f32 = false; }
// device rows in, statistics updated (used by the fused entry points)
int normalizer_update_dev(gcrl_normalizer* z, const float* x_dev, int n, int ld, hipStream_t st) {
  hipLaunchKernelGGL(norm_update_kernel, dim3((z->size + 63) / 64), dim3(64), 0, st, x_dev, n, ld, z->size, z->mean, z->var, z->count, z->mode());
  normalizer_updated(z);
  hipLaunchKernelGGL(norm_count_kernel, dim3(1), dim3(1), 0, st, z->count, n);
  GCRL_HIP(hipGetLastError());
  return GCRL_OK;
}
// out[:, col0 : col0 + D] = normalize(x) (z == null: copy)
int normalizer_apply_dev(const gcrl_normalizer* z, const float* x_dev, int n, int ld, int D, float* out_dev, int ld_out, int col0, hipStream_t st) {
  hipLaunchKernelGGL(norm_apply_kernel, dim3((n * D + 255) / 256), dim3(256), 0, st, x_dev, n, ld, D, z ? z->mean : nullptr,
                     z ? z->var : nullptr, z ? z->clip : 0.0, out_dev, ld_out, col0, z ? z->mode() : 0);
  GCRL_HIP(hipGetLastError());
  return GCRL_OK;
}
}  // namespace gcrl

extern "C" {

gcrl_normalizer* gcrl_normalizer_create(int size, double clip_range, double eps, int device) {
  if (size < 1 || gcrl_device_count() <= device || device < 0) {
    gcrl::fail(size < 1 ? GCRL_ERR_ARG : GCRL_ERR_HIP, "gcrl_normalizer_create: size %d, device %d (there is no CPU fallback)", size, device);
    return nullptr;
  }
  gcrl_normalizer* z = new gcrl_normalizer;
  z->size = size; z->clip = clip_range; z->device = device;
  std::vector<double> ones((size_t)size, 1.0), zeros((size_t)size, 0.0);
  bool ok = hipSetDevice(device) == hipSuccess && hipMalloc((void**)&z->mean, size * sizeof(double)) == hipSuccess &&
            hipMalloc((void**)&z->var, size * sizeof(double)) == hipSuccess && hipMalloc((void**)&z->count, sizeof(double)) == hipSuccess &&
            hipMemcpy(z->mean, zeros.data(), size * sizeof(double), hipMemcpyHostToDevice) == hipSuccess &&
            hipMemcpy(z->var, ones.data(), size * sizeof(double), hipMemcpyHostToDevice) == hipSuccess &&
            hipMemcpy(z->count, &eps, sizeof(double), hipMemcpyHostToDevice) == hipSuccess;   // self.count = eps (src/utils.py:72)
  if (!ok) { gcrl::fail(GCRL_ERR_HIP, "gcrl_normalizer_create: device allocation failed"); gcrl_normalizer_destroy(z); return nullptr; }
  return z;
}

void gcrl_normalizer_destroy(gcrl_normalizer* z) {
  if (!z) return;
  (void)hipDeviceSynchronize();
  for (void* p : {(void*)z->mean, (void*)z->var, (void*)z->count, (void*)z->xdev}) if (p) (void)hipFree(p);
  if (z->xpin) (void)hipHostFree(z->xpin);
  delete z;
}

int gcrl_normalizer_size(const gcrl_normalizer* z) { return z ? z->size : 0; }

int gcrl_normalizer_update(gcrl_normalizer* z, const float* x, int n, int ld, int on_device, void* stream) {
  GCRL_CHECK_ARG(z && x && n >= 1 && ld >= z->size, "gcrl_normalizer_update: bad arguments");
  hipStream_t st = pick_stream(stream);
  const float* xd = x;
  if (!on_device) {
    if (int rc = ensure_staging(z, (size_t)n * z->size)) return rc;
    GCRL_HIP(hipStreamSynchronize(st));   // the pinned staging is reused call after call
    for (int i = 0; i < n; ++i) std::memcpy(z->xpin + (size_t)i * z->size, x + (size_t)i * ld, sizeof(float) * z->size);
    GCRL_HIP(hipMemcpyAsync(z->xdev, z->xpin, (size_t)n * z->size * sizeof(float), hipMemcpyHostToDevice, st));
    xd = z->xdev; ld = z->size;
  }
  return gcrl::normalizer_update_dev(z, xd, n, ld, st);
}

int gcrl_normalizer_normalize(gcrl_normalizer* z, const float* x, int n, int ld, int on_device, float* out, int ld_out,
                              int out_on_device, void* stream) {
  GCRL_CHECK_ARG(z && x && out && n >= 1 && ld >= z->size && ld_out >= z->size, "gcrl_normalizer_normalize: bad arguments");
  hipStream_t st = pick_stream(stream);
  const float* xd = x;
  const size_t nd = (size_t)n * z->size;
  if (!on_device || !out_on_device) {
    if (int rc = ensure_staging(z, 2 * nd)) return rc;
    GCRL_HIP(hipStreamSynchronize(st));
  }
  if (!on_device) {
    for (int i = 0; i < n; ++i) std::memcpy(z->xpin + (size_t)i * z->size, x + (size_t)i * ld, sizeof(float) * z->size);
    GCRL_HIP(hipMemcpyAsync(z->xdev, z->xpin, nd * sizeof(float), hipMemcpyHostToDevice, st));
    xd = z->xdev; ld = z->size;
  }
  float* od = out_on_device ? out : z->xdev + nd;
  const int ldo = out_on_device ? ld_out : z->size;
  if (int rc = gcrl::normalizer_apply_dev(z, xd, n, ld, z->size, od, ldo, 0, st)) return rc;
  if (!out_on_device) {
    GCRL_HIP(hipMemcpyAsync(z->xpin + nd, od, nd * sizeof(float), hipMemcpyDeviceToHost, st));
    GCRL_HIP(hipStreamSynchronize(st));
    for (int i = 0; i < n; ++i) std::memcpy(out + (size_t)i * ld_out, z->xpin + nd + (size_t)i * z->size, sizeof(float) * z->size);
  }
  return GCRL_OK;
}

int gcrl_normalizer_get(gcrl_normalizer* z, double* mean, double* var, double* count) {
  GCRL_CHECK_ARG(z, "gcrl_normalizer_get: null handle");
  GCRL_HIP(hipDeviceSynchronize());
  if (mean) GCRL_HIP(hipMemcpy(mean, z->mean, z->size * sizeof(double), hipMemcpyDeviceToHost));
  if (var) GCRL_HIP(hipMemcpy(var, z->var, z->size * sizeof(double), hipMemcpyDeviceToHost));
  if (count) GCRL_HIP(hipMemcpy(count, z->count, sizeof(double), hipMemcpyDeviceToHost));
  return GCRL_OK;
}

int gcrl_normalizer_set(gcrl_normalizer* z, const double* mean, const double* var, double count, double clip_range) {
  GCRL_CHECK_ARG(z && mean && var, "gcrl_normalizer_set: null argument");
  GCRL_HIP(hipDeviceSynchronize());
  GCRL_HIP(hipMemcpy(z->mean, mean, z->size * sizeof(double), hipMemcpyHostToDevice));
  GCRL_HIP(hipMemcpy(z->var, var, z->size * sizeof(double), hipMemcpyHostToDevice));
  GCRL_HIP(hipMemcpy(z->count, &count, sizeof(double), hipMemcpyHostToDevice));
  z->clip = clip_range;
  return GCRL_OK;
}

int gcrl_normalizer_set_float32(gcrl_normalizer* z, int on) {
  GCRL_CHECK_ARG(z, "gcrl_normalizer_set_float32: null handle");
  GCRL_HIP(hipDeviceSynchronize());
  z->f32 = on != 0;
  return GCRL_OK;
}

int gcrl_normalizer_is_float32(const gcrl_normalizer* z) { return (z && z->f32) ? 1 : 0; }

int gcrl_normalizer_set_rows_float64(gcrl_normalizer* z, int on) {
  GCRL_CHECK_ARG(z, "gcrl_normalizer_set_rows_float64: null handle");
  z->rows64 = on != 0;   // (host state read when a launch is enqueued: no synchronisation needed)
  return GCRL_OK;
}

}  // extern "C"
