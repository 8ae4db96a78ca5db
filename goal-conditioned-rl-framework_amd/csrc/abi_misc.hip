// abi_misc.hip — small C-ABI helpers: stand-alone sort op, hipEvent timing, raw device memory.
#include <cstdlib>
#include <cstring>
#include <vector>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <map>
#include <string>

#include <fcntl.h>
#include <sys/file.h>
#include <sys/stat.h>
#include <unistd.h>

#include "common.h"
#include "gemm_mfma.h"
#include "meet.h"
#include "ops.h"

// ---- host side of meet.h ---------------------------------------------------------------------------------------------------
namespace gcrl {
namespace {
int g_shared = -1;   // -1: not decided yet (GCRL_SHARED_GPU in the environment)
}
bool meet_device_shared() {
  if (g_shared < 0) {
    const char* e = std::getenv("GCRL_SHARED_GPU");
    g_shared = (e && *e && *e != '0') ? 1 : 0;
  }
  return g_shared == 1;
}
void meet_set_device_shared(bool on) { g_shared = on ? 1 : 0; }

// Is another PROCESS using this device through this library?  (the launch forms with in-kernel waits were switched off
// for ranks that share a GPU only under DataParallelUpdater; anyone else had to set GCRL_SHARED_GPU by hand.)  Every process that
// creates a handle on a device holds a SHARED advisory lock on a per-device presence file (named after the PCI bus id); whoever
// cannot convert it into an exclusive lock — non-blocking — is not alone, and from then on counts the device as shared.  Probed
// at handle creation and every few update calls (two system calls), so the process that was there first notices a later arrival
// too.  Not seen: a process with another /tmp, or another library on the same GPU (GCRL_SHARED_GPU=1 remains for those).
// GCRL_NO_DEVICE_LOCK=1 switches the probe off.
bool meet_probe_device(int device) {
  static std::map<int, int> fds;
  static const bool off = std::getenv("GCRL_NO_DEVICE_LOCK") != nullptr;
  if (off) return false;
  auto it = fds.find(device);
  if (it == fds.end()) {
    char bus[32] = {0};
    int fd = -2;
    if (hipDeviceGetPCIBusId(bus, (int)sizeof(bus), device) == hipSuccess && bus[0]) {
      for (char* c = bus; *c; ++c) if (*c == ':' || *c == '.') *c = '_';
      std::string path = std::string(std::getenv("TMPDIR") ? std::getenv("TMPDIR") : "/tmp") + "/gcrl_amd_gpu_" + bus + ".lock";
      fd = ::open(path.c_str(), O_RDWR | O_CREAT | O_CLOEXEC, 0666);
      if (fd >= 0) { (void)::fchmod(fd, 0666); (void)::flock(fd, LOCK_SH); } else fd = -2;
    } else {
      (void)hipGetLastError();
    }
    it = fds.emplace(device, fd).first;
  }
  if (it->second < 0) return false;
  if (::flock(it->second, LOCK_EX | LOCK_NB) == 0) {   // alone: back to the shared lock every process holds
    (void)::flock(it->second, LOCK_SH);
    return false;
  }
  (void)::flock(it->second, LOCK_SH);   // (a failed conversion drops the old lock on Linux: take it again)
  g_shared = 1;
  return true;
}

long long meet_capacity(const void* kernel, int threads, size_t lds_bytes) {
  if (meet_device_shared()) return 0;
  int dev = 0, cus = 0, per_cu = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, threads, lds_bytes) != hipSuccess) { (void)hipGetLastError(); return 0; }
  // LDS: the query divides 160 KB by the request, the hardware allocates in larger granules (32 256 B -> 4 resident, 31 744 B -> 5:
  // tools/occupancy_probe.hip) — recount with 1 KB granules and one granule held back
  hipFuncAttributes fa;
  if (hipFuncGetAttributes(&fa, kernel) == hipSuccess) {
    const size_t lds = ((fa.sharedSizeBytes + lds_bytes + 1023) / 1024) * 1024;
    if (lds > 0) per_cu = std::min<long long>(per_cu, (long long)((160 * 1024 - 1024) / lds));
  }
  return (long long)cus * std::max(0, per_cu);
}
}  // namespace gcrl

namespace {
hipStream_t as_stream(void* s) {
  if (!s || s == GCRL_STREAM_LEGACY) return (hipStream_t) nullptr;
  return (hipStream_t)s;
}
__global__ __launch_bounds__(256) void hash_normal_fill_kernel(unsigned long long seed, unsigned long long ctr0, long long n, float* out) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
    out[i] = gcrl::hash_normal(seed, ctr0 + (unsigned long long)i);
}
}  // namespace

extern "C" {

int gcrl_sort_truncate_mean(const float* in_dev, int64_t rows, int width, int drop, float* sorted_dev,
                            float* mean_dev, void* stream) {
  return gcrl::launch_sort_truncate_mean(as_stream(stream), in_dev, rows, width, drop, sorted_dev, mean_dev);
}

int gcrl_gemm_f32(const float* a, int64_t a_rs, int64_t a_cs, const float* b, int64_t b_rs, int64_t b_cs,
                  float* c, int64_t c_rs, const float* bias, int M, int N, int K, int act, int shape,
                  void* stream) {
  GCRL_CHECK_ARG(act >= 0 && act <= 3 && shape >= 0 && shape <= 5, "gcrl_gemm_f32: bad act/shape");
  gcrl::GemmDesc d;
  std::memset(&d, 0, sizeof(d));
  d.A = a; d.a_rs = a_rs; d.a_cs = a_cs;
  d.B = b; d.b_rs = b_rs; d.b_cs = b_cs;
  d.C = c; d.c_rs = c_rs;
  d.bias = bias;
  d.M = M; d.N = N; d.K = K;
  d.epi = act;
  return gcrl::launch_gemm_batch(as_stream(stream), &d, 1, shape);
}

int gcrl_gemm_dw_split_f32(const float* g, int64_t ldg, const float* x, int64_t ldx, float* dw, float* db, int out, int in, int batch,
                           int ksplit, float* sumsq, void* stream) {
  GCRL_CHECK_ARG(g && x && dw && db && out >= 1 && in >= 1 && batch >= 1 && ldg >= out && ldx >= in && ksplit >= 1 && ksplit <= 64,
                 "gcrl_gemm_dw_split_f32: bad arguments");
  hipStream_t st = as_stream(stream);
  const long long tiles = (long long)((out + 63) / 64) * ((in + 63) / 64);
  float *part = nullptr, *ss = nullptr;
  unsigned int* tick = nullptr;
  gcrl::GemmDesc d;
  std::memset(&d, 0, sizeof(d));
  d.A = g; d.a_rs = 1; d.a_cs = ldg;
  d.B = x; d.b_rs = ldx; d.b_cs = 1;
  d.C = dw; d.c_rs = in;
  d.M = out; d.N = in + 1; d.K = batch;
  d.ones_col = 1; d.col_out = db;
  d.ksplit = ksplit;
  int rc = GCRL_OK;
  auto cleanup = [&]() { (void)hipStreamSynchronize(st); if (part) (void)hipFree(part); if (tick) (void)hipFree(tick); if (ss) (void)hipFree(ss); };
  if (ksplit > 1) {
    GCRL_HIP(hipMalloc((void**)&part, (size_t)tiles * ksplit * gcrl::kTiledPartStride * sizeof(float)));
    if (hipMalloc((void**)&tick, (size_t)tiles * gcrl::kTicketStride * sizeof(unsigned int)) != hipSuccess ||
        hipMemsetAsync(tick, 0, (size_t)tiles * gcrl::kTicketStride * sizeof(unsigned int), st) != hipSuccess) { cleanup(); return gcrl::fail(GCRL_ERR_HIP, "gcrl_gemm_dw_split_f32: scratch"); }
    d.kpart = part; d.kticket = tick;
  }
  if (sumsq) {
    if (hipMalloc((void**)&ss, (size_t)tiles * 4 * sizeof(float)) != hipSuccess ||
        hipMemsetAsync(ss, 0, (size_t)tiles * 4 * sizeof(float), st) != hipSuccess) { cleanup(); return gcrl::fail(GCRL_ERR_HIP, "gcrl_gemm_dw_split_f32: scratch"); }
    d.sumsq_out = ss;
  }
  rc = gcrl::launch_gemm_batch(st, &d, 1, 4);
  if (!rc && ksplit > 1) rc = gcrl::launch_gemm_batch(st, &d, 1, 4);   // twice: the tickets must come back to zero by themselves
  if (!rc && sumsq) {
    std::vector<float> h((size_t)tiles * 4);
    if (hipMemcpyAsync(h.data(), ss, h.size() * sizeof(float), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) rc = GCRL_ERR_HIP;
    float tot = 0.f;
    for (float v : h) tot += v;
    if (!rc && hipMemcpyAsync(sumsq, &tot, sizeof(float), hipMemcpyHostToDevice, st) != hipSuccess) rc = GCRL_ERR_HIP;
  }
  cleanup();
  return rc;
}

int gcrl_bn_relu_fwd_f32(const float* z, int B, int H, const float* gamma, const float* beta, float* h, float* xhat, float* invstd,
                         float* running_mean, float* running_var, float* scratch, void* stream) {
  GCRL_CHECK_ARG(z && gamma && beta && h && running_mean && running_var && scratch && B >= 1 && H >= 4, "gcrl_bn_relu_fwd_f32: bad arguments");
  return gcrl::launch_bn_relu_fwd(as_stream(stream), z, B, H, gamma, beta, h, xhat, invstd, running_mean, running_var, scratch);
}

int gcrl_bn_relu_bwd_f32(const float* dh, const float* xhat, const float* invstd, const float* gamma, const float* beta, int B, int H,
                         float* dz, float* dgamma, float* dbeta, float* scratch, void* stream) {
  GCRL_CHECK_ARG(dh && xhat && invstd && gamma && beta && dz && dgamma && dbeta && scratch && B >= 1 && H >= 4,
                 "gcrl_bn_relu_bwd_f32: bad arguments");
  return gcrl::launch_bn_relu_bwd(as_stream(stream), dh, nullptr, xhat, invstd, gamma, beta, B, H, dz, dgamma, dbeta, scratch);
}

// row-group exchange scratch of the stand-alone slab entries (allocated on first use, process lifetime: test entry points)
static int slab_scratch(int H, float** xchg, unsigned int** bar) {
  static float* x = nullptr; static unsigned int* b = nullptr; static int cap = 0;
  if (H > cap) {
    GCRL_HIP(hipDeviceSynchronize());
    if (x) (void)hipFree(x);
    if (b) (void)hipFree(b);
    GCRL_HIP(hipMalloc((void**)&x, (size_t)gcrl::bn_slab_xchg_floats(H) * sizeof(float)));
    GCRL_HIP(hipMalloc((void**)&b, (size_t)gcrl::bn_slab_bar_words(H) * sizeof(unsigned int)));
    GCRL_HIP(hipMemset(b, 0, (size_t)gcrl::bn_slab_bar_words(H) * sizeof(unsigned int)));
    GCRL_HIP(hipDeviceSynchronize());
    cap = H;
  }
  *xchg = x; *bar = b;
  return GCRL_OK;
}
// The meeting counters are monotonic and must be a multiple of a launch's row-group count when it starts (meet.h): an agent's
// own counters always are (its row-group count never changes); this process-wide scratch serves launches of ANY shape, so
// every launch starts from zeroed counters (stream-ordered)
static int slab_scratch_reset(int H, float* xchg, unsigned int* bar, hipStream_t st) { return gcrl::bn_slab_scratch_reset(xchg, bar, H, st); }

int gcrl_bn_linear_slab_fwd_f32(const float* x, int64_t ldx, const float* w, const float* bias, const float* gamma, const float* beta,
                                int B, int H, int K, float* h, float* xhat, float* invstd, float* bstat, int row_split, void* stream) {
  GCRL_CHECK_ARG(x && w && bias && gamma && beta && h && bstat && K >= 1 && ldx >= K && gcrl::bn_slab_ok(B, H),
                 "gcrl_bn_linear_slab_fwd_f32: bad arguments (B <= 512, H a multiple of 16)");
  gcrl::BnSlabFwd f;
  std::memset(&f, 0, sizeof(f));
  f.n = 1;
  f.p[0] = gcrl::BnSlabFwdProb{x, 0, h, xhat, invstd, bstat};
  f.W = w; f.bias = bias; f.gamma = gamma; f.beta = beta; f.ldx = ldx; f.B = B; f.H = H; f.K = K;
  if (row_split > 1) {
    f.rsplit = row_split;
    if (int rc = slab_scratch(H, &f.xchg, &f.bar)) return rc;
    if (int rc = slab_scratch_reset(H, f.xchg, f.bar, as_stream(stream))) return rc;
  }
  return gcrl::launch_bn_linear_fwd_slab(as_stream(stream), f);
}

int gcrl_bn_linear_slab_bwd_f32(const float* g_up, int64_t ldg, int K_up, const float* w_up, float* xhat_dz, const float* invstd,
                                const float* gamma, const float* beta, int B, int H, float* dgamma, float* dbeta, int row_split, void* stream) {
  GCRL_CHECK_ARG(g_up && w_up && xhat_dz && invstd && gamma && beta && dgamma && dbeta && K_up >= 1 && ldg >= K_up && gcrl::bn_slab_ok(B, H),
                 "gcrl_bn_linear_slab_bwd_f32: bad arguments (B <= 512, H a multiple of 16)");
  gcrl::BnSlabBwd b;
  std::memset(&b, 0, sizeof(b));
  b.nup = 1;
  b.G[0] = g_up; b.ldg[0] = ldg; b.K[0] = K_up; b.W[0] = w_up; b.ldw[0] = H;
  b.xhat_dz = xhat_dz; b.invstd = invstd; b.gamma = gamma; b.beta = beta; b.dgamma = dgamma; b.dbeta = dbeta;
  b.B = B; b.H = H;
  if (row_split > 1) {
    b.rsplit = row_split;
    if (int rc = slab_scratch(H, &b.xchg, &b.bar)) return rc;
    if (int rc = slab_scratch_reset(H, b.xchg, b.bar, as_stream(stream))) return rc;
  }
  return gcrl::launch_bn_linear_bwd_slab(as_stream(stream), b);
}

int gcrl_set_shared_device(int shared) {
  gcrl::meet_set_device_shared(shared != 0);
  return GCRL_OK;
}

int gcrl_hash_normal_fill(uint64_t seed, uint64_t ctr0, int64_t n, float* out_dev, void* stream) {
  GCRL_CHECK_ARG(out_dev && n >= 1, "gcrl_hash_normal_fill: bad arguments");
  const int blocks = (int)std::min<int64_t>((n + 255) / 256, 4096);
  hipLaunchKernelGGL(hash_normal_fill_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), (unsigned long long)seed,
                     (unsigned long long)ctr0, (long long)n, out_dev);
  GCRL_HIP(hipGetLastError());
  return GCRL_OK;
}

void* gcrl_event_create(void) {
  hipEvent_t ev = nullptr;
  if (hipEventCreate(&ev) != hipSuccess) { gcrl::fail(GCRL_ERR_HIP, "hipEventCreate failed"); return nullptr; }
  return (void*)ev;
}
void gcrl_event_destroy(void* ev) { if (ev) (void)hipEventDestroy((hipEvent_t)ev); }
int gcrl_event_record(void* ev, void* stream) {
  GCRL_CHECK_ARG(ev, "gcrl_event_record: null event");
  GCRL_HIP(hipEventRecord((hipEvent_t)ev, as_stream(stream)));
  return GCRL_OK;
}
int gcrl_event_elapsed_ms(void* ev_start, void* ev_stop, float* ms_out) {
  GCRL_CHECK_ARG(ev_start && ev_stop && ms_out, "gcrl_event_elapsed_ms: null argument");
  GCRL_HIP(hipEventSynchronize((hipEvent_t)ev_stop));
  GCRL_HIP(hipEventElapsedTime(ms_out, (hipEvent_t)ev_start, (hipEvent_t)ev_stop));
  return GCRL_OK;
}
int gcrl_stream_synchronize(void* stream) {
  GCRL_HIP(hipStreamSynchronize(as_stream(stream)));
  return GCRL_OK;
}
void* gcrl_malloc(size_t bytes) {
  void* p = nullptr;
  if (hipMalloc(&p, bytes) != hipSuccess) { gcrl::fail(GCRL_ERR_HIP, "hipMalloc(%zu) failed", bytes); return nullptr; }
  return p;
}
void gcrl_free(void* p) { if (p) (void)hipFree(p); }
int gcrl_memcpy_h2d(void* dst, const void* src, size_t bytes, void* stream) {
  GCRL_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, as_stream(stream)));
  GCRL_HIP(hipStreamSynchronize(as_stream(stream)));
  return GCRL_OK;
}
int gcrl_memcpy_d2h(void* dst, const void* src, size_t bytes, void* stream) {
  GCRL_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, as_stream(stream)));
  GCRL_HIP(hipStreamSynchronize(as_stream(stream)));
  return GCRL_OK;
}

}  // extern "C"
