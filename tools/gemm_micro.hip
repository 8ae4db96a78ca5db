// gemm_micro.hip — stand-alone harness for the LDS-tiled GEMM kernel (csrc/gemm_tiled.h), plus the two hot instantiations as kernels of
// their own (argument 5 = "direct").
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I goal-conditioned-rl-framework_amd/csrc -I include \
//         tools/gemm_micro.hip -o tools/gemm_micro && tools/gemm_micro [M N K] [dx]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#include <cmath>
#include <functional>

// ablations: -DABL_NOREAD / -DABL_NOSTORE / -DABL_NOFETCH / -DABL_NOBARRIER drop one ingredient of the fused k-step (results meaningless)
#ifdef ABL_NOREAD
#define GCRL_ABL_READ(p) make_float4(0.5f + (float)(((size_t)(p)) & 15), 1.25f, -0.75f, 2.f)
#endif
#ifdef ABL_NOSTORE
#define GCRL_ABL_STORE(x) do { } while (0)
#endif
#ifdef ABL_NOFETCH
#define GCRL_ABL_FETCH(x) do { } while (0)
#endif
#ifdef ABL_NOBARRIER
#define GCRL_ABL_BARRIER() do { } while (0)
#endif
#ifdef STAMPS
__device__ unsigned long long g_stamps[8 * 8192];
#define GCRL_STAMP(i) do { if (threadIdx.x == 0) g_stamps[8 * blockIdx.x + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#endif
#include "gemm_tiled.h"

// the two hot instantiations alone (smaller code objects to read: hipcc -S --cuda-device-only)
__global__ __launch_bounds__(256, 5) void k_fwd(gcrl::GemmBatch gb) {
  __shared__ __attribute__((aligned(16))) float lds[4 * gcrl::kTB * gcrl::kLDT];
  gcrl::gemm_tiled_body<gcrl::FETCH_KC, gcrl::FETCH_KC>(gb.d[0], gcrl::xcd_tile_of((int)blockIdx.x, (int)gridDim.x), lds, lds + 2 * gcrl::kTB * gcrl::kLDT);
}
__global__ __launch_bounds__(256, 5) void k_dx(gcrl::GemmBatch gb) {
  __shared__ __attribute__((aligned(16))) float lds[4 * gcrl::kTB * gcrl::kLDT];
  gcrl::gemm_tiled_body<gcrl::FETCH_KC, gcrl::FETCH_RC>(gb.d[0], gcrl::xcd_tile_of((int)blockIdx.x, (int)gridDim.x), lds, lds + 2 * gcrl::kTB * gcrl::kLDT);
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

namespace gcrl { int fail(int code, const char*, ...) { return code; } }

template <int BYTES>
__global__ __launch_bounds__(256) void k_lds_probe(float* out) {
  __shared__ float buf[BYTES / 4];
  buf[threadIdx.x] = (float)threadIdx.x;
  __syncthreads();
  out[threadIdx.x] = buf[(threadIdx.x * 7) % (BYTES / 4)];
}
template <int BYTES>
void probe() {
  int n = 0;
  hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_lds_probe<BYTES>, 256, 0);
  printf("  static LDS %6d B per workgroup of 256 threads: %d workgroups per CU (runtime occupancy query)\n", BYTES, n);
}

int main(int argc, char** argv) {
  if (argc > 1 && !strcmp(argv[1], "occupancy")) {
    int n = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_fwd, 256, 0); printf("k_fwd: %d workgroups per CU\n", n);
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, gcrl::gemm_tiled_kernel, 256, 0); printf("gemm_tiled_kernel: %d workgroups per CU\n", n);
    probe<32768>(); probe<32512>(); probe<32256>(); probe<31744>(); probe<30720>(); probe<28672>(); probe<27136>(); probe<26624>(); probe<24576>(); probe<20480>(); probe<16384>();
    hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
    printf("sharedMemPerMultiprocessor %zu, sharedMemPerBlock %zu, maxSharedMemoryPerMultiProcessor %zu\n", pr.sharedMemPerMultiprocessor, pr.sharedMemPerBlock, pr.maxSharedMemoryPerMultiProcessor);
    return 0;
  }
  const int M = argc > 3 ? atoi(argv[1]) : 10240, N = argc > 3 ? atoi(argv[2]) : 512, K = argc > 3 ? atoi(argv[3]) : 512;
  const bool dx = argc > 4 && !strcmp(argv[4], "dx");
  const bool direct = argc > 5 && !strcmp(argv[5], "direct");
  const bool arow0 = argc > 6 && !strcmp(argv[6], "arow0");   // experiment: every A row aliases row 0 (A always cache-resident; results meaningless)
  float *A, *B, *C, *bias;
  CK(hipMalloc(&A, (size_t)M * K * 4)); CK(hipMalloc(&B, (size_t)N * K * 4)); CK(hipMalloc(&C, (size_t)M * N * 4)); CK(hipMalloc(&bias, N * 4));
  std::vector<float> h((size_t)M * K);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u >> 16) & 255) / 256.f - 0.5f;
  CK(hipMemcpy(A, h.data(), (size_t)M * K * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(B, h.data(), (size_t)N * K * 4, hipMemcpyHostToDevice));
  CK(hipMemset(bias, 0, N * 4));
  gcrl::GemmBatch gb; memset(&gb, 0, sizeof(gb));
  gb.n = 1;
  gcrl::GemmDesc& d = gb.d[0];
  d.A = A; d.a_rs = K; d.a_cs = 1;
  if (!dx) { d.B = B; d.b_rs = 1; d.b_cs = K; }            // forward: B(k, n) = W[n][k]
  else { d.B = B; d.b_rs = N; d.b_cs = 1; }                 // dX: B(k, n) = W[k][n]  (row-contiguous)
  d.C = C; d.c_rs = N; d.bias = bias; d.M = M; d.N = N; d.K = K; d.epi = 1;
  d.a_vec = 1; d.b_vec = !dx; d.a_rvec = 0; d.b_rvec = dx;
  if (arow0) d.a_rs = 0;
  d.tiles_n = (N + 63) / 64; d.ntiles = ((M + 63) / 64) * d.tiles_n; d.tile0 = 0;
  // "dw": P problems dW|db = G^T [X | 1]  (M = out, N = in + 1, K = batch; both operands row-contiguous, synthesised ones column):
  // argv: out in batch dw P    e.g. 512 512 2048 dw 10 = the hidden-layer dW problems of TQC's five critics in one launch
  const bool dw = argc > 4 && !strcmp(argv[4], "dw");
  int total_tiles = d.ntiles;
  double flops = 2.0 * M * N * K;
  std::function<void()> dw_check;
  if (dw) {
    const int P = argc > 5 ? atoi(argv[5]) : 10, out = M, in = N, batch = K;
    float *G, *X, *dW, *db;
    CK(hipMalloc(&G, (size_t)batch * out * 4)); CK(hipMalloc(&X, (size_t)batch * in * 4));
    CK(hipMalloc(&dW, (size_t)P * out * in * 4)); CK(hipMalloc(&db, (size_t)P * out * 4));
    CK(hipMemcpy(G, h.data(), (size_t)batch * out * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(X, h.data(), (size_t)batch * in * 4, hipMemcpyHostToDevice));
    const int S = argc > 6 ? atoi(argv[6]) : 1;      // split of the reduction over S workgroups per tile
    gb.n = P; total_tiles = 0;
    for (int i = 0; i < P; ++i) {
      gcrl::GemmDesc& q = gb.d[i]; memset(&q, 0, sizeof(q));
      q.A = G; q.a_rs = 1; q.a_cs = out; q.B = X; q.b_rs = in; q.b_cs = 1; q.C = dW + (size_t)i * out * in; q.c_rs = in;
      q.M = out; q.N = in + 1; q.K = batch; q.ones_col = 1; q.col_out = db + (size_t)i * out;
      q.a_rvec = 1; q.b_rvec = 1;
      q.tiles_n = (in + 63) / 64;
      const int tl = ((q.M + 63) / 64) * q.tiles_n;
      q.ksplit = S;
      if (S > 1) {
        CK(hipMalloc(&q.kpart, (size_t)tl * S * gcrl::kTiledPartStride * 4));
        CK(hipMalloc(&q.kticket, (size_t)tl * 4 * gcrl::kTicketStride)); CK(hipMemset(q.kticket, 0, (size_t)tl * 4 * gcrl::kTicketStride));
      }
      q.ntiles = tl * S; q.tile0 = total_tiles; total_tiles += q.ntiles;
    }
    flops = 2.0 * P * out * (in + 1) * batch;
    dw_check = [=]() {   // problem 0 and P-1 against a host fp64 sum (sampled entries) — the split / ticket path must not lose a partial
      std::vector<float> hw((size_t)out * in), hb(out), hg((size_t)batch * out), hx((size_t)batch * in);
      double worst = 0;
      CK(hipMemcpy(hg.data(), G, hg.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hx.data(), X, hx.size() * 4, hipMemcpyDeviceToHost));
      for (int pi : {0, P - 1}) {
        CK(hipMemcpy(hw.data(), dW + (size_t)pi * out * in, hw.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hb.data(), db + (size_t)pi * out, hb.size() * 4, hipMemcpyDeviceToHost));
        for (int smp = 0; smp < 400; ++smp) {
          const int m = (smp * 37 + 5) % out, n = (smp * 101 + 3) % in;
          double r = 0, rb = 0;
          for (int k = 0; k < batch; ++k) { r += (double)hg[(size_t)k * out + m] * hx[(size_t)k * in + n]; rb += hg[(size_t)k * out + m]; }
          worst = std::max(worst, std::abs(r - hw[(size_t)m * in + n]));
          worst = std::max(worst, std::abs(rb - hb[m]));
        }
      }
      printf("  dW check (400 sampled entries + their db, problems 0 and %d): worst |err| %.3g\n", P - 1, worst);
    };
  }
  hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  auto launch = [&]() {
    if (!direct) hipLaunchKernelGGL(gcrl::gemm_tiled_kernel, dim3(total_tiles), dim3(256), 0, st, gb);
    else if (dx) hipLaunchKernelGGL(k_dx, dim3(d.ntiles), dim3(256), 0, st, gb);
    else hipLaunchKernelGGL(k_fwd, dim3(d.ntiles), dim3(256), 0, st, gb);
  };
  for (int i = 0; i < 5; ++i) launch();
  CK(hipStreamSynchronize(st));
  const int reps = 30;
  CK(hipEventRecord(a, st));
  for (int i = 0; i < reps; ++i) launch();
  CK(hipEventRecord(b, st)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  const double us = ms * 1e3 / reps;
  if (dw_check) dw_check();
  printf("%s%s M=%d N=%d K=%d (%d tiles): %.1f us  %.1f TFLOP/s\n", dw ? "dW " : (dx ? "dX " : "fwd"), direct ? " (direct)" : "", M, N, K, total_tiles, us, flops / us / 1e6);
#ifdef STAMPS
  {   // one more launch, then the stamps: 100 MHz wall clock, relative to the earliest workgroup start
    launch(); CK(hipStreamSynchronize(st));
    std::vector<unsigned long long> z(8 * (size_t)d.ntiles);
    CK(hipMemcpyFromSymbol(z.data(), HIP_SYMBOL(g_stamps), z.size() * 8));
    unsigned long long t0 = ~0ull;
    for (int i = 0; i < d.ntiles; ++i) t0 = std::min(t0, z[8 * i]);
    const char* nm[6] = {"workgroup start", "main loop begins", "main loop ends", "results stored", "tile in LDS", "epilogue barrier"};
    for (int k : {0, 1, 2, 4, 5, 3}) {
      std::vector<double> v;
      for (int i = 0; i < d.ntiles; ++i) v.push_back((z[8 * i + k] - t0) * 0.01);
      std::sort(v.begin(), v.end());
      printf("  %-18s us after the first workgroup started: min %6.2f  p10 %6.2f  median %6.2f  p75 %6.2f  p80 %6.2f  p85 %6.2f  p90 %6.2f  max %6.2f\n", nm[k], v.front(), v[v.size() / 10], v[v.size() / 2], v[v.size() * 3 / 4], v[v.size() * 8 / 10], v[v.size() * 85 / 100], v[v.size() * 9 / 10], v.back());
      if (k == 0) { int late = 0; for (double x : v) late += x > 8.0; printf("  workgroups that started more than 8 us after the first: %d of %d\n", late, (int)v.size()); }
    }
  }
#endif
  return 0;
}
