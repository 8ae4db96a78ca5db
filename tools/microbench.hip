// microbench.hip — measurement aid (not part of the product): launch floors and GEMM body times.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I goal-conditioned-rl-framework_amd/csrc \
//         tools/microbench.hip goal-conditioned-rl-framework_amd/build/gemm_mfma.o \
//         goal-conditioned-rl-framework_amd/build/lr_sched.o -o /tmp/microbench && /tmp/microbench
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#include "gemm_mfma.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void tiny_kernel(int* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1; }
__global__ void tiny_kernel_host(float* hostmapped) { if (threadIdx.x == 0 && blockIdx.x == 0) hostmapped[0] += 1.f; }

// register-only MFMA loop: the fp32 matrix-core ceiling of this box at the clock it actually holds
__global__ __launch_bounds__(256) void mfma_peak_kernel(float* out, int iters) {
  typedef float v4f __attribute__((ext_vector_type(4)));
  v4f acc[4];
  for (int q = 0; q < 4; ++q) acc[q] = (v4f){0.f, 0.f, 0.f, 0.f};
  float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f + 1.f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[q], 0, 0, 0);
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
}

template <typename F>
float time_graph(hipStream_t st, hipStream_t cap, int reps, F enqueue) {
  hipGraph_t g; hipGraphExec_t ex;
  hipStreamBeginCapture(cap, hipStreamCaptureModeThreadLocal);
  enqueue(cap);
  hipStreamEndCapture(cap, &g);
  hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 5; ++i) hipGraphLaunch(ex, st);
  hipStreamSynchronize(st);
  hipEventRecord(a, st);
  for (int i = 0; i < reps; ++i) hipGraphLaunch(ex, st);
  hipEventRecord(b, st);
  hipEventSynchronize(b);
  float ms = 0; hipEventElapsedTime(&ms, a, b);
  hipGraphExecDestroy(ex); hipGraphDestroy(g);
  return ms * 1e3f / reps;  // us per graph launch
}

int main() {
  hipStream_t st, cap;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&cap, hipStreamNonBlocking));
  int* d; CK(hipMalloc(&d, 256)); CK(hipMemset(d, 0, 256));
  float* hm; CK(hipHostMalloc((void**)&hm, 256, hipHostMallocMapped)); hm[0] = 0;
  float* hmd; CK(hipHostGetDevicePointer((void**)&hmd, hm, 0));

  const int chain = 32;
  float t = time_graph(st, cap, 200, [&](hipStream_t s) { for (int i = 0; i < chain; ++i) hipLaunchKernelGGL(tiny_kernel, dim3(1), dim3(64), 0, s, d); });
  printf("graph: %d dependent tiny kernels (1 WG): %.2f us/kernel\n", chain, t / chain);
  t = time_graph(st, cap, 200, [&](hipStream_t s) { for (int i = 0; i < chain; ++i) hipLaunchKernelGGL(tiny_kernel, dim3(256), dim3(256), 0, s, d); });
  printf("graph: %d dependent tiny kernels (256 WG): %.2f us/kernel\n", chain, t / chain);
  t = time_graph(st, cap, 200, [&](hipStream_t s) { for (int i = 0; i < chain; ++i) hipLaunchKernelGGL(tiny_kernel_host, dim3(1), dim3(64), 0, s, hmd); });
  printf("graph: %d dependent tiny kernels writing host-mapped memory: %.2f us/kernel\n", chain, t / chain);
  {  // eager
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 100; ++i) hipLaunchKernelGGL(tiny_kernel, dim3(1), dim3(64), 0, st, d);
    hipStreamSynchronize(st);
    hipEventRecord(a, st);
    for (int i = 0; i < 2000; ++i) hipLaunchKernelGGL(tiny_kernel, dim3(1), dim3(64), 0, st, d);
    hipEventRecord(b, st); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("eager: 2000 tiny kernels: %.2f us/kernel\n", ms * 1e3f / 2000);
  }

  {  // do two independent branches of one graph overlap?  (fork/join with events during capture)
    float *A2, *B2, *C2a, *C2b, *bias2;
    CK(hipMalloc(&A2, 256 * 256 * 4)); CK(hipMalloc(&B2, 256 * 256 * 4)); CK(hipMalloc(&C2a, 256 * 256 * 4)); CK(hipMalloc(&C2b, 256 * 256 * 4)); CK(hipMalloc(&bias2, 1024));
    CK(hipMemset(A2, 0, 256 * 256 * 4)); CK(hipMemset(B2, 0, 256 * 256 * 4)); CK(hipMemset(bias2, 0, 1024));
    hipStream_t side; CK(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
    hipEvent_t fork, join; hipEventCreateWithFlags(&fork, hipEventDisableTiming); hipEventCreateWithFlags(&join, hipEventDisableTiming);
    auto chain = [&](hipStream_t s, float* Cout, int n) {
      for (int i = 0; i < n; ++i) {
        gcrl::GemmDesc c; memset(&c, 0, sizeof(c));
        c.A = A2; c.a_rs = 256; c.a_cs = 1; c.B = B2; c.b_rs = 1; c.b_cs = 256; c.C = Cout; c.c_rs = 256; c.bias = bias2; c.M = 256; c.N = 256; c.K = 256; c.epi = 1;
        gcrl::launch_gemm_batch(s, &c, 1, 1);
      }
    };
    float one = time_graph(st, cap, 100, [&](hipStream_t s) { chain(s, C2a, 16); });
    float seq = time_graph(st, cap, 100, [&](hipStream_t s) { chain(s, C2a, 16); chain(s, C2b, 16); });
    float par = time_graph(st, cap, 100, [&](hipStream_t s) {
      hipEventRecord(fork, s); hipStreamWaitEvent(side, fork, 0);
      chain(s, C2a, 16); chain(side, C2b, 16);
      hipEventRecord(join, side); hipStreamWaitEvent(s, join, 0);
    });
    printf("graph branches: one chain of 16 gemms %.1f us; 32 sequential %.1f us; two parallel chains of 16 %.1f us\n", one, seq, par);
  }

  {
    float* o; CK(hipMalloc(&o, 1024 * 256 * 4));
    for (int blocks : {256, 512, 1024}) {
      const int iters = 4096;
      hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
      hipLaunchKernelGGL(mfma_peak_kernel, dim3(blocks), dim3(256), 0, st, o, iters);
      hipEventRecord(a, st);
      hipLaunchKernelGGL(mfma_peak_kernel, dim3(blocks), dim3(256), 0, st, o, iters);
      hipEventRecord(b, st); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      const double flop = (double)blocks * 4 /*waves*/ * iters * 4 * (2.0 * 16 * 16 * 4);
      printf("mfma f32 16x16x4 register loop, %4d blocks x 4 waves: %.1f TFLOP/s (%.1f us)\n", blocks, flop / (ms * 1e-3) * 1e-12, ms * 1e3);
    }
  }

  // GEMM bodies at the hot shapes
  struct Shape { int M, N, K; const char* what; };
  std::vector<Shape> shapes = {{256, 256, 256, "fwd hidden B=256 H=256"}, {256, 256, 27, "fwd first layer K=27"},
                               {256, 1, 256, "fwd last layer N=1"},      {256, 257, 256, "dW hidden (TN, K=B=256)"},
                               {2048, 512, 512, "fwd hidden B=2048 H=512"}, {1024, 256, 256, "fwd hidden B=1024 H=256"}};
  float *A, *B, *C, *bias;
  CK(hipMalloc(&A, 2048 * 544 * 4)); CK(hipMalloc(&B, 2048 * 544 * 4)); CK(hipMalloc(&C, 2048 * 512 * 4)); CK(hipMalloc(&bias, 4096));
  CK(hipMemset(A, 0, 2048 * 512 * 4)); CK(hipMemset(B, 0, 2048 * 512 * 4)); CK(hipMemset(bias, 0, 4096));
  for (auto& sh : shapes) {
    for (int shape = 1; shape <= 4; ++shape) {
      for (int form = 0; form < 3; ++form) {
        gcrl::GemmDesc dsc;
        memset(&dsc, 0, sizeof(dsc));
        if (form == 0) { dsc.A = A; dsc.a_rs = sh.K; dsc.a_cs = 1; dsc.B = B; dsc.b_rs = 1; dsc.b_cs = sh.K; }       // NT
        else if (form == 2) { dsc.A = A; dsc.a_rs = sh.K + 16; dsc.a_cs = 1; dsc.B = B; dsc.b_rs = 1; dsc.b_cs = sh.K + 16; }  // NT, padded ld
        else { dsc.A = A; dsc.a_rs = 1; dsc.a_cs = sh.M; dsc.B = B; dsc.b_rs = sh.N; dsc.b_cs = 1; }                    // TN-like strided
        dsc.C = C; dsc.c_rs = sh.N; dsc.bias = bias; dsc.M = sh.M; dsc.N = sh.N; dsc.K = sh.K; dsc.epi = 1;
        const int chainN = 16;
        float us = time_graph(st, cap, 100, [&](hipStream_t s) { for (int i = 0; i < chainN; ++i) { gcrl::GemmDesc c = dsc; gcrl::launch_gemm_batch(s, &c, 1, shape); } });
        printf("gemm %-28s M=%4d N=%3d K=%3d shape=%d %s: %.2f us/launch  (%.2f TFLOP/s)\n", sh.what, sh.M, sh.N, sh.K, shape,
               form == 0 ? "NT " : (form == 1 ? "str" : "NTp"), us / chainN, 2.0 * sh.M * sh.N * sh.K / (us / chainN) * 1e-6);
      }
    }
  }
  return 0;
}
