// microbench: row-block chain of H x H layers in one launch (rowchain.h) — numerics vs a CPU
// double reference and time per launch / per layer.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include "rowchain.h"
using namespace gcrl;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int RG>
__global__ __launch_bounds__(kRowThreads) void chain_kernel(const float* X, int B, int H, const float* Wt, const float* bias,
                                                           int NL, float* Y) {
  extern __shared__ float lds[];
  constexpr int R = 4 * RG;
  const int ldx = H + 4;
  float* xs0 = lds;
  float* xs1 = xs0 + R * ldx;
  float* part = xs1 + R * ldx;
  const int r0 = blockIdx.x * R;
  for (int i = threadIdx.x; i < R * H; i += kRowThreads) {
    const int r = i / H, c = i - r * H;
    xs0[r * ldx + c] = (r0 + r < B) ? X[(long long)(r0 + r) * H + c] : 0.f;
  }
  __syncthreads();
  float* a = xs0; float* b = xs1;
  for (int l = 0; l < NL; ++l) {
    rows_linear<RG>(a, ldx, H, Wt + (long long)l * H * H, H, H, bias + l * H, EPI_LEAKY, part, b, ldx,
                    l == NL - 1 ? Y + (long long)r0 * H : nullptr, H, min(R, B - r0));
    float* t = a; a = b; b = t;
  }
}

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 256, H = argc > 2 ? atoi(argv[2]) : 256, NL = argc > 3 ? atoi(argv[3]) : 8;
  std::vector<float> X((size_t)B * H), W((size_t)NL * H * H), bias((size_t)NL * H);
  srand(1);
  for (auto& v : X) v = rand() / (float)RAND_MAX - 0.5f;
  for (auto& v : W) v = (rand() / (float)RAND_MAX - 0.5f) * 0.15f;
  for (auto& v : bias) v = (rand() / (float)RAND_MAX - 0.5f) * 0.1f;
  float *dX, *dW, *db, *dY;
  CK(hipMalloc(&dX, X.size() * 4)); CK(hipMalloc(&dW, W.size() * 4)); CK(hipMalloc(&db, bias.size() * 4)); CK(hipMalloc(&dY, X.size() * 4));
  CK(hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(db, bias.data(), bias.size() * 4, hipMemcpyHostToDevice));
  // CPU reference
  std::vector<double> cur(X.begin(), X.end()), nxt(cur.size());
  for (int l = 0; l < NL; ++l) {
    for (int r = 0; r < B; ++r)
      for (int c = 0; c < H; ++c) {
        double s = bias[l * H + c];
        for (int j = 0; j < H; ++j) s += cur[(size_t)r * H + j] * W[((size_t)l * H + j) * H + c];
        nxt[(size_t)r * H + c] = s > 0 ? s : 0.01 * s;
      }
    cur.swap(nxt);
  }
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int RGsel = 1; RGsel <= 4; RGsel *= 2) {
    const int R = 4 * RGsel;
    const size_t lds = (size_t)(2 * R * (H + 4) + 4 * R * kRowChunk) * 4;
    const int grid = (B + R - 1) / R;
    auto launch = [&]() {
      if (RGsel == 1) hipLaunchKernelGGL(chain_kernel<1>, dim3(grid), dim3(kRowThreads), lds, st, dX, B, H, dW, db, NL, dY);
      else if (RGsel == 2) hipLaunchKernelGGL(chain_kernel<2>, dim3(grid), dim3(kRowThreads), lds, st, dX, B, H, dW, db, NL, dY);
      else hipLaunchKernelGGL(chain_kernel<4>, dim3(grid), dim3(kRowThreads), lds, st, dX, B, H, dW, db, NL, dY);
    };
    if (RGsel == 1) CK(hipFuncSetAttribute((const void*)chain_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    if (RGsel == 2) CK(hipFuncSetAttribute((const void*)chain_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    if (RGsel == 4) CK(hipFuncSetAttribute((const void*)chain_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CK(hipMemsetAsync(dY, 0, X.size() * 4, st));
    launch();
    CK(hipStreamSynchronize(st));
    std::vector<float> Y(X.size());
    CK(hipMemcpy(Y.data(), dY, Y.size() * 4, hipMemcpyDeviceToHost));
    double maxerr = 0, maxref = 0;
    for (size_t i = 0; i < Y.size(); ++i) { maxerr = fmax(maxerr, fabs(Y[i] - cur[i])); maxref = fmax(maxref, fabs(cur[i])); }
    for (int i = 0; i < 20; ++i) launch();
    CK(hipEventRecord(e0, st));
    const int reps = 200;
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("rows/WG %2d grid %4d lds %6zu B: max|err| %.3e (max|ref| %.3f)  %.2f us/launch  %.2f us/layer\n", R, grid, lds, maxerr, maxref,
           ms * 1e3 / reps, ms * 1e3 / reps / NL);
  }
  return 0;
}
