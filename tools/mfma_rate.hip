// clocks per v_mfma_f32_4x4x1 / 16x16x4 with NACC independent accumulators (one wave per SIMD)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));
template <int NACC, int KIND>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* clk, int iters) {
  v4f acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (v4f){0.f, 0.f, 0.f, 0.f};
  float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
  const unsigned long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 16 / NACC; ++r)
#pragma unroll
      for (int i = 0; i < NACC; ++i) {
        if (KIND == 0) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[i], 0, 0, 0);
        else acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
      }
  }
  const unsigned long long t1 = clock64();
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}
template <int NACC, int KIND> void run(float* out, unsigned long long* clk, int waves) {
  const int iters = 1000;
  hipLaunchKernelGGL((k<NACC, KIND>), dim3(64), dim3(64 * waves), 0, 0, out, clk, iters);
  hipLaunchKernelGGL((k<NACC, KIND>), dim3(64), dim3(64 * waves), 0, 0, out, clk, iters);
  hipDeviceSynchronize();
  unsigned long long h; hipMemcpy(&h, clk, 8, hipMemcpyDeviceToHost);
  printf("%s NACC=%2d waves/WG=%d: %.2f clk per MFMA per wave\n", KIND == 0 ? "4x4x1  " : "16x16x4", NACC, waves, (double)h / (iters * 16.0));
}
int main() {
  float* out; unsigned long long* clk;
  hipMalloc(&out, 64 * 512 * 4); hipMalloc(&clk, 8);
  run<1, 0>(out, clk, 4); run<2, 0>(out, clk, 4); run<4, 0>(out, clk, 4); run<8, 0>(out, clk, 4); run<16, 0>(out, clk, 4);
  run<4, 0>(out, clk, 8); run<16, 0>(out, clk, 8);
  run<1, 1>(out, clk, 4); run<4, 1>(out, clk, 4); run<16, 1>(out, clk, 4);
  return 0;
}
