// mfma_ceiling.hip — measurement aid (not part of the product): the fp32 matrix-core rate this box SUSTAINS.
// Register-only loops of v_mfma_f32_16x16x4_f32 / v_mfma_f32_4x4x1_16b_f32 on every SIMD of the chip (1024 waves x k),
// non-trivial operands, run back to back for >= 1 s before the timed launches (the chip lowers its clock under matrix
// load; a 0.3 ms launch from idle reads high).  Reports TFLOP/s by the host clock over long launches and the in-kernel
// clock (s_memtime ticks per 100 MHz s_memrealtime tick).
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_ceiling.hip -o /tmp/mfma_ceiling && /tmp/mfma_ceiling
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <vector>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));

// v_mfma_f32_32x32x2_f32: NACC independent 32x32 accumulators (16 registers each) per wave
template <int NACC>
__global__ __launch_bounds__(256) void loop32(float* out, unsigned long long* stamps, int iters) {
  v16f acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  float a = 0.5f + (threadIdx.x % 13) * 0.03125f, b = 0.25f + (threadIdx.x % 7) * 0.0625f;
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8 / NACC; ++r)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    a = -a;
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
  out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = c1 - c0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int KIND>
__global__ __launch_bounds__(256) void loop(float* out, unsigned long long* stamps, int iters) {
  v4f acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = (v4f){0.f, 0.f, 0.f, 0.f};
  float a = 0.5f + (threadIdx.x % 13) * 0.03125f, b = 0.25f + (threadIdx.x % 7) * 0.0625f;
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (KIND == 0) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
      else acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[i], 0, 0, 0);
    }
    a = -a;   // keep the accumulators bounded
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = c1 - c0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int KIND>
void run(const char* name, double flop_per_mfma, int blocks, float* out, unsigned long long* stamps) {
  const int waves = blocks * 4;
  auto launch = [&](int iters) {
    if (KIND == 2) hipLaunchKernelGGL((loop32<1>), dim3(blocks), dim3(256), 0, 0, out, stamps, iters);
    else if (KIND == 3) hipLaunchKernelGGL((loop32<2>), dim3(blocks), dim3(256), 0, 0, out, stamps, iters);
    else hipLaunchKernelGGL((loop<(KIND == 1 ? 1 : 0)>), dim3(blocks), dim3(256), 0, 0, out, stamps, iters);
  };
  // warm up for ~1.5 s
  auto t0 = std::chrono::steady_clock::now();
  while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < 1.0) { launch(200000); hipDeviceSynchronize(); }
  for (int iters : {200000}) {
    hipDeviceSynchronize();
    auto a = std::chrono::steady_clock::now();
    launch(iters);
    hipDeviceSynchronize();
    const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - a).count();
    std::vector<unsigned long long> h(2 * blocks);
    hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost);
    double clk = 0;
    for (int i = 0; i < blocks; ++i) clk += (double)h[2 * i] / (double)h[2 * i + 1] * 100e6;
    clk /= blocks;
    const double tf = (double)iters * 8 * flop_per_mfma * waves / sec / 1e12;
    printf("%s  %d blocks x 4 waves, %8d iters: %8.3f ms  %7.1f TFLOP/s  in-kernel clock %.2f GHz  (%.1f clk per MFMA per wave)\n", name, blocks,
           iters, sec * 1e3, tf, clk / 1e9, clk * sec / ((double)iters * 8));
  }
}

int main() {
  float* out; unsigned long long* stamps;
  hipMalloc(&out, (size_t)4096 * 256 * 4); hipMalloc(&stamps, 4096 * 16);
  for (int b : {256, 512, 1024, 2048}) run<0>("16x16x4 f32", 2.0 * 16 * 16 * 4, b, out, stamps);     // 1, 2, 4, 8 waves per SIMD
  for (int b : {256, 512, 768, 1024, 1280, 1536, 2048}) run<2>("32x32x2 f32, 1 accumulator ", 2.0 * 32 * 32 * 2, b, out, stamps);
  for (int b : {256, 512, 1024}) run<3>("32x32x2 f32, 2 accumulators", 2.0 * 32 * 32 * 2, b, out, stamps);
  for (int b : {256, 512, 1024, 2048}) run<1>("4x4x1x16 f32", 2.0 * 4 * 4 * 1 * 16, b, out, stamps);
  return 0;
}
