// occupancy_probe.hip — how many 256-thread workgroups does a CU REALLY hold at a given static LDS size / register count?
// 1280 workgroups (5 per CU) each spin ~30 us; a workgroup that starts more than 8 us after the first one did not fit.
//   hipcc --offload-arch=gfx950 -O3 tools/occupancy_probe.hip -o tools/occupancy_probe && tools/occupancy_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

template <int BYTES, int REGS>
__global__ __launch_bounds__(256) void spin(unsigned long long* stamps, float* out) {
  __shared__ float buf[BYTES / 4];
  float r[REGS];
#pragma unroll
  for (int i = 0; i < REGS; ++i) r[i] = threadIdx.x * 0.5f + i;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  buf[threadIdx.x] = (float)threadIdx.x;
  __syncthreads();
  while (__builtin_amdgcn_s_memrealtime() - t0 < 3000) {   // 30 us at 100 MHz
#pragma unroll
    for (int i = 0; i < REGS; ++i) r[i] = r[i] * 1.0001f + buf[(threadIdx.x + i) & 255];
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < REGS; ++i) s += r[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) stamps[blockIdx.x] = t0;
}

template <int BYTES, int REGS>
void run(unsigned long long* d_st, float* d_out, int blocks) {
  int occ = 0;
  hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, spin<BYTES, REGS>, 256, 0);
  hipFuncAttributes fa; hipFuncGetAttributes(&fa, (const void*)spin<BYTES, REGS>);
  hipLaunchKernelGGL((spin<BYTES, REGS>), dim3(blocks), dim3(256), 0, 0, d_st, d_out);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(blocks);
  hipMemcpy(h.data(), d_st, blocks * 8, hipMemcpyDeviceToHost);
  const unsigned long long t0 = *std::min_element(h.begin(), h.end());
  int late = 0;
  for (auto t : h) late += (t - t0) > 800;
  printf("LDS %6d B, %3d VGPRs (numRegs), %4d workgroups: occupancy query %d per CU; started late: %4d  -> resident at once: %.2f per CU\n", BYTES, fa.numRegs, blocks, occ,
         late, (blocks - late) / 256.0);
}

int main() {
  unsigned long long* d_st; float* d_out;
  hipMalloc(&d_st, 4096 * 8); hipMalloc(&d_out, 4096 * 256 * 4);
  run<32768, 8>(d_st, d_out, 1280);
  run<32256, 8>(d_st, d_out, 1280);
  run<31744, 8>(d_st, d_out, 1280);
  run<30720, 8>(d_st, d_out, 1280);
  run<28672, 8>(d_st, d_out, 1280);
  run<24576, 8>(d_st, d_out, 1536);
  run<16384, 8>(d_st, d_out, 2048);
  run<16384, 64>(d_st, d_out, 1280);
  run<16384, 72>(d_st, d_out, 1280);
  run<16384, 80>(d_st, d_out, 1280);
  run<16384, 88>(d_st, d_out, 1280);
  run<1024, 8>(d_st, d_out, 2048);
  return 0;
}
