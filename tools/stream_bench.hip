// stream_bench: how fast can ONE workgroup per CU stream a 256 KB weight matrix (256 rows x 1 KB, each wave a quarter of
// the rows, every workgroup the same matrix — the row-chain kernel's access pattern) and feed it to 4x4x1 MFMAs?
//   mode 0: 16-byte buffer loads into VGPRs, two stages of 8 rows in flight per wave (rowchain.h's loop)
//   mode 1: global_load_lds_dwordx4 (gfx950: the data goes straight to LDS), D stages of 8 rows in flight per wave,
//           operands back through ds_read_b128
// Prints the time per 256 KB pass per workgroup.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned int v4u __attribute__((ext_vector_type(4)));

__device__ inline __amdgpu_buffer_rsrc_t rsrc_of(const float* p, int bytes) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, bytes, 0x00020000);
}

template <int MODE, int D>
__global__ __launch_bounds__(256) void stream_kernel(const float* W, int nmat, int npass, float* out, unsigned long long* clk, int sync_each) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  constexpr int U = 8;
  v4f acc[4];
  for (int q = 0; q < 4; ++q) acc[q] = (v4f){0.f, 0.f, 0.f, 0.f};
  const float xa = 1.0f + lane * 1e-3f;
  __syncthreads();
  const unsigned long long t0 = wall_clock64();
  for (int p = 0; p < npass; ++p) {
    if (sync_each) __syncthreads();   // a layer boundary: no wave starts the next matrix before all finished this one
    const float* M = W + (size_t)(p % nmat) * 65536 + (size_t)wave * 64 * 256;   // this wave's 64 rows
    if (MODE == 0) {
      const __amdgpu_buffer_rsrc_t rs = rsrc_of(M, 64 * 1024);
      v4u wa[U], wb[U];
      auto load = [&](v4u (&w)[U], int j) {
#pragma unroll
        for (int u = 0; u < U; ++u) w[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, 16 * lane + (j + u) * 1024, 0, 0);
      };
      auto compute = [&](const v4u (&w)[U]) {
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
          for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_4x4x1f32(xa, __uint_as_float(w[u][q]), acc[q], 0, 0, 0);
      };
      load(wa, 0);
      for (int j = 0; j < 64; j += 2 * U) {
        load(wb, j + U);
        compute(wa);
        load(wa, (j + 2 * U) & 63);
        compute(wb);
      }
    } else {
      // per wave: D stages of U rows (1 KB each) in LDS
      float* ring = lds + (size_t)wave * D * U * 256;
      auto issue = [&](int stage, int j) {
#pragma unroll
        for (int u = 0; u < U; ++u)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(M + (size_t)(j + u) * 256 + 4 * lane),
                                           (__attribute__((address_space(3))) void*)(ring + (stage * U + u) * 256), 16, 0, 0);
      };
      auto consume = [&](int stage) {
        v4f w[U];
#pragma unroll
        for (int u = 0; u < U; ++u) w[u] = *(const v4f*)(ring + (stage * U + u) * 256 + 4 * lane);
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
          for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_4x4x1f32(xa, w[u][q], acc[q], 0, 0, 0);
      };
      constexpr int NS = 64 / U;   // stages per pass
#pragma unroll
      for (int s = 0; s < D - 1; ++s) issue(s, s * U);
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        if (s + D - 1 < NS) issue((s + D - 1) % D, (s + D - 1) * U);
        // wait until stage s has landed: at most (stages issued after s) * U loads may remain outstanding
        const int later = (s + D - 1 < NS ? D - 1 : NS - 1 - s);
        if (later >= 3) __builtin_amdgcn_s_waitcnt(0x0f70 | (3 * U & 0xf) | (((3 * U) >> 4) << 14));
        else if (later == 2) __builtin_amdgcn_s_waitcnt(0x0f70 | (2 * U & 0xf) | (((2 * U) >> 4) << 14));
        else if (later == 1) __builtin_amdgcn_s_waitcnt(0x0f70 | (U & 0xf) | ((U >> 4) << 14));
        else __builtin_amdgcn_s_waitcnt(0x0f70);
        consume(s % D);
      }
    }
  }
  const unsigned long long t1 = wall_clock64();
  float s = 0.f;
  for (int q = 0; q < 4; ++q) s += acc[q][0] + acc[q][1] + acc[q][2] + acc[q][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t0; clk[2 * blockIdx.x + 1] = t1; }
}

template <int MODE, int D>
void run(const char* tag, const float* dW, int nmat, int nwg, int npass, float* dout, unsigned long long* dclk, int sync_each) {
  const size_t lds = MODE == 0 ? 0 : (size_t)4 * D * 8 * 1024;
  if (lds > 64 * 1024) CK(hipFuncSetAttribute((const void*)stream_kernel<MODE, D>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  for (int it = 0; it < 3; ++it) {
    hipLaunchKernelGGL((stream_kernel<MODE, D>), dim3(nwg), dim3(256), lds, 0, dW, nmat, npass, dout, dclk, sync_each);
    CK(hipDeviceSynchronize());
  }
  std::vector<unsigned long long> clk(2 * nwg);
  CK(hipMemcpy(clk.data(), dclk, clk.size() * 8, hipMemcpyDeviceToHost));
  double sum = 0, mx = 0;
  for (int b = 0; b < nwg; ++b) { const double us = (clk[2 * b + 1] - clk[2 * b]) / 100.0; sum += us; if (us > mx) mx = us; }   // 100 MHz
  printf("%-36s %s %3d workgroups: %.3f us per 256 KB pass (mean), %.3f (slowest)\n", tag, sync_each ? "barrier/pass" : "free-running", nwg, sum / nwg / npass, mx / npass);
}

int main() {
  const int nmat = 6, npass = 60;
  float* dW; float* dout; unsigned long long* dclk;
  CK(hipMalloc(&dW, (size_t)nmat * 65536 * 4)); CK(hipMalloc(&dout, 256 * 256 * 4)); CK(hipMalloc(&dclk, 2 * 256 * 8));
  std::vector<float> h((size_t)nmat * 65536);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.f - 0.5f;
  CK(hipMemcpy(dW, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  for (int sync_each : {0, 1})
    for (int nwg : {8, 128}) {
      run<0, 2>("VGPR loads, 2 stages of 8 rows", dW, nmat, nwg, npass, dout, dclk, sync_each);
      run<1, 2>("LDS-DMA loads, 2 stages of 8 rows", dW, nmat, nwg, npass, dout, dclk, sync_each);
      run<1, 3>("LDS-DMA loads, 3 stages of 8 rows", dW, nmat, nwg, npass, dout, dclk, sync_each);
      run<1, 4>("LDS-DMA loads, 4 stages of 8 rows", dW, nmat, nwg, npass, dout, dclk, sync_each);
    }
  return 0;
}
