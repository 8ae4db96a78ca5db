// which XCD does workgroup b run on?  (HW_REG_XCC_ID; MI355X guide: workgroups are dealt round-robin over the 8 XCDs)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void census(unsigned int* out) {
  if (threadIdx.x == 0) out[blockIdx.x] = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15u;
}
int main() {
  unsigned int* d; hipMalloc(&d, 4096 * 4);
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(census, dim3(768), dim3(256), 0, 0, d);
    unsigned int h[768]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("launch %d:", rep);
    for (int i = 0; i < 24; ++i) printf(" %u", h[i]);
    int bad = 0; for (int i = 8; i < 768; ++i) bad += h[i] != h[i - 8];
    printf("  ... blocks b and b+8 on different XCDs: %d of 760\n", bad);
  }
  return 0;
}
