// gemm_mfma.hip — launcher of the batched fp32-MFMA GEMM (kernel: gemm_mfma.h).
#include "gemm_mfma.h"

namespace gcrl {

namespace {
template <int TM, int TN, int KSPLIT>
int launch_shape(hipStream_t st, GemmBatch& gb) {
  int tiles = 0;
  for (int i = 0; i < gb.n; ++i) {
    GemmDesc& d = gb.d[i];
    const int tm = (d.M + 16 * TM - 1) / (16 * TM);
    d.tiles_n = (d.N + 16 * TN - 1) / (16 * TN);
    d.ntiles = tm * d.tiles_n;
    d.tile0 = tiles;
    tiles += d.ntiles;
    if (KSPLIT == 1) tiles = (tiles + 3) & ~3;  // a workgroup's 4 waves stay inside one problem
  }
  const int grid = (KSPLIT == 4) ? tiles : (tiles + 3) / 4;
  hipLaunchKernelGGL((gemm_batch_kernel<TM, TN, KSPLIT>), dim3(grid), dim3(256), 0, st, gb);
  GCRL_HIP(hipGetLastError());
  return GCRL_OK;
}
}  // namespace

int launch_gemm_batch(hipStream_t st, GemmDesc* descs, int n, int shape) {
  GCRL_CHECK_ARG(n >= 1 && n <= kMaxProb, "launch_gemm_batch: %d problems (max %d)", n, kMaxProb);
  GemmBatch gb;
  gb.n = n;
  long long tiles16 = 0;
  int kmax = 0;
  for (int i = 0; i < n; ++i) {
    GemmDesc& d = descs[i];
    GCRL_CHECK_ARG(d.M >= 1 && d.N >= 1 && d.K >= 1 && d.A && d.B && d.C, "launch_gemm_batch: bad problem %d (M=%d N=%d K=%d)", i, d.M, d.N, d.K);
    GCRL_CHECK_ARG(!d.ones_col || (d.N >= 2 && d.col_out), "launch_gemm_batch: ones_col needs N >= 2 and col_out");
    d.a_vec = (d.a_cs == 1 && d.a_rs % 4 == 0 && ((uintptr_t)d.A & 15) == 0);
    d.b_vec = (d.b_rs == 1 && d.b_cs % 4 == 0 && ((uintptr_t)d.B & 15) == 0);
    tiles16 += (long long)((d.M + 15) / 16) * ((d.N + 15) / 16);
    kmax = d.K > kmax ? d.K : kmax;
    gb.d[i] = d;
  }
  // latency-bound sizes: one 16x16 tile per workgroup, K split over its 4 waves (fills the
  // chip with <= 1024 tiles); otherwise one tile per wave; large problems: 32x32 per wave.
  if (shape == 1) return launch_shape<1, 1, 4>(st, gb);
  if (shape == 2) return launch_shape<1, 1, 1>(st, gb);
  if (shape == 3) return launch_shape<2, 2, 1>(st, gb);
  if (tiles16 <= 1024 && kmax >= 64) return launch_shape<1, 1, 4>(st, gb);
  if (tiles16 <= 8192) return launch_shape<1, 1, 1>(st, gb);
  return launch_shape<2, 2, 1>(st, gb);
}

}  // namespace gcrl
