// sac_heads.h — SACActorModel's two heads and its sampling as ONE launch (ops_sac.hip heads_sample_kernel; round 5).
// Reference: mean_head / log_std_head (src/model.py:114-115, :121-123), sample() (:125-141).  Before: one batched GEMM launch
// for the heads of both inputs (6.3 us at cfg 5) and one sampling launch (7.1 us: a thread per ROW walking its actions through
// fp64 exp / tanh / log); here a workgroup owns 16 rows — the two heads' 16 x 16 tiles by the batched launch's own tile body
// (gemm_mfma.h gemm_batch_tile<1, 1, 4>: the same bits), then one thread per (row, action) for the sampling arithmetic and one
// per row for the log-prob sum in action order (the same bits again).
#pragma once
#include "gemm_mfma.h"
#include "ops.h"

namespace gcrl {

struct HeadsSampleArgs {
  int n;                 // inputs: 1 or 2
  GemmDesc mean[2];      // [B, H] x mean_head -> head[:, 0:A]        (agent.hip fwd(): bias, no activation)
  GemmDesc lstd[2];      // [B, H] x log_std_head -> head[:, Apad:Apad+A]
  TanhGaussArgs tg[2];   // mu / ls_raw must be the two GEMMs' outputs; tg[0].run: the running-statistics rider (one extra workgroup)
};
int launch_heads_sample(hipStream_t st, HeadsSampleArgs& h);

}  // namespace gcrl
