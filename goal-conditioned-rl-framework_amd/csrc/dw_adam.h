// dw_adam.h — launch arguments of the fused weight-gradient + optimiser launch (dw_adam.hip; ops.h describes the protocol).
#pragma once
#include "gemm_mfma.h"
#include "ops.h"

namespace gcrl {

struct DwAdamNet {
  DwAdamNetArgs o;
  GemmDesc d[kFusedMaxLayers];   // the net's dW | db problems (agent.hip bwd_dw), in launch order; tile0 / tiles_n / ntiles by the launcher
};
struct DwAdamArgs {
  int nnets;
  float beta2, w1, w2, eps, tau, one_m_tau;
  float* metrics;          // host-mapped metric records
  CtrlBlock* advance;      // non-null: thread 0 of workgroup (0, 0) advances the control block for the next step
  unsigned int* status;    // host-visible status word (meet.h) or null
  int leaders;             // 1: the slots of a net of >= 64 tiles are swept by its first eight workgroups only (dw_adam.hip); 0: by every workgroup
  int poll_gate;           // no poll of the norm slots before the launch is this many 10-ns ticks old
  int poll_first_sleep, poll_sleep;   // s_sleep arguments (64-clock units) before the first poll of the norm slots / between re-polls of the missing ones
#ifdef GCRL_OF_STAMPS
  unsigned long long* stamps;
#endif
  DwAdamNet net[2];
};
// every problem must be one the batched launch would run on its k-split 16x16 form (gemm_shape_of(d) == 1); fills the tile
// bookkeeping; refuses (GCRL_ERR_STATE) when the launch's workgroups are not all resident at once
int launch_dw_adam(hipStream_t st, DwAdamArgs& a);

}  // namespace gcrl
