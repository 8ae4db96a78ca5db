// ops.h — non-GEMM kernels of the update step: TD target + critic loss, actor-loss pieces,
// tanh-Gaussian head, BatchNorm1d, sort/truncate, gradient norm, Adam/AdamW + clip + Polyak.
#pragma once
#include <hip/hip_runtime.h>

#include "common.h"

namespace gcrl {

constexpr int kMaxCritics = 8;
constexpr int kMetricFloats = 32;   // one metrics record per update step (host-mapped memory)
constexpr int kMetricSlots = 4096;
// record layout
enum {
  MET_CRITIC_LOSS = 0,   // [0..7]  per-critic loss
  MET_CRITIC_GRAD = 8,   // [8..15] per-critic post-clip gradient norm
  MET_TD = 16,
  MET_Q = 17,
  MET_ACTOR_LOSS = 18,
  MET_ACTOR_GRAD = 19,
  MET_ALPHA_LOSS = 20,
  MET_ALPHA = 21
};

// Per-step scalars.  The host uploads a table of these (one per step of an update_n call); the
// first kernel of each step copies table[cursor++] to the fixed `cur` slot that all other
// kernels of the (graph-replayed) step read.
struct StepCtrl {
  float step_size_actor, bc2s_actor, decay_actor;    // lr/(1-b1^t), sqrt(1-b2^t), 1-lr*wd
  float step_size_critic, bc2s_critic, decay_critic;
  float step_size_alpha, bc2s_alpha, decay_alpha;
  float grad_scale;
  int batch_slot;     // which pre-gathered batch this step consumes
  int metrics_slot;
  int do_alpha;       // SAC/TQC: step > alpha_min_steps
  unsigned int rng_hi, rng_lo;  // device-RNG counter base of this step
  int pad;
};

struct CtrlBlock {   // device layout: cursor, then cur / prev, then the table
  int cursor;
  int pad[3];
  StepCtrl cur;    // the step whose critic phase is running
  StepCtrl prev;   // software-pipelined DDPG: the step whose actor phase shares the launches
  // second copy, read by the optimiser launch of the row-block path: one thread of that launch
  // can then advance cur/prev for the next step (nobody in that launch reads them), and one thread
  // of the next step's first launch refreshes the copy — no begin_step launch, no atomics
  StepCtrl cur_b, prev_b;
  StepCtrl table[1];
};

// cur <- table[cursor++]; with shift: prev <- cur first
int launch_begin_step(hipStream_t st, CtrlBlock* cb, int shift = 0);

enum { LOSS_MSE = 0, LOSS_SMOOTH_L1 = 1 };
enum { TGT_DDPG = 0, TGT_MIN = 1, TGT_MIN_ENT = 2, TGT_TRUNC_ENT = 3 };

struct TdLossArgs {
  const StepCtrl* cur;
  const float* r;  const float* d;   // + cur->batch_slot * slot_stride
  long long slot_stride;
  const float* qt;          // [C][B] target-critic outputs
  const float* q;           // [C][B] online-critic outputs
  const float* logp_next;   // [B] (entropy targets) or null
  const float* alpha_dev;   // device scalar alpha (TQC) or null -> alpha_const
  float alpha_const;        // SAC: literal 0.2 (src/agent.py:569)
  float* dq;                // [C][B]  dLoss_c/dq_c
  const float* w;           // [B] importance-sampling weights of a prioritised batch (src/agent.py:1320-1325 etc.) or null
  float* td_abs;            // [B] max_c |q_c - y| per sample (what PERBuffer.update_priorities consumes) or null
  float* metrics;           // host-mapped records
  int B, C, drop, target_kind, loss_kind;
  float gamma, clamp_lo;
  // layer-per-launch steps whose last optimiser launch advances the control block (AdamArgs::advance): this launch — before
  // every optimiser launch of its step — refreshes the copies cur_b / prev_b that launch reads; null otherwise
  CtrlBlock* refresh;
  // multi-workgroup form (B >= 1024): 64 * ceil(B/256) floats of partial sums and a zeroed ticket word on its own line; null: one workgroup
  float* part; unsigned int* ticket;
};
int launch_td_loss(hipStream_t st, const TdLossArgs& a);

// mean over [C][B] (actor loss of DDPG/TD3 = -mean(Q), TQC post-update q_value metric)
int launch_mean_metric(hipStream_t st, const StepCtrl* cur, const float* x, int n, float scale,
                       float* metrics, int metric_index);
int launch_fill(hipStream_t st, float* x, long long n, float v);

// TD3 target-policy smoothing (src/agent.py:174-179): act = clamp(act + clamp(eps*pn, +-nc), -1, 1)
// act: [B, A] rows with stride ld inside the next-state critic input; eps: injected randn or
// null (device RNG).
int launch_td3_smooth(hipStream_t st, const StepCtrl* cur, float* act, long long slot_stride,
                      int ld, int B, int A, const float* eps, float policy_noise,
                      float noise_clamp, unsigned long long seed);

// counter-hash normals of the device-RNG mode (TD3 smoothing noise, SAC eps when not injected): the ONE definition; stands
// in for torch.randn_like (src/agent.py:175) and Normal.rsample's eps (src/model.py:134) when no noise is injected.
// Box-Muller over two 24-bit uniforms cut from one 64-bit counter hash: u1 in (0,1] (so log is finite), u2 in [0,1).
// Restated in oracle/device_rng_oracle.py; exposed for tests as gcrl_hash_normal_fill (abi_misc.hip).
__device__ inline unsigned long long mix64d(unsigned long long z) {
  z += 0x9e3779b97f4a7c15ull;
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return z ^ (z >> 31);
}
__device__ inline float hash_normal(unsigned long long seed, unsigned long long ctr) {
  const unsigned long long h = mix64d(mix64d(seed) + ctr);
  const float u1 = ((float)((h >> 40) + 1)) * (1.0f / 16777217.0f);  // (0,1]  (the literal rounds to 2^24: scale 2^-24, u1 = 1 occurs)
  const float u2 = (float)((h >> 8) & 0xffffff) * (1.0f / 16777216.0f);
  return sqrtf(-2.0f * logf(u1)) * cosf(6.2831853071795864f * u2);
}

// ---- optimiser -------------------------------------------------------------------------
constexpr int kNormBlocks = 64;
constexpr int kMaxTransposed = 8;
constexpr int kMaxAdamSeg = 2 * kMaxTransposed + 2;
// A net's parameter vector cut into block-sized work items: FLAT segments (256 consecutive elements per
// block) and TILED [rows][cols] weight matrices (one 16x16 tile per block) whose updated values are also
// written, through an LDS transpose, to an [cols][rows] copy (rowchain.h streams those): 64-byte runs
// instead of 4-byte scattered stores (the scattered form cost 3.2 us of a 10.8 us launch).
struct AdamSeg { long long beg, dst; int rows, cols; int blk0, nblk; int tiled, pad; };
// a block of the segmented launch: kAdamPerThread elements per thread — 1 024 consecutive elements of a flat segment, or a
// kAdamTile x kAdamTile tile of a hidden layer's weight as four 16 x 16 sub-tiles (the launchers count `nblk` accordingly)
constexpr int kAdamPerThread = 4, kAdamTile = 32;
// partial[net][kNormBlocks] = sum of squares of block-strided chunks of g[net][n]
int launch_sumsq(hipStream_t st, const float* g, long long n, long long net_stride, int nets,
                 float* partial);

int launch_sumsq2(hipStream_t st, const float* g0, long long n0, float* partial0, const float* g1, long long n1,
                  float* partial1);

// log-alpha AdamW step (src/agent.py:532-546) as a scalar rider on another launch: thread 0 of block (0,0) of the
// actor's optimiser launch performs it (alpha_update_kernel's phase 1), saving that launch.  log_alpha == null: off
struct AlphaStep {
  float* log_alpha; float* m; float* v; float* alpha; const float* grad;
  float beta2, w1, w2, eps;
  float* metrics;
};

__device__ inline void alpha_step(const AlphaStep& a, const StepCtrl& c) {
  if (!c.do_alpha) return;
  const float g = *a.grad * c.grad_scale;
  float p = *a.log_alpha;
  if (c.decay_alpha != 1.0f) p = __fmul_rn(p, c.decay_alpha);
  float m = *a.m, v = *a.v;
  m = __fadd_rn(m, __fmul_rn(a.w1, __fsub_rn(g, m)));
  v = __fadd_rn(__fmul_rn(v, a.beta2), __fmul_rn(__fmul_rn(a.w2, g), g));
  // (sqrtf: correctly rounded under hipcc's default -fhip-fp32-correctly-rounded-divide-sqrt, as torch's CPU sqrt is; __fsqrt_rn is the
  // NATIVE square root in this toolchain unless OCML_BASIC_ROUNDED_OPERATIONS is defined — 1 ulp off)
  const float denom = __fadd_rn(__fdiv_rn(sqrtf(v), c.bc2s_alpha), a.eps);
  p = __fadd_rn(p, __fdiv_rn(__fmul_rn(-c.step_size_alpha, m), denom));
  *a.log_alpha = p; *a.m = m; *a.v = v;
  *a.alpha = expf(p);
  a.metrics[(long long)c.metrics_slot * kMetricFloats + MET_ALPHA] = expf(p);
}

struct AdamArgs {
  const StepCtrl* cur;
  int which;            // 0 actor, 1 critic, 2 alpha : selects the StepCtrl triple
  float* p; const float* g; float* m; float* v;  // net i at + i*net_stride
  float* target;        // Polyak destination (null: none), same stride
  long long n, net_stride;
  int nets;
  const float* partial; // sum-of-squares partials of net i at partial + i*part_stride, nparts each
  int nparts; long long part_stride;  // (launch_sumsq: kNormBlocks; fused into the dW epilogues: per-tile)
  float clip[kMaxCritics];   // per net max_norm; < 0: no clipping (TD3 critic_1, src/agent.py:201)
  float beta2, w1, w2, eps;   // w = fp32(1 - beta) formed in double on the host, like torch
  float tau, one_m_tau;
  int polyak;           // 1: target = tau*p + (1-tau)*target after the step
  float* metrics;       // post-clip grad norm -> metrics[slot][metric_index + net]
  int metric_index;
  // optional fused scalar metric: metrics[slot][mean_index] = mean_scale * mean(mean_x[0..mean_n))
  // (actor loss -Q.mean() of DDPG/TD3, saving a launch); computed by block (0,0)
  const float* mean_x; int mean_n; float mean_scale; int mean_index;
  // optional TD metrics of a DDPG critic step whose loss was formed inside the row-block kernel:
  // critic loss mean((q-y)^2), mean |q-y|, mean q from the per-row q / y it left behind (block (0,0))
  // (td_q: [td_C][td_n], td_loss_kind LOSS_*; the sums td_loss_kernel would have formed)
  const float* td_q; const float* td_y; int td_n; int td_C; int td_loss_kind;
  // optional [in][out] copies of weight matrices (rowchain.h streams them in the forward pass):
  // element (o, k) of a tiled segment also lands at wt[seg.dst + k*rows + o].  With n_seg > 0 the
  // launch covers the net by segments (seg_blocks blocks) instead of a flat grid-stride loop
  // optional end-of-step control advance: thread 0 of block (0,0) does what begin_step(shift)
  // would do for the next step (prev <- cur, cur <- table[cursor++]), saving that launch.  Legal
  // only when this launch reads the copies cur_b / prev_b (CtrlBlock)
  CtrlBlock* advance;
  AlphaStep alpha;
  // net i's copies live at wt + i*wt_net_stride; wt_target: the same for the Polyak destination
  float* wt; float* wt_target; long long wt_net_stride;
  int n_seg, seg_blocks;
  AdamSeg seg[kMaxAdamSeg];
};
int launch_adam(hipStream_t st, const AdamArgs& a);
int launch_adam_pair(hipStream_t st, const AdamArgs& a0, const AdamArgs& a1);  // two single-net steps, one launch
int launch_polyak(hipStream_t st, const float* p, float* target, long long n, double tau);

// ---- dw_adam.hip: weight gradients + global-norm clip + Adam(W) (+ Polyak, [in][out] copies, metrics, control advance) of up
// to two plain MLPs in ONE launch (round 5).  The reference's sequence is backward -> clip_grad_norm_ -> optimizer.step()
// (src/agent.py:1326-1333 critic, :1288-1300 actor); nothing but the global norm stands between a gradient element and its
// parameter's step.  A workgroup owns one 16x16 tile of one layer's dW | db problem (gemm_mfma.h: the SAME tile body as the
// batched GEMM launch, K split over its four waves), requests its tile's p / m / v / target first, publishes the tile's sum of
// squares into its own slot, re-loads every slot of its net until none is the "not written yet" pattern — the data is the flag,
// no atomics — sums them in slot order (the order adam_kernel sums the GEMM launch's partials: results are the same bits as the
// two-launch form), and steps the elements it still holds.  Every workgroup of the launch must be resident at once (launcher
// checks, dw_adam_capacity); a wait is bounded and reported (meet.h).
constexpr int kFusedMaxLayers = 5;      // dW problems per net (hidden layers + head)
constexpr int kFusedMaxSlotsPerThread = 4;   // a net has at most 256 * this many tiles
struct GemmDesc;
struct DwAdamLayer {
  long long pw, pb;      // offsets of the layer's weight [out][in] and bias [out] inside the net's parameter vector
  long long wt_dst;      // offset of the weight's [in][out] copy inside wt (hidden layers); < 0: none
  int slot0, pad;        // the layer's first norm slot (adam_kernel's order: by layer, then by tile)
};
struct DwAdamNetArgs {
  int nl, ntiles;        // dW problems; 16x16 tiles of all of them = workgroups = norm slots
  DwAdamLayer lay[kFusedMaxLayers];
  const StepCtrl* cur; int which;   // as AdamArgs (the control block's copy: this launch may advance the original)
  float *p, *m, *v, *target;        // the net's parameter vector, Adam moments, Polyak destination (null: none)
  float *wt, *wt_target;            // [in][out] copies of the net / of the Polyak destination
  float clip; int polyak; int metric_index;
  // per-tile sums of squares as fp64 bit patterns, two arrays of slot_stride words used by alternate launches of this net
  // (`seq` counts them); all-ones = "not written yet": a workgroup writes its slot of this launch's array and puts its slot of
  // the other array back
  unsigned long long* slots; int slot_stride; unsigned int* seq;
  const float* mean_x; int mean_n; float mean_scale; int mean_index;           // riders, as AdamArgs
  const float* td_q; const float* td_y; int td_n, td_C, td_loss_kind;
};
long long dw_adam_capacity();   // workgroups of the launch resident at once on the current device (0: shared device / query failed)

// ---- SAC / TQC pieces (ops_sac.hip) -------------------------------------------------------
// nn.BatchNorm1d in training mode followed by ReLU (src/model.py:106-108): z [B,H] -> h [B,H];
// saves xhat [B,H] and invstd [H] when `xhat` is non-null; updates running_mean/var
// (momentum 0.1, unbiased variance) in place.
// scratch: 2 * ceil(B/64) * H floats (per-row-block partial statistics)
int launch_bn_relu_fwd(hipStream_t st, const float* z, int B, int H, const float* gamma,
                       const float* beta, float* h, float* xhat, float* invstd,
                       float* running_mean, float* running_var, float* scratch);
// The same for up to two independent inputs of the SAME layer in one pair of launches (SAC / TQC: actor.sample(next_state)
// of the critic phase and actor.sample(states) of the actor phase use the same actor parameters, src/agent.py:558 and
// :514, so the two forwards run layer by layer in the same launches).  The running statistics take problem 0's batch
// first, then problem 1's — the order of the reference's two forward calls.
struct BnFwdProb { const float* z; float* h; float* xhat; float* invstd; float* scratch; };
// Data-parallel SyncBN (new design: the reference is single-process): the batch statistics of BatchNorm1d are
// those of the CONCATENATED batch of all ranks (rank r's rows = rows [r*B, (r+1)*B) of it), so that G ranks x B rows equal
// 1 rank x G*B rows for the BatchNorm actors too.  Every rank writes its row-block partials into its slots of a
// [world][...] array, zeros the other ranks' slots, `exchange` sums the array over the ranks in place (an all-gather spelt
// as an all-reduce: exact, the other addends are zeros), and every rank merges ALL partials in the same fixed order — the
// order a single process with the big batch uses.  scratch then holds 2 * world * ceil(B/64) * H floats per problem.
struct BnSync {
  int world = 1, rank = 0;
  int (*exchange)(float* dev, long long n, hipStream_t st, void* user) = nullptr;
  void* user = nullptr;
  // where the summed array is after `exchange` (null: in place).  The in-engine peer-to-peer exchange (xchg_ipc.hip) leaves its
  // result in the receive buffer its peers write, never in the array the ranks read from each other
  const float* (*result)(const float* dev, void* user) = nullptr;
};
// rows_per_part: 64 = this launch computes the row-block partials itself (bn_stats launch) — unless `stats_done`: the LDS-tiled
// GEMM that produced z left them in `scratch` (GemmDesc::bn_part, 64-row tiles); 16 = the k-split GEMM that produced z left them
// there; then only the apply launch runs.  scratch: 2 * ceil(B/rows_per_part) * H floats
int launch_bn_relu_fwd_multi(hipStream_t st, const BnFwdProb* probs, int nprob, int B, int H, const float* gamma,
                             const float* beta, float* running_mean, float* running_var, int rows_per_part = 64,
                             const BnSync* sync = nullptr, bool stats_done = false);
constexpr int kBnFusedRows = 16;      // GEMM tile height
constexpr int kBnFusedMaxParts = 32;  // what bn_relu_apply gathers through LDS
// eval mode (running statistics): select_action path
int launch_bn_relu_eval(hipStream_t st, const float* z, int B, int H, const float* gamma,
                        const float* beta, const float* running_mean, const float* running_var,
                        float* h);
// backward of BN(train)+ReLU: dh -> dz, dgamma, dbeta
int launch_bn_relu_bwd(hipStream_t st, const float* dh, const float* dh2, const float* xhat, const float* invstd,
                       const float* gamma, const float* beta, int B, int H, float* dz, float* dgamma, float* dbeta,
                       float* scratch, float* sumsq_out = nullptr,    // sumsq_out[ceil(H/64)]: sum of squares of dgamma | dbeta per column block
                       const BnSync* sync = nullptr);   // sync: dz from the sums over every rank's rows; dgamma | dbeta stay this rank's share

// ---- bn_slab.hip: [Linear -> BatchNorm1d(train) -> ReLU] as one launch per layer and direction (B <= 512) ----
// Forward, one or two inputs of the same layer: h = relu(bn(X W^T + b)); xhat / invstd saved when non-null; the batch
// statistics (mean [H], biased variance [H]) go to bstat — the running statistics are updated by the step's tanh-Gaussian
// launch (BnRunning below), in input order.
struct BnSlabFwdProb { const float* X; long long x_slot; float* h; float* xhat; float* invstd; float* bstat; };
struct BnSlabFwd {
  BnSlabFwdProb p[2]; int n;
  const int* slot;                        // X rows at X + (*slot) * x_slot (null / x_slot 0: none)
  const float *W, *bias, *gamma, *beta;   // W [H][K]
  long long ldx;
  int B, H, K;
  // rsplit > 1 (and B > 128): the rows split over ceil(B/128) workgroups per slab that exchange their column partials through
  // `xchg` (bn_slab_xchg_floats(H) floats) and wait on `bar` (bn_slab_bar_words(H) words); both initialised by bn_slab_scratch_reset — bn_slab.hip
  // `status`: host-visible word that takes MEET_ERR_BN_SLAB when a wait inside the launch times out (meet.h; may be null)
  int rsplit; float* xchg; unsigned int* bar; unsigned int* status;
};
// Backward of layer l: dh = sum_u G[u] . W[u] (the consuming layers: W[u] is [K[u]][ldw] row-major, its first H columns used),
// ReLU mask from xhat, dz written over xhat, dgamma / dbeta, sumsq_out[H/16] (sum of squares of dgamma | dbeta per slab).
struct BnSlabBwd {
  const float* G[2]; long long ldg[2]; int K[2]; const float* W[2]; long long ldw[2]; int nup;
  float* xhat_dz;
  const float *invstd, *gamma, *beta;
  float *dgamma, *dbeta, *sumsq_out;
  int B, H;
  int rsplit; float* xchg; unsigned int* bar; unsigned int* status;   // as in BnSlabFwd
  // the top layer of the SAC / TQC actor (nup == 2: G[0] / G[1] are the two heads' output gradients): the launch FORMS them itself from the critics'
  // action gradients (fold_tg: the arguments of launch_tanh_gauss_bwd, whose gmu / gls must be G[0] / G[1]) and, with fold_sel / fold_al, carries
  // the selection + log-alpha block of launch_tanh_gauss_bwd_select as one more workgroup.  Row-split form only (bn_slab_bwd_can_fold).
  const struct TanhGaussBwdArgs* fold_tg; const struct ActorSelArgs* fold_sel; const struct AlphaArgs* fold_al;
};
bool bn_slab_bwd_can_fold(int B, int H, int A);
bool bn_slab_ok(int B, int H);
long long bn_slab_xchg_floats(int H);
long long bn_slab_bar_words(int H);
bool bn_slab_data_flag();   // the row groups exchange through words that are their own flags (default) / through a counter meeting (GCRL_SLAB_MEET=1)
int bn_slab_scratch_reset(float* xchg, unsigned int* bar, int H, hipStream_t st);   // all words "not written yet", counters zero
int bn_slab_row_split(int B, int H, int n_inputs);   // row groups the launchers use when a layer asks for the split (1: none) on this device
int launch_bn_linear_fwd_slab(hipStream_t st, const BnSlabFwd& f);
int launch_bn_linear_bwd_slab(hipStream_t st, const BnSlabBwd& b);
// running_mean / running_var of `layers` BatchNorm layers from the batch statistics the slab launches left
// (bstat[i]: [layers][2][H], input 0's batch first, then input 1's: the order of the reference's two forward calls)
struct BnRunning { const float* bstat[2]; int n; float* rmean; float* rvar; int layers, H, B; };

// SACActorModel.sample (src/model.py:125-141): mean/log_std head outputs -> action + log-prob.
struct TanhGaussArgs {
  const StepCtrl* cur;
  const float* mu; const float* ls_raw; int ld_head;  // [B,A] head outputs
  const float* eps;        // injected N(0,1) [B,A] or null (device RNG, stream id rng_stream)
  float* act; long long act_slot_stride; int ld_act;  // destination rows (inside a critic input)
  float* logp;             // [B]
  float* save_eps; float* save_std;   // [B,A] (null in the no-grad pass)
  int B, A;
  int deterministic;       // act = tanh(mu), no log-prob
  unsigned long long seed; int rng_stream;
  BnRunning run;           // layers > 0: one extra workgroup of the launch updates the running statistics
};
int launch_tanh_gauss_fwd(hipStream_t st, const TanhGaussArgs& a);
// (the two heads' GEMMs AND the sampling of one or two inputs as one launch: sac_heads.h)
int launch_tanh_gauss_fwd2(hipStream_t st, const TanhGaussArgs& a0, const TanhGaussArgs& a1);   // two heads, one launch

// actor loss of SAC/TQC: L = mean(alpha*logp - sel(q_c)) with sel = min (C=2) or mean of the
// lowest C-drop of the sorted ensemble (src/agent.py:516-521, :916-925).  Writes
// dq[c][b] = dL/dq_c(b) and the loss metric.
struct ActorSelArgs {
  const StepCtrl* cur;
  const float* q;      // [C][B]
  const float* logp;   // [B]
  const float* alpha_dev; float alpha_const;
  float* dq;           // [C][B]
  float* metrics;
  int B, C, drop;
  // rider (launch_actor_select_alpha only): metrics[mean_index] = mean(mean_x[0 .. mean_n)) by a second workgroup — the q_value
  // metric of the re-evaluated critics (launch_mean_metric's sums), one single-workgroup launch less per step
  const float* mean_x; int mean_n, mean_index;
  // multi-workgroup form of launch_actor_select_alpha (B >= 1024): 8 * ceil(B/256) floats of partial sums and a zeroed ticket word; null: one workgroup
  float* part; unsigned int* ticket;
};
int launch_actor_select(hipStream_t st, const ActorSelArgs& a);

// backward of the tanh-Gaussian head: action grads of the critics -> grads of the two heads.
struct TanhGaussBwdArgs {
  const float* dact;   // [C][B][ld_dact] dL/da from each critic's input gradient
  int C, ld_dact; long long dact_stride;
  const float* act; long long act_slot_stride; int ld_act; const StepCtrl* cur;
  const float* eps; const float* std; const float* ls_raw; int ld_head;
  const float* alpha_dev; float alpha_const;
  float* gmu; float* gls; int ld_g;   // [B,A]
  int B, A;
};
int launch_tanh_gauss_bwd(hipStream_t st, const TanhGaussBwdArgs& a);

// alpha_update (src/agent.py:532-546, :936-949): AdamW on the scalar log_alpha
struct AlphaArgs {
  const StepCtrl* cur;
  const float* logp; int B;
  float target_entropy;
  float* log_alpha; float* m; float* v; float* alpha;  // device scalars
  float* grad_out;   // the scalar gradient (for get("grad:log_alpha") and DP)
  float beta2, w1, w2, eps;
  float* metrics;
  int phase;  // 0: gradient + loss metric only, 1: optimiser step only, 2: both
};
int launch_alpha_update(hipStream_t st, const AlphaArgs& a);
// actor_select + the gradient / loss-metric half of alpha_update (phase 0) as one launch
int launch_actor_select_alpha(hipStream_t st, const ActorSelArgs& a, const AlphaArgs& al);
// tanh_gauss_bwd + (as one extra block) actor_select + the gradient half of alpha_update
int launch_tanh_gauss_bwd_select(hipStream_t st, const TanhGaussBwdArgs& a, const ActorSelArgs& s, const AlphaArgs& al);

// ---- distributional TQC (BASELINE.json configs[3]: "25 quantiles x 2 critics, top-2 truncate"; Kuznetsov et al. 2020).
// NOT the reference's TQC (an ensemble of scalar critics, SURVEY.md headline facts): no reference parity, pinned to
// oracle/quantile_tqc_oracle.py only.  Every critic outputs Q atoms; C*Q <= 64 so a row's pooled atoms sort in ONE wavefront.
struct QuantileArgs {
  const StepCtrl* cur;
  const float* r; const float* d; long long slot_stride;   // + cur->batch_slot * slot_stride
  const float* zt;         // [C][B][Q] target-critic atoms of (ns, a')
  const float* z;          // [C][B][Q] online-critic atoms of (s, a)
  const float* logp_next;  // [B]
  const float* alpha_dev;  // device scalar alpha
  float* y;                // [B][64] kept target atoms (K = C*Q - C*drop per row)
  float* dz;               // [C][B][Q] dLoss_c / dz_c
  float* row_loss;         // [C][B] per-row loss sums (reduced by the metrics pass)
  float* row_td;           // [B] max_c |mean_K y - mean_Q z_c|
  float* metrics;
  int B, C, Q, drop;       // drop: atoms dropped PER CRITIC (top drop*C of the pooled, sorted atoms)
  float gamma;
};
// pooled sort + truncation -> y; quantile-Huber loss (kappa = 1) of every critic against y and its gradient; metrics
int launch_quantile_td(hipStream_t st, const QuantileArgs& a);
// mean of z over all atoms of all critics per row: actor objective of the distributional variant
//   loss = mean_b(alpha * logp_b - mean_{c,q} z_c(s_b, pi(s_b))_q);  dz = -1 / (B*C*Q) everywhere
struct QuantileActorArgs {
  const StepCtrl* cur; const float* z; const float* logp; const float* alpha_dev; float* dz; float* metrics; int B, C, Q;
};
int launch_quantile_actor(hipStream_t st, const QuantileActorArgs& a);

// one wavefront per row: bitonic sort of width<=64 values by cross-lane exchange, mean of the
// lowest width-drop
int launch_sort_truncate_mean(hipStream_t st, const float* in, long long rows, int width, int drop,
                              float* sorted, float* mean);

}  // namespace gcrl
