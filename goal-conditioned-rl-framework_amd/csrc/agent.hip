// agent.hip — the actor-critic update engine behind gcrl_agent_*.
//
// Replaces, for all four reference agents, update() and everything it calls:
//   DDPG      src/agent.py:1288-1343 (actor_update, critic_update), :1378-1404 (update)
//   TD3Agent  src/agent.py:149-251, :281-317
//   SACAgent  src/agent.py:513-639, :659-699
//   TQCAgent  src/agent.py:912-1042, :1062-1100
// and the networks of src/model.py (Actor, Critic, SACActorModel).
//
// One update step is a fixed sequence of launches (batched MFMA GEMMs + small fused
// element-wise / reduction kernels) split in three phases at the two points where a
// data-parallel run exchanges gradients:
//   phase 0  begin_step, target pass + online critic pass, TD target/loss, critic backward
//   phase 1  critic clip+Adam(W)(+Polyak), [TQC metric re-evaluation], actor pass through the
//            stepped critics, actor backward, log-alpha gradient
//   phase 2  actor clip+Adam(W)(+Polyak), log-alpha step
// The sequence is captured once per variant into a hipGraph and replayed; everything that
// changes between steps (learning rates, bias corrections, batch slot, metrics slot, RNG
// counters) travels through a device-resident StepCtrl table, so no pointer is re-bound.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <map>
#include <mutex>
#include <random>
#include <string>
#include <vector>

#include "dw_adam.h"
#include "gemm_mfma.h"
#include "her_ring.h"
#include "meet.h"
#include "ops.h"
#include "rowchain.h"
#include "sac_heads.h"
#include "xchg_ipc.h"

using namespace gcrl;

namespace {

constexpr int kMaxStepsPerCall = 128;
constexpr int kCtrlSlots = 4;
constexpr int kEventRing = 256;
constexpr double kBeta1 = 0.9, kBeta2 = 0.999, kAdamEps = 1e-8, kWeightDecay = 0.01;

struct Lin { int in, out; long long w, b; };

struct NetSpec {
  int in_dim = 0, H = 0, L = 0, out_dim = 0;
  bool sac = false;
  std::vector<Lin> lin;               // L hidden layers, then head(s)
  std::vector<long long> bn_g, bn_b;  // per hidden layer (sac)
  long long numel = 0;
};

// parameter order = torch module.parameters(): Linear w,b [, BatchNorm w,b] per block, heads
NetSpec make_net(int in, int H, int L, int out, bool sac) {
  NetSpec n;
  n.in_dim = in; n.H = H; n.L = L; n.out_dim = out; n.sac = sac;
  long long off = 0;
  for (int l = 0; l < L; ++l) {
    const int k = l == 0 ? in : H;
    n.lin.push_back({k, H, off, off + (long long)k * H});
    off += (long long)k * H + H;
    if (sac) { n.bn_g.push_back(off); n.bn_b.push_back(off + H); off += 2 * H; }
  }
  const int heads = sac ? 2 : 1;  // mean_head, log_std_head (src/model.py:114-115)
  for (int h = 0; h < heads; ++h) {
    n.lin.push_back({H, out, off, off + (long long)H * out});
    off += (long long)H * out + out;
  }
  n.numel = off;
  return n;
}

long long align_up(long long x, long long a) { return (x + a - 1) / a * a; }

struct UploadBlock {  // device image of one call's upload: control table, then batch indices
  CtrlBlock cb;       // cb.table[0] is table entry 0 ...
  StepCtrl more[kMaxStepsPerCall - 1];
  // uint32_t idx[...] follows
};

}  // namespace

struct StepPlan { int variant; int tuple_len; };
// kind 0: phase `b` of planned step `a`; kind 1: pipelined DDPG segment (a = what, b = seg).
// reduce/reduce_n: gradient block to all-reduce after the segment (none when reduce_n == 0)
struct DpSeg { int kind, a, b; float* reduce; long long reduce_n; };

struct gcrl_agent {
  gcrl_agent_config cfg;
  int S = 0, A = 0, H = 0, L = 0, B = 0, C = 0, ldx = 0, Mmax = 0, Apad = 0;
  int Q = 1;   // atoms per critic (distributional TQC variant: n_quantiles > 1; the reference's agents: 1)
  bool has_target_actor = false, sac = false;
  NetSpec actor, critic;
  hipStream_t stream = nullptr, cap_stream = nullptr;

  float *params = nullptr, *grads = nullptr, *adam_m = nullptr, *adam_v = nullptr;
  long long n_params = 0, n_grads = 0;
  long long off_actor = 0, off_tactor = 0, off_critic = 0, off_tcritic = 0, off_logalpha = 0;
  long long goff_critic = 0, goff_actor = 0, goff_alpha = 0;
  long long critic_stride = 0;
  float *bn_rmean = nullptr, *bn_rvar = nullptr;  // [L*H]
  float* alpha_dev = nullptr;                     // exp(log_alpha) as of the last alpha step

  float* work = nullptr;
  float *sa = nullptr, *nsa = nullptr, *spa = nullptr, *rbuf = nullptr, *dbuf = nullptr;
  long long slot_x = 0, slot_rd = 0;
  float *hTA[2] = {}, *hA = nullptr, *hC = nullptr, *hTC = nullptr, *gC = nullptr, *gA[2] = {};
  float *hC2 = nullptr, *gC2 = nullptr, *bn_part = nullptr;
  // multi-workgroup reductions (td_loss, actor_select_alpha at B >= 1024): [64 nb] + [8 nb] partial sums, then two ticket words on their own lines
  float* red_scratch = nullptr;
  float* red_td() const { return red_scratch; }
  float* red_sel() const { return red_scratch + 64LL * ((B + 255) / 256); }
  unsigned int* red_ticket(int i) const { return reinterpret_cast<unsigned int*>(red_scratch + 72LL * ((B + 255) / 256) + 32 * i); }
  float *q = nullptr, *qt = nullptr, *q2 = nullptr, *dq = nullptr, *dq2 = nullptr, *dact = nullptr;
  float *zA = nullptr, *xhatA = nullptr, *invstdA = nullptr, *headA = nullptr, *ghead = nullptr, *dh2 = nullptr;
  float *zN = nullptr, *hN = nullptr, *headN = nullptr, *bn_partN = nullptr;   // scratch of the co-scheduled actor.sample(next_state) forward
  float *logp = nullptr, *logp_next = nullptr, *epsbuf = nullptr, *stdbuf = nullptr;
  float *noise_in = nullptr, *eps_next_in = nullptr, *eps_cur_in = nullptr, *norm_partial = nullptr;
  float *w_in = nullptr, *td_abs = nullptr;   // prioritised replay: IS weights of the batch [B], per-sample |td| [B]
  float *act_in = nullptr, *act_tmp[2] = {};
  float *qy = nullptr, *q_row_loss = nullptr, *q_row_td = nullptr;   // distributional TQC: kept target atoms [B][64], per-row sums
  float* pi_buf = nullptr;   // SAC row-chain path: pi(s) [B][Apad] (the layer-per-launch paths keep it in spa's action columns)
  float* act_pinned = nullptr;   // host staging of gcrl_agent_act_host
  char *oa_pinned = nullptr, *oa_dev = nullptr;   // staging of gcrl_agent_observe_act (raw rows, noise, actions)
  // ... and of its inline form (rowchain.h RowActInline): float64 actions + one flag per workgroup, host-visible
  char* act_fl_host = nullptr; char* act_fl_dev = nullptr; unsigned long long act_seq = 0;
  size_t oa_bytes = 0;
  // row-block DDPG path (rowchain.h): [in][out] weight copies of actor | target actor | critic 0 |
  // target critic 0, per-layer gradient buffers, TD targets
  bool rowchain = false, wt_dirty = true;
  BnSync bn_sync;             // data-parallel SyncBN (gcrl_agent_dp_sync_bn): world > 1 -> batch statistics over every rank's rows
  gcrl_dp* bn_sync_dp = nullptr;
  gcrl_exchange_fn bn_sync_fn = nullptr;
  void* bn_sync_user = nullptr;
  float* bn_sync_buf = nullptr;
  long long bn_sync_cap = 0;   // floats allocated
  gcrl_xchg* bn_xchg_h = nullptr;   // SyncBN partials through the in-engine peer-to-peer exchange (round 5; not owned): graphs stay on
  // dW problems at batch >= 1024 on the LDS-tiled form with the reduction split over dw_split_[c|a] workgroups per tile
  // (gemm_tiled.h; 1: off): partial tiles and tickets per net and layer
  int dw_split_c = 1, dw_split_a = 1;
  float *dw_part = nullptr, *dw_tick = nullptr;
  std::vector<long long> dwp_off_c, dwp_off_a, dwt_off_c, dwt_off_a;
  long long dwp_cstride = 0, dwt_cstride = 0, dwp_actor = 0, dwt_actor = 0;
  bool dw_batch_off = false;  // GCRL_NO_DW_BATCH=1: a large ensemble's dW problems stay with their layers' dX launches (A/B knob)
  // [Linear -> BatchNorm -> ReLU] of the SAC / TQC actor as one launch per layer and direction (bn_slab.hip): B <= 512, no
  // SyncBN (GCRL_NO_BN_SLAB=1: the GEMM + BatchNorm launches).  bn_bstat: batch statistics [input][L][2][H]
  bool bn_slab = false;
  bool slab_on() const { return bn_slab && bn_sync.world <= 1; }
  float* bn_bstat = nullptr;
  float *bn_xchg = nullptr, *bn_bar = nullptr;   // row-group exchange of the slab launches (bn_slab.hip): partials, barrier words
  // round 5: with 64-row workgroups (bn_slab.hip) the narrow launches — the first layer's forward (K = state_dim), the top layer's backward (K = 2 x
  // action_dim) — gain from the row split as well: SAC cfg 5 157.9 -> 155.6 us/step (profiles/r05_ab_slab_waves.txt); GCRL_NO_SLAB_SPLIT_ALL=1: round 3's rule (K >= 128 only)
  bool bn_split_all = std::getenv("GCRL_NO_SLAB_SPLIT_ALL") == nullptr;
  bool tg_fold_off = std::getenv("GCRL_NO_TG_FOLD") != nullptr;   // A/B knob: the sampling backward as its own launch (round 4's form)
  // round 5: the actor's heads + sampling inside the row-chain launches that consume the actions (rowchain.h HeadsFold); GCRL_NO_HEADS_FOLD=1: the
  // heads + sampling launch of sac_heads.h.  hf_*: what sac_actor_forwards leaves for the chain launches of the same step
  bool heads_fold_off = std::getenv("GCRL_NO_HEADS_FOLD") != nullptr;
  const float *hf_eps_next = nullptr, *hf_eps_cur = nullptr;
  int hf_nf = 0;
  int bn_rsplit = 1;          // > 1: K >= 128 slab launches split their rows over ceil(B/128) workgroups (GCRL_NO_BN_RSPLIT=1: off)
  int bn_slots = 0;           // sum-of-squares slots of one BatchNorm layer's dgamma | dbeta (16-column slabs)
  bool heads_fused_off = false;   // GCRL_NO_HEADS_FUSED=1: the BatchNorm actor's heads and its sampling as two launches (rounds 1-4)
  bool red_off = false;       // GCRL_NO_MB_REDUCE=1: single-workgroup td_loss / actor_select_alpha at every batch size
  bool layer_adv_off = false; // GCRL_NO_LAYER_ADV=1: begin_step launches on the layer-per-launch path as in rounds 1-3
  bool bn_fused_tiled = true; // GCRL_NO_BN_TILED_STATS=1 turns it off: the LDS-tiled GEMM's epilogue leaves the 64-row BatchNorm partials (no bn_stats launch)
  bool bn_fused = false;      // GCRL_BN_FUSED=1: BatchNorm statistics out of the producing GEMM's epilogue instead of bn_stats launches
  bool split_k = false;       // TD3: critic phase as role-parallel launches (agent_rowchain.inc)
  bool split_roles = false;   // twin-critic phases as role-parallel launches (rowchain.h launch_rowchain_split)
  int n_cus = 0;              // compute units of the device (residency checks of the launches whose workgroups meet)
  bool rc_merge = false;      // ... forward and backward part in ONE launch each (part 3; GCRL_NO_RC_MERGE=1: two launches)
  bool rc_merge_k = false;    // TD3 (split_k): the critic phase's two launches as one, producers / consumers form (meet.h)
  bool ddpg_ksplit = false, ddpg_ksplit_can = false;   // DDPG: the critic phase as two roles of the fused launch (rowchain.hip, k_split)
  // the dW | db GEMMs, the clip and the optimiser step of the row-chain DDPG step as ONE launch (dw_adam.hip; GCRL_NO_OPT_FUSE=1: the
  // two launches): per net [critic 0 | actor] two arrays of of_stride 64-bit norm slots and a launch count on its own line
  bool opt_fuse = false, opt_fuse_can = false;
  float *of_slots = nullptr, *of_seq = nullptr;
  long long of_stride = 0;
  float* rc_bar = nullptr;    // meeting counters of the row blocks [2][nblk][32 words]
  long long rc_bar_words = 0;
  // weight-slice form of the DDPG launch (rowtile.hip): a 16 x 16 tile of every layer per workgroup, hand-offs inside the launch
  bool rowtile = false, rowtile_can = false;
  float *rt_xb = nullptr, *rt_qpart = nullptr, *rt_ctr = nullptr, *rt_xid = nullptr;
  long long rt_xb_floats = 0, rt_part_floats = 0;
  long long rt_ctr_words = 0;   // 64-bit words
  // host-visible status word of the launches whose workgroups wait for each other (meet.h): a timed-out wait sets a bit, the
  // next host synchronisation of this handle returns GCRL_ERR_STATE, zeroes the counters and clears it (meet_check below)
  unsigned int *status_host = nullptr, *status_dev = nullptr;
  // data-parallel gradient exchange inside the launch sequence (xchg_ipc.hip; gcrl_agent_set_exchange): segments 0..C-1 = the
  // critics, C = the actor, C+1 = log_alpha (BatchNorm actors).  Not owned by the handle.
  gcrl_xchg* xchg = nullptr;
  bool xchg_sep_norm = false;   // GCRL_XCHG_SEPARATE_NORM=1 (A/B and the bitwise test against the RCCL / gloo exchange): the clip norm from a sum-of-squares launch over the reduced gradients instead of the exchange kernel's partials
  float xchg_scale() const { return xchg ? 1.0f / (float)gcrl_xchg_world(xchg) : 1.0f; }
  int split_rg[4] = {1, 1, 1, 1};
  int row_rg = 1, row_ldl = 0;
  float *wt = nullptr, *rc_gC = nullptr, *rc_gA = nullptr, *ybuf = nullptr;
  long long wt_net[4] = {};   // offsets of actor | target actor | critic 0 | target critic 0 inside wt
  long long wt_cstride = 0;   // critic c / target critic c at wt_net[2|3] + c*wt_cstride
  // measurement hooks (gcrl_agent_profile_*): event pair + device clock around each row-block launch
  bool prof = false;
  static constexpr int kProfPairs = 64;
  hipEvent_t prof_a[kProfPairs] = {}, prof_b[kProfPairs] = {};
  unsigned long long* prof_clk = nullptr;
  int prof_used = 0;
  int64_t prof_launches = 0;
  double prof_ms = 0, prof_ticks = 0;
  float *parts_c = nullptr, *parts_a = nullptr;   // fused-norm partials: [C][nparts_c], [nparts_a]
  int nparts_c = 0, nparts_a = 0, part_off_bn = 0;   // part_off_bn: first BatchNorm slot of parts_a (L x ceil(H/64))
  std::vector<int> part_off_c, part_off_a;        // per-layer offsets inside a net's partials

  char* upload_dev = nullptr;
  size_t upload_bytes = 0;
  char* upload_pinned[kCtrlSlots] = {};
  hipEvent_t upload_ev[kCtrlSlots] = {};
  int next_upload = 0;
  float *metrics_host = nullptr, *metrics_dev = nullptr;
  int64_t next_ticket = 0;
  int64_t fetched_upto = -1;   // newest ticket whose record is in the host mirror
  std::vector<int> ticket_len;
  hipEvent_t call_ev[kEventRing] = {};
  int64_t call_last_ticket[kEventRing];
  int64_t calls = 0;

  int64_t t_actor = 0, t_critic = 0, t_alpha = 0;
  double lr_actor = 0, lr_critic = 0;
  uint64_t rng_ctr = 0;
  int pending_variant = 0;  // variant of the step whose phases are being issued one by one
  // update_n: batches 1..n-1 are drawn / uploaded / gathered AFTER the first step's launches have been issued (the host
  // draws ~2 us per batch from the MT stream: with all n batches up front the GPU idled ~40 us at the start of a cycle)
  struct { gcrl_her* her = nullptr; int n = 0; int slot = 0; int next = 0; } deferred;   // batches [next, n) not yet drawn
  int head_batches = 2;       // batches drawn and gathered before a call's first launch (the rest: behind that many queued steps)
  std::vector<StepPlan> dp_plans;  // steps of the data-parallel cycle begun by gcrl_agent_dp_begin
  std::vector<DpSeg> dp_segs;      // ... as segments separated by gradient exchanges
  size_t dp_pos = 0;

  std::map<int, hipGraphExec_t> graphs;
  std::map<std::string, std::pair<float*, long long>> names;

  hipStream_t pick(void* s) const {
    if (!s) return stream;
    if (s == GCRL_STREAM_LEGACY) return (hipStream_t) nullptr;
    return (hipStream_t)s;
  }
  CtrlBlock* ctrl() const { return (CtrlBlock*)upload_dev; }
  uint32_t* idx_dev() const { return (uint32_t*)(upload_dev + sizeof(UploadBlock)); }
  const StepCtrl* cur() const { return &ctrl()->cur; }
  const StepCtrl* prev() const { return &ctrl()->prev; }
  const int* slot_ptr() const { return &ctrl()->cur.batch_slot; }
  float* P_actor() const { return params + off_actor; }
  float* P_tactor() const { return params + off_tactor; }
  float* P_critic(int c) const { return params + off_critic + c * critic_stride; }
  float* P_tcritic(int c) const { return params + off_tcritic + c * critic_stride; }
  float* P_logalpha() const { return params + off_logalpha; }
  float* G_actor() const { return grads + goff_actor; }
  float* G_critic(int c) const { return grads + goff_critic + c * critic_stride; }
  float* hC_at(int c, int l) const { return hC + ((long long)c * L + l) * B * H; }
  float* hTC_at(int c, int i) const { return hTC + ((long long)c * 2 + i) * B * H; }
  float* gC_at(int c, int i) const { return gC + ((long long)c * 2 + i) * B * H; }
  // second activation / gradient set of critic 0: the actor phase of step i runs in the same
  // launches as the critic phase of step i+1 (software-pipelined DDPG)
  float* hC2_at(int l) const { return hC2 + (long long)l * B * H; }
  float* gC2_at(int i) const { return gC2 + (long long)i * B * H; }
  float* hA_at(int l) const { return hA + (long long)l * B * H; }
  int n_actor_critics() const { return (cfg.kind == GCRL_AGENT_DDPG || cfg.kind == GCRL_AGENT_TD3) ? 1 : C; }
};

namespace {

#define TRY(x) do { if (int rc__ = (x)) return rc__; } while (0)

// ---------------------------------------------------------------- small kernels of this file
__global__ void pack_batch_kernel(const float* s, int ld_s, const float* a, int ld_a, const float* r,
                                  const float* ns, int ld_ns, const float* d, int B, int S, int A, int ldx,
                                  float* sa, float* nsa, float* spa, float* rb, float* db) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int W = 2 * S + A + 2;
  if (i >= B * W) return;
  const int b = i / W, c = i - b * W;
  if (c < S) {
    const float v = s[(long long)b * ld_s + c];
    sa[(long long)b * ldx + c] = v;
    if (spa) spa[(long long)b * ldx + c] = v;
  } else if (c < S + A) sa[(long long)b * ldx + c] = a[(long long)b * ld_a + (c - S)];
  else if (c < W - 2) nsa[(long long)b * ldx + (c - S - A)] = ns[(long long)b * ld_ns + (c - S - A)];
  else if (c == W - 2) rb[b] = r[b];
  else db[b] = d[b];
}

// select_action's arithmetic after the network (src/agent.py:1345-1366, :253-270): mode 0 clip(tanh(x), -1, 1) (DDPG eval),
// 1 clip(tanh(x) + noise, -1, 1) (DDPG / TD3 exploration; float32 tanh + float64 noise like numpy), 2 x as it is
__global__ void act_post_kernel(const float* x, int ld, int n, int A, const double* noise, int mode, double* out) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * A) return;
  const int i = t / A, j = t - i * A;
  const float v = x[(long long)i * ld + j];
  double r = (double)v;
  if (mode != 2) {
    r = (double)tanhf(v);
    if (mode == 1 && noise) r += noise[t];
    r = fmin(fmax(r, -1.0), 1.0);
  }
  out[t] = r;
}

__global__ void copy_rows_kernel(const float* src, int ld_src, float* dst, int ld_dst, int rows, int cols) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * cols) return;
  const int r = i / cols, c = i - r * cols;
  dst[(long long)r * ld_dst + c] = src[(long long)r * ld_src + c];
}

// ---------------------------------------------------------------- descriptor builders
GemmDesc blank() {
  GemmDesc d;
  std::memset(&d, 0, sizeof(d));
  return d;
}
// Y[B,out] = act(X[B,in] W^T + b)
GemmDesc fwd(const float* X, long long ldx, const float* P, const Lin& ln, float* Y, long long ldy, int B, int epi) {
  GemmDesc d = blank();
  d.A = X; d.a_rs = ldx; d.a_cs = 1;
  d.B = P + ln.w; d.b_rs = 1; d.b_cs = ln.in;
  d.C = Y; d.c_rs = ldy;
  d.bias = P + ln.b;
  d.M = B; d.N = ln.out; d.K = ln.in;
  d.epi = epi;
  return d;
}
// dX[B,ncols] = (G[B,out] . W[out, col0:col0+ncols]) * act'(Hprev)
GemmDesc bwd_dx(const float* G, long long ldg, const float* P, const Lin& ln, int col0, int ncols, float* dX,
                long long lddx, int B, int mul, const float* Hprev, long long ldh) {
  GemmDesc d = blank();
  d.A = G; d.a_rs = ldg; d.a_cs = 1;
  d.B = P + ln.w + col0; d.b_rs = ln.in; d.b_cs = 1;
  d.C = dX; d.c_rs = lddx;
  d.M = B; d.N = ncols; d.K = ln.out;
  d.mul = mul; d.H = Hprev; d.h_rs = ldh;
  return d;
}
// dW[out,in] = G^T X ; db[out] = colsum(G)      (X = the layer's input [B,in])
GemmDesc bwd_dw(const float* G, long long ldg, const float* X, long long ldx, float* Gp, const Lin& ln, int B) {
  GemmDesc d = blank();
  d.A = G; d.a_rs = 1; d.a_cs = ldg;
  d.B = X; d.b_rs = ldx; d.b_cs = 1;
  d.C = Gp + ln.w; d.c_rs = ln.in;
  d.M = ln.out; d.N = ln.in + 1; d.K = B;
  d.ones_col = 1; d.col_out = Gp + ln.b;
  return d;
}

// a dW problem of critic `c` (or the actor: c < 0), layer l, onto the split LDS-tiled form when the agent's configuration
// asks for it (build(): a function of the shapes only — a problem's summation order never changes from step to step)
void dw_split_form(gcrl_agent* a, GemmDesc& dw, int c, int l) {
  const int S = c >= 0 ? a->dw_split_c : a->dw_split_a;
  if (S <= 1) return;
  dw.shape_hint = 4;
  dw.ksplit = S;
  dw.kpart = a->dw_part + (c >= 0 ? (long long)c * a->dwp_cstride + a->dwp_off_c[l] : a->dwp_actor + a->dwp_off_a[l]);
  dw.kticket = reinterpret_cast<unsigned int*>(a->dw_tick + (c >= 0 ? (long long)c * a->dwt_cstride + a->dwt_off_c[l] : a->dwt_actor + a->dwt_off_a[l]));
}

struct Launches {  // problems grouped by launch index
  std::vector<std::vector<GemmDesc>> steps;
  void add(size_t at, const GemmDesc& d) {
    if (steps.size() <= at) steps.resize(at + 1);
    steps[at].push_back(d);
  }
  int run(hipStream_t st) {
    for (auto& v : steps)
      for (size_t o = 0; o < v.size(); o += kMaxProb)
        TRY(launch_gemm_batch(st, v.data() + o, (int)std::min<size_t>(kMaxProb, v.size() - o)));
    return GCRL_OK;
  }
};

typedef float* (*HidFn)(gcrl_agent*, int, int);
float* hid_TA(gcrl_agent* a, int, int l) { return a->hTA[l & 1]; }
float* hid_A(gcrl_agent* a, int, int l) { return a->hA_at(l); }
float* hid_C(gcrl_agent* a, int c, int l) { return a->hC_at(c, l); }
float* hid_TC(gcrl_agent* a, int c, int l) { return a->hTC_at(c, l & 1); }
float* hid_ACT(gcrl_agent* a, int, int l) { return a->act_tmp[l & 1]; }
float* hid_C2(gcrl_agent* a, int, int l) { return a->hC2_at(l); }

// plain MLP chain (Actor / Critic of src/model.py): layer l goes to launch at+l.
// x_slot / out_slot: per-batch-slot strides when input / output live in the batch array.
void chain_mlp(gcrl_agent* a, Launches& ls, size_t at, const NetSpec& net, const float* P, const float* X0,
               long long ldx0, long long x_slot, HidFn hid, int key, float* out, long long ld_out,
               long long out_slot, int out_epi, int rows, const int* slot = nullptr) {
  if (!slot) slot = a->slot_ptr();
  for (int l = 0; l <= net.L; ++l) {
    const float* X = l == 0 ? X0 : hid(a, key, l - 1);
    float* Y = l < net.L ? hid(a, key, l) : out;
    GemmDesc d = fwd(X, l == 0 ? ldx0 : net.H, P, net.lin[l], Y, l < net.L ? net.H : ld_out, rows,
                     l < net.L ? EPI_LEAKY : out_epi);
    if (l == 0 && x_slot) { d.slot = slot; d.a_slot = x_slot; }
    if (l == net.L && out_slot) { d.slot = slot; d.c_slot = out_slot; }
    ls.add(at + l, d);
  }
}

// SACActorModel forward (src/model.py:118-141): [Linear -> BatchNorm1d(train) -> ReLU] x L, two heads, tanh-Gaussian
// sample — for ONE input or for TWO inputs co-scheduled layer by layer in the same launches.  The reference calls
// actor.sample(next_state) in the critic phase (src/agent.py:558, no_grad) and actor.sample(states) in the actor phase
// (:514) with the SAME actor parameters (the actor steps only afterwards), so on actor steps the two forwards are
// independent: one GEMM launch carries both layers-l problems, one BatchNorm launch pair both batches (running
// statistics: next_state's batch first, then the state's, as in the reference's call order).  11 launches instead of 22.
// (Round 3, tried: the two heads computed inside the tanh-Gaussian launch, four lanes per row — one launch less, but SAC cfg 5
// 211.0 vs 207.7 us/step and TQC cfg 4 unchanged: the sampling launch grew by what the GEMM launch had cost.  Not kept.)
struct ActorFwd {
  const float* X0; long long x_slot;   // input rows (+ batch_slot * x_slot)
  bool save;                           // keep xhat / invstd / eps / std for the backward
  float* act_dst; long long act_slot; int ld_act;
  float* logp; const float* eps; int rng_stream;
  float *z, *h, *head, *bn_part;       // z [B,H] scratch, h [L][B][H] (save) or [2][B][H] ping-pong, head [B][2*Apad]
};

bool heads_fold_on(const gcrl_agent* a);
int sac_actor_forward_multi(gcrl_agent* a, hipStream_t st, const ActorFwd* f, int nf, Launches* extra) {
  const NetSpec& net = a->actor;
  const float* P = a->P_actor();
  const int B = a->B, H = a->H;
  const long long BH = (long long)B * H;
  auto hbuf = [&](const ActorFwd& q, int l) { return q.save ? q.h + (long long)l * BH : q.h + (long long)(l & 1) * BH; };
  const bool slab = a->slab_on();
  for (int l = 0; l < net.L && slab; ++l) {   // one launch per layer: GEMM, batch statistics, normalise, ReLU (bn_slab.hip)
    BnSlabFwd sf;
    std::memset(&sf, 0, sizeof(sf));
    sf.n = nf;
    for (int i = 0; i < nf; ++i)
      sf.p[i] = BnSlabFwdProb{l == 0 ? f[i].X0 : hbuf(f[i], l - 1), l == 0 ? f[i].x_slot : 0, hbuf(f[i], l),
                              f[i].save ? a->xhatA + (long long)l * BH : nullptr, f[i].save ? a->invstdA + (long long)l * H : nullptr,
                              a->bn_bstat + ((long long)i * net.L + l) * 2 * H};
    sf.slot = a->slot_ptr();
    sf.W = P + net.lin[l].w; sf.bias = P + net.lin[l].b; sf.gamma = P + net.bn_g[l]; sf.beta = P + net.bn_b[l];
    sf.ldx = l == 0 ? a->ldx : H;
    sf.B = B; sf.H = H; sf.K = net.lin[l].in;
    sf.rsplit = (sf.K >= 128 || a->bn_split_all) ? a->bn_rsplit : 1; sf.xchg = a->bn_xchg; sf.bar = reinterpret_cast<unsigned int*>(a->bn_bar); sf.status = a->status_dev;
    TRY(launch_bn_linear_fwd_slab(st, sf));
    if (extra && (size_t)l < extra->steps.size()) {   // (co-scheduled critic chains of the launch-per-layer schedule: their own launch here)
      std::vector<GemmDesc> v = extra->steps[l];
      for (size_t o = 0; o < v.size(); o += kMaxProb) TRY(launch_gemm_batch(st, v.data() + o, (int)std::min<size_t>(kMaxProb, v.size() - o)));
    }
  }
  for (int l = 0; l < net.L && !slab; ++l) {
    std::vector<GemmDesc> v;
    bool fused_stats = false, fused_tiled = false;
    for (int i = 0; i < nf; ++i) {
      const float* X = l == 0 ? f[i].X0 : hbuf(f[i], l - 1);
      GemmDesc d = fwd(X, l == 0 ? a->ldx : H, P, net.lin[l], f[i].z, H, B, EPI_NONE);
      if (l == 0 && f[i].x_slot) { d.slot = a->slot_ptr(); d.a_slot = f[i].x_slot; }
      // BatchNorm statistics out of this GEMM's epilogue when its form allows (16-row partials, at most 32 of them)
      if (i == 0) fused_stats = a->bn_fused && a->bn_sync.world <= 1 && gemm_shape_of(d) == 1 && (B + kBnFusedRows - 1) / kBnFusedRows <= kBnFusedMaxParts;
      // ... or of the LDS-tiled form's (64-row partials: what bn_stats_kernel would compute from z; TQC's B = 2048, H = 512)
      if (i == 0) fused_tiled = !fused_stats && a->bn_fused_tiled && a->bn_sync.world <= 1 && gemm_shape_of(d) == 4 && H % 4 == 0;
      if (fused_stats || fused_tiled) d.bn_part = f[i].bn_part;
      v.push_back(d);
    }
    if (extra && (size_t)l < extra->steps.size()) v.insert(v.end(), extra->steps[l].begin(), extra->steps[l].end());
    for (size_t o = 0; o < v.size(); o += kMaxProb) TRY(launch_gemm_batch(st, v.data() + o, (int)std::min<size_t>(kMaxProb, v.size() - o)));
    BnFwdProb pb[2];
    for (int i = 0; i < nf; ++i)
      pb[i] = BnFwdProb{f[i].z, hbuf(f[i], l), f[i].save ? a->xhatA + (long long)l * BH : nullptr,
                        f[i].save ? a->invstdA + (long long)l * H : nullptr, f[i].bn_part};
    TRY(launch_bn_relu_fwd_multi(st, pb, nf, B, H, P + net.bn_g[l], P + net.bn_b[l], a->bn_rmean + (long long)l * H,
                                 a->bn_rvar + (long long)l * H, fused_stats ? kBnFusedRows : 64, a->bn_sync.world > 1 ? &a->bn_sync : nullptr, fused_tiled));
  }
  const int ldh = 2 * a->Apad;
  std::vector<GemmDesc> v;
  for (int i = 0; i < nf; ++i) {
    v.push_back(fwd(hbuf(f[i], net.L - 1), H, P, net.lin[net.L], f[i].head, ldh, B, EPI_NONE));                // mean
    v.push_back(fwd(hbuf(f[i], net.L - 1), H, P, net.lin[net.L + 1], f[i].head + a->Apad, ldh, B, EPI_NONE));  // log_std
  }
  const bool co_scheduled = extra && (size_t)net.L < extra->steps.size() && !extra->steps[net.L].empty();
  // round 5: the heads and the sampling as ONE launch (sac_heads.h) unless other problems ride in the heads' launch (the layer-per-
  // launch path's co-scheduled critic chains) or the head problems are not on the k-split 16x16 form (GCRL_NO_HEADS_FUSED=1: A/B knob)
  const bool heads_fused = !a->heads_fused_off && !co_scheduled && gemm_shape_of(v[0]) == 1;
  if (heads_fold_on(a) && slab && !co_scheduled) {   // the chain launches form the heads' outputs and sample themselves (rowchain.h HeadsFold)
    a->hf_eps_next = f[0].eps; a->hf_eps_cur = nf == 2 ? f[1].eps : nullptr; a->hf_nf = nf;
    return GCRL_OK;
  }
  if (!heads_fused) {
    if (co_scheduled) v.insert(v.end(), extra->steps[net.L].begin(), extra->steps[net.L].end());
    for (size_t o = 0; o < v.size(); o += kMaxProb) TRY(launch_gemm_batch(st, v.data() + o, (int)std::min<size_t>(kMaxProb, v.size() - o)));
  }
  TanhGaussArgs tg[2];
  for (int i = 0; i < nf; ++i) {
    std::memset(&tg[i], 0, sizeof(tg[i]));
    tg[i].cur = a->cur();
    tg[i].mu = f[i].head; tg[i].ls_raw = f[i].head + a->Apad; tg[i].ld_head = 2 * a->Apad;
    tg[i].eps = f[i].eps;
    tg[i].act = f[i].act_dst; tg[i].act_slot_stride = f[i].act_slot; tg[i].ld_act = f[i].ld_act ? f[i].ld_act : a->ldx;
    tg[i].logp = f[i].logp;
    tg[i].save_eps = f[i].save ? a->epsbuf : nullptr;
    tg[i].save_std = f[i].save ? a->stdbuf : nullptr;
    tg[i].B = B; tg[i].A = a->A;
    tg[i].seed = a->cfg.seed; tg[i].rng_stream = f[i].rng_stream;
  }
  if (slab)   // the running statistics of every layer, from the batch statistics the slab launches left: input 0's, then input 1's
    tg[0].run = BnRunning{{a->bn_bstat, nf == 2 ? a->bn_bstat + (long long)net.L * 2 * H : nullptr}, nf, a->bn_rmean, a->bn_rvar, net.L, H, B};
  if (heads_fused) {
    HeadsSampleArgs hs;
    std::memset(&hs, 0, sizeof(hs));
    hs.n = nf;
    for (int i = 0; i < nf; ++i) { hs.mean[i] = v[2 * i]; hs.lstd[i] = v[2 * i + 1]; hs.tg[i] = tg[i]; }
    return launch_heads_sample(st, hs);
  }
  if (nf == 2) TRY(launch_tanh_gauss_fwd2(st, tg[0], tg[1]));
  else TRY(launch_tanh_gauss_fwd(st, tg[0]));
  return GCRL_OK;
}

// the two forwards of a step: actor.sample(next_state) -> action columns of nsa + logp_next (scratch buffers), and —
// with_cur — actor.sample(states) -> pi(s) + logp, activations saved for the actor's backward
int sac_actor_forwards(gcrl_agent* a, hipStream_t st, int variant, bool with_cur, Launches* extra);

// V_FUSED_NORM: gradient sum-of-squares partials come out of the dW GEMM epilogues (whole step in
// one graph); off in data-parallel runs, where the norm is of the all-reduced gradients
// V_PRE / V_ADV (SAC on the row-block path, update_n): the step's LAST optimiser launch also advances the control block
// for the next step (V_ADV; it reads the cur_b copy the first row-block launch refreshed), so that step starts without
// a begin_step launch (V_PRE)
// V_XCHG (round 4): the gradients are all-reduced over the ranks INSIDE the launch sequence (xchg_ipc.hip) — after the critic
// backward and after the actor backward — and the exchange kernel leaves the sum-of-squares partials of the REDUCED gradients:
// no dW-epilogue partials (they would be the local gradients'), no sumsq launch
enum { V_ACTOR = 1, V_POLYAK_C = 2, V_POLYAK_A = 4, V_NOISE = 8, V_EPSN = 16, V_EPSC = 32, V_FUSED_NORM = 64, V_WEIGHTS = 128,
       V_PRE = 256, V_ADV = 512, V_XCHG = 1024 };
// how the clip norm reaches the optimiser launches of a step issued by the update entry points
int norm_bits(const gcrl_agent* a) { return a->xchg ? V_XCHG : V_FUSED_NORM; }
// exchange of the critic gradients (which = 1), the actor's (+ log_alpha) (2), or both blocks at once (3: overlapped DDPG step)
int xchg_launch(gcrl_agent* a, hipStream_t st, int which) {
  const int C = a->C, na = a->sac ? 2 : 1;
  const int seg0 = (which & 1) ? 0 : C, nseg = ((which & 1) ? C : 0) + ((which & 2) ? na : 0);
  return gcrl_xchg_allreduce(a->xchg, seg0, nseg, (void*)st);
}
// where the optimiser reads a gradient vector: the arena, or — once an exchange over more than one rank has run — the same
// offset of the exchange's fine-grained receive buffer (xchg_ipc.hip: peers never write into the arena)
const float* xg(const gcrl_agent* a, const float* p) { return a->xchg ? gcrl_xchg_result(a->xchg) + (p - a->grads) : p; }
int xchg_parts(gcrl_agent* a, bool critic, const float** p, int* n) { return gcrl_xchg_seg_parts(a->xchg, critic ? 0 : a->C, p, n); }

int adam_common(gcrl_agent* a, AdamArgs& ad);
int finish_deferred_draw(gcrl_agent* a, hipStream_t st);

int sac_actor_forwards(gcrl_agent* a, hipStream_t st, int variant, bool with_cur, Launches* extra) {
  const int S = a->S;
  ActorFwd f[2];
  f[0] = ActorFwd{a->nsa, a->slot_x, false, a->nsa + S, a->slot_x, 0, a->logp_next, (variant & V_EPSN) ? a->eps_next_in : nullptr, 1,
                  a->zN, a->hN, a->headN, a->bn_partN};
  if (a->rowchain)   // s is read from sa's rows, pi(s) goes to its own [B][Apad] matrix (no spa on this path)
    f[1] = ActorFwd{a->sa, a->slot_x, true, a->pi_buf, 0, a->Apad, a->logp, (variant & V_EPSC) ? a->eps_cur_in : nullptr, 2,
                    a->zA, a->hA, a->headA, a->bn_part};
  else
    f[1] = ActorFwd{a->spa, a->slot_x, true, a->spa + S, a->slot_x, 0, a->logp, (variant & V_EPSC) ? a->eps_cur_in : nullptr, 2,
                    a->zA, a->hA, a->headA, a->bn_part};
  return sac_actor_forward_multi(a, st, f, with_cur ? 2 : 1, extra);
}

// hipGraph replay pays when a step is many launches (31 per DDPG step launch-per-layer: 1.8 us per
// dependent kernel in a graph vs ~5 us issued one by one); the row-block step is 3-7 launches issued
// from a native loop, where plain launches measured 1.5-4 % faster than replaying graphs (a graph
// launch itself idles the GPU ~8 us).  use_graph = 2 forces graphs everywhere.
// (SAC keeps ~35 BatchNorm / head launches per step around its two row-block launches: graphs stay on)
// (SyncBN: the statistics exchanges sit between a step's launches — plain launches only)
bool graph_on(const gcrl_agent* a) {
  if (a->bn_sync.world > 1 && !a->bn_xchg_h) return false;   // (an exchange that is a kernel of the sequence replays like any other)
  return a->cfg.use_graph >= 2 || (a->cfg.use_graph == 1 && (!a->rowchain || a->sac));
}

#include "agent_rowchain.inc"

// ---------------------------------------------------------------- phase 0
int enqueue_phase0(gcrl_agent* a, hipStream_t st, int variant) {
  const int kind = a->cfg.kind, B = a->B, C = a->C, S = a->S, L = a->L, H = a->H;
  if (!(variant & V_PRE)) TRY(launch_begin_step(st, a->ctrl()));
  if (a->rowchain) {
    // row-block form: the whole critic phase up to the input gradients in one launch, then every dW|db
    // (SAC: the BatchNorm actor samples the next action first, into the action columns of nsa)
    if (a->sac) TRY(sac_actor_forwards(a, st, variant, (variant & V_ACTOR) != 0, nullptr));
    const PipeCtx kc{a->cur(), a->slot_ptr()};
    TRY(rc_launch_chain(a, st, kc, kc, 1, (variant & V_NOISE) ? a->noise_in : nullptr));
    if (of_phase_critics(a, variant)) return of_launch_critics(a, st, variant);   // dW | db + clip + optimiser step of every critic: one launch (dw_adam.hip)
    Launches dw;
    rc_add_dw(a, dw, kc, true, (variant & V_FUSED_NORM) != 0);
    TRY(dw.run(st));
    if (variant & V_XCHG) TRY(xchg_launch(a, st, 1));
    return GCRL_OK;
  }
  Launches crit;  // online critics on [s|a], activations kept for the backward
  for (int c = 0; c < C; ++c)
    chain_mlp(a, crit, 0, a->critic, a->P_critic(c), a->sa, a->ldx, a->slot_x, hid_C, c, a->q + (long long)c * B * a->Q, a->Q, 0, EPI_NONE, B);
  if (!a->sac) {
    // target actor on ns, co-scheduled with the online critics; its tanh output lands in the
    // action columns of the target critics' input rows
    Launches ls = crit;
    chain_mlp(a, ls, 0, a->actor, a->P_tactor(), a->nsa, a->ldx, a->slot_x, hid_TA, 0, a->nsa + S, a->ldx, a->slot_x, EPI_TANH, B);
    TRY(ls.run(st));
    if (kind == GCRL_AGENT_TD3)
      TRY(launch_td3_smooth(st, a->cur(), a->nsa + S, a->slot_x, a->ldx, B, a->A,
                            (variant & V_NOISE) ? a->noise_in : nullptr, (float)a->cfg.policy_noise,
                            (float)a->cfg.noise_clamp, a->cfg.seed));
  } else {
    // actor.sample(next_state) under no_grad, BatchNorm in training mode (src/agent.py:558, :961)
    TRY(sac_actor_forwards(a, st, variant, (variant & V_ACTOR) != 0, &crit));
  }
  Launches tc;
  for (int c = 0; c < C; ++c)
    chain_mlp(a, tc, 0, a->critic, a->P_tcritic(c), a->nsa, a->ldx, a->slot_x, hid_TC, c, a->qt + (long long)c * B * a->Q, a->Q, 0, EPI_NONE, B);
  TRY(tc.run(st));

  TdLossArgs td;
  std::memset(&td, 0, sizeof(td));
  td.cur = a->cur();
  td.r = a->rbuf; td.d = a->dbuf; td.slot_stride = a->slot_rd;
  td.qt = a->qt; td.q = a->q; td.dq = a->dq;
  td.metrics = a->metrics_dev;
  td.td_abs = a->td_abs;
  td.w = (variant & V_WEIGHTS) ? a->w_in : nullptr;
  td.B = B; td.C = C; td.drop = 0;
  td.gamma = (float)a->cfg.gamma;
  td.loss_kind = LOSS_MSE;
  if (!a->red_off) { td.part = a->red_td(); td.ticket = a->red_ticket(0); }
  td.refresh = (variant & V_ADV) ? a->ctrl() : nullptr;   // (layer_adv: the step's last optimiser launch reads the copies and advances)
  switch (kind) {
    case GCRL_AGENT_DDPG: td.target_kind = TGT_DDPG; td.clamp_lo = (float)(-1.0 / (1.0 - a->cfg.gamma)); break;
    case GCRL_AGENT_TD3: td.target_kind = TGT_MIN; td.loss_kind = LOSS_SMOOTH_L1; break;
    case GCRL_AGENT_SAC: td.target_kind = TGT_MIN_ENT; td.logp_next = a->logp_next; td.alpha_const = 0.2f; break;
    default: td.target_kind = TGT_TRUNC_ENT; td.logp_next = a->logp_next; td.alpha_dev = a->alpha_dev; td.drop = a->cfg.top_drop; break;
  }
  if (a->Q > 1) {   // distributional variant: pooled-atom sort + truncation, quantile-Huber loss (ops.h QuantileArgs)
    QuantileArgs qa;
    std::memset(&qa, 0, sizeof(qa));
    qa.cur = a->cur(); qa.r = a->rbuf; qa.d = a->dbuf; qa.slot_stride = a->slot_rd;
    qa.zt = a->qt; qa.z = a->q; qa.logp_next = a->logp_next; qa.alpha_dev = a->alpha_dev;
    qa.y = a->qy; qa.dz = a->dq; qa.row_loss = a->q_row_loss; qa.row_td = a->q_row_td; qa.metrics = a->metrics_dev;
    qa.B = B; qa.C = C; qa.Q = a->Q; qa.drop = a->cfg.top_drop; qa.gamma = (float)a->cfg.gamma;
    TRY(launch_quantile_td(st, qa));
  } else {
    TRY(launch_td_loss(st, td));
  }

  // critic backward, layer L..0: dW|db and dX of a layer share a launch — unless the ensemble is large enough for its
  // dW problems to fill the chip on their own (TQC at B = 2048, H = 512: 5 critics x 2 hidden layers x 72 tiles of 64x64):
  // then the dX chain runs first, every layer's gradient kept (rc_gC: one buffer per layer instead of the two ping-pong
  // ones), and ALL dW|db problems follow in one batch on the LDS-tiled form.  A dW problem alone is a long reduction
  // (K = batch) into a mid-sized output: it wants the k-split 16x16 form, which re-reads its operands from L2 for every
  // 16x16 tile (4 flop per byte) and runs the 16x16x4 instruction — 13 such launches were 30 % of TQC's step.
  const long long dw_tiles = (long long)((H + 63) / 64) * ((H + 1 + 63) / 64);
  const bool dw_batch = !a->dw_batch_off && L >= 2 && B >= 1024 && B % 16 == 0 && (long long)C * (L - 1) * dw_tiles >= 512;
  auto Gbuf = [&](int c, int l) -> float* {            // gradient w.r.t. the output of hidden layer l (l < L)
    return dw_batch ? a->rc_gC + ((long long)c * L + l) * B * H : a->gC_at(c, (l + 1) & 1);
  };
  Launches bw;
  for (int c = 0; c < C; ++c) {
    float* Gp = a->G_critic(c);
    const float* P = a->P_critic(c);
    for (int l = L; l >= 0; --l) {
      const size_t at = (size_t)(L - l);
      const float* G = l == L ? a->dq + (long long)c * B * a->Q : Gbuf(c, l);
      const long long ldg = l == L ? a->Q : H;
      GemmDesc dw = bwd_dw(G, ldg, l == 0 ? a->sa : a->hC_at(c, l - 1), l == 0 ? a->ldx : H, Gp, a->critic.lin[l], B);
      if (l == 0) { dw.slot = a->slot_ptr(); dw.b_slot = a->slot_x; }
      if (variant & V_FUSED_NORM) dw.sumsq_out = a->parts_c + (long long)c * a->nparts_c + a->part_off_c[l];
      // (tried: each hidden layer's dW problems, split four ways, inside that layer's dX launch — two launches of 2 560 work
      // items instead of dX, dX, dW batch: 1 226.5 vs 1 225.8 us/step, no difference; the batch stays)
      if (dw_batch && dw.M >= 64 && dw.N >= 64) { dw.shape_hint = 4; dw_split_form(a, dw, c, l); }
      bw.add(!dw_batch ? at : (dw.shape_hint ? (size_t)L + 1 : (size_t)L + 2), dw);   // the big ones together, in ONE launch
      if (l > 0)
        bw.add(at, bwd_dx(G, ldg, P, a->critic.lin[l], 0, H, Gbuf(c, l - 1), H, B, MUL_DLEAKY, a->hC_at(c, l - 1), H));
    }
  }
  TRY(bw.run(st));
  if (variant & V_XCHG) TRY(xchg_launch(a, st, 1));
  return GCRL_OK;
}

int adam_common(gcrl_agent* a, AdamArgs& ad) {
  ad.cur = a->cur();
  ad.beta2 = (float)kBeta2;
  ad.w1 = (float)(1.0 - kBeta1);
  ad.w2 = (float)(1.0 - kBeta2);
  ad.eps = (float)kAdamEps;
  ad.tau = (float)a->cfg.tau;
  ad.one_m_tau = (float)(1.0 - a->cfg.tau);
  ad.metrics = a->metrics_dev;
  ad.partial = a->norm_partial;
  ad.nparts = kNormBlocks;
  ad.part_stride = kNormBlocks;
  return GCRL_OK;
}

// ---------------------------------------------------------------- phase 1
int enqueue_phase1_body(gcrl_agent* a, hipStream_t st, int variant);
int enqueue_phase1(gcrl_agent* a, hipStream_t st, int variant) {
  TRY(enqueue_phase1_body(a, st, variant));
  if ((variant & V_XCHG) && (variant & V_ACTOR)) TRY(xchg_launch(a, st, 2));   // the actor's (+ log_alpha) gradients are complete
  return GCRL_OK;
}
int enqueue_phase1_body(gcrl_agent* a, hipStream_t st, int variant) {
  const int kind = a->cfg.kind, B = a->B, C = a->C, S = a->S, A = a->A, L = a->L, H = a->H;
  // critic optimiser: global-norm clip + Adam(W) (+ Polyak into the target critics)
  const bool fused = (variant & V_FUSED_NORM) != 0, xc = (variant & V_XCHG) != 0 && !a->xchg_sep_norm;
  const bool xv = (variant & V_XCHG) != 0;   // the gradients the optimiser consumes are the exchanged ones
  if (!fused && !xc) TRY(launch_sumsq(st, xv ? xg(a, a->G_critic(0)) : a->G_critic(0), a->critic.numel, a->critic_stride, C, a->norm_partial));
  if (!(a->rowchain && of_phase_critics(a, variant))) {   // (else: the critics were stepped by phase 0's last launch)
    AdamArgs ad;
    std::memset(&ad, 0, sizeof(ad));
    adam_common(a, ad);
    if (fused) { ad.partial = a->parts_c; ad.nparts = a->nparts_c; ad.part_stride = a->nparts_c; }
    if (xc) { TRY(xchg_parts(a, true, &ad.partial, &ad.nparts)); ad.part_stride = ad.nparts; }
    ad.which = 1;
    ad.p = a->P_critic(0); ad.g = xv ? xg(a, a->G_critic(0)) : a->G_critic(0);
    ad.m = a->adam_m + a->goff_critic; ad.v = a->adam_v + a->goff_critic;
    ad.target = a->P_tcritic(0);
    ad.n = a->critic.numel; ad.net_stride = a->critic_stride; ad.nets = C;
    for (int c = 0; c < kMaxCritics; ++c) ad.clip[c] = (float)a->cfg.grad_clip;
    if (kind == GCRL_AGENT_TD3) ad.clip[0] = -1.f;  // critic_1 is not clipped (src/agent.py:201)
    ad.polyak = (variant & V_POLYAK_C) ? 1 : 0;
    ad.metric_index = MET_CRITIC_GRAD;
    if (a->rowchain) rc_adam_extras(a, ad, true);
    if ((variant & V_ADV) && !(variant & V_ACTOR)) { ad.cur = &a->ctrl()->cur_b; ad.advance = a->ctrl(); }   // last launch of this step
    TRY(launch_adam(st, ad));
  }
  // q_value metric from the UPDATED critics (src/agent.py:1016-1019): a full forward of the ensemble on [s | a] that feeds
  // nothing but a logged number.  On actor steps of the layer-per-launch schedule it rides in the launches of the stepped
  // critics' forward on [s | pi(s)] below (same parameters, layer by layer: ten problems per launch instead of two chains
  // of five — four launches and their ramps less per step)
  Launches re;
  const bool re_merged = kind == GCRL_AGENT_TQC && (variant & V_ACTOR) && !a->rowchain && 2 * C <= kMaxProb;
  if (kind == GCRL_AGENT_TQC) {
    for (int c = 0; c < C; ++c)
      chain_mlp(a, re, 0, a->critic, a->P_critic(c), a->sa, a->ldx, a->slot_x, hid_TC, c, a->qt + (long long)c * B * a->Q, a->Q, 0, EPI_NONE, B);
    if (!re_merged) {
      TRY(re.run(st));
      TRY(launch_mean_metric(st, a->cur(), a->qt, C * B * a->Q, 1.0f, a->metrics_dev, MET_Q));
    }
  }
  if (kind == GCRL_AGENT_DDPG && (variant & V_POLYAK_A)) {  // before the actor step (src/agent.py:1397-1401)
    TRY(launch_polyak(st, a->P_actor(), a->P_tactor(), a->actor.numel, a->cfg.tau));
    if (a->rowchain) {
      const RowNet r = make_rownet(a, a->actor, a->P_tactor(), 1);
      TRY(launch_wt_rebuild(st, r, const_cast<float*>(r.Wt)));
    }
  }
  if (!(variant & V_ACTOR)) return GCRL_OK;
  if (a->rowchain && !a->sac) {
    // row-block form of the actor phase: actor, stepped critic 0, both input-gradient chains; then dW|db
    const PipeCtx pc{a->cur(), a->slot_ptr()};
    TRY(rc_launch_chain(a, st, pc, pc, 2));
    if (of_phase_actor(a, variant)) return of_launch_actor(a, st, variant);   // ... and the actor's (phase 2 then has nothing to do)
    Launches dw;
    rc_add_dw(a, dw, pc, false, (variant & V_FUSED_NORM) != 0);
    return dw.run(st);
  }

  const int nac = a->n_actor_critics();
  // actor forward on s; its action lands in the action columns of spa
  if (!a->sac) {
    Launches af;
    chain_mlp(a, af, 0, a->actor, a->P_actor(), a->spa, a->ldx, a->slot_x, hid_A, 0, a->spa + S, a->ldx, a->slot_x, EPI_TANH, B);
    TRY(af.run(st));
  } else {
    // (actor.sample(states) already ran, co-scheduled with actor.sample(next_state), in phase 0: sac_actor_forwards)
  }
  if (a->rowchain) {   // SAC: both stepped critics, the min-selection gradient and their action gradients, one launch
    const PipeCtx pc{a->cur(), a->slot_ptr()};
    TRY(rc_launch_chain(a, st, pc, pc, 2));
  }
  // stepped critic(s) on [s | pi(s)]
  Launches c2;
  for (int c = 0; c < (a->rowchain ? 0 : nac); ++c)
    chain_mlp(a, c2, 0, a->critic, a->P_critic(c), a->spa, a->ldx, a->slot_x, hid_C, c, a->q2 + (long long)c * B * a->Q, a->Q, 0, EPI_NONE, B);
  if (re_merged)
    for (size_t i = 0; i < re.steps.size(); ++i)
      for (const GemmDesc& d : re.steps[i]) c2.add(i, d);
  TRY(c2.run(st));
  const bool met_rider = re_merged && a->sac && a->Q == 1 && !a->rowchain;   // then the mean rides on the selection launch below
  if (re_merged && !met_rider) TRY(launch_mean_metric(st, a->cur(), a->qt, C * B * a->Q, 1.0f, a->metrics_dev, MET_Q));
  bool sel_deferred = false;
  ActorSelArgs as_d;
  AlphaArgs al_d;
  if (a->sac) {
    ActorSelArgs as;
    std::memset(&as, 0, sizeof(as));
    as.cur = a->cur(); as.q = a->q2; as.logp = a->logp; as.dq = a->dq2; as.metrics = a->metrics_dev;
    as.B = B; as.C = nac;
    if (kind == GCRL_AGENT_SAC) { as.alpha_const = 0.2f; as.drop = 0; }
    else { as.alpha_dev = a->alpha_dev; as.drop = a->cfg.top_drop; }
    // ... and, in the same launch, the log-alpha gradient + loss metric (the optimiser step itself is in phase 2)
    AlphaArgs al;
    std::memset(&al, 0, sizeof(al));
    al.cur = a->cur(); al.logp = a->logp; al.B = B;
    al.target_entropy = kind == GCRL_AGENT_SAC ? -0.5f * (float)A : -(float)A;  // src/agent.py:424, :820
    al.log_alpha = a->P_logalpha(); al.m = a->adam_m + a->goff_alpha; al.v = a->adam_v + a->goff_alpha;
    al.alpha = a->alpha_dev; al.grad_out = a->grads + a->goff_alpha;
    al.metrics = a->metrics_dev; al.phase = 0;
    if (a->Q > 1) {   // distributional variant: the actor maximises the mean of ALL atoms (no truncation on this side)
      QuantileActorArgs qa{a->cur(), a->q2, a->logp, a->alpha_dev, a->dq2, a->metrics_dev, B, nac, a->Q};
      TRY(launch_quantile_actor(st, qa));
      TRY(launch_alpha_update(st, al));
    } else if (a->rowchain) {
      sel_deferred = true;   // metrics + log-alpha gradient only on this path: rides on the tanh-Gaussian backward launch below
      as_d = as; al_d = al;
    } else {
      if (met_rider) { as.mean_x = a->qt; as.mean_n = C * B * a->Q; as.mean_index = MET_Q; }
      if (!a->red_off) { as.part = a->red_sel(); as.ticket = a->red_ticket(1); }
      TRY(launch_actor_select_alpha(st, as, al));
    }
  }
  // input gradient of the critic(s) down to the action columns
  Launches cb;
  for (int c = 0; c < (a->rowchain ? 0 : nac); ++c) {
    const float* P = a->P_critic(c);
    for (int l = L; l >= 1; --l) {
      const float* G = l == L ? a->dq2 + (long long)c * B * a->Q : a->gC_at(c, l & 1);
      cb.add((size_t)(L - l), bwd_dx(G, l == L ? a->Q : H, P, a->critic.lin[l], 0, H, a->gC_at(c, (l - 1) & 1), H, B,
                                     MUL_DLEAKY, a->hC_at(c, l - 1), H));
    }
    GemmDesc d0 = bwd_dx(a->gC_at(c, 0), H, P, a->critic.lin[0], S, A, a->dact + (long long)c * B * a->Apad, a->Apad, B,
                         a->sac ? MUL_NONE : MUL_DTANH, a->sac ? nullptr : a->spa + S, a->ldx);
    if (!a->sac) { d0.slot = a->slot_ptr(); d0.h_slot = a->slot_x; }
    cb.add((size_t)L, d0);
  }
  TRY(cb.run(st));

  float* Ga = a->G_actor();
  const float* Pa = a->P_actor();
  if (!a->sac) {
    Launches ab;
    for (int l = L; l >= 0; --l) {
      const size_t at = (size_t)(L - l);
      const float* G = l == L ? a->dact : a->gA[l & 1];
      const long long ldg = l == L ? a->Apad : H;
      GemmDesc dw = bwd_dw(G, ldg, l == 0 ? a->spa : a->hA_at(l - 1), l == 0 ? a->ldx : H, Ga, a->actor.lin[l], B);
      if (l == 0) { dw.slot = a->slot_ptr(); dw.b_slot = a->slot_x; }
      if (variant & V_FUSED_NORM) dw.sumsq_out = a->parts_a + a->part_off_a[l];
      ab.add(at, dw);
      if (l > 0) ab.add(at, bwd_dx(G, ldg, Pa, a->actor.lin[l], 0, H, a->gA[(l - 1) & 1], H, B, MUL_DLEAKY, a->hA_at(l - 1), H));
    }
    TRY(ab.run(st));
  } else {
    TanhGaussBwdArgs tb;
    std::memset(&tb, 0, sizeof(tb));
    tb.dact = a->dact; tb.C = nac; tb.ld_dact = a->Apad; tb.dact_stride = (long long)B * a->Apad;
    tb.act = a->spa + S; tb.act_slot_stride = a->slot_x; tb.ld_act = a->ldx; tb.cur = a->cur();
    if (a->rowchain) { tb.act = a->pi_buf; tb.act_slot_stride = 0; tb.ld_act = a->Apad; }
    tb.eps = a->epsbuf; tb.std = a->stdbuf; tb.ls_raw = a->headA + a->Apad; tb.ld_head = 2 * a->Apad;
    if (kind == GCRL_AGENT_SAC) tb.alpha_const = 0.2f; else tb.alpha_dev = a->alpha_dev;
    tb.gmu = a->ghead; tb.gls = a->ghead + a->Apad; tb.ld_g = 2 * a->Apad;
    tb.B = B; tb.A = A;
    // round 5: the top layer's backward slab forms the heads' gradients itself and carries the selection block (bn_slab.hip FOLD) — one launch less
    const bool tg_fold = a->slab_on() && a->bn_rsplit > 1 && a->bn_split_all && !a->tg_fold_off && bn_slab_bwd_can_fold(B, H, A);
    if (tg_fold) { /* inside the slab launch below */ }
    else if (sel_deferred) TRY(launch_tanh_gauss_bwd_select(st, tb, as_d, al_d));
    else TRY(launch_tanh_gauss_bwd(st, tb));
    if (a->slab_on()) {
      // one launch per BatchNorm layer: the dX GEMM(s) of the consumer(s), the ReLU mask, BatchNorm's backward; dz_l takes
      // xhat_l's place.  Then every dW | db of the actor in one launch.
      const long long BH = (long long)B * H;
      const int ldg = 2 * a->Apad;
      for (int l = L - 1; l >= 0; --l) {
        BnSlabBwd sb;
        std::memset(&sb, 0, sizeof(sb));
        if (l == L - 1) {
          sb.nup = 2;
          sb.G[0] = a->ghead; sb.G[1] = a->ghead + a->Apad; sb.ldg[0] = sb.ldg[1] = ldg; sb.K[0] = sb.K[1] = A;
          sb.W[0] = Pa + a->actor.lin[L].w; sb.W[1] = Pa + a->actor.lin[L + 1].w; sb.ldw[0] = sb.ldw[1] = H;
        } else {
          sb.nup = 1;
          sb.G[0] = a->xhatA + (long long)(l + 1) * BH; sb.ldg[0] = H; sb.K[0] = H;
          sb.W[0] = Pa + a->actor.lin[l + 1].w; sb.ldw[0] = H;
        }
        sb.xhat_dz = a->xhatA + (long long)l * BH; sb.invstd = a->invstdA + (long long)l * H;
        sb.gamma = Pa + a->actor.bn_g[l]; sb.beta = Pa + a->actor.bn_b[l];
        sb.dgamma = Ga + a->actor.bn_g[l]; sb.dbeta = Ga + a->actor.bn_b[l];
        sb.sumsq_out = (variant & V_FUSED_NORM) ? a->parts_a + a->part_off_bn + l * a->bn_slots : nullptr;
        sb.B = B; sb.H = H;
        if (l == L - 1 && tg_fold) { sb.fold_tg = &tb; if (sel_deferred) { sb.fold_sel = &as_d; sb.fold_al = &al_d; } }
        sb.rsplit = (l == L - 1 && !a->bn_split_all) ? 1 : a->bn_rsplit; sb.xchg = a->bn_xchg; sb.bar = reinterpret_cast<unsigned int*>(a->bn_bar); sb.status = a->status_dev;
        TRY(launch_bn_linear_bwd_slab(st, sb));
      }
      std::vector<GemmDesc> v;
      v.push_back(bwd_dw(a->ghead, ldg, a->hA_at(L - 1), H, Ga, a->actor.lin[L], B));
      v.push_back(bwd_dw(a->ghead + a->Apad, ldg, a->hA_at(L - 1), H, Ga, a->actor.lin[L + 1], B));
      if (variant & V_FUSED_NORM) { v[0].sumsq_out = a->parts_a + a->part_off_a[L]; v[1].sumsq_out = a->parts_a + a->part_off_a[L + 1]; }
      for (int l = L - 1; l >= 0; --l) {
        GemmDesc dw = bwd_dw(a->xhatA + (long long)l * BH, H, l == 0 ? (a->rowchain ? a->sa : a->spa) : a->hA_at(l - 1), l == 0 ? a->ldx : H, Ga,
                             a->actor.lin[l], B);
        if (l == 0) { dw.slot = a->slot_ptr(); dw.b_slot = a->slot_x; }
        if (variant & V_FUSED_NORM) dw.sumsq_out = a->parts_a + a->part_off_a[l];
        v.push_back(dw);
      }
      for (size_t o = 0; o < v.size(); o += kMaxProb) TRY(launch_gemm_batch(st, v.data() + o, (int)std::min<size_t>(kMaxProb, v.size() - o)));
      return GCRL_OK;
    }
    {
      const int ldg = 2 * a->Apad;
      std::vector<GemmDesc> v;
      v.push_back(bwd_dw(a->ghead, ldg, a->hA_at(L - 1), H, Ga, a->actor.lin[L], B));
      v.push_back(bwd_dw(a->ghead + a->Apad, ldg, a->hA_at(L - 1), H, Ga, a->actor.lin[L + 1], B));
      if (variant & V_FUSED_NORM) { v[0].sumsq_out = a->parts_a + a->part_off_a[L]; v[1].sumsq_out = a->parts_a + a->part_off_a[L + 1]; }
      v.push_back(bwd_dx(a->ghead, ldg, Pa, a->actor.lin[L], 0, H, a->gA[0], H, B, MUL_NONE, nullptr, 0));
      v.push_back(bwd_dx(a->ghead + a->Apad, ldg, Pa, a->actor.lin[L + 1], 0, H, a->dh2, H, B, MUL_NONE, nullptr, 0));
      TRY(launch_gemm_batch(st, v.data(), (int)v.size()));
    }
    for (int l = L - 1; l >= 0; --l) {
      const float* dh = l == L - 1 ? a->gA[0] : a->gA[(l + 1) & 1];
      // (dh for layer l<L-1 was written by the dX of layer l+1 into gA[(l+1)&1])
      TRY(launch_bn_relu_bwd(st, dh, l == L - 1 ? a->dh2 : nullptr, a->xhatA + (long long)l * B * H,
                             a->invstdA + (long long)l * H, Pa + a->actor.bn_g[l], Pa + a->actor.bn_b[l], B, H, a->zA,
                             Ga + a->actor.bn_g[l], Ga + a->actor.bn_b[l], a->bn_part,
                             (variant & V_FUSED_NORM) ? a->parts_a + a->part_off_bn + l * a->bn_slots : nullptr,
                             a->bn_sync.world > 1 ? &a->bn_sync : nullptr));
      std::vector<GemmDesc> v;
      GemmDesc dw = bwd_dw(a->zA, H, l == 0 ? (a->rowchain ? a->sa : a->spa) : a->hA_at(l - 1), l == 0 ? a->ldx : H, Ga, a->actor.lin[l], B);
      if (l == 0) { dw.slot = a->slot_ptr(); dw.b_slot = a->slot_x; }
      if (variant & V_FUSED_NORM) dw.sumsq_out = a->parts_a + a->part_off_a[l];
      if (dw.M >= 64 && dw.N > 64) dw_split_form(a, dw, -1, l);   // (then the layer's dW and dX share ONE LDS-tiled launch)
      v.push_back(dw);
      if (l > 0) v.push_back(bwd_dx(a->zA, H, Pa, a->actor.lin[l], 0, H, a->gA[l & 1], H, B, MUL_NONE, nullptr, 0));
      TRY(launch_gemm_batch(st, v.data(), (int)v.size()));
    }
  }
  return GCRL_OK;
}

// ---------------------------------------------------------------- phase 2
int enqueue_phase2(gcrl_agent* a, hipStream_t st, int variant) {
  if (!(variant & V_ACTOR)) return GCRL_OK;
  if (a->rowchain && of_phase_actor(a, variant)) return GCRL_OK;   // stepped by phase 1's last launch
  const int kind = a->cfg.kind;
  const bool fused = (variant & V_FUSED_NORM) != 0, xc = (variant & V_XCHG) != 0 && !a->xchg_sep_norm;   // (BatchNorm gradients: their launches leave partials too)
  const bool xv = (variant & V_XCHG) != 0;
  const float* g_actor = xv ? xg(a, a->G_actor()) : a->G_actor();
  const float* g_alpha = xv ? xg(a, a->grads + a->goff_alpha) : a->grads + a->goff_alpha;
  if (!fused && !xc) TRY(launch_sumsq(st, g_actor, a->actor.numel, 0, 1, a->norm_partial));
  AdamArgs ad;
  std::memset(&ad, 0, sizeof(ad));
  adam_common(a, ad);
  if (fused) { ad.partial = a->parts_a; ad.nparts = a->nparts_a; ad.part_stride = a->nparts_a; }
  if (xc) { TRY(xchg_parts(a, false, &ad.partial, &ad.nparts)); ad.part_stride = ad.nparts; }
  ad.which = 0;
  ad.p = a->P_actor(); ad.g = g_actor;
  ad.m = a->adam_m + a->goff_actor; ad.v = a->adam_v + a->goff_actor;
  ad.n = a->actor.numel; ad.net_stride = 0; ad.nets = 1;
  for (int c = 0; c < kMaxCritics; ++c) ad.clip[c] = (float)a->cfg.grad_clip;
  if (kind == GCRL_AGENT_TD3 && (variant & V_POLYAK_A)) { ad.target = a->P_tactor(); ad.polyak = 1; }
  ad.metric_index = MET_ACTOR_GRAD;
  if (!a->sac) {  // actor_loss = -Q.mean() (dq2 holds the constant -1/B), folded into this launch
    ad.mean_x = a->q2; ad.mean_n = a->B; ad.mean_scale = -1.0f; ad.mean_index = MET_ACTOR_LOSS;
  }
  if (a->rowchain && !a->sac) rc_adam_extras(a, ad, false);
  const bool alpha_rider = a->sac;   // the log-alpha step rides on the actor's optimiser launch (round 4: on every path)
  if (alpha_rider) {
    ad.alpha = AlphaStep{a->P_logalpha(), a->adam_m + a->goff_alpha, a->adam_v + a->goff_alpha, a->alpha_dev, g_alpha,
                         (float)kBeta2, (float)(1.0 - kBeta1), (float)(1.0 - kBeta2), (float)kAdamEps, a->metrics_dev};
  }
  if (variant & V_ADV) { ad.cur = &a->ctrl()->cur_b; ad.advance = a->ctrl(); }   // the step's last launch (row-block paths only)
  TRY(launch_adam(st, ad));
  if (a->sac && !alpha_rider) {
    AlphaArgs al;
    std::memset(&al, 0, sizeof(al));
    al.cur = a->cur(); al.logp = a->logp; al.B = a->B;
    al.log_alpha = a->P_logalpha(); al.m = a->adam_m + a->goff_alpha; al.v = a->adam_v + a->goff_alpha;
    al.alpha = a->alpha_dev; al.grad_out = const_cast<float*>(g_alpha);   // (phase 1 of the launch only reads it)
    al.beta2 = (float)kBeta2; al.w1 = (float)(1.0 - kBeta1); al.w2 = (float)(1.0 - kBeta2); al.eps = (float)kAdamEps;
    al.metrics = a->metrics_dev; al.phase = 1;
    TRY(launch_alpha_update(st, al));
  }
  return GCRL_OK;
}

int enqueue_phases(gcrl_agent* a, hipStream_t st, int variant, int mask) {
  if (mask & 1) TRY(enqueue_phase0(a, st, variant));
  if (mask & 2) TRY(enqueue_phase1(a, st, variant));
  if (mask & 4) TRY(enqueue_phase2(a, st, variant));
  return GCRL_OK;
}

// `count` consecutive steps of the same variant as ONE graph: a graph launch leaves the GPU idle for ~8.6 us (measured
// between the steps of SAC's one-graph-per-step sequence: 4 % of its 200 us step), so a trainer cycle's run of
// identical steps is replayed with a single launch
int run_step(gcrl_agent* a, hipStream_t st, int variant, int mask, int count = 1) {
  if (a->rowchain && a->wt_dirty) TRY(rc_rebuild_wt(a, st));
  if (!graph_on(a)) {
    for (int c = 0; c < count; ++c) TRY(enqueue_phases(a, st, variant, mask));
    return GCRL_OK;
  }
  static_assert(2 * V_XCHG <= (1 << 12), "graph key: the variant flags must stay below the phase-mask bits");
  const int key = variant | (mask << 12) | (count > 1 ? (0x40000000 | (count << 16)) : 0);
  auto it = a->graphs.find(key);
  if (it == a->graphs.end()) {
    hipGraph_t g = nullptr;
    GCRL_HIP(hipStreamBeginCapture(a->cap_stream, hipStreamCaptureModeThreadLocal));
    int rc = GCRL_OK;
    for (int c = 0; c < count && !rc; ++c) rc = enqueue_phases(a, a->cap_stream, variant, mask);
    hipError_t e = hipStreamEndCapture(a->cap_stream, &g);
    if (rc) { if (g) (void)hipGraphDestroy(g); return rc; }
    GCRL_HIP(e);
    hipGraphExec_t ex = nullptr;
    GCRL_HIP(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
    (void)hipGraphDestroy(g);
    it = a->graphs.emplace(key, ex).first;
  }
  GCRL_HIP(hipGraphLaunch(it->second, st));
  return GCRL_OK;
}

#include "agent_pipeline.inc"

// ---------------------------------------------------------------- per-step host bookkeeping

StepPlan plan_step(gcrl_agent* a, int64_t step, float grad_scale, int batch_slot, int64_t ticket, StepCtrl* sc) {
  const gcrl_agent_config& c = a->cfg;
  StepPlan p{0, 0};
  const bool do_actor = (step % c.ac_update_freq) == 0;
  if (do_actor) p.variant |= V_ACTOR;
  switch (c.kind) {
    case GCRL_AGENT_DDPG:
      if (step % c.polyak_every == 0) p.variant |= V_POLYAK_C | V_POLYAK_A;
      p.tuple_len = do_actor ? 6 : 4;
      break;
    case GCRL_AGENT_TD3:
      p.variant |= V_POLYAK_C;
      if (do_actor) p.variant |= V_POLYAK_A;
      p.tuple_len = do_actor ? 8 : 6;
      break;
    case GCRL_AGENT_SAC:
      if (step % c.gradient_step == 0) p.variant |= V_POLYAK_C;
      p.tuple_len = do_actor ? 9 : 6;
      break;
    default:
      p.variant |= V_POLYAK_C;
      p.tuple_len = do_actor ? 9 : 6;
      break;
  }
  const bool adamw = c.kind != GCRL_AGENT_DDPG;  // DDPG: Adam; others AdamW, weight_decay 0.01
  std::memset(sc, 0, sizeof(*sc));
  auto fill = [&](int64_t t, double lr, float* step_size, float* bc2s, float* decay) {
    *step_size = (float)(lr / (1.0 - std::pow(kBeta1, (double)t)));
    *bc2s = (float)std::sqrt(1.0 - std::pow(kBeta2, (double)t));
    *decay = adamw ? (float)(1.0 - lr * kWeightDecay) : 1.0f;
  };
  // critics step every update; lr = value after (t-1) scheduler steps
  a->t_critic++;
  fill(a->t_critic, a->lr_critic, &sc->step_size_critic, &sc->bc2s_critic, &sc->decay_critic);
  a->lr_critic = gcrl_cosine_lr_next(a->lr_critic, c.critic_lr, c.critic_lr_min, c.cr_scheduler_steps, a->t_critic);
  sc->step_size_actor = 0.f; sc->bc2s_actor = 1.f; sc->decay_actor = 1.f;
  sc->step_size_alpha = 0.f; sc->bc2s_alpha = 1.f; sc->decay_alpha = 1.f;
  if (do_actor) {
    a->t_actor++;
    fill(a->t_actor, a->lr_actor, &sc->step_size_actor, &sc->bc2s_actor, &sc->decay_actor);
    a->lr_actor = gcrl_cosine_lr_next(a->lr_actor, c.actor_lr, c.actor_lr_min, c.ac_scheduler_steps, a->t_actor);
    if (a->sac && (double)step > c.alpha_min_steps) {
      sc->do_alpha = 1;
      a->t_alpha++;
      fill(a->t_alpha, c.alpha_lr, &sc->step_size_alpha, &sc->bc2s_alpha, &sc->decay_alpha);
    }
  }
  sc->grad_scale = grad_scale;
  sc->batch_slot = batch_slot;
  sc->metrics_slot = (int)(ticket % kMetricSlots);
  sc->rng_hi = (unsigned int)(a->rng_ctr >> 32);
  sc->rng_lo = (unsigned int)(a->rng_ctr & 0xffffffffu);
  a->rng_ctr += (uint64_t)a->B * 16;
  return p;
}

int stage_injected(gcrl_agent* a, const gcrl_update_inputs* in, hipStream_t st, int* vbits) {
  const int B = a->B, S = a->S, A = a->A;
  if (in->s_dev) {
    GCRL_CHECK_ARG(in->a_dev && in->r_dev && in->ns_dev && in->d_dev, "update: injected batch needs s, a, r, ns, d");
    GCRL_CHECK_ARG(in->ld_s >= S && in->ld_ns >= S && in->ld_a >= A, "update: injected batch row stride too small");
    const int n = B * (2 * S + A + 2);
    hipLaunchKernelGGL(pack_batch_kernel, dim3((n + 255) / 256), dim3(256), 0, st, in->s_dev, in->ld_s, in->a_dev,
                       in->ld_a, in->r_dev, in->ns_dev, in->ld_ns, in->d_dev, B, S, A, a->ldx, a->sa, a->nsa, a->rowchain ? nullptr : a->spa,
                       a->rbuf, a->dbuf);
    GCRL_HIP(hipGetLastError());
  }
  const size_t nb = (size_t)B * A * sizeof(float);
  if (in->noise_dev) { GCRL_HIP(hipMemcpyAsync(a->noise_in, in->noise_dev, nb, hipMemcpyDeviceToDevice, st)); *vbits |= V_NOISE; }
  if (in->eps_next_dev) { GCRL_HIP(hipMemcpyAsync(a->eps_next_in, in->eps_next_dev, nb, hipMemcpyDeviceToDevice, st)); *vbits |= V_EPSN; }
  if (in->eps_cur_dev) { GCRL_HIP(hipMemcpyAsync(a->eps_cur_in, in->eps_cur_dev, nb, hipMemcpyDeviceToDevice, st)); *vbits |= V_EPSC; }
  if (in->weights_host) {
    GCRL_CHECK_ARG(!a->rowchain && a->Q == 1, "update: importance-sampling weights need the layer-per-launch schedule (pipeline_steps = 0) and scalar critics");
    GCRL_HIP(hipMemcpyAsync(a->w_in, in->weights_host, (size_t)B * sizeof(float), hipMemcpyHostToDevice, st));   // pageable source: staged synchronously
    *vbits |= V_WEIGHTS;
  }
  return GCRL_OK;
}

// phase-0 entry of `n` steps: control table + indices upload, batch gather / pack
// Draw and gather the batches a call left for later ([deferred.next, n); same MT stream order as drawing them all up
// front).  Called once the head's steps have been issued, so that the GPU has more work queued than the host needs for the
// draw, and the upload + gather run in order behind them.  THREE steps (round 3: with two — 110 us of GPU work — the trace of a 20-step call still showed 25 us of idle GPU before the
// main gather, in every call and trainer cycle: the six launches of two steps cost the host ~30 us before it can start drawing).  (Round-2 history: gathering everything up
// front delayed step 0 by the whole draw; deferring batches 1.. behind step 0 left the GPU idle for 21 us per call while
// the host was still drawing; a second stream for the deferred part removed the wait on the driver's 20-step line but
// cost 1-3 % in steady state and depended on how HIP maps streams to hardware queues.)
int finish_deferred_draw(gcrl_agent* a, hipStream_t st) {
  gcrl_her* her = a->deferred.her;
  if (!her) return GCRL_OK;
  a->deferred.her = nullptr;
  const int n = a->deferred.n, B = a->B, first = a->deferred.next;
  uint32_t* idx = (uint32_t*)(a->upload_pinned[a->deferred.slot] + sizeof(UploadBlock));
  for (int i = first; i < n; ++i) TRY(gcrl_mt_sample_indices(her->rng, (uint32_t)her->len, (uint32_t)B, idx + (size_t)i * B));
  GCRL_HIP(hipMemcpyAsync(a->idx_dev() + (size_t)first * B, idx + (size_t)first * B, (size_t)(n - first) * B * sizeof(uint32_t),
                          hipMemcpyHostToDevice, st));
  GCRL_HIP(hipEventRecord(a->upload_ev[a->deferred.slot], st));
  return her_gather_update(her, a->idx_dev() + (size_t)first * B, (int64_t)(n - first) * B, a->sa + first * a->slot_x,
                           a->nsa + first * a->slot_x, a->rowchain ? nullptr : a->spa + first * a->slot_x, a->ldx,
                           a->rbuf + first * a->slot_rd, a->dbuf + first * a->slot_rd, st);
}

int order_after_other_handles(gcrl_agent* a, hipStream_t st);
int begin_call(gcrl_agent* a, gcrl_her* her, int64_t step0, int n, const gcrl_update_inputs* in, float grad_scale,
               hipStream_t st, std::vector<StepPlan>& plans, int64_t* tickets, int32_t* lens, bool defer_rest = false,
               bool pre_advanced = false) {
  TRY(finish_deferred_draw(a, st));   // (never pending here; cheap safety)
  TRY(order_after_other_handles(a, st));
  GCRL_CHECK_ARG(n >= 1 && n <= kMaxStepsPerCall && n <= a->Mmax, "update: n=%d steps per call (max %d)", n, std::min(kMaxStepsPerCall, a->Mmax));
  const bool injected = in && in->s_dev;
  GCRL_CHECK_ARG(injected || her, "update: neither a replay ring nor an injected batch was given");
  GCRL_CHECK_ARG(!injected || n == 1, "update: an injected batch drives exactly one step");
  if (!injected) {
    GCRL_CHECK_ARG(her->S == a->S && her->A == a->A, "update: ring dims (S=%d,A=%d) differ from the agent's (S=%d,A=%d)", her->S, her->A, a->S, a->A);
    if (her->len < a->B) return fail(GCRL_ERR_NOT_ENOUGH, "[ERROR] Not enough in buffer to sample");
  }
  if (!injected && in && in->idx_host)     // every argument is checked before a ticket, slot or plan is consumed
    for (int i = 0; i < a->B; ++i)
      GCRL_CHECK_ARG((int64_t)in->idx_host[i] < her->len, "update: row index %u outside the ring (len %lld)", in->idx_host[i], (long long)her->len);
  const int slot = a->next_upload;
  a->next_upload = (slot + 1) % kCtrlSlots;
  GCRL_HIP(hipEventSynchronize(a->upload_ev[slot]));
  UploadBlock* ub = (UploadBlock*)a->upload_pinned[slot];
  uint32_t* idx = (uint32_t*)(a->upload_pinned[slot] + sizeof(UploadBlock));
  ub->cb.cursor = pre_advanced ? 1 : 0;   // pre_advanced: cur = table[0] travels in the block itself, the first step launches no begin_step
  plans.resize(n);
  StepCtrl* table = ub->cb.table;
  for (int i = 0; i < n; ++i) {
    const int64_t ticket = a->next_ticket++;
    plans[i] = plan_step(a, step0 + i, grad_scale, i, ticket, &table[i]);
    a->ticket_len[ticket % kMetricSlots] = plans[i].tuple_len;
    if (tickets) tickets[i] = ticket;
    if (lens) lens[i] = plans[i].tuple_len;
  }
  ub->cb.cur = table[0];
  size_t bytes = sizeof(UploadBlock);
  const bool device_rng = !injected && her->cfg.rng_mode != GCRL_RNG_CPYTHON_MT;
  const bool explicit_idx = !injected && in && in->idx_host;
  if (explicit_idx) {   // prioritised replay: the caller drew the rows
    for (int i = 0; i < a->B; ++i) idx[i] = in->idx_host[i];
    bytes += (size_t)a->B * sizeof(uint32_t);
  } else if (!injected && !device_rng) {
    const int head = a->head_batches;
    defer_rest = defer_rest && n > head;
    const int now = defer_rest ? head : n;   // same MT stream order either way: the head's batches now, the rest once that many steps are issued
    for (int i = 0; i < now; ++i)
      TRY(gcrl_mt_sample_indices(her->rng, (uint32_t)her->len, (uint32_t)a->B, idx + (size_t)i * a->B));
    bytes += (size_t)now * a->B * sizeof(uint32_t);
    if (defer_rest) { a->deferred.her = her; a->deferred.n = n; a->deferred.slot = slot; a->deferred.next = now; }
  } else defer_rest = false;
  if (device_rng && !explicit_idx) {   // the gather kernel computes the indices itself: nothing to draw or upload here
    her->last_gen = IdxGen{her->cfg.seed, her->draws_done, (uint32_t)her->len, a->B, feistel_half_bits((uint32_t)her->len)};
    her->draws_done += n;
  }
  const bool host_idx = !injected && !(device_rng && !explicit_idx);
  const int64_t rows_now = (int64_t)(a->deferred.her ? a->deferred.next : n) * a->B;
  if (!injected && (!host_idx || rows_now <= (int64_t)a->head_batches * a->B)) {
    // One launch starts the call: the gather reads its (<= 2 B) indices straight from the pinned block and carries the
    // control block to the device (header + the n table entries in use) — before, two staged copies and their launch
    // gaps (19 us) preceded the first gather.
    const size_t cb_bytes = (offsetof(UploadBlock, cb) + offsetof(CtrlBlock, table) + (size_t)n * sizeof(StepCtrl) + 15) & ~(size_t)15;
    TRY(her_gather_update(her, host_idx ? idx : nullptr, rows_now, a->sa, a->nsa, a->rowchain ? nullptr : a->spa, a->ldx, a->rbuf,
                          a->dbuf, st, ub, a->upload_dev, cb_bytes));
    GCRL_HIP(hipEventRecord(a->upload_ev[slot], st));
  } else {
    GCRL_HIP(hipMemcpyAsync(a->upload_dev, ub, bytes, hipMemcpyHostToDevice, st));
    GCRL_HIP(hipEventRecord(a->upload_ev[slot], st));
    if (!injected)
      TRY(her_gather_update(her, host_idx ? a->idx_dev() : nullptr, rows_now, a->sa, a->nsa, a->rowchain ? nullptr : a->spa, a->ldx,
                            a->rbuf, a->dbuf, st));
  }
  return GCRL_OK;
}

// Launch forms whose workgroups wait for each other need every workgroup of the launch resident at once — also against the OTHER
// handles of this process: two agents' update calls queued on their own streams could otherwise run such launches side by side
// (the admission was a promise by the caller).  Per device, the handle that queued update work last and the event that
// closes it; another handle's next call waits for that event on the device (no host wait; nothing happens while one handle is used).
struct WaitOwner { gcrl_agent* a = nullptr; hipEvent_t ev = nullptr; };
std::mutex g_wait_mu;
WaitOwner g_wait_owner[16];

int order_after_other_handles(gcrl_agent* a, hipStream_t st) {
  std::lock_guard<std::mutex> lk(g_wait_mu);
  WaitOwner& w = g_wait_owner[a->cfg.device & 15];
  if (w.a && w.a != a && w.ev) GCRL_HIP(hipStreamWaitEvent(st, w.ev, 0));
  return GCRL_OK;
}

int end_call(gcrl_agent* a, hipStream_t st) {
  const int e = (int)(a->calls % kEventRing);
  GCRL_HIP(hipEventRecord(a->call_ev[e], st));
  a->call_last_ticket[e] = a->next_ticket - 1;
  a->calls++;
  {
    std::lock_guard<std::mutex> lk(g_wait_mu);
    g_wait_owner[a->cfg.device & 15] = WaitOwner{a, a->call_ev[e]};
  }
  return GCRL_OK;
}

// A wait inside a launch timed out (meet.h): the launch poisoned its result with NaN and set a bit of the status word.
// Called after every host synchronisation of the handle: the error surfaces ONCE, the meeting counters are zeroed (a
// timed-out round may have left them off a multiple of the arrival count) and the next launch works again.  The reference
// raises on any failed step (src/agent.py:659-699).
int rowtile_reset(gcrl_agent* a);
int optfuse_reset(gcrl_agent* a);
int meet_check(gcrl_agent* a) {
  if (!a->status_host) return GCRL_OK;
  const unsigned int bits = __atomic_load_n(a->status_host, __ATOMIC_ACQUIRE);
  if (!bits) return GCRL_OK;
  (void)hipDeviceSynchronize();
  if (a->bn_bar && a->bn_xchg) (void)bn_slab_scratch_reset(a->bn_xchg, reinterpret_cast<unsigned int*>(a->bn_bar), a->H, nullptr);
  if (a->rc_bar && a->rc_bar_words) (void)hipMemset(a->rc_bar, 0, (size_t)a->rc_bar_words * sizeof(unsigned int));
  (void)rowtile_reset(a);
  (void)optfuse_reset(a);
  if (bits & (MEET_ERR_XCHG_READY | MEET_ERR_XCHG_DONE)) {
    (void)hipDeviceSynchronize();
    __atomic_store_n(a->status_host, 0u, __ATOMIC_RELEASE);
    return fail(GCRL_ERR_STATE, "the in-engine gradient exchange timed out waiting for a peer (status 0x%x:%s%s): a rank is missing, late by more than the bounded wait (2^24 polls: "
                                "about a minute), or enqueued a different exchange sequence; this step's gradients are NaN on this rank.  Re-synchronise the ranks and call "
                                "gcrl_xchg_reset on each (Python: DataParallelUpdater.recover(), collective), then reload the last checkpoint",
                bits, (bits & MEET_ERR_XCHG_READY) ? " peers' gradients not ready" : "", (bits & MEET_ERR_XCHG_DONE) ? " peers' chunks not delivered" : "");
  }
  (void)hipDeviceSynchronize();
  __atomic_store_n(a->status_host, 0u, __ATOMIC_RELEASE);
  return fail(GCRL_ERR_STATE, "a wait between workgroups inside a launch timed out (status 0x%x:%s%s%s); the affected step's statistics / gradients are NaN. "
                              "The device is probably shared with other work: set GCRL_SHARED_GPU=1 (or gcrl_set_shared_device) to use the launch forms without such waits",
              bits, (bits & MEET_ERR_BN_SLAB) ? " BatchNorm slab row groups" : "", (bits & MEET_ERR_ROWCHAIN) ? " row-chain roles" : "",
              (bits & MEET_ERR_DW_ADAM) ? " gradient-norm slots of the fused optimiser launch" : "");
}

// the fused optimiser launch's norm slots: "not written yet" everywhere, launch counts at zero (creation; after a timed-out wait)
int optfuse_reset(gcrl_agent* a) {
  if (!a->of_slots) return GCRL_OK;
  GCRL_HIP(hipMemset(a->of_slots, 0xFF, (size_t)(a->C + 1) * 2 * a->of_stride * sizeof(unsigned long long)));
  GCRL_HIP(hipMemset(a->of_seq, 0, (size_t)(a->C + 1) * 32 * sizeof(unsigned int)));
  return GCRL_OK;
}

// the weight-slice launch's hand-off words: "not written yet" everywhere, counters at zero (creation; after a timed-out wait)
int rowtile_reset(gcrl_agent* a) {
  if (a->rt_ctr && a->rt_ctr_words) GCRL_HIP(hipMemset(a->rt_ctr, 0, (size_t)a->rt_ctr_words * sizeof(unsigned long long)));
  if (a->rt_xb && a->rt_xb_floats) GCRL_HIP(hipMemset(a->rt_xb, 0xFF, (size_t)a->rt_xb_floats * sizeof(float)));
  if (a->rt_qpart && a->rt_part_floats) GCRL_HIP(hipMemset(a->rt_qpart, 0xFF, (size_t)a->rt_part_floats * sizeof(float)));
  return GCRL_OK;
}

int bytes_alloc(float** p, long long n) {
  GCRL_HIP(hipMalloc((void**)p, (size_t)n * sizeof(float)));
  GCRL_HIP(hipMemset(*p, 0, (size_t)n * sizeof(float)));
  return GCRL_OK;
}

// the weight-slice DDPG launch (rowtile.hip): GCRL_ROWTILE=1 turns it on, GCRL_NO_ROWTILE=1 off (the default until its full-size
// parity and speed are recorded)
bool rowtile_enabled() {
  if (std::getenv("GCRL_NO_ROWTILE")) return false;
  return std::getenv("GCRL_ROWTILE") != nullptr;
}

int build(gcrl_agent* a) {
  const gcrl_agent_config& c = a->cfg;
  GCRL_HIP(hipSetDevice(c.device));
  (void)meet_probe_device(c.device);   // another process of this library on the device: no launch form with an in-kernel wait (meet.h)
  GCRL_HIP(hipStreamCreateWithFlags(&a->stream, hipStreamNonBlocking));
  GCRL_HIP(hipStreamCreateWithFlags(&a->cap_stream, hipStreamNonBlocking));
  const int S = a->S, A = a->A, H = a->H, L = a->L, B = a->B, C = a->C;
  a->actor = make_net(S, H, L, A, a->sac);
  a->critic = make_net(S + A, H, L, a->Q, false);
  a->critic_stride = align_up(a->critic.numel, 64);
  const long long na = align_up(a->actor.numel, 64);
  // params: actor | target_actor | critics | target critics | log_alpha
  long long off = 0;
  a->off_actor = off; off += na;
  a->off_tactor = off; if (a->has_target_actor) off += na;
  a->off_critic = off; off += C * a->critic_stride;
  a->off_tcritic = off; off += C * a->critic_stride;
  a->off_logalpha = off; off += 64;
  a->n_params = off;
  // grads / moments: critics | actor | log_alpha   (each DP exchange block is contiguous)
  off = 0;
  a->goff_critic = off; off += C * a->critic_stride;
  a->goff_actor = off; off += na;
  a->goff_alpha = off; off += 64;
  a->n_grads = off;
  TRY(bytes_alloc(&a->params, a->n_params));
  TRY(bytes_alloc(&a->grads, a->n_grads));
  TRY(bytes_alloc(&a->adam_m, a->n_grads));
  TRY(bytes_alloc(&a->adam_v, a->n_grads));
  TRY(bytes_alloc(&a->bn_rmean, (long long)std::max(1, L * H)));
  TRY(bytes_alloc(&a->bn_rvar, (long long)std::max(1, L * H)));
  TRY(bytes_alloc(&a->alpha_dev, 64));

  // fused-norm partial slots: a dW problem [out, in+1] has at most ceil(out/16)*ceil((in+1)/16)
  // tiles of 16x16 and 4 finishing waves per tile
  auto part_layout = [](const NetSpec& net, std::vector<int>& off) {
    int total = 0;
    off.clear();
    for (const Lin& ln : net.lin) {
      off.push_back(total);
      total += ((ln.out + 15) / 16) * ((ln.in + 1 + 15) / 16) * 4;
    }
    return total;
  };
  a->nparts_c = part_layout(a->critic, a->part_off_c);
  a->nparts_a = part_layout(a->actor, a->part_off_a);
  // dgamma | dbeta of every BatchNorm layer: one slot per 16-column slab (bn_slab.hip; the GEMM + BatchNorm launches use the
  // first ceil(H/64) of a layer's slots, the rest stay zero)
  a->bn_slots = (H + 15) / 16;
  a->bn_slab = a->sac && bn_slab_ok(B, H) && !std::getenv("GCRL_NO_BN_SLAB");
  {
    // (the row groups of a slab wait for each other inside the launch: all (H/16) x ceil(B/128) x 2 workgroups of 512 threads
    // must be resident at once — by the kernels' own occupancy on a device this process has to itself: bn_slab_row_split, meet.h)
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c.device) != hipSuccess) cus = 0;
    a->n_cus = cus;
    a->bn_rsplit = (a->bn_slab && B > 128 && !std::getenv("GCRL_NO_BN_RSPLIT") && bn_slab_row_split(B, H, 2) > 1) ? 4 : 1;
    GCRL_HIP(hipHostMalloc((void**)&a->status_host, 64, hipHostMallocMapped));
    std::memset(a->status_host, 0, 64);
    GCRL_HIP(hipHostGetDevicePointer((void**)&a->status_dev, a->status_host, 0));
  }
  if (a->sac) { a->part_off_bn = a->nparts_a; a->nparts_a += L * a->bn_slots; }
  // work buffers
  const long long BH = (long long)B * H;
  std::vector<std::pair<float**, long long>> wants = {
      {&a->sa, a->Mmax * a->slot_x}, {&a->nsa, a->Mmax * a->slot_x}, {&a->spa, a->Mmax * a->slot_x},
      {&a->rbuf, (long long)a->Mmax * B}, {&a->dbuf, (long long)a->Mmax * B},
      {&a->hTA[0], BH}, {&a->hTA[1], BH}, {&a->hA, L * BH}, {&a->hC, (long long)C * L * BH},
      {&a->hTC, (long long)C * 2 * BH}, {&a->gC, (long long)C * 2 * BH}, {&a->gA[0], BH}, {&a->gA[1], BH},
      {&a->q, (long long)C * B * a->Q}, {&a->qt, (long long)C * B * a->Q}, {&a->q2, (long long)C * B * a->Q}, {&a->dq, (long long)C * B * a->Q},
      {&a->dq2, (long long)C * B * a->Q}, {&a->qy, (long long)B * 64}, {&a->q_row_loss, (long long)C * B}, {&a->q_row_td, B}, {&a->dact, (long long)C * B * a->Apad}, {&a->zA, BH}, {&a->xhatA, L * BH},
      {&a->invstdA, (long long)L * H}, {&a->headA, (long long)B * 2 * a->Apad}, {&a->ghead, (long long)B * 2 * a->Apad},
      {&a->dh2, BH}, {&a->logp, B}, {&a->logp_next, B}, {&a->epsbuf, (long long)B * A}, {&a->stdbuf, (long long)B * A},
      {&a->noise_in, (long long)B * A}, {&a->eps_next_in, (long long)B * A}, {&a->eps_cur_in, (long long)B * A},
      {&a->norm_partial, (long long)kMaxCritics * kNormBlocks}, {&a->act_in, (long long)B * a->ldx},
      {&a->act_tmp[0], BH}, {&a->act_tmp[1], BH},
      {&a->parts_c, (long long)C * a->nparts_c}, {&a->parts_a, (long long)a->nparts_a},
      {&a->hC2, (long long)C * L * BH}, {&a->gC2, 2 * BH}, {&a->bn_part, 2LL * ((B + 15) / 16) * H},
      {&a->rc_gC, (long long)C * L * BH}, {&a->rc_gA, L * BH}, {&a->ybuf, B}, {&a->pi_buf, (long long)B * a->Apad},
      {&a->w_in, B}, {&a->td_abs, B}, {&a->red_scratch, 72LL * ((B + 255) / 256) + 64},
      {&a->bn_bstat, 2LL * 2 * L * H}, {&a->bn_xchg, bn_slab_xchg_floats(H)}, {&a->bn_bar, bn_slab_bar_words(H)}, {&a->zN, BH}, {&a->hN, 2 * BH}, {&a->headN, (long long)B * 2 * a->Apad}, {&a->bn_partN, 2LL * ((B + 15) / 16) * H}};
  // row-block path: plain DDPG nets whose rows fit the 16-byte column ownership
  {
    const long long per_a = (long long)round_up(S, 4) * H + (long long)(L - 1) * H * H;
    const long long per_c = (long long)a->ldx * H + (long long)(L - 1) * H * H;
    a->wt_cstride = align_up(per_c, 64);
    a->wt_net[0] = 0; a->wt_net[1] = align_up(per_a, 64);
    a->wt_net[2] = 2 * align_up(per_a, 64); a->wt_net[3] = a->wt_net[2] + C * a->wt_cstride;
    wants.push_back({&a->wt, a->wt_net[3] + C * a->wt_cstride});
    a->row_ldl = round_up(std::max(H, a->ldx), 4) + 4;
    // fewest rows per block that keep a launch within one wave of blocks (DDPG launches both phases together)
    const int phases = c.kind == GCRL_AGENT_DDPG ? 2 : 1;
    a->row_rg = 1;
    while (a->row_rg < 4 && phases * ((B + 4 * a->row_rg - 1) / (4 * a->row_rg)) > 256) a->row_rg *= 2;
    if (const char* e = std::getenv("GCRL_ROW_RG")) a->row_rg = std::max(1, std::min(4, std::atoi(e)));   // experiment knob
    a->rowchain = (c.kind == GCRL_AGENT_DDPG || c.kind == GCRL_AGENT_TD3 || c.kind == GCRL_AGENT_SAC) && H % 4 == 0 &&
                  c.pipeline_steps >= 2 &&
                  rowchain_lds_bytes(a->row_rg, a->row_ldl, A, H, C) <= 160 * 1024;
    // SAC (the BatchNorm actor runs outside the chain kernels, both phases are critic-only): split by roles while the
    // fused form leaves CUs idle, i.e. up to ~one workgroup per CU per role pair
    a->split_roles = a->rowchain && c.kind == GCRL_AGENT_SAC && C == 2 && (B + 4 * a->row_rg - 1) / (4 * a->row_rg) <= 256 &&
                     !std::getenv("GCRL_NO_SPLIT_ROLES");
    for (int i = 0; i < 4; ++i) a->split_rg[i] = a->row_rg;
    {
      const long long nblk = (B + 4 * a->row_rg - 1) / (4 * a->row_rg);
      // (workgroups that wait for each other inside a launch must all be resident at once: rowchain_merge_ok asks the kernel's
      // occupancy at this LDS size and refuses on a shared device)
      a->rc_merge = a->split_roles && !std::getenv("GCRL_NO_RC_MERGE") && !std::getenv("GCRL_SPLIT_RG") &&
                    rowchain_merge_ok(a->row_rg, a->row_ldl, A, H, C, B);
      if (a->rc_merge) { a->rc_bar_words = 2 * nblk * 32; wants.push_back({&a->rc_bar, a->rc_bar_words}); }
    }
    // TD3 once the batch fills the chip (cfg 3: 183.5 -> 178.4 us/step; below that the fused launch is the shorter chain)
    a->split_k = a->rowchain && c.kind == GCRL_AGENT_TD3 && C == 2 && (B + 4 * a->row_rg - 1) / (4 * a->row_rg) >= 256 &&
                 !std::getenv("GCRL_NO_SPLIT_TD3");
    // ... and its two launches (forward | backward) as ONE: the online-critic workgroups go on to their backward chains once the
    // target roles of their rows have reported in (producers / consumers: no residency requirement, meet.h)
    a->rc_merge_k = a->split_k && !meet_device_shared() && !std::getenv("GCRL_NO_RC_MERGE");
    if (a->rc_merge_k && !a->rc_bar_words) {
      const long long nblk = (B + 4 * a->row_rg - 1) / (4 * a->row_rg);
      a->rc_bar_words = 2 * nblk * 32;
      wants.push_back({&a->rc_bar, a->rc_bar_words});
    }
    // DDPG, one critic: the weight-slice launch (rowtile.hip) when all its 3 * (B/16) * (H/16) workgroups are resident at once
    a->rowtile_can = a->rowchain && c.kind == GCRL_AGENT_DDPG && C == 1 && rowtile_shape_ok(B, H, L, S, A, C);
    a->rowtile = a->rowtile_can && rowtile_enabled() && rowtile_ok(B, H, L, S, A, C);
    if (a->rowtile_can) {
      a->rt_ctr_words = rowtile_ctr_words(B, L);
      a->rt_xb_floats = rowtile_xb_floats(B, H, L); a->rt_part_floats = rowtile_part_floats(B, H);
      wants.push_back({&a->rt_xb, a->rt_xb_floats});
      wants.push_back({&a->rt_qpart, a->rt_part_floats}); wants.push_back({&a->rt_ctr, 2 * a->rt_ctr_words});
      wants.push_back({&a->rt_xid, 3LL * (B / 16) * 32});
    }
    // DDPG: target chain and online critic of the critic phase in their own workgroups of the fused launch while all three roles'
    // workgroups fit the chip at once (one per CU)
    {
      const long long nblk = (B + 4 * a->row_rg - 1) / (4 * a->row_rg);
      static const int per_cu = std::getenv("GCRL_DDPG_KSPLIT_PER_CU") ? std::atoi(std::getenv("GCRL_DDPG_KSPLIT_PER_CU")) : 1;   // experiment knob
      a->ddpg_ksplit_can = a->rowchain && c.kind == GCRL_AGENT_DDPG && C == 1 && 3 * nblk <= per_cu * std::max(a->n_cus, 1);
      a->ddpg_ksplit = a->ddpg_ksplit_can && !meet_device_shared() && !std::getenv("GCRL_NO_DDPG_KSPLIT");
      if (a->ddpg_ksplit_can && !a->rc_bar_words) { a->rc_bar_words = 2 * nblk * 32; wants.push_back({&a->rc_bar, a->rc_bar_words}); }
    }
    // DDPG: dW | db + clip + optimiser as one launch whose workgroups (one per 16x16 gradient tile of both nets) are all resident
    // at once; every problem on the k-split 16x16 form the two-launch path uses for it (same bits either way)
    {
      auto tiles16 = [&](const NetSpec& net, bool* form1) {
        long long t = 0;
        for (const Lin& ln : net.lin) {
          GemmDesc d = bwd_dw(a->grads, 1, a->grads, 1, a->grads, ln, B);
          if (gemm_shape_of(d) != 1) *form1 = false;
          t += (long long)((ln.out + 15) / 16) * ((ln.in + 1 + 15) / 16);
        }
        return t;
      };
      bool form1 = true;
      const long long tc = tiles16(a->critic, &form1), ta = tiles16(a->actor, &form1);
      const long long cap = dw_adam_capacity();
      a->of_stride = align_up(std::max(tc, ta) + 8, 32);   // (the last eight words of an array: the leaders' result words)
      // DDPG: critic | actor in one launch of the overlapped step; TD3 / SAC: the C critics in one launch (TD3: then the actor alone)
      const bool kind_ok = (c.kind == GCRL_AGENT_DDPG && C == 1) || ((c.kind == GCRL_AGENT_TD3 || c.kind == GCRL_AGENT_SAC) && C == 2);
      a->opt_fuse_can = a->rowchain && kind_ok && L + 1 <= kFusedMaxLayers && form1 && B < 2048 &&
                        std::max(tc, ta) <= 256LL * kFusedMaxSlotsPerThread && 2 * std::max(tc, ta) <= cap;
      a->opt_fuse = a->opt_fuse_can && !meet_device_shared() && !std::getenv("GCRL_NO_OPT_FUSE");
      if (a->opt_fuse_can) {
        wants.push_back({&a->of_slots, (long long)(C + 1) * 2 * a->of_stride * 2});
        wants.push_back({&a->of_seq, (long long)(C + 1) * 32});
      }
    }
    a->head_batches = 3;   // (TD3 at batch 2048: the host draws 38 x 2048 indices in ~390 us, more than two 170 us steps)
    // round 5, DDPG's overlapped step at 50 us: three steps of queued work no longer cover the host's six launches + the draw (a 20-step call showed
    // 23 us of idle GPU before the main gather); four do: 57.0 -> 55.9 us/step on the 20-step line, the steady state unchanged
    // (profiles/r05_ab_head_batches.txt; reading the indices straight from the pinned block instead of the staged copy: no gain)
    if (c.kind == GCRL_AGENT_DDPG && c.pipeline_steps != 0) a->head_batches = 4;
    if (const char* e = std::getenv("GCRL_HEAD_BATCHES")) a->head_batches = std::max(1, std::min(8, std::atoi(e)));   // experiment knob
    a->dw_batch_off = std::getenv("GCRL_NO_DW_BATCH") != nullptr;
    a->bn_fused_tiled = std::getenv("GCRL_NO_BN_TILED_STATS") == nullptr;
    a->layer_adv_off = std::getenv("GCRL_NO_LAYER_ADV") != nullptr;
    a->red_off = std::getenv("GCRL_NO_MB_REDUCE") != nullptr;
    a->heads_fused_off = std::getenv("GCRL_NO_HEADS_FUSED") != nullptr;
    a->bn_fused = std::getenv("GCRL_BN_FUSED") != nullptr;   // measured equal at cfg 5 (204.6 vs 203.6 us/step): off by default
    if (const char* e = std::getenv("GCRL_SPLIT_RG"))   // experiment knob: four digits, rows/4 per workgroup of the four launches
      for (int i = 0; i < 4 && e[i]; ++i) a->split_rg[i] = e[i] - '0';
  }
  // split dW reductions (gemm_tiled.h): at batch >= 1024 a dW problem is a long reduction into few 64x64 tiles; S workgroups
  // per tile, S = the power of two (<= 8) that brings the phase's launch to about five workgroups per CU (1280)
  {
    auto tiles_of = [](const Lin& ln) { return (long long)((ln.out + 63) / 64) * ((ln.in + 63) / 64); };
    auto layout = [&](const NetSpec& net, int S, std::vector<long long>& po, std::vector<long long>& to, long long* pn, long long* tn) {
      long long p = 0, t = 0;
      po.clear(); to.clear();
      for (const Lin& ln : net.lin) {
        po.push_back(p); to.push_back(t);
        p += tiles_of(ln) * S * kTiledPartStride;
        t += align_up(tiles_of(ln) * kTicketStride, 64);
      }
      *pn = align_up(p, 64); *tn = t;
    };
    auto pick = [&](long long phase_tiles) {
      int S = 1;
      while (S < 8 && phase_tiles * (S * 2) <= 1280 && B / (16 * S * 2) >= 8) S *= 2;
      return S;
    };
    // experiment knob: the split LDS-tiled dW form at any batch size.  Round 4, headline (B = 256), same box: 57.8 us/step with the
    // 16x16 k-split form, 66.5 (S = 2) / 65.8 (S = 4) with the split tiled form, 74.1 with the unsplit one (GCRL_DW_TILED): stays off
    const bool anyb = std::getenv("GCRL_DW_SPLIT_ANYB") != nullptr;
    const bool on = (B >= 1024 || anyb) && B % 32 == 0 && !std::getenv("GCRL_NO_DW_SPLIT");
    // rc_add_dw (DDPG / TD3 on the row-block path: every dW problem of a phase in one launch) at batch >= 2048: TD3 cfg 3
    // 173.8 -> 168.7 us/step (the launch 31.2 -> 27.0 us by rocprofv3 at S = 8, 25.5 at S = 16; element-wise operands — the
    // head's M = 1, the first layer's 27 columns — go through the pipelined k-loop too, in the plain loop they set the launch's
    // length at one memory round trip per k-step).  Not at batch 1024 (DDPG cfg 2: 61.8 -> 67.8 us/step, 8 k-steps per split).
    const bool chain_dw = a->rowchain && (c.kind == GCRL_AGENT_DDPG || c.kind == GCRL_AGENT_TD3) && (B >= 2048 || anyb);
    long long tc = 0, ta = 0, big_c = 0;
    for (const Lin& ln : a->critic.lin) { tc += tiles_of(ln); if (ln.out >= 64 && ln.in >= 64) big_c += tiles_of(ln); }
    for (const Lin& ln : a->actor.lin) ta += tiles_of(ln);
    const bool ens = !a->dw_batch_off && L >= 2 && (long long)C * (L - 1) * ((H + 63) / 64) * ((H + 1 + 63) / 64) >= 512;   // enqueue_critic's dw_batch
    if (on && chain_dw) { a->dw_split_c = pick(C * tc); a->dw_split_a = pick(ta); }
    else if (on && ens) a->dw_split_c = pick(C * big_c);
    // the BatchNorm actor at batch >= 2048 (TQC cfg 4): a hidden layer's dW joins its dX problem's LDS-tiled launch
    if (on && a->sac && B >= 2048 && !a->rowchain && !std::getenv("GCRL_NO_DW_SPLIT_ACTOR")) a->dw_split_a = pick((long long)((H + 63) / 64) * ((H + 63) / 64));
    if (const char* e = std::getenv("GCRL_DW_SPLIT")) {   // experiment knob
      const int S = std::max(1, std::min(16, std::atoi(e)));
      if (a->dw_split_c > 1) a->dw_split_c = S;
      if (a->dw_split_a > 1) a->dw_split_a = S;
    }
    long long pc = 0, tcn = 0, pa = 0, tan = 0;
    layout(a->critic, a->dw_split_c, a->dwp_off_c, a->dwt_off_c, &pc, &tcn);
    layout(a->actor, a->dw_split_a, a->dwp_off_a, a->dwt_off_a, &pa, &tan);
    a->dwp_cstride = pc; a->dwt_cstride = tcn;
    a->dwp_actor = (long long)C * pc; a->dwt_actor = (long long)C * tcn;
    if (a->dw_split_c > 1 || a->dw_split_a > 1) {
      wants.push_back({&a->dw_part, (long long)C * pc + pa});
      wants.push_back({&a->dw_tick, (long long)C * tcn + tan});     // (unsigned int tickets; the arena is zero-filled)
    }
  }
  long long total = 0;
  for (auto& w : wants) total += align_up(w.second, 64);
  TRY(bytes_alloc(&a->work, total));
  long long used = 0;
  for (auto& w : wants) { *w.first = a->work + used; used += align_up(w.second, 64); }
  TRY(rowtile_reset(a));
  TRY(optfuse_reset(a));
  if (a->bn_xchg && a->bn_bar) { TRY(bn_slab_scratch_reset(a->bn_xchg, reinterpret_cast<unsigned int*>(a->bn_bar), H, nullptr)); GCRL_HIP(hipDeviceSynchronize()); }

  // upload block + pinned mirrors, metrics, events
  a->upload_bytes = sizeof(UploadBlock) + (size_t)kMaxStepsPerCall * B * sizeof(uint32_t);
  GCRL_HIP(hipMalloc((void**)&a->upload_dev, a->upload_bytes));
  GCRL_HIP(hipMemset(a->upload_dev, 0, a->upload_bytes));
  for (int i = 0; i < kCtrlSlots; ++i) {
    GCRL_HIP(hipHostMalloc((void**)&a->upload_pinned[i], a->upload_bytes, hipHostMallocDefault));
    GCRL_HIP(hipEventCreateWithFlags(&a->upload_ev[i], hipEventDisableTiming));
  }
  for (int i = 0; i < kEventRing; ++i) {
    GCRL_HIP(hipEventCreateWithFlags(&a->call_ev[i], hipEventDisableTiming));
    a->call_last_ticket[i] = -1;
  }
  // metrics live in device memory (a kernel that writes host-mapped memory costs +0.9 us at
  // its boundary, measured); the host mirror is refreshed in bulk when a ticket is asked for
  GCRL_HIP(hipHostMalloc((void**)&a->metrics_host, (size_t)kMetricSlots * kMetricFloats * sizeof(float), hipHostMallocDefault));
  std::memset(a->metrics_host, 0, (size_t)kMetricSlots * kMetricFloats * sizeof(float));
  GCRL_HIP(hipMalloc((void**)&a->metrics_dev, (size_t)kMetricSlots * kMetricFloats * sizeof(float)));
  GCRL_HIP(hipMemset(a->metrics_dev, 0, (size_t)kMetricSlots * kMetricFloats * sizeof(float)));
  a->ticket_len.assign(kMetricSlots, 0);

  // (the arenas were zero-filled on the null stream; the agent's streams are non-blocking, i.e. NOT ordered after it: without
  // this wait the fills below can land before the arena's memset does and be wiped by it — seen once as an actor gradient of
  // exactly 0 for a whole run, in a test that passes on its own)
  GCRL_HIP(hipDeviceSynchronize());
  // constant upstream gradient of -Q.mean()
  TRY(launch_fill(a->stream, a->dq2, (long long)C * B * a->Q, -1.0f / (float)B));
  TRY(launch_fill(a->stream, a->alpha_dev, 1, 1.0f));  // exp(log_alpha = 0)
  GCRL_HIP(hipStreamSynchronize(a->stream));

  // names
  auto reg = [&](const std::string& k, float* p, long long n) { a->names[k] = {p, n}; };
  reg("actor", a->P_actor(), a->actor.numel);
  reg("grad:actor", a->G_actor(), a->actor.numel);
  reg("adam_m:actor", a->adam_m + a->goff_actor, a->actor.numel);
  reg("adam_v:actor", a->adam_v + a->goff_actor, a->actor.numel);
  if (a->has_target_actor) reg("target_actor", a->P_tactor(), a->actor.numel);
  for (int i = 0; i < C; ++i) {
    const std::string s = std::to_string(i);
    reg("critic_" + s, a->P_critic(i), a->critic.numel);
    reg("target_critic_" + s, a->P_tcritic(i), a->critic.numel);
    reg("grad:critic_" + s, a->G_critic(i), a->critic.numel);
    reg("adam_m:critic_" + s, a->adam_m + a->goff_critic + i * a->critic_stride, a->critic.numel);
    reg("adam_v:critic_" + s, a->adam_v + a->goff_critic + i * a->critic_stride, a->critic.numel);
  }
  reg("td_abs", a->td_abs, B);
  if (a->sac) {
    reg("log_alpha", a->P_logalpha(), 1);
    reg("grad:log_alpha", a->grads + a->goff_alpha, 1);
    reg("adam_m:log_alpha", a->adam_m + a->goff_alpha, 1);
    reg("adam_v:log_alpha", a->adam_v + a->goff_alpha, 1);
    reg("alpha", a->alpha_dev, 1);
    reg("bn_running_mean", a->bn_rmean, (long long)L * H);
    reg("bn_running_var", a->bn_rvar, (long long)L * H);
  }
  return GCRL_OK;
}

void xavier_fill(std::vector<float>& p, const NetSpec& net, std::mt19937_64& gen, bool touch_bn) {
  for (const Lin& ln : net.lin) {
    const double bound = std::sqrt(6.0 / (double)(ln.in + ln.out));  // nn.init.xavier_uniform_
    std::uniform_real_distribution<double> u(-bound, bound);
    for (long long i = 0; i < (long long)ln.in * ln.out; ++i) p[ln.w + i] = (float)u(gen);
    for (int i = 0; i < ln.out; ++i) p[ln.b + i] = 0.01f;  // bias.data.fill_(0.01)
  }
  if (touch_bn)
    for (size_t l = 0; l < net.bn_g.size(); ++l)
      for (int i = 0; i < net.H; ++i) { p[net.bn_g[l] + i] = 1.f; p[net.bn_b[l] + i] = 0.f; }
}

}  // namespace

extern "C" {

gcrl_agent* gcrl_agent_create(const gcrl_agent_config* cfg) {
  auto bad = [](const char* m) -> gcrl_agent* { fail(GCRL_ERR_ARG, "gcrl_agent_create: %s", m); return nullptr; };
  if (!cfg) return bad("null config");
  if (cfg->kind < GCRL_AGENT_DDPG || cfg->kind > GCRL_AGENT_TQC) return bad("unknown agent kind");
  if (cfg->obs_dim < 1 || cfg->ac_dim < 1 || cfg->ac_dim > 16) return bad("obs_dim >= 1 and 1 <= ac_dim <= 16 required");
  if (cfg->hidden_dim < 1 || cfg->layer_count < 1 || cfg->layer_count > 8) return bad("hidden_dim >= 1, 1 <= layer_count <= 8 required");
  if (cfg->batch_size < 1 || cfg->batch_size > 65536) return bad("batch_size must be 1..65536");
  if (cfg->ac_update_freq < 1 || cfg->gradient_step < 1 || cfg->polyak_every < 1) return bad("ac_update_freq, gradient_step, polyak_every must be >= 1");
  int C = 1;
  if (cfg->kind == GCRL_AGENT_TD3 || cfg->kind == GCRL_AGENT_SAC) C = 2;
  if (cfg->kind == GCRL_AGENT_TQC) {
    C = cfg->num_critics;
    if (C < 2 || C > kMaxCritics) return bad("TQC needs 2..8 critics");
    if (cfg->n_quantiles > 1) {   // distributional variant: top_drop atoms per critic are dropped from the pooled, sorted atoms
      if (C * cfg->n_quantiles > 64) return bad("num_critics * n_quantiles must be <= 64 (one wavefront sorts a row's pooled atoms)");
      if (cfg->top_drop < 0 || cfg->top_drop >= cfg->n_quantiles) return bad("top_drop must be in [0, n_quantiles)");
    } else if (cfg->top_drop < 0 || cfg->top_drop >= C) return bad("top_drop must be in [0, num_critics)");
  } else if (cfg->n_quantiles > 1) return bad("n_quantiles > 1 is the distributional TQC variant only");
  int ndev = gcrl_device_count();
  if (ndev <= 0 || cfg->device < 0 || cfg->device >= ndev) {
    fail(GCRL_ERR_HIP, "gcrl_agent_create: no usable HIP device (count=%d, requested %d); there is no CPU fallback", ndev, cfg->device);
    return nullptr;
  }
  gcrl_agent* a = new gcrl_agent;
  a->cfg = *cfg;
  a->S = cfg->obs_dim; a->A = cfg->ac_dim; a->H = cfg->hidden_dim; a->L = cfg->layer_count;
  a->B = cfg->batch_size; a->C = C;
  a->Q = cfg->n_quantiles > 1 ? cfg->n_quantiles : 1;
  a->ldx = round_up(a->S + a->A, 4);
  a->Apad = round_up(a->A, 4);
  a->Mmax = std::min(kMaxStepsPerCall, std::max(1, cfg->gradient_step));
  a->slot_x = (long long)a->B * a->ldx;
  a->slot_rd = a->B;
  a->sac = cfg->kind == GCRL_AGENT_SAC || cfg->kind == GCRL_AGENT_TQC;
  a->has_target_actor = !a->sac;
  a->lr_actor = cfg->actor_lr;
  a->lr_critic = cfg->critic_lr;
  if (build(a) != GCRL_OK) { gcrl_agent_destroy(a); return nullptr; }
  if (gcrl_agent_init_weights(a, cfg->seed, 1) != GCRL_OK) { gcrl_agent_destroy(a); return nullptr; }
  return a;
}

void gcrl_agent_destroy(gcrl_agent* a) {
  if (!a) return;
  {
    std::lock_guard<std::mutex> lk(g_wait_mu);
    for (WaitOwner& w : g_wait_owner) if (w.a == a) w = WaitOwner{};
  }
  if (a->stream) (void)hipStreamSynchronize(a->stream);
  (void)hipDeviceSynchronize();
  for (auto& kv : a->graphs) (void)hipGraphExecDestroy(kv.second);
  float* bufs[] = {a->params, a->grads, a->adam_m, a->adam_v, a->bn_rmean, a->bn_rvar, a->alpha_dev, a->work};
  for (float* p : bufs) if (p) (void)hipFree(p);
  if (a->upload_dev) (void)hipFree(a->upload_dev);
  for (int i = 0; i < kCtrlSlots; ++i) {
    if (a->upload_pinned[i]) (void)hipHostFree(a->upload_pinned[i]);
    if (a->upload_ev[i]) (void)hipEventDestroy(a->upload_ev[i]);
  }
  for (int i = 0; i < kEventRing; ++i) if (a->call_ev[i]) (void)hipEventDestroy(a->call_ev[i]);
  if (a->status_host) (void)hipHostFree(a->status_host);
  if (a->metrics_host) (void)hipHostFree(a->metrics_host);
  if (a->metrics_dev) (void)hipFree(a->metrics_dev);
  if (a->prof_clk) (void)hipFree(a->prof_clk);
  if (a->bn_sync_buf) (void)hipFree(a->bn_sync_buf);
  if (a->act_pinned) (void)hipHostFree(a->act_pinned);
  if (a->oa_pinned) (void)hipHostFree(a->oa_pinned);
  if (a->act_fl_host) (void)hipHostFree(a->act_fl_host);
  if (a->oa_dev) (void)hipFree(a->oa_dev);
  for (int i = 0; i < gcrl_agent::kProfPairs; ++i) {
    if (a->prof_a[i]) (void)hipEventDestroy(a->prof_a[i]);
    if (a->prof_b[i]) (void)hipEventDestroy(a->prof_b[i]);
  }
  if (a->stream) (void)hipStreamDestroy(a->stream);
  if (a->cap_stream) (void)hipStreamDestroy(a->cap_stream);
  delete a;
}

void* gcrl_agent_stream(const gcrl_agent* a) { return a ? (void*)a->stream : nullptr; }

int64_t gcrl_agent_numel(const gcrl_agent* a, const char* name) {
  if (!a || !name) return -1;
  auto it = a->names.find(name);
  return it == a->names.end() ? -1 : it->second.second;
}

static int find_vec(gcrl_agent* a, const char* name, float** ptr, int64_t* numel) {
  GCRL_CHECK_ARG(a && name && ptr, "gcrl_agent_dev_ptr: null argument");
  auto it = a->names.find(name);
  GCRL_CHECK_ARG(it != a->names.end(), "unknown vector name '%s'", name);
  *ptr = it->second.first;
  if (numel) *numel = it->second.second;
  return GCRL_OK;
}

int gcrl_agent_dev_ptr(gcrl_agent* a, const char* name, float** ptr, int64_t* numel) {
  TRY(find_vec(a, name, ptr, numel));
  a->wt_dirty = true;   // the caller may write through the pointer (data-parallel parameter broadcast)
  return GCRL_OK;
}

int gcrl_agent_get(gcrl_agent* a, const char* name, float* dst, int64_t n) {
  float* p = nullptr; int64_t numel = 0;
  TRY(find_vec(a, name, &p, &numel));
  GCRL_CHECK_ARG(dst && n == numel, "gcrl_agent_get('%s'): n=%lld but the vector has %lld elements", name, (long long)n, (long long)numel);
  GCRL_HIP(hipDeviceSynchronize());
  TRY(meet_check(a));
  GCRL_HIP(hipMemcpy(dst, p, (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
  return GCRL_OK;
}

int gcrl_agent_set(gcrl_agent* a, const char* name, const float* src, int64_t n) {
  float* p = nullptr; int64_t numel = 0;
  TRY(find_vec(a, name, &p, &numel));
  GCRL_CHECK_ARG(src && n == numel, "gcrl_agent_set('%s'): n=%lld but the vector has %lld elements", name, (long long)n, (long long)numel);
  GCRL_HIP(hipDeviceSynchronize());
  GCRL_HIP(hipMemcpy(p, src, (size_t)n * sizeof(float), hipMemcpyHostToDevice));
  a->wt_dirty = true;
  if (std::string(name) == "log_alpha") {  // keep alpha = exp(log_alpha) coherent (src/agent.py:106, :878)
    const float al = std::exp(src[0]);
    GCRL_HIP(hipMemcpy(a->alpha_dev, &al, sizeof(float), hipMemcpyHostToDevice));
  }
  return GCRL_OK;
}

int gcrl_agent_init_weights(gcrl_agent* a, uint64_t seed, int recreate_alpha) {
  GCRL_CHECK_ARG(a, "gcrl_agent_init_weights: null handle");
  GCRL_HIP(hipDeviceSynchronize());
  a->wt_dirty = true;
  std::mt19937_64 gen(seed * 0x9e3779b97f4a7c15ull + 12345);
  const bool first = a->t_critic == 0 && a->t_actor == 0;
  std::vector<float> host;
  auto push = [&](const NetSpec& net, float* dev, bool bn_too) -> int {
    host.assign((size_t)net.numel, 0.f);
    if (!bn_too && net.sac) GCRL_HIP(hipMemcpy(host.data(), dev, (size_t)net.numel * sizeof(float), hipMemcpyDeviceToHost));
    xavier_fill(host, net, gen, bn_too);
    GCRL_HIP(hipMemcpy(dev, host.data(), (size_t)net.numel * sizeof(float), hipMemcpyHostToDevice));
    return GCRL_OK;
  };
  // reset() re-initialises Linear layers of every network, targets included, independently
  // (src/agent.py:1461-1465); BatchNorm affine/statistics only at construction
  TRY(push(a->actor, a->P_actor(), first));
  if (a->has_target_actor) TRY(push(a->actor, a->P_tactor(), first));
  for (int c = 0; c < a->C; ++c) TRY(push(a->critic, a->P_critic(c), false));
  for (int c = 0; c < a->C; ++c) TRY(push(a->critic, a->P_tcritic(c), false));
  if (first) {
    TRY(gcrl_agent_hard_update_targets(a));
    std::vector<float> ones((size_t)std::max(1, a->L * a->H), 1.f);
    GCRL_HIP(hipMemcpy(a->bn_rvar, ones.data(), ones.size() * sizeof(float), hipMemcpyHostToDevice));
    GCRL_HIP(hipMemset(a->bn_rmean, 0, ones.size() * sizeof(float)));
  }
  if (a->sac && recreate_alpha) {
    const float zero = 0.f, one = 1.f;
    GCRL_HIP(hipMemcpy(a->P_logalpha(), &zero, sizeof(float), hipMemcpyHostToDevice));
    GCRL_HIP(hipMemcpy(a->alpha_dev, &one, sizeof(float), hipMemcpyHostToDevice));
    GCRL_HIP(hipMemset(a->adam_m + a->goff_alpha, 0, sizeof(float)));
    GCRL_HIP(hipMemset(a->adam_v + a->goff_alpha, 0, sizeof(float)));
    a->t_alpha = 0;
  }
  GCRL_HIP(hipDeviceSynchronize());   // (null-stream memsets vs the agent's non-blocking streams)
  return GCRL_OK;
}

int gcrl_agent_hard_update_targets(gcrl_agent* a) {
  GCRL_CHECK_ARG(a, "gcrl_agent_hard_update_targets: null handle");
  GCRL_HIP(hipDeviceSynchronize());
  a->wt_dirty = true;
  if (a->has_target_actor)
    GCRL_HIP(hipMemcpy(a->P_tactor(), a->P_actor(), (size_t)a->actor.numel * sizeof(float), hipMemcpyDeviceToDevice));
  GCRL_HIP(hipMemcpy(a->P_tcritic(0), a->P_critic(0), (size_t)a->C * a->critic_stride * sizeof(float), hipMemcpyDeviceToDevice));
  return GCRL_OK;
}

// ---- full resume state (SURVEY.md §8f-2 extension: the reference saves weights + normalisers only, src/env.py:430-440)
namespace {
struct AgentStateHeader {
  uint32_t magic, version;
  int32_t kind, S, A, H, L, B, C, pad;
  int64_t n_params, n_grads, bn_n;
  int64_t t_actor, t_critic, t_alpha;
  double lr_actor, lr_critic;
  uint64_t rng_ctr;
};
constexpr uint32_t kAgentMagic = 0x4c524347u;   // "GCRL"
size_t agent_state_bytes(const gcrl_agent* a) {
  const long long bn = std::max(1, a->L * a->H);
  return sizeof(AgentStateHeader) + (size_t)(a->n_params + 2 * a->n_grads + 2 * bn + 1) * sizeof(float);
}
}  // namespace

int64_t gcrl_agent_state_size(const gcrl_agent* a) { return a ? (int64_t)agent_state_bytes(a) : -1; }

int gcrl_agent_save_state(gcrl_agent* a, void* dst_host, int64_t n) {
  GCRL_CHECK_ARG(a && dst_host && n == (int64_t)agent_state_bytes(a), "gcrl_agent_save_state: buffer must be gcrl_agent_state_size() bytes");
  GCRL_HIP(hipDeviceSynchronize());
  TRY(meet_check(a));
  const long long bn = std::max(1, a->L * a->H);
  AgentStateHeader h{kAgentMagic, 1, a->cfg.kind, a->S, a->A, a->H, a->L, a->B, a->C, 0, a->n_params, a->n_grads, bn,
                     a->t_actor, a->t_critic, a->t_alpha, a->lr_actor, a->lr_critic, a->rng_ctr};
  char* o = (char*)dst_host;
  std::memcpy(o, &h, sizeof(h)); o += sizeof(h);
  auto put = [&](const float* dev, long long cnt) -> int {
    GCRL_HIP(hipMemcpy(o, dev, (size_t)cnt * sizeof(float), hipMemcpyDeviceToHost));
    o += (size_t)cnt * sizeof(float);
    return GCRL_OK;
  };
  TRY(put(a->params, a->n_params)); TRY(put(a->adam_m, a->n_grads)); TRY(put(a->adam_v, a->n_grads));
  TRY(put(a->bn_rmean, bn)); TRY(put(a->bn_rvar, bn)); TRY(put(a->alpha_dev, 1));
  return GCRL_OK;
}

int gcrl_agent_load_state(gcrl_agent* a, const void* src_host, int64_t n) {
  GCRL_CHECK_ARG(a && src_host && n == (int64_t)agent_state_bytes(a), "gcrl_agent_load_state: the blob does not have this agent's size");
  AgentStateHeader h;
  std::memcpy(&h, src_host, sizeof(h));
  GCRL_CHECK_ARG(h.magic == kAgentMagic && h.version == 1, "gcrl_agent_load_state: not an agent state blob");
  GCRL_CHECK_ARG(h.kind == a->cfg.kind && h.S == a->S && h.A == a->A && h.H == a->H && h.L == a->L && h.C == a->C &&
                     h.n_params == a->n_params && h.n_grads == a->n_grads,
                 "gcrl_agent_load_state: the blob was saved by a different agent shape");
  GCRL_HIP(hipDeviceSynchronize());
  const char* o = (const char*)src_host + sizeof(h);
  auto get = [&](float* dev, long long cnt) -> int {
    GCRL_HIP(hipMemcpy(dev, o, (size_t)cnt * sizeof(float), hipMemcpyHostToDevice));
    o += (size_t)cnt * sizeof(float);
    return GCRL_OK;
  };
  TRY(get(a->params, a->n_params)); TRY(get(a->adam_m, a->n_grads)); TRY(get(a->adam_v, a->n_grads));
  TRY(get(a->bn_rmean, h.bn_n)); TRY(get(a->bn_rvar, h.bn_n)); TRY(get(a->alpha_dev, 1));
  a->t_actor = h.t_actor; a->t_critic = h.t_critic; a->t_alpha = h.t_alpha;
  a->lr_actor = h.lr_actor; a->lr_critic = h.lr_critic; a->rng_ctr = h.rng_ctr;
  a->wt_dirty = true;
  return GCRL_OK;
}

// update_target_network(hard_update=False, tau) (src/agent.py:1259-1271 and its copies): every target <- tau*net + (1-tau)*target
int gcrl_agent_soft_update_targets(gcrl_agent* a, double tau, void* stream) {
  GCRL_CHECK_ARG(a, "gcrl_agent_soft_update_targets: null handle");
  hipStream_t st = a->pick(stream);
  if (a->has_target_actor) TRY(launch_polyak(st, a->P_actor(), a->P_tactor(), a->actor.numel, tau));
  for (int c = 0; c < a->C; ++c) TRY(launch_polyak(st, a->P_critic(c), a->P_tcritic(c), a->critic.numel, tau));
  if (a->rowchain) TRY(rc_rebuild_wt(a, st, true));
  return GCRL_OK;
}

int gcrl_agent_profile_enable(gcrl_agent* a, int on) {
  GCRL_CHECK_ARG(a, "gcrl_agent_profile_enable: null handle");
  if (on && !a->prof_clk) {
    GCRL_HIP(hipMalloc((void**)&a->prof_clk, 2 * gcrl_agent::kProfPairs * sizeof(unsigned long long)));
    for (int i = 0; i < gcrl_agent::kProfPairs; ++i) {
      GCRL_HIP(hipEventCreate(&a->prof_a[i]));
      GCRL_HIP(hipEventCreate(&a->prof_b[i]));
    }
  }
  if (!on) TRY(prof_drain(a));
  else { a->prof_launches = 0; a->prof_ms = 0; a->prof_ticks = 0; a->prof_used = 0; }
  a->prof = on != 0;
  return GCRL_OK;
}

int gcrl_agent_profile_read(gcrl_agent* a, int64_t* launches_out, double* total_ms_out, double* device_clock_ms_out) {
  GCRL_CHECK_ARG(a, "gcrl_agent_profile_read: null handle");
  TRY(prof_drain(a));
  if (launches_out) *launches_out = a->prof_launches;
  if (total_ms_out) *total_ms_out = a->prof_ms;
  if (device_clock_ms_out) *device_clock_ms_out = a->prof_ticks / 1e5;   // wall_clock64: 100 MHz
  return GCRL_OK;
}

int gcrl_agent_update(gcrl_agent* a, gcrl_her* her, int64_t step, const gcrl_update_inputs* in, int64_t* ticket_out, void* stream) {
  GCRL_CHECK_ARG(a, "gcrl_agent_update: null handle");
  hipStream_t st = a->pick(stream);
  std::vector<StepPlan> plans;
  int32_t len = 0;
  TRY(begin_call(a, her, step, 1, in, a->xchg_scale(), st, plans, ticket_out, &len));
  int variant = plans[0].variant | norm_bits(a);
  if (in) TRY(stage_injected(a, in, st, &variant));
  if (a->rowchain && a->cfg.kind == GCRL_AGENT_DDPG && variant == (V_ACTOR | norm_bits(a))) {
    TRY(run_ddpg_pipe(a, st, 1));   // the phases of a plain DDPG step, row-block kernels
    TRY(run_ddpg_pipe(a, st, 2));
  } else {
    TRY(run_step(a, st, variant, 7));
  }
  TRY(end_call(a, st));
  return len;
}

int gcrl_agent_update_n(gcrl_agent* a, gcrl_her* her, int64_t step0, int n, int64_t* tickets_out, int32_t* lens_out, void* stream) {
  GCRL_CHECK_ARG(a && her, "gcrl_agent_update_n: null handle");
  GCRL_CHECK_ARG(n >= 1, "gcrl_agent_update_n: n must be >= 1");
  // a process that arrived on this device after the handle was built: its kernels hold CUs, so no more waits inside launches
  if ((a->calls & 31) == 0 && !meet_device_shared()) { const int rc = gcrl_agent_get_meetings(a); if (rc < 0) return rc; }   // (probes, and switches the forms off)
  hipStream_t st = a->pick(stream);
  const int chunk = std::min(kMaxStepsPerCall, a->Mmax);
  for (int done = 0; done < n; done += chunk) {
    const int m = std::min(chunk, n - done);
    std::vector<StepPlan> plans;
    const bool ddpg_pipe = a->cfg.kind == GCRL_AGENT_DDPG && a->cfg.pipeline_steps != 0;
    // SAC / TD3 on the row-block path (the call's last step advances into table[m]: never read); round 4: also the BatchNorm-actor
    // agents on the layer-per-launch path (TQC's cfg 4) when every step is an actor step — then the actor's optimiser launch is the
    // step's last one, and the TD-loss launch refreshes the copies it reads
    const bool layer_adv = a->sac && !a->rowchain && a->Q == 1 && a->cfg.ac_update_freq == 1 && !a->layer_adv_off;
    const int adv = ((a->rowchain && !ddpg_pipe) || layer_adv) ? V_ADV : 0;
    const bool pre = ddpg_pipe || adv != 0;                 // these paths start from the uploaded cur: no begin_step launch at all
    TRY(begin_call(a, her, step0 + done, m, nullptr, a->xchg_scale(), st, plans, tickets_out ? tickets_out + done : nullptr,
                   lens_out ? lens_out + done : nullptr, /*defer_rest=*/true, pre));
    if (ddpg_pipe) {
      // plain actor steps overlap pairwise: P(i) shares its launches with K(i+1)
      std::vector<int> variants(m);
      for (int i = 0; i < m; ++i) variants[i] = plans[i].variant;
      TRY(run_steps_ddpg(a, st, variants.data(), m, /*first_pre=*/true));
    } else {
      auto var_of = [&](int i) { return plans[i].variant | norm_bits(a) | adv | (adv ? V_PRE : 0); };
      int i = 0;
      while (i < m) {
        int j = i + 1;
        // steps 0 and 1 go out one by one (the rest of the call's batches is drawn once both are queued); after that a
        // run of identical steps is one graph launch
        // (at most 8 steps per graph: replaying graphs of 1 600 kernel nodes showed occasional 25-120 us stalls between nodes)
        if (!a->deferred.her)
          while (j < m && j - i < 8 && var_of(j) == var_of(i)) ++j;
        TRY(run_step(a, st, var_of(i), 7, j - i));
        if (j >= a->head_batches) TRY(finish_deferred_draw(a, st));   // the head's steps are in flight: now draw and gather the other batches
        i = j;
      }
    }
    TRY(finish_deferred_draw(a, st));
    TRY(end_call(a, st));
  }
  return GCRL_OK;
}

int gcrl_agent_update_phase(gcrl_agent* a, gcrl_her* her, int64_t step, int phase, const gcrl_update_inputs* in,
                            float grad_scale, int64_t* ticket_out, void* stream) {
  GCRL_CHECK_ARG(a && phase >= 0 && phase <= 2, "gcrl_agent_update_phase: bad arguments");
  hipStream_t st = a->pick(stream);
  int len = 0;
  if (phase == 0) {
    std::vector<StepPlan> plans;
    int32_t l = 0;
    TRY(begin_call(a, her, step, 1, in, a->xchg ? a->xchg_scale() : grad_scale, st, plans, ticket_out, &l));
    int variant = plans[0].variant | (a->xchg ? V_XCHG : 0);
    if (in) TRY(stage_injected(a, in, st, &variant));
    a->pending_variant = variant;
    len = l;
  }
  TRY(run_step(a, st, a->pending_variant, 1 << phase));
  if (phase == 2) TRY(end_call(a, st));
  return len;
}

int gcrl_agent_dp_begin(gcrl_agent* a, gcrl_her* her, int64_t step0, int n, float grad_scale, int64_t* tickets_out,
                        int32_t* lens_out, void* stream) {
  GCRL_CHECK_ARG(a && her, "gcrl_agent_dp_begin: null handle");
  if (a->xchg) return fail(GCRL_ERR_STATE, "gcrl_agent_dp_begin: an in-engine exchange is attached (gcrl_agent_set_exchange): the update entry points exchange by themselves");
  hipStream_t st = a->pick(stream);
  a->dp_plans.clear();
  TRY(begin_call(a, her, step0, n, nullptr, grad_scale, st, a->dp_plans, tickets_out, lens_out));
  dp_build_schedule(a);
  return GCRL_OK;
}

int gcrl_agent_dp_run(gcrl_agent* a, float** reduce_ptr_out, int64_t* reduce_numel_out, void* stream) {
  GCRL_CHECK_ARG(a && reduce_ptr_out && reduce_numel_out, "gcrl_agent_dp_run: null argument");
  GCRL_CHECK_ARG(a->dp_pos < a->dp_segs.size(), "gcrl_agent_dp_run: no data-parallel cycle in progress");
  hipStream_t st = a->pick(stream);
  const DpSeg sg = a->dp_segs[a->dp_pos++];
  if (sg.kind == 0) TRY(run_step(a, st, a->dp_plans[sg.a].variant, 1 << sg.b));
  else TRY(run_ddpg_pipe(a, st, sg.a, sg.b));
  *reduce_ptr_out = sg.reduce;
  *reduce_numel_out = sg.reduce_n;
  if (a->dp_pos < a->dp_segs.size()) return 1;
  a->dp_plans.clear();
  a->dp_segs.clear();
  a->dp_pos = 0;
  TRY(end_call(a, st));
  return 0;
}

int gcrl_agent_dp_run_all(gcrl_agent* a, gcrl_dp* d, void* stream) {
  GCRL_CHECK_ARG(a && d, "gcrl_agent_dp_run_all: null handle");
  for (;;) {
    float* ptr = nullptr;
    int64_t n = 0;
    const int more = gcrl_agent_dp_run(a, &ptr, &n, stream);
    if (more < 0) return more;
    if (n > 0) TRY(gcrl_dp_allreduce_sum(d, ptr, n, stream ? stream : (void*)a->stream));
    if (!more) return GCRL_OK;
  }
}

// BnSync exchange: the in-engine peer-to-peer exchange over the partials' own arena (segments: problem 0's array, problem 1's),
// the library-owned communicator when one was given, else the caller's function
static const float* bn_sync_result(const float* dev, void* user) {
  gcrl_agent* a = (gcrl_agent*)user;
  return a->bn_xchg_h ? gcrl_xchg_result(a->bn_xchg_h) + (dev - a->bn_sync_buf) : dev;
}
static int bn_sync_exchange(float* dev, long long n, hipStream_t st, void* user) {
  gcrl_agent* a = (gcrl_agent*)user;
  if (a->bn_xchg_h) {
    const long long n1 = a->bn_sync_cap / 2, off = dev - a->bn_sync_buf;
    if (off % n1 != 0 || n % n1 != 0 || off + n > a->bn_sync_cap) return gcrl::fail(GCRL_ERR_STATE, "SyncBN: exchange of %lld floats at %lld is not a whole segment", n, off);
    return gcrl_xchg_allreduce(a->bn_xchg_h, (int)(off / n1), (int)(n / n1), (void*)st);
  }
  if (a->bn_sync_dp) return gcrl_dp_allreduce_sum(a->bn_sync_dp, dev, (int64_t)n, (void*)st);
  if (!a->bn_sync_fn) return gcrl::fail(GCRL_ERR_STATE, "SyncBN: no exchange function");
  if (a->bn_sync_fn(dev, (int64_t)n, (void*)st, a->bn_sync_user) != 0) return gcrl::fail(GCRL_ERR_STATE, "SyncBN: the exchange function reported a failure");
  return GCRL_OK;
}

static int dp_sync_bn_setup(gcrl_agent* a, int world, int rank, gcrl_dp* dp, gcrl_exchange_fn fn, void* user, gcrl_xchg* bx);

int gcrl_agent_dp_sync_bn(gcrl_agent* a, int world, int rank, gcrl_dp* dp, gcrl_exchange_fn fn, void* user) {
  GCRL_CHECK_ARG(a, "gcrl_agent_dp_sync_bn: null handle");
  GCRL_CHECK_ARG(world == 1 || dp || fn, "gcrl_agent_dp_sync_bn: a communicator or an exchange function is required");
  return dp_sync_bn_setup(a, world, rank, dp, fn, user, nullptr);
}

// The partials' arena and an exchange handle over it (two segments: the two co-scheduled forwards' arrays; the backward's partials
// reuse the second).  The caller connects it like the gradient exchange (gcrl_xchg_handles / _connect / _selftest) and then hands
// it to gcrl_agent_dp_sync_bn_xchg.  Not owned by the agent.
gcrl_xchg* gcrl_agent_bn_xchg_create(gcrl_agent* a, int rank, int world) {
  if (!a || !a->sac || world < 2 || world > 8 || rank < 0 || rank >= world) { fail(GCRL_ERR_ARG, "gcrl_agent_bn_xchg_create: a BatchNorm actor and 2 <= world <= 8 required"); return nullptr; }
  if (hipDeviceSynchronize() != hipSuccess) { fail(GCRL_ERR_HIP, "gcrl_agent_bn_xchg_create: device error"); return nullptr; }
  const long long n1 = 2LL * world * ((a->B + 63) / 64) * a->H;
  if (2 * n1 != a->bn_sync_cap) {   // (the arena's size is part of the exchange's layout: exactly two segments)
    if (a->bn_sync_buf) { (void)hipFree(a->bn_sync_buf); a->bn_sync_buf = nullptr; a->bn_sync_cap = 0; }
    if (bytes_alloc(&a->bn_sync_buf, 2 * n1) != GCRL_OK) return nullptr;
    a->bn_sync_cap = 2 * n1;
  }
  const int64_t off[2] = {0, n1}, n[2] = {n1, n1};
  return gcrl_xchg_create(a->bn_sync_buf, 2 * n1, off, n, 2, rank, world, a->cfg.device);
}

int gcrl_agent_dp_sync_bn_xchg(gcrl_agent* a, int world, int rank, gcrl_xchg* bx) {
  GCRL_CHECK_ARG(a && bx && world >= 2 && gcrl_xchg_world(bx) == world, "gcrl_agent_dp_sync_bn_xchg: a connected exchange handle of this world is required");
  return dp_sync_bn_setup(a, world, rank, nullptr, nullptr, nullptr, bx);
}

static int dp_sync_bn_setup(gcrl_agent* a, int world, int rank, gcrl_dp* dp, gcrl_exchange_fn fn, void* user, gcrl_xchg* bx) {
  GCRL_CHECK_ARG(world >= 1 && world <= 15 && rank >= 0 && rank < world, "gcrl_agent_dp_sync_bn: world must be 1..15 and 0 <= rank < world");
  GCRL_CHECK_ARG(world == 1 || a->sac, "gcrl_agent_dp_sync_bn: only the BatchNorm actors (SAC / TQC) have statistics to synchronise");
  GCRL_HIP(hipDeviceSynchronize());
  for (auto& kv : a->graphs) (void)hipGraphExecDestroy(kv.second);   // captured steps hold the old buffers and have no exchanges
  a->graphs.clear();
  a->bn_sync = BnSync{};
  a->bn_sync_dp = nullptr; a->bn_sync_fn = nullptr; a->bn_sync_user = nullptr; a->bn_xchg_h = nullptr;
  if (a->sac && a->parts_a)   // (the slab launches and the BatchNorm launches fill different subsets of a layer's sum-of-squares slots)
    GCRL_HIP(hipMemset(a->parts_a + a->part_off_bn, 0, (size_t)a->L * a->bn_slots * sizeof(float)));
  GCRL_HIP(hipDeviceSynchronize());
  if (world == 1) return GCRL_OK;
  // partial statistics of both co-scheduled forwards, every rank's slots, adjacent: ONE exchange per BatchNorm layer and pass
  const long long n1 = 2LL * world * ((a->B + 63) / 64) * a->H;
  GCRL_CHECK_ARG(!bx || a->bn_sync_cap == 2 * n1, "gcrl_agent_dp_sync_bn_xchg: the handle was created for another world / batch");
  if (2 * n1 > a->bn_sync_cap) {   // a later call may name a larger world (2 -> 4): the buffer grows with it
    if (a->bn_sync_buf) { (void)hipFree(a->bn_sync_buf); a->bn_sync_buf = nullptr; a->bn_sync_cap = 0; }
    TRY(bytes_alloc(&a->bn_sync_buf, 2 * n1));
    a->bn_sync_cap = 2 * n1;
  } else {
    GCRL_HIP(hipMemset(a->bn_sync_buf, 0, (size_t)a->bn_sync_cap * sizeof(float)));   // (other ranks' slots must start from zero)
  }
  a->bn_partN = a->bn_sync_buf;          // (problem 0 of sac_actor_forwards, then problem 1)
  a->bn_part = a->bn_sync_buf + n1;      // (the old allocations stay owned by the handle's free list)
  a->bn_sync.world = world; a->bn_sync.rank = rank;
  a->bn_sync.exchange = bn_sync_exchange; a->bn_sync.user = a;
  a->bn_sync_dp = dp; a->bn_sync_fn = fn; a->bn_sync_user = user;
  a->bn_xchg_h = bx;
  if (bx) { a->bn_sync.result = bn_sync_result; gcrl_xchg_set_status(bx, a->status_dev); }
  GCRL_HIP(hipDeviceSynchronize());
  return GCRL_OK;
}

int gcrl_agent_dp_phase(gcrl_agent* a, int i, int phase, void* stream) {
  GCRL_CHECK_ARG(a && phase >= 0 && phase <= 2, "gcrl_agent_dp_phase: bad arguments");
  GCRL_CHECK_ARG(i >= 0 && i < (int)a->dp_plans.size(), "gcrl_agent_dp_phase: step %d outside the begun cycle of %d", i, (int)a->dp_plans.size());
  return run_step(a, a->pick(stream), a->dp_plans[i].variant, 1 << phase);
}

int gcrl_agent_dp_end(gcrl_agent* a, void* stream) {
  GCRL_CHECK_ARG(a, "gcrl_agent_dp_end: null handle");
  a->dp_plans.clear();
  return end_call(a, a->pick(stream));
}

gcrl_xchg* gcrl_agent_xchg_create(gcrl_agent* a, int rank, int world) {
  if (!a) { fail(GCRL_ERR_ARG, "gcrl_agent_xchg_create: null handle"); return nullptr; }
  std::vector<int64_t> off, n;
  for (int c = 0; c < a->C; ++c) { off.push_back(a->goff_critic + c * a->critic_stride); n.push_back(a->critic.numel); }
  off.push_back(a->goff_actor); n.push_back(a->actor.numel);
  if (a->sac) { off.push_back(a->goff_alpha); n.push_back(1); }
  gcrl_xchg* x = gcrl_xchg_create(a->grads, a->n_grads, off.data(), n.data(), (int)off.size(), rank, world, a->cfg.device);
  if (x) gcrl_xchg_set_status(x, a->status_dev);
  return x;
}

int gcrl_agent_set_exchange(gcrl_agent* a, gcrl_xchg* x) {
  GCRL_CHECK_ARG(a, "gcrl_agent_set_exchange: null handle");
  GCRL_HIP(hipDeviceSynchronize());
  for (auto& kv : a->graphs) (void)hipGraphExecDestroy(kv.second);   // captured steps hold the other sequence
  a->graphs.clear();
  a->xchg = x;
  a->xchg_sep_norm = std::getenv("GCRL_XCHG_SEPARATE_NORM") != nullptr;
  return GCRL_OK;
}

int gcrl_agent_set_meetings(gcrl_agent* a, int on) {
  GCRL_CHECK_ARG(a, "gcrl_agent_set_meetings: null handle");
  GCRL_HIP(hipDeviceSynchronize());
  for (auto& kv : a->graphs) (void)hipGraphExecDestroy(kv.second);   // captured steps hold the old launch forms
  a->graphs.clear();
  const gcrl_agent_config& c = a->cfg;
  const bool want = on != 0 && !meet_device_shared();
  a->bn_rsplit = (want && a->bn_slab && a->B > 128 && !std::getenv("GCRL_NO_BN_RSPLIT") && bn_slab_row_split(a->B, a->H, 2) > 1) ? 4 : 1;
  a->rc_merge = want && a->rc_bar && a->split_roles && !std::getenv("GCRL_NO_RC_MERGE") && !std::getenv("GCRL_SPLIT_RG") &&
                rowchain_merge_ok(a->row_rg, a->row_ldl, c.ac_dim, a->H, a->C, a->B);
  a->rc_merge_k = want && a->rc_bar && a->split_k && !std::getenv("GCRL_NO_RC_MERGE");
  a->ddpg_ksplit = want && a->ddpg_ksplit_can && a->rc_bar && !std::getenv("GCRL_NO_DDPG_KSPLIT");
  a->rowtile = want && a->rowtile_can && rowtile_enabled() && rowtile_ok(a->B, a->H, a->L, a->S, a->A, a->C);
  a->opt_fuse = want && a->opt_fuse_can && !std::getenv("GCRL_NO_OPT_FUSE");
  return (a->bn_rsplit > 1 ? 1 : 0) | ((a->rc_merge || a->rc_merge_k || a->ddpg_ksplit) ? 2 : 0) | (a->rowtile ? 4 : 0) | (a->opt_fuse ? 8 : 0);
}

int gcrl_agent_get_meetings(gcrl_agent* a) {
  GCRL_CHECK_ARG(a, "gcrl_agent_get_meetings: null handle");
  if (!meet_device_shared() && meet_probe_device(a->cfg.device)) {   // (someone arrived since: the forms go off now)
    const int rc = gcrl_agent_set_meetings(a, 0);
    if (rc < 0) return rc;
  }
  return (a->bn_rsplit > 1 ? 1 : 0) | ((a->rc_merge || a->rc_merge_k || a->ddpg_ksplit) ? 2 : 0) | (a->rowtile ? 4 : 0) | (a->opt_fuse ? 8 : 0);
}

int gcrl_agent_debug_meet_fault(gcrl_agent* a) {
  GCRL_CHECK_ARG(a, "gcrl_agent_debug_meet_fault: null handle");
  GCRL_HIP(hipDeviceSynchronize());
  // one meeting counter off its multiple-of-arrivals state (7: not a multiple of 2, 3 or 4, and 3 of 4 arrivers of a 4-way point
  // land in the NEXT round): those arrivers compute a target that is never reached and time out (~1 s) — what a workgroup
  // kept off the chip would cause
  const unsigned long long one = 7;
  float* words = a->rc_merge ? a->rc_bar : (a->bn_rsplit > 1 ? a->bn_bar : nullptr);
  if (a->rowtile) {   // the first-arrival counter of the actor-phase role's row block 0: +7 of its H / 16 arrivals
    GCRL_HIP(hipMemcpy(a->rt_ctr, &one, sizeof(one), hipMemcpyHostToDevice));
    return GCRL_OK;
  }
  if (a->rc_merge_k || (a->ddpg_ksplit && !a->rowtile)) {   // producers / consumers: a consumer's own launch count far ahead of its producers' counter
    const unsigned long long far = 1ull << 40;
    GCRL_HIP(hipMemcpy(a->rc_bar + 4, &far, sizeof(far), hipMemcpyHostToDevice));    // (64-bit word 2 of row block 0's line: critic 0's consumer)
    return GCRL_OK;
  }
  if (a->opt_fuse) {   // the fused optimiser launch: one workgroup's norm slot never arrives (once)
    const unsigned int on = 1;
    GCRL_HIP(hipMemcpy(reinterpret_cast<unsigned int*>(a->of_seq) + 1, &on, sizeof(on), hipMemcpyHostToDevice));   // (critic 0's fault word)
    return GCRL_OK;
  }
  GCRL_CHECK_ARG(words, "gcrl_agent_debug_meet_fault: this agent's launches contain no waits (meetings off or not applicable)");
  if (words == a->bn_bar && bn_slab_data_flag()) {   // the slab exchange whose words are their own flags: row group 1 of slab 0 withholds its words (once)
    const unsigned int on = 1;
    GCRL_HIP(hipMemcpy(reinterpret_cast<unsigned int*>(a->bn_bar) + 1, &on, sizeof(on), hipMemcpyHostToDevice));
    return GCRL_OK;
  }
  GCRL_HIP(hipMemcpy(words, &one, sizeof(one), hipMemcpyHostToDevice));
  return GCRL_OK;
}

int gcrl_agent_grad_ptr(gcrl_agent* a, int phase, float** ptr, int64_t* numel) {
  GCRL_CHECK_ARG(a && ptr && numel && (phase == 0 || phase == 1), "gcrl_agent_grad_ptr: phase must be 0 or 1");
  if (phase == 0) { *ptr = a->grads + a->goff_critic; *numel = a->C * a->critic_stride; }
  else { *ptr = a->grads + a->goff_actor; *numel = (a->goff_alpha - a->goff_actor) + (a->sac ? 1 : 0); }
  return GCRL_OK;
}

int gcrl_agent_metrics(gcrl_agent* a, int64_t ticket, double* out, int n) {
  GCRL_CHECK_ARG(a && out, "gcrl_agent_metrics: null argument");
  GCRL_CHECK_ARG(ticket >= 0 && ticket < a->next_ticket && ticket >= a->next_ticket - kMetricSlots, "gcrl_agent_metrics: ticket %lld is not live", (long long)ticket);
  const int len = a->ticket_len[ticket % kMetricSlots];
  GCRL_CHECK_ARG(n == len, "gcrl_agent_metrics: this step's tuple has %d entries, %d requested", len, n);
  // wait for the call that produced the ticket (the oldest recorded call covering it)
  int64_t best = -1;
  const int64_t lo = std::max<int64_t>(0, a->calls - kEventRing);
  for (int64_t cidx = lo; cidx < a->calls; ++cidx)
    if (a->call_last_ticket[cidx % kEventRing] >= ticket) { best = cidx; break; }
  if (ticket > a->fetched_upto) {
    int64_t upto = ticket;
    if (best >= 0) {
      GCRL_HIP(hipEventSynchronize(a->call_ev[best % kEventRing]));
      upto = a->call_last_ticket[best % kEventRing];
    } else {
      GCRL_HIP(hipDeviceSynchronize());
      upto = a->next_ticket - 1;
    }
    // copy every finished, not yet mirrored record in (at most two) contiguous pieces
    int64_t from = std::max<int64_t>(a->fetched_upto + 1, upto - kMetricSlots + 1);
    while (from <= upto) {
      const int64_t s0 = from % kMetricSlots;
      const int64_t cnt = std::min<int64_t>(upto - from + 1, kMetricSlots - s0);
      GCRL_HIP(hipMemcpy(a->metrics_host + s0 * kMetricFloats, a->metrics_dev + s0 * kMetricFloats,
                         (size_t)cnt * kMetricFloats * sizeof(float), hipMemcpyDeviceToHost));
      from += cnt;
    }
    a->fetched_upto = upto;
    TRY(meet_check(a));
  }
  const float* m = a->metrics_host + (ticket % kMetricSlots) * kMetricFloats;
  const int C = a->C;
  auto meanv = [&](int base) { double s = 0; for (int c = 0; c < C; ++c) s += (double)m[base + c]; return s / C; };
  const bool act = (a->cfg.kind == GCRL_AGENT_DDPG) ? len == 6 : (a->cfg.kind == GCRL_AGENT_TD3 ? len == 8 : len == 9);
  int o = 0;
  switch (a->cfg.kind) {
    case GCRL_AGENT_DDPG:  // (critic_loss, [ac_loss,] td_error, q_value, critic_grad [, ac_grad])
      out[o++] = m[MET_CRITIC_LOSS];
      if (act) out[o++] = m[MET_ACTOR_LOSS];
      out[o++] = m[MET_TD]; out[o++] = m[MET_Q]; out[o++] = m[MET_CRITIC_GRAD];
      if (act) out[o++] = m[MET_ACTOR_GRAD];
      break;
    case GCRL_AGENT_TD3:
    case GCRL_AGENT_SAC:
    default: {
      const bool tqc = a->cfg.kind == GCRL_AGENT_TQC;
      const double l1 = tqc ? meanv(MET_CRITIC_LOSS) : m[MET_CRITIC_LOSS];
      const double l2 = tqc ? l1 : m[MET_CRITIC_LOSS + 1];
      const double g1 = tqc ? meanv(MET_CRITIC_GRAD) : m[MET_CRITIC_GRAD];
      const double g2 = tqc ? g1 : m[MET_CRITIC_GRAD + 1];
      out[o++] = l1; out[o++] = l2;
      if (act) out[o++] = m[MET_ACTOR_LOSS];
      out[o++] = m[MET_TD]; out[o++] = m[MET_Q]; out[o++] = g1; out[o++] = g2;
      if (act) out[o++] = m[MET_ACTOR_GRAD];
      if (act && a->cfg.kind != GCRL_AGENT_TD3) out[o++] = m[MET_ALPHA_LOSS];
      break;
    }
  }
  return GCRL_OK;
}

int gcrl_agent_act(gcrl_agent* a, const float* obs, int n, int ld_obs, float* out, int ld_out, const float* eps, void* stream) {
  GCRL_CHECK_ARG(a && obs && out && n >= 1 && ld_obs >= a->S && ld_out >= a->A, "gcrl_agent_act: bad arguments");
  hipStream_t st = a->pick(stream);
  const int H = a->H, L = a->L;
  if (a->rowchain && !a->sac && !eps) {   // one row-block launch: hidden layers and tanh head for 4 rows per workgroup
    if (a->wt_dirty) TRY(rc_rebuild_wt(a, st));
    RowActArgs ra;
    std::memset(&ra, 0, sizeof(ra));
    ra.actor = make_rownet(a, a->actor, a->P_actor(), 0);
    ra.obs = obs; ra.ld_obs = ld_obs; ra.out = out; ra.ld_out = ld_out;
    ra.n = n; ra.S = a->S; ra.A = a->A; ra.ldl = a->row_ldl;
    return launch_rowchain_act(st, ra);
  }
  for (int r0 = 0; r0 < n; r0 += a->B) {
    const int rows = std::min(a->B, n - r0);
    const float* X = obs + (long long)r0 * ld_obs;
    float* Y = out + (long long)r0 * ld_out;
    if (!a->sac) {
      Launches ls;
      chain_mlp(a, ls, 0, a->actor, a->P_actor(), X, ld_obs, 0, hid_ACT, 0, Y, ld_out, 0, EPI_TANH, rows);
      TRY(ls.run(st));
    } else {
      const float* P = a->P_actor();
      for (int l = 0; l < L; ++l) {
        GemmDesc d = fwd(l == 0 ? X : a->act_tmp[(l - 1) & 1], l == 0 ? ld_obs : H, P, a->actor.lin[l], a->zA, H, rows, EPI_NONE);
        TRY(launch_gemm_batch(st, &d, 1));
        TRY(launch_bn_relu_eval(st, a->zA, rows, H, P + a->actor.bn_g[l], P + a->actor.bn_b[l], a->bn_rmean + (long long)l * H,
                                a->bn_rvar + (long long)l * H, a->act_tmp[l & 1]));
      }
      const int ldh = 2 * a->Apad;
      GemmDesc hd[2] = {fwd(a->act_tmp[(L - 1) & 1], H, P, a->actor.lin[L], a->headA, ldh, rows, EPI_NONE),
                        fwd(a->act_tmp[(L - 1) & 1], H, P, a->actor.lin[L + 1], a->headA + a->Apad, ldh, rows, EPI_NONE)};
      TRY(launch_gemm_batch(st, hd, 2));
      TanhGaussArgs tg;
      std::memset(&tg, 0, sizeof(tg));
      tg.cur = a->cur();
      tg.mu = a->headA; tg.ls_raw = a->headA + a->Apad; tg.ld_head = ldh;
      tg.eps = eps ? eps + (long long)r0 * a->A : nullptr;
      tg.act = Y; tg.act_slot_stride = 0; tg.ld_act = ld_out;
      tg.B = rows; tg.A = a->A;
      tg.deterministic = eps ? 0 : 1;
      TRY(launch_tanh_gauss_fwd(st, tg));
    }
  }
  return GCRL_OK;
}

// select_action from host arrays in one call (src/agent.py:1345-1366 builds a tensor, moves it to the device,
// runs the actor and copies the result back): pinned staging, H2D, gcrl_agent_act, D2H, stream sync.
int gcrl_agent_act_host(gcrl_agent* a, const float* obs_host, int n, int ld_obs, float* out_host, int ld_out, void* stream) {
  GCRL_CHECK_ARG(a && obs_host && out_host && n >= 1 && n <= a->B && ld_obs >= a->S && ld_out >= a->A,
                 "gcrl_agent_act_host: bad arguments (n must be 1..batch_size)");
  hipStream_t st = a->pick(stream);
  if (!a->act_pinned) GCRL_HIP(hipHostMalloc((void**)&a->act_pinned, (size_t)a->B * (a->ldx + a->Apad) * sizeof(float), hipHostMallocDefault));
  float* pin_in = a->act_pinned;
  float* pin_out = a->act_pinned + (size_t)a->B * a->ldx;
  for (int r = 0; r < n; ++r) std::memcpy(pin_in + (size_t)r * a->S, obs_host + (size_t)r * ld_obs, (size_t)a->S * sizeof(float));
  GCRL_HIP(hipMemcpyAsync(a->act_in, pin_in, (size_t)n * a->S * sizeof(float), hipMemcpyHostToDevice, st));
  TRY(gcrl_agent_act(a, a->act_in, n, a->S, a->dact, a->Apad, nullptr, stream));
  GCRL_HIP(hipMemcpyAsync(pin_out, a->dact, (size_t)n * a->Apad * sizeof(float), hipMemcpyDeviceToHost, st));
  GCRL_HIP(hipStreamSynchronize(st));
  for (int r = 0; r < n; ++r) std::memcpy(out_host + (size_t)r * ld_out, pin_out + (size_t)r * a->Apad, (size_t)a->A * sizeof(float));
  return GCRL_OK;
}

// One vector-env step of the acting side as ONE call: raw observation / desired-goal rows from the
// host -> normalize_state_batch (src/agent.py:1435-1447) with the device normalisers (null: that part stays raw) ->
// actor -> select_action's post-processing (modes: act_post_kernel; SAC / TQC: noise = the rsample eps, null = eval)
// -> float64 actions on the host, as the reference returns them.  The epsilon-random branch of DDPG
// (src/agent.py:1348) is the caller's: it consumes the shared Python `random` stream.
int gcrl_agent_observe_act(gcrl_agent* a, gcrl_normalizer* nz_obs, gcrl_normalizer* nz_dg, const float* obs_host, int obs_dim,
                           const float* dg_host, int goal_dim, int n, const double* noise_host, int mode, double* out_host,
                           void* stream) {
  GCRL_CHECK_ARG(a && obs_host && dg_host && out_host && n >= 1 && n <= a->B, "gcrl_agent_observe_act: bad arguments (n must be 1..batch_size)");
  GCRL_CHECK_ARG(obs_dim >= 1 && goal_dim >= 0 && obs_dim + goal_dim == a->S, "gcrl_agent_observe_act: obs_dim %d + goal_dim %d != %d", obs_dim, goal_dim, a->S);
  GCRL_CHECK_ARG(mode >= 0 && mode <= 2, "gcrl_agent_observe_act: mode must be 0, 1 or 2");
  // the kernels index mean[j] / var[j] for j < obs_dim (goal_dim): a normaliser of another size is an argument error
  // (the reference raises numpy's broadcast error, src/utils.py:95-97), never an out-of-bounds device access
  GCRL_CHECK_ARG(!nz_obs || gcrl_normalizer_size(nz_obs) == obs_dim, "gcrl_agent_observe_act: observation normaliser of size %d for obs_dim %d", gcrl_normalizer_size(nz_obs), obs_dim);
  GCRL_CHECK_ARG(!nz_dg || gcrl_normalizer_size(nz_dg) == goal_dim, "gcrl_agent_observe_act: goal normaliser of size %d for goal_dim %d", gcrl_normalizer_size(nz_dg), goal_dim);
  hipStream_t st = a->pick(stream);
  const int D = obs_dim, G = goal_dim, A = a->A;
  const size_t f_raw = (size_t)a->B * (D + G + A), d_cnt = (size_t)a->B * A;
  const size_t bytes = f_raw * sizeof(float) + 2 * d_cnt * sizeof(double) + 64;
  if (!a->oa_pinned) {
    GCRL_HIP(hipHostMalloc((void**)&a->oa_pinned, bytes, hipHostMallocDefault));
    GCRL_HIP(hipMalloc((void**)&a->oa_dev, bytes));
    a->oa_bytes = bytes;
  }
  // layout: doubles first (alignment): noise [B*A], out [B*A]; then floats: obs, dg, eps
  double* p_noise = (double*)a->oa_pinned; double* p_out = p_noise + d_cnt; float* p_f = (float*)(p_out + d_cnt);
  double* d_noise = (double*)a->oa_dev; double* d_out = d_noise + d_cnt; float* d_f = (float*)(d_out + d_cnt);
  if (a->rowchain && !a->sac && n * a->S <= kActInlineFloats && n * A <= kActInlineNoise && !std::getenv("GCRL_ACT_STAGED")) {
    // ONE launch and nothing else (round 4): the raw rows and the noise travel INSIDE the kernel arguments, the float64 actions come
    // back through host-visible memory followed by a flag per workgroup, and the host waits for the flags — no staged copies, no
    // stream synchronisation.  (Before: two copies up, the launch, a copy down, hipStreamSynchronize = 30 us per vector step.)
    if (a->wt_dirty) TRY(rc_rebuild_wt(a, st));
    constexpr size_t kFlagOff = 2048;
    if (!a->act_fl_host) {
      GCRL_HIP(hipHostMalloc((void**)&a->act_fl_host, 4096, hipHostMallocMapped));
      std::memset(a->act_fl_host, 0, 4096);
      GCRL_HIP(hipHostGetDevicePointer((void**)&a->act_fl_dev, a->act_fl_host, 0));
    }
    RowActInline ri;
    RowActArgs& ra = ri.base;
    std::memset(&ra, 0, sizeof(ra));
    ra.actor = make_rownet(a, a->actor, a->P_actor(), 0);
    ra.ld_obs = a->S; ra.out = a->dact; ra.ld_out = a->Apad;
    ra.n = n; ra.S = a->S; ra.A = A; ra.ldl = a->row_ldl;
    ra.D = D;
    gcrl::normalizer_view(nz_obs, &ra.nz_mean, &ra.nz_var, nullptr, &ra.nz_clip, &ra.nz_mode);
    gcrl::normalizer_view(nz_dg, &ra.nzg_mean, &ra.nzg_var, nullptr, &ra.nzg_clip, &ra.nzg_mode);
    ra.post = mode == 1 ? 1 : (mode == 0 ? 2 : 3);
    for (int i = 0; i < n; ++i) {
      std::memcpy(ri.obs_inl + (size_t)i * a->S, obs_host + (size_t)i * D, sizeof(float) * D);
      std::memcpy(ri.obs_inl + (size_t)i * a->S + D, dg_host + (size_t)i * G, sizeof(float) * G);
    }
    ri.with_noise = (noise_host && mode == 1) ? 1 : 0;
    if (ri.with_noise) std::memcpy(ri.noise_inl, noise_host, sizeof(double) * n * A);
    ri.out_host = reinterpret_cast<double*>(a->act_fl_dev);
    ri.flag_host = reinterpret_cast<unsigned long long*>(a->act_fl_dev + kFlagOff);
    ri.seq = ++a->act_seq;
    TRY(launch_rowchain_act_inline(st, ri));
    const int nwg = (n + 3) / 4;
    volatile unsigned long long* flags = reinterpret_cast<volatile unsigned long long*>(a->act_fl_host + kFlagOff);
    bool seen = false;
    for (long spin = 0; spin < 4000000 && !seen; ++spin) {      // (~ms: then the ordinary synchronisation says what happened)
      seen = true;
      for (int w = 0; w < nwg; ++w) seen = seen && flags[w] == ri.seq;
      if (!seen) __builtin_ia32_pause();
    }
    if (!seen) GCRL_HIP(hipStreamSynchronize(st));
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    std::memcpy(out_host, a->act_fl_host, sizeof(double) * n * A);
    return GCRL_OK;
  }
  if (a->rowchain && !a->sac) {
    // ONE launch: raw [observation | goal] rows up, normalisation in the row-chain act kernel's prologue, select_action's
    // tanh / noise / clip in its epilogue
    if (a->wt_dirty) TRY(rc_rebuild_wt(a, st));
    for (int i = 0; i < n; ++i) {
      std::memcpy(p_f + (size_t)i * a->S, obs_host + (size_t)i * D, sizeof(float) * D);
      std::memcpy(p_f + (size_t)i * a->S + D, dg_host + (size_t)i * G, sizeof(float) * G);
    }
    const bool with_noise = noise_host && mode == 1;
    if (with_noise) std::memcpy(p_noise, noise_host, sizeof(double) * n * A);
    // noise [B*A doubles], out [B*A doubles] and the rows are contiguous in the staging block: one copy covers what is used
    if (with_noise) GCRL_HIP(hipMemcpyAsync(d_noise, p_noise, sizeof(double) * n * A, hipMemcpyHostToDevice, st));
    GCRL_HIP(hipMemcpyAsync(d_f, p_f, sizeof(float) * (size_t)n * a->S, hipMemcpyHostToDevice, st));
    RowActArgs ra;
    std::memset(&ra, 0, sizeof(ra));
    ra.actor = make_rownet(a, a->actor, a->P_actor(), 0);
    ra.obs = d_f; ra.ld_obs = a->S; ra.out = a->dact; ra.ld_out = a->Apad;
    ra.n = n; ra.S = a->S; ra.A = A; ra.ldl = a->row_ldl;
    ra.D = D;
    gcrl::normalizer_view(nz_obs, &ra.nz_mean, &ra.nz_var, nullptr, &ra.nz_clip, &ra.nz_mode);
    gcrl::normalizer_view(nz_dg, &ra.nzg_mean, &ra.nzg_var, nullptr, &ra.nzg_clip, &ra.nzg_mode);
    ra.post = mode == 1 ? 1 : (mode == 0 ? 2 : 3);
    ra.noise = with_noise ? d_noise : nullptr; ra.out64 = d_out;
    TRY(launch_rowchain_act(st, ra));
  } else {
  std::memcpy(p_f, obs_host, sizeof(float) * n * D);
  std::memcpy(p_f + (size_t)n * D, dg_host, sizeof(float) * n * G);
  const bool eps_act = a->sac && noise_host;
  if (noise_host) {
    if (eps_act) for (int i = 0; i < n * A; ++i) p_f[(size_t)n * (D + G) + i] = (float)noise_host[i];
    else std::memcpy(p_noise, noise_host, sizeof(double) * n * A);
  }
  if (noise_host && !eps_act) GCRL_HIP(hipMemcpyAsync(d_noise, p_noise, sizeof(double) * n * A, hipMemcpyHostToDevice, st));
  GCRL_HIP(hipMemcpyAsync(d_f, p_f, sizeof(float) * ((size_t)n * (D + G) + (eps_act ? (size_t)n * A : 0)), hipMemcpyHostToDevice, st));
  TRY(gcrl::normalizer_apply_dev(nz_obs, d_f, n, D, D, a->act_in, a->S, 0, st));
  if (G) TRY(gcrl::normalizer_apply_dev(nz_dg, d_f + (size_t)n * D, n, G, G, a->act_in, a->S, D, st));
  TRY(gcrl_agent_act(a, a->act_in, n, a->S, a->dact, a->Apad, eps_act ? d_f + (size_t)n * (D + G) : nullptr, stream));
  const int pm = a->sac ? 2 : mode;
  hipLaunchKernelGGL(act_post_kernel, dim3((n * A + 255) / 256), dim3(256), 0, st, a->dact, a->Apad, n, A,
                     (noise_host && !eps_act) ? d_noise : nullptr, pm, d_out);
  GCRL_HIP(hipGetLastError());
  }
  GCRL_HIP(hipMemcpyAsync(p_out, d_out, sizeof(double) * n * A, hipMemcpyDeviceToHost, st));
  GCRL_HIP(hipStreamSynchronize(st));
  std::memcpy(out_host, p_out, sizeof(double) * n * A);
  return GCRL_OK;
}

}  // extern "C"
