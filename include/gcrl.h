/*
 * gcrl.h — C ABI of libgcrl_hip.so: MI355X-native HER replay + actor-critic update engine.
 *
 * The reference (CodeKnight314/Goal-Conditioned-RL-Framework) has NO native/FFI layer: its
 * boundary is the duck-typed Python surface that src/env.py uses on `agent` and `agent.buffer`
 * (SURVEY.md §8b).  Each entry point below therefore cites the reference *Python* method it
 * replaces; the Python classes in goal-conditioned-rl-framework_amd/src/ (same names and
 * signatures as the reference's) are thin ctypes callers of these functions.
 *
 * Conventions
 *   - plain C: pointers + sizes, no torch / C++ types.  `*_dev` = device (HBM) pointer,
 *     `*_host` = host pointer.  `stream` is a hipStream_t passed as void*: NULL = the
 *     handle's own (non-blocking) stream, GCRL_STREAM_LEGACY = HIP's legacy default stream
 *     (what torch.cuda.current_stream().cuda_stream == 0 means).
 *   - every function returning int returns GCRL_OK (0) or a negative gcrl_status; nothing
 *     throws across the boundary.  gcrl_last_error() gives the message (thread-local).
 *   - one calling thread per handle (the reference's trainer is single-threaded,
 *     src/env.py:334-406); all device work of a handle is stream-ordered.
 *   - there is NO CPU fallback: every device entry point fails with GCRL_ERR_HIP when no
 *     gfx950 device is usable.  Host-only entry points (gcrl_mt_*, gcrl_cosine_lr_*) work
 *     without a GPU.
 */
#ifndef GCRL_H
#define GCRL_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GCRL_ABI_VERSION 1
#define GCRL_STREAM_LEGACY ((void*)1)

typedef enum gcrl_status {
  GCRL_OK = 0,
  GCRL_ERR_ARG = -1,        /* bad argument / unsupported shape */
  GCRL_ERR_HIP = -2,        /* HIP runtime error (message has hipGetErrorString) */
  GCRL_ERR_NOT_ENOUGH = -3, /* sample(B) with len < B  (reference: assert, src/buffer.py:122) */
  GCRL_ERR_STATE = -4       /* call made in the wrong state */
} gcrl_status;

const char* gcrl_last_error(void);
int gcrl_abi_version(void);
/* number of usable HIP devices (0 when none); never fails */
int gcrl_device_count(void);

/* ------------------------------------------------------------------------------------------
 * CPython-exact Mersenne Twister.  Replaces the stdlib `random` calls the reference makes on
 * the hot path: random.randint (src/buffer.py:153) and random.sample (src/buffer.py:124);
 * random.random (src/agent.py:1348) shares the same stream.  Algorithm: CPython 3.10
 * Lib/random.py (_randbelow_with_getrandbits, sample, randrange) + Modules/_randommodule.c
 * (init_by_array, genrand_uint32, genrand_res53).  Host-only.
 * ---------------------------------------------------------------------------------------- */
typedef struct gcrl_mt gcrl_mt;

gcrl_mt* gcrl_mt_create(void);
void gcrl_mt_destroy(gcrl_mt* mt);
/* random.seed(n) for a non-negative int n < 2**64 */
int gcrl_mt_seed(gcrl_mt* mt, uint64_t seed);
/* state[0..623] = MT words, state[624] = index; same layout as random.getstate()[1] */
int gcrl_mt_get_state(const gcrl_mt* mt, uint32_t* state625);
int gcrl_mt_set_state(gcrl_mt* mt, const uint32_t* state625);
uint32_t gcrl_mt_getrandbits(gcrl_mt* mt, int k /* 1..32 */);
uint32_t gcrl_mt_randbelow(gcrl_mt* mt, uint32_t n /* >= 1 */);
int64_t gcrl_mt_randint(gcrl_mt* mt, int64_t a, int64_t b); /* a + randbelow(b-a+1) */
double gcrl_mt_random(gcrl_mt* mt);
/* index stream of random.sample(population_of_len_n, k): out[i] = index chosen i-th.
 * Both the pool path (n <= setsize(k)) and the set path are reproduced. */
int gcrl_mt_sample_indices(gcrl_mt* mt, uint32_t n, uint32_t k, uint32_t* out_host);
/* draw order of HERBuffer.apply_her (src/buffer.py:146-153): for i in 0..T-2, k_future times
 * randint(i+1, T-1); nothing for the last step.  out has k_future*(T-1) entries. */
int gcrl_mt_future_indices(gcrl_mt* mt, int T, int k_future, uint8_t* out_host);

/* ------------------------------------------------------------------------------------------
 * torch.optim.lr_scheduler.CosineAnnealingLR (recursive/"chainable" form) as the reference
 * configures it (src/agent.py:1203-1212 etc.).  Host-only, double precision like torch.
 * lr_after_k_steps: lr used by the optimiser step number k+1 (k scheduler.step() calls done).
 * ---------------------------------------------------------------------------------------- */
double gcrl_cosine_lr_next(double lr_now, double base_lr, double eta_min, int64_t t_max,
                           int64_t last_epoch_after_step);

/* ------------------------------------------------------------------------------------------
 * HER replay ring in HBM.  Replaces class HERBuffer (src/buffer.py:92-179).
 * Layout (fp32): one 64-byte-aligned packed record per transition,
 * [s(S)|a(A)|pad4][ns(S)|pad4][r|d|pad] (every field group 16-byte aligned),
 * records contiguous in arrival order (DESIGN.md "HBM layout" explains why not five field
 * arrays: a random row gather would touch ~3x the 128-B lines).  sample() returns the five
 * dense field matrices of the reference.  The stored dg/ag columns of the reference's tuples
 * are never read by sample() (src/buffer.py:125) and are not materialised.
 * ---------------------------------------------------------------------------------------- */
typedef struct gcrl_her gcrl_her;

/* GCRL_REWARD_HOST: any other callable.  The reference calls whatever was injected (src/env.py:105) once per relabelled row
 * (src/buffer.py:166); with this kind the flush does the same through gcrl_her_set_reward_callback: the picks and the goal
 * swap stay on the device, the rewards of a flush launch's relabel slots are computed by the callback on the host (from a
 * host mirror of the staged achieved goals) and uploaded with the launch. */
enum { GCRL_REWARD_SPARSE = 0, GCRL_REWARD_DENSE = 1, GCRL_REWARD_HOST = 2 };
enum { GCRL_RNG_CPYTHON_MT = 0, GCRL_RNG_DEVICE = 1 };

typedef struct gcrl_her_config {
  int32_t state_dim;   /* S: obs + time feature + goal (goal = LAST goal_dim entries) */
  int32_t action_dim;  /* A */
  int32_t goal_dim;    /* G */
  int64_t capacity;    /* max_mem_len (deque maxlen, src/buffer.py:101) */
  int32_t nenvs;       /* per-env staging areas (src/buffer.py:102) */
  int32_t k_future;    /* relabels per step (src/buffer.py:151); 1..64 here (gcrl_her_create refuses more: the flush kernel gives a
                        * 16-lane group per stored row and keeps a step's picks in one wave) — the reference has no such limit */
  int32_t flush_len;   /* 50: literal in src/buffer.py:117 (max_eps_len is ignored there) */
  int32_t reward_kind; /* GCRL_REWARD_*: stands in for the injected compute_reward
                          (src/env.py:105, src/buffer.py:166): sparse = -(||ag-g||2 > thr) */
  float reward_threshold; /* 0.05 for all Panda tasks */
  int32_t device;
  int32_t rng_mode;    /* GCRL_RNG_* */
  uint64_t seed;       /* used when the handle owns its RNG */
} gcrl_her_config;

/* rng: shared CPython-exact generator (may be NULL: the handle creates one seeded with
 * cfg->seed).  The ring never frees a shared rng. */
gcrl_her* gcrl_her_create(const gcrl_her_config* cfg, gcrl_mt* rng);
void gcrl_her_destroy(gcrl_her* h);
/* compute_reward for GCRL_REWARD_HOST rings: out[i] = compute_reward(achieved[i], goal[i], {}) for n pairs of goal_dim floats
 * (row-major), in the reference's call order (src/buffer.py:151-166: step-major, relabel-minor).  Returns 0, or non-zero to
 * abort the flush (the push call then fails with GCRL_ERR_STATE).  Called on the pushing thread, inside the push call. */
typedef int (*gcrl_reward_fn)(const float* achieved, const float* goal, int n, int goal_dim, float* out, void* user);
int gcrl_her_set_reward_callback(gcrl_her* h, gcrl_reward_fn fn, void* user);
int64_t gcrl_her_len(const gcrl_her* h);       /* HERBuffer.__len__  src/buffer.py:137 */
int64_t gcrl_her_head(const gcrl_her* h);      /* physical row of logical index 0 */
int32_t gcrl_her_staged(const gcrl_her* h, int env);
void* gcrl_her_stream(const gcrl_her* h);

/* HERBuffer.push (src/buffer.py:110-119).  state/next_state: S floats, either device
 * pointers (the reference passes device tensors, src/env.py:217-219) or host pointers
 * (flag *_on_device).  action (A), desired_goal/achieved_goal (G): host floats.
 * Stages the transition; when `done` or the env has flush_len staged transitions, runs the
 * HER relabel+flush kernel (apply_her) and clears the staging area.
 * Returns the number of ring rows appended (0 = staged only) or a negative status. */
int64_t gcrl_her_push(gcrl_her* h, int env, const float* state, int state_on_device,
                      const float* action_host, const float* next_state,
                      int next_state_on_device, float reward, int done,
                      const float* desired_goal_host, const float* achieved_goal_host,
                      void* stream);

/* ReplayBuffer.push / PERBuffer.push (src/buffer.py:13-14, :46-48): ONE row appended at once (no staging, no relabel,
 * visible to the next sample), deque(maxlen) eviction.  state / next_state: S floats on the device or the host. */
int64_t gcrl_her_append(gcrl_her* h, const float* state, int state_on_device, const float* action_host, float reward,
                        const float* next_state, int next_state_on_device, int done, void* stream);

/* One step of a vector env (the loop of src/env.py:192-201 as ONE call): transition i belongs to
 * env env0+i.  states / next_states: device matrices [n][ld] (what _process_step builds,
 * src/env.py:189-190); actions [n][A], rewards [n], dones [n] (0/1 bytes), achieved goals [n][G]:
 * host.  One payload upload + one staging launch; all envs that finish on this step are relabelled
 * and flushed together (a segment scan over their row counts places each episode), in env order
 * with the RNG draws in env order — identical rows and RNG state to n gcrl_her_push calls.
 * Returns the ring rows appended. */
int64_t gcrl_her_push_batch(gcrl_her* h, int env0, int n, const float* states_dev, int ld_s,
                            const float* actions_host, const float* next_states_dev, int ld_ns,
                            const float* rewards_host, const uint8_t* dones_host,
                            const float* achieved_goals_host, void* stream);

/* Whole-episode variant (one H2D copy + one flush launch): T transitions of env `env` given as
 * host arrays s[T][S], a[T][A], ns[T][S], r[T], d[T] (0/1), ag[T][G].  Equivalent to T calls
 * of gcrl_her_push whose last one triggers the flush; any partially staged episode of that env
 * must be empty.  future_idx (k_future*(T-1) entries, draw order of src/buffer.py:146-153) may
 * be NULL: then drawn from the handle's RNG. */
int64_t gcrl_her_push_episode(gcrl_her* h, int env, int T, const float* s_host,
                              const float* a_host, const float* ns_host, const float* r_host,
                              const float* d_host, const float* ag_host,
                              const uint8_t* future_idx_host, void* stream);

/* HERBuffer.sample (src/buffer.py:121-135), M batches per launch.  Logical indices
 * (0 = oldest row) are drawn by random.sample semantics from the handle's RNG, or taken from
 * idx_host (M*B entries) when non-NULL.  Outputs are device pointers; row strides ld_* in
 * floats (ld_s >= S etc.; rewards/dones are [M*B] contiguous = the reference's [B,1]).
 * drawn_idx_host (optional, M*B) receives the indices used. */
int gcrl_her_sample(gcrl_her* h, int B, int M, const uint32_t* idx_host,
                    float* out_s_dev, int ld_s, float* out_a_dev, int ld_a, float* out_r_dev,
                    float* out_ns_dev, int ld_ns, float* out_d_dev,
                    uint32_t* drawn_idx_host, void* stream);

/* Measurement: when enabled, every gather launch of this ring (gcrl_her_sample and the update
 * engine's batch gather) is bracketed by two hipEvents on the stream it is launched on.
 * gcrl_her_profile_read waits for the pending events and returns, since enabling, for the LARGEST launch size seen (a
 * trainer cycle's main gather; the one-batch head launch that lets step 0 start early is not counted), the number
 * of gather launches, their summed hipEvent time (ms, includes the event/dispatch overhead of a
 * bracketed launch: ~3-4 us more than rocprofv3's kernel duration) and the rows they gathered.  device_clock_ms_out is
 * always 0 now: rounds 1-2 stamped the device wall clock in the first / last blocks, which excludes dispatch and drain and
 * read ~half of the profiler's duration — bench.py takes the kernel's duration from rocprofv3 itself instead. */
int gcrl_her_profile_enable(gcrl_her* h, int on);
int gcrl_her_profile_read(gcrl_her* h, int64_t* launches_out, double* total_ms_out,
                          int64_t* rows_out, double* device_clock_ms_out);

/* Full resume state of the ring (extension of SURVEY.md §8f-2: the reference checkpoints weights + normalisers only,
 * src/env.py:430-440): every stored row in logical order, the per-env staged partial episodes, the counters of the
 * device-RNG mode.  The CPython-MT stream is saved with gcrl_mt_get_state.  Load into a ring of the same shape (any
 * capacity >= the saved row count); logical indices — what sample() draws — are preserved. */
int64_t gcrl_her_state_size(const gcrl_her* h);
int gcrl_her_save_state(gcrl_her* h, void* dst_host, int64_t nbytes);
int gcrl_her_load_state(gcrl_her* h, const void* src_host, int64_t nbytes);

/* Test/debug: copy `n` ring rows starting at logical index `first` to host arrays
 * (any may be NULL).  Synchronises the handle's stream. */
int gcrl_her_read_rows(gcrl_her* h, int64_t first, int64_t n, float* s_host, float* a_host,
                       float* ns_host, float* r_host, float* d_host);

/* ------------------------------------------------------------------------------------------
 * Update engine.  Replaces DDPG / TD3Agent / SACAgent / TQCAgent .update() and what it calls
 * (src/agent.py:1378-1404, :281-317, :659-699, :1062-1100) plus the networks of src/model.py.
 * ---------------------------------------------------------------------------------------- */
typedef struct gcrl_agent gcrl_agent;

enum { GCRL_AGENT_DDPG = 0, GCRL_AGENT_TD3 = 1, GCRL_AGENT_SAC = 2, GCRL_AGENT_TQC = 3 };

typedef struct gcrl_agent_config {
  int32_t kind;           /* GCRL_AGENT_* */
  int32_t obs_dim;        /* S = obs + goal, the `obs_dim` ctor argument (src/env.py:120-127) */
  int32_t ac_dim;         /* A */
  int32_t hidden_dim;     /* BaseAgentConfig.hidden_dim (src/utils.py:11) */
  int32_t layer_count;    /* BaseAgentConfig.layer_count */
  int32_t batch_size;
  int32_t num_critics;    /* TQC: 5 (src/agent.py:789); DDPG 1; TD3/SAC 2 */
  int32_t top_drop;       /* TQC: 2 (src/agent.py:790) */
  int32_t ac_update_freq;
  int32_t gradient_step;  /* ctor argument; SAC Polyak cadence (src/agent.py:681) */
  int32_t polyak_every;   /* DDPG: literal 40 (src/agent.py:1397) */
  /* hyper-parameters are Python floats (doubles) in the reference's pydantic config */
  double gamma, tau;
  double grad_clip;       /* < 0: no clipping (grad_clip None) */
  double policy_noise, noise_clamp; /* TD3 target smoothing (src/agent.py:174-179) */
  double actor_lr, actor_lr_min, critic_lr, critic_lr_min, alpha_lr;
  int64_t ac_scheduler_steps, cr_scheduler_steps;
  double alpha_min_steps; /* SACAgentConfig.alpha_min_steps is a float (src/utils.py:39) */
  int32_t device;
  int32_t use_graph;      /* 0: plain launches; 1: replay the step as a hipGraph where that pays (not on the
                             3-7-launch row-block path, measured faster issued one by one); 2: always */
  uint64_t seed;          /* device RNG for TD3 noise / SAC eps when not injected */
  int32_t pipeline_steps; /* DDPG: 0 = one launch per layer and phase after phase; 1 = co-schedule the
                             actor phase of step i with the critic phase of step i+1 in
                             gcrl_agent_update_n (they are independent: the critic phase reads the
                             TARGET actor; same arithmetic, fewer launches); 2 = additionally run
                             each phase's forward / input-gradient chain as ONE row-block launch
                             (csrc/rowchain.h; needs hidden_dim % 4 == 0, else falls back to 1) */
  int32_t n_quantiles;    /* TQC only; > 1 selects the DISTRIBUTIONAL variant of BASELINE.json configs[3] ("25 quantiles x 2
                             critics"): every critic outputs n_quantiles atoms, the num_critics*n_quantiles (<= 64) pooled target
                             atoms are sorted in one wavefront, the top top_drop PER CRITIC dropped, quantile-Huber loss.  NOT the
                             reference's TQC (scalar-critic ensemble): no reference parity, pinned to its oracle restatement.
                             0 / 1: the reference's semantics */
} gcrl_agent_config;

gcrl_agent* gcrl_agent_create(const gcrl_agent_config* cfg);

/* Measurement (bench.py's roofline leg): while enabled, gcrl_agent_update_n issues the row-block
 * DDPG step (pipeline_steps = 2) as plain launches instead of graph replays and brackets every
 * overlapped row-block launch (actor phase of step i || critic phase of step i+1) with two hipEvents
 * on its stream and a device wall-clock stamp pair.  _read waits for the pending events and returns
 * the launches since enabling, their summed hipEvent time (ms, includes the dispatch overhead of a
 * bracketed launch) and their summed in-kernel time (last block end - first block start, the
 * duration rocprofv3 reports).  Results are unchanged; only the issue path differs. */
int gcrl_agent_profile_enable(gcrl_agent* a, int on);
int gcrl_agent_profile_read(gcrl_agent* a, int64_t* launches_out, double* total_ms_out,
                            double* device_clock_ms_out);
void gcrl_agent_destroy(gcrl_agent* a);
void* gcrl_agent_stream(const gcrl_agent* a);

/* Networks are addressed by name: "actor", "target_actor", "critic_<i>", "target_critic_<i>"
 * (i from 0), "log_alpha".  A network's parameters are ONE flat fp32 vector in
 * torch `module.parameters()` order (weight [out,in] row-major, then bias, per layer; SAC
 * actor: Linear w,b, BatchNorm w,b per block, then mean_head, log_std_head — src/model.py),
 * so state_dict tensors are contiguous slices of it.  Buffers "bn_running_mean"/
 * "bn_running_var" are separate named vectors [L*H]. */
int64_t gcrl_agent_numel(const gcrl_agent* a, const char* name); /* <0: unknown name */
int gcrl_agent_get(gcrl_agent* a, const char* name, float* dst_host, int64_t n);
int gcrl_agent_set(gcrl_agent* a, const char* name, const float* src_host, int64_t n);
/* gradient of the last update() for "actor" / "critic_<i>" (pre-clip), optimiser moments
 * "adam_m:<net>" / "adam_v:<net>" — through the same get call. */

/* Xavier-uniform weights, bias 0.01 (src/model.py:39-42), targets hard-copied
 * (src/agent.py:1257-1258); device RNG seeded from cfg->seed.  Statistically, not bitwise,
 * equal to torch's initialisation: parity tests load explicit parameters with gcrl_agent_set.
 * Also what reset() does (src/agent.py:1461-1465): Linear layers only; optimiser moments,
 * schedulers and BN statistics are kept. `recreate_alpha` = SAC/TQC reset (src/agent.py:767). */
int gcrl_agent_init_weights(gcrl_agent* a, uint64_t seed, int recreate_alpha);
/* update_target_network(hard_update=True) */
int gcrl_agent_hard_update_targets(gcrl_agent* a);

/* update_target_network(hard_update=False, tau): every target <- tau*net + (1-tau)*target
 * (src/agent.py:1259-1271, :117-132, :487-496, :888-895) */
int gcrl_agent_soft_update_targets(gcrl_agent* a, double tau, void* stream);
/* Full resume state of the agent (extension): parameters and targets, Adam moments, BatchNorm running
 * statistics, alpha, optimiser step counts and scheduler positions, device-RNG counter.  One host blob of
 * gcrl_agent_state_size() bytes; loads only into an agent of the same shape. */
int64_t gcrl_agent_state_size(const gcrl_agent* a);
int gcrl_agent_save_state(gcrl_agent* a, void* dst_host, int64_t nbytes);
int gcrl_agent_load_state(gcrl_agent* a, const void* src_host, int64_t nbytes);

/* Optional injected inputs for one update (parity tests; all device pointers, may be NULL). */
typedef struct gcrl_update_inputs {
  const float* s_dev;   int32_t ld_s;    /* [B, S]  explicit batch instead of sampling */
  const float* a_dev;   int32_t ld_a;
  const float* r_dev;
  const float* ns_dev;  int32_t ld_ns;
  const float* d_dev;
  const float* noise_dev;      /* TD3: randn_like(action) [B, A]   (src/agent.py:175) */
  const float* eps_next_dev;   /* SAC/TQC: rsample eps for actor.sample(next_state) [B, A] */
  const float* eps_cur_dev;    /* SAC/TQC: rsample eps for actor.sample(states)     [B, A] */
  /* prioritised replay (PERBuffer, src/buffer.py:38-89): the caller draws the batch (np.random.choice over the priorities,
   * host) and passes the logical row indices and the importance-sampling weights; the critic losses become
   * (weights * loss).mean() (src/agent.py:193-197, :577-581, :993-997, :1320-1325).  Layer-per-launch schedule only
   * (pipeline_steps = 0).  The per-sample |td| of the step is the named vector "td_abs" (gcrl_agent_get). */
  const uint32_t* idx_host;    /* [B] logical ring indices instead of a random.sample draw (needs `her`) */
  const float* weights_host;   /* [B] */
} gcrl_update_inputs;

/* One agent.update(step).  Batch: sampled from `her` (HERBuffer.sample semantics) unless
 * inputs->s_dev is given.  Asynchronous: returns after enqueueing.  Returns the length of the
 * reference's return tuple for this step (DDPG 6/4, TD3 8/6, SAC & TQC 9/6 — the caller
 * dispatches on it, src/env.py:448-506) or a negative status.  `ticket_out` identifies the
 * metrics slot of this step. */
int gcrl_agent_update(gcrl_agent* a, gcrl_her* her, int64_t step,
                      const gcrl_update_inputs* inputs, int64_t* ticket_out, void* stream);
/* `n` consecutive updates step0..step0+n-1 (the loop of src/env.py:384-385; the buffer is not
 * mutated inside it, so all n batches are drawn and gathered by ONE gather launch).
 * tickets_out[n], tuple_len_out[n] optional. */
int gcrl_agent_update_n(gcrl_agent* a, gcrl_her* her, int64_t step0, int n,
                        int64_t* tickets_out, int32_t* tuple_len_out, void* stream);
/* Metrics of a ticket, in the reference's tuple order, as fp32 (waits for that step only).
 * n = tuple length returned by the update. */
int gcrl_agent_metrics(gcrl_agent* a, int64_t ticket, double* out_host, int n);

/* Data-parallel hooks: the step is split at the two gradient exchanges.
 * phase 0: sample + critic fwd/bwd  -> caller all-reduces grads of "critic_*"
 * phase 1: critic clip/optimiser/Polyak + actor fwd/bwd -> caller all-reduces "actor" grads
 * phase 2: actor clip/optimiser (+alpha, + actor Polyak)
 * gcrl_agent_grad_ptr returns the device pointer + numel of the flat gradient block of the
 * phase (all critics contiguous / actor (+log_alpha grad)). `grad_scale` multiplies the
 * gradients before clipping (1/world after an all-reduce sum). */
int gcrl_agent_update_phase(gcrl_agent* a, gcrl_her* her, int64_t step, int phase,
                            const gcrl_update_inputs* inputs, float grad_scale,
                            int64_t* ticket_out, void* stream);
int gcrl_agent_grad_ptr(gcrl_agent* a, int phase, float** ptr_dev_out, int64_t* numel_out);
/* Data-parallel trainer cycle: plan n steps at once — one index upload and ONE gather launch for
 * all n batches, as gcrl_agent_update_n does — then run step i's phases 0,1,2 in order with the
 * caller's two gradient all-reduces in between; gcrl_agent_dp_end closes the cycle. */
int gcrl_agent_dp_begin(gcrl_agent* a, gcrl_her* her, int64_t step0, int n, float grad_scale,
                        int64_t* tickets_out, int32_t* tuple_len_out, void* stream);
int gcrl_agent_dp_phase(gcrl_agent* a, int i, int phase, void* stream);
int gcrl_agent_dp_end(gcrl_agent* a, void* stream);
/* Engine-scheduled form of the same cycle: after gcrl_agent_dp_begin, call gcrl_agent_dp_run
 * repeatedly.  Each call enqueues the next segment of the cycle and names the gradient block the
 * caller must all-reduce (sum over ranks) before the next call (*reduce_numel_out == 0: nothing to
 * exchange).  Returns 1 while segments remain, 0 when the cycle is complete (it is then closed),
 * negative on error.  The engine picks the schedule: three phases per step in general; for runs
 * of plain DDPG steps the software-pipelined form, where the critic gradients of step i+1 and the
 * actor gradients of step i are adjacent in memory and travel in ONE all-reduce per step. */
int gcrl_agent_dp_run(gcrl_agent* a, float** reduce_ptr_out, int64_t* reduce_numel_out, void* stream);
/* In-engine gradient exchange: an RCCL communicator owned by the library (one process per GPU; RCCL is bound at
 * run time from `rccl_path` — the librccl the host process already uses — or the default soname when NULL).
 * Rank 0 makes the 128-byte id and hands it to the other ranks by any side channel (torch.distributed's store,
 * MPI, a file).  New design: the reference has no distributed code. */
typedef struct gcrl_dp gcrl_dp;
int gcrl_dp_unique_id(uint8_t* id128_out, const char* rccl_path);
gcrl_dp* gcrl_dp_create(int rank, int world, const uint8_t* id128, int device, const char* rccl_path);
void gcrl_dp_destroy(gcrl_dp* d);
int gcrl_dp_world(const gcrl_dp* d);
int gcrl_dp_allreduce_sum(gcrl_dp* d, float* buf_dev, int64_t n, void* stream); /* in place, fp32 */
int gcrl_dp_broadcast(gcrl_dp* d, float* buf_dev, int64_t n, int root, void* stream);
/* Test hook of the replay ring's deque(maxlen) bookkeeping (csrc/ring_book.h; what collections.deque(maxlen=max_len) does for
 * the reference's buffers, src/buffer.py:95, :8-20).  Host-only: for append i of `appends[0..n)` rows to a ring of `capacity`
 * rows (initially `len0` rows, head `head0`) reports the physical row its first row goes to and how many of its rows are
 * overwritten inside the same append; returns the final head / len. */
int gcrl_ringbook_sim(int64_t capacity, int64_t head0, int64_t len0, const int64_t* appends, int n, int64_t* tail_out, int64_t* skip_out,
                      int64_t* head_out, int64_t* len_out);
/* ------------------------------------------------------------------------------------------
 * The gradient exchange INSIDE the engine's launch sequence (csrc/xchg_ipc.hip; round 4).  New design: the reference is
 * single-process, BASELINE.json's north star asks for the all-reduce of actor/critic gradients over xGMI.  One process per GPU of
 * ONE node (world <= 8); every rank's gradient arena is mapped by every peer through HIP IPC handles and one kernel per exchange
 * performs a two-shot all-reduce peer to peer: the owner of a 1024-float chunk (chunk c: rank c mod world) adds the chunk of
 * every rank in RANK ORDER (bitwise identical replicas), forms its sum of squares, and writes both back to every rank in
 * place (into a fine-grained receive buffer of the arena's layout: gcrl_xchg_read) — the optimiser launch reads the reduced
 * gradients and the clip norm's partials, no separate norm launch, and the
 * exchange is a plain kernel node: hipGraph replay, control-advance riders and multi-step graphs stay on.
 *   gcrl_xchg_create    over an arena (a hipMalloc base pointer, 16-byte aligned) cut into `nseg` segments (offset / length in
 *                       floats: the nets); every rank must pass the same layout
 *   gcrl_xchg_handles   this rank's record (GCRL_XCHG_HANDLE_BYTES) for the peers; gather the records of all ranks in rank
 *                       order (torch.distributed's store / all_gather) and pass them to gcrl_xchg_connect on every rank
 *   gcrl_xchg_allreduce in-place sum over the ranks of segments [seg0, seg0 + nseg), stream-ordered; EVERY rank must enqueue
 *                       the same sequence of exchanges.  Waits are bounded (~1 s): a rank that never arrives leaves NaN and
 *                       a status bit that the owner's next synchronising call reports (GCRL_ERR_STATE)
 *   gcrl_agent_xchg_create / gcrl_agent_set_exchange   the same over an agent's own gradient arena (segments: every critic, the
 *                       actor, log_alpha), and attaching it: from then on EVERY update entry point (gcrl_agent_update,
 *                       _update_n, _update_phase) exchanges the critic gradients after the critic backward and the actor
 *                       (+ log_alpha) gradients after the actor backward — one exchange per overlapped DDPG step — and scales
 *                       by 1 / world inside the optimiser launches.  Reference semantics kept: the actor loss goes through the
 *                       STEPPED critic (src/agent.py:1389-1401, :548-639).  NULL detaches.
 * xGMI time is unmeasured (1-GPU boxes); world-size-1 cost: profiles/r04_dp_overhead_world1.json. */
#define GCRL_XCHG_HANDLE_BYTES 256
typedef struct gcrl_xchg gcrl_xchg;
gcrl_xchg* gcrl_xchg_create(float* arena_dev, int64_t arena_floats, const int64_t* seg_off, const int64_t* seg_n, int nseg, int rank,
                            int world, int device);
void gcrl_xchg_destroy(gcrl_xchg* x);
int gcrl_xchg_handles(gcrl_xchg* x, uint8_t* out, int64_t n);
int gcrl_xchg_connect(gcrl_xchg* x, const uint8_t* all_ranks_records, int64_t n);
int gcrl_xchg_world(const gcrl_xchg* x);
int gcrl_xchg_allreduce(gcrl_xchg* x, int seg0, int nseg, void* stream);
/* sums of squares of segment `seg`'s reduced 1024-float chunks as the last exchange left them (what the optimiser launch sums
 * for the clip norm); returns how many (synchronises the device; tests) */
int gcrl_xchg_get_partials(gcrl_xchg* x, int seg, float* out_host, int n);
/* the reduced values of [first, first + n) arena floats as the last exchange left them, to the host (synchronises; tests) */
int gcrl_xchg_read(gcrl_xchg* x, int64_t first, int64_t n, float* out_host);
/* collective self-test after gcrl_xchg_connect (every rank calls it): exchanges a known pattern through segment 0 and checks
 * the sum; GCRL_ERR_STATE when a peer's mapping, layout or arrival is wrong — the host side then falls back to another exchange
 * before training starts (src/dp.py).  Leaves the arena as it found it. */
int gcrl_xchg_selftest(gcrl_xchg* x, void* stream);
int gcrl_xchg_reset(gcrl_xchg* x);   /* counters back to zero after a reported failure (call on every rank, ranks synchronised around it) */
gcrl_xchg* gcrl_agent_xchg_create(gcrl_agent* a, int rank, int world);
int gcrl_agent_set_exchange(gcrl_agent* a, gcrl_xchg* x);
/* The whole data-parallel trainer cycle begun by gcrl_agent_dp_begin as ONE host call: every segment of the
 * engine's schedule and, after each, the all-reduce(sum) of the gradient block it names, all enqueued on `stream`
 * (what the caller of gcrl_agent_dp_run does one Python round trip at a time). */
int gcrl_agent_dp_run_all(gcrl_agent* a, gcrl_dp* d, void* stream);
/* SyncBN for the data-parallel BatchNorm actors (SACAgent / TQCAgent; new design — torch's SyncBatchNorm is what a DDP port
 * of src/model.py:100-108 would use).  After this call the batch statistics of every BatchNorm1d forward and backward are
 * those of the CONCATENATED batch of all `world` ranks (rank r's rows = rows [r*B, (r+1)*B) of it), so that G ranks x B rows
 * equal 1 rank x G*B rows for these agents as well: per BatchNorm layer and pass the ranks' row-block partials
 * (2 * world * ceil(B/64) * H floats) are summed over the ranks — by the library-owned communicator `dp` when given, else by
 * `fn` (an in-place all-reduce(sum) of n device floats, stream-ordered on `stream` or synchronous; returns 0) — and merged in
 * the order a single process with the big batch uses.  dgamma / dbeta stay each rank's share (the gradient exchange sums
 * them).  Steps then run as plain launches (no hipGraph replay).  Call it before the first update; world = 1 switches it off. */
typedef int (*gcrl_exchange_fn)(float* buf_dev, int64_t n, void* stream, void* user);
int gcrl_agent_dp_sync_bn(gcrl_agent* a, int world, int rank, gcrl_dp* dp, gcrl_exchange_fn fn, void* user);
/* The same with the partials exchanged by the in-engine peer-to-peer kernel (round 5): gcrl_agent_bn_xchg_create allocates the
 * partials' arena and returns an exchange handle over it (connect it like the gradient exchange: gcrl_xchg_handles / _connect /
 * _selftest; the caller owns it and destroys it after switching SyncBN off); gcrl_agent_dp_sync_bn_xchg switches SyncBN on
 * through it.  The exchange is then a kernel of the step's launch sequence: hipGraph replay and multi-step graphs stay on.
 * New design (the reference is single-process, src/model.py:107 BatchNorm1d on one batch). */
gcrl_xchg* gcrl_agent_bn_xchg_create(gcrl_agent* a, int rank, int world);
int gcrl_agent_dp_sync_bn_xchg(gcrl_agent* a, int world, int rank, gcrl_xchg* bx);
/* Launches whose workgroups WAIT for each other inside the kernel (csrc/meet.h: the row groups of a BatchNorm slab, the role
 * workgroups of a twin-critic row block) are admitted only when every workgroup of the launch is resident at once — judged by
 * the kernel's occupancy on a device this process has to itself.  New design (the reference is eager PyTorch: no such forms).
 *   gcrl_set_shared_device(1)   process-wide (also GCRL_SHARED_GPU=1 in the environment, read at first use): the device is
 *                               shared with other processes / streams that hold CUs (several ranks on one GPU, another
 *                               library's collectives) — handles created afterwards use the launch forms without waits;
 *   gcrl_agent_set_meetings     switches an existing handle (0: off; 1: on where admissible); captured graphs are dropped;
 *                               returns a bit mask of the forms now active (1 slab row groups, 2 row-chain roles: merged phases,
 *                               DDPG's two-role critic phase; 4 the opt-in weight-slice DDPG launch, GCRL_ROWTILE=1; 8 the
 *                               fused dW | db + clip + optimiser launch of the row-chain agents, csrc/dw_adam.hip — the reference's
 *                               backward -> clip_grad_norm_ -> optimizer.step(), src/agent.py:1326-1333, :1288-1300, :199-222);
 *   gcrl_agent_get_meetings     the same mask, changing nothing;
 *   a wait that times out (~1 s) poisons that launch's statistics / gradients with NaN AND is reported: the next call that
 *   synchronises the handle (gcrl_agent_metrics, _get, _save_state) returns GCRL_ERR_STATE once, after which the handle works
 *   again — the reference raises on any failed step (src/agent.py:659-699);
 *   gcrl_agent_debug_meet_fault injects such a failure into the handle's next launch (tests). */
int gcrl_set_shared_device(int shared);
int gcrl_agent_set_meetings(gcrl_agent* a, int on);
int gcrl_agent_get_meetings(gcrl_agent* a);
int gcrl_agent_debug_meet_fault(gcrl_agent* a);
/* device pointer of a named vector (parameters / grads), for zero-copy interop */
int gcrl_agent_dev_ptr(gcrl_agent* a, const char* name, float** ptr_dev_out, int64_t* numel_out);

/* Batched actor inference for select_action (src/agent.py:1345-1366, :641-647): obs_dev
 * [n, S] -> out_dev [n, A]: DDPG/TD3 the network output (already tanh, src/model.py:24);
 * SAC/TQC tanh(mean) in eval mode (BatchNorm running statistics) when eps_dev is NULL,
 * else tanh(mean + std*eps). */
int gcrl_agent_act(gcrl_agent* a, const float* obs_dev, int n, int ld_obs, float* out_dev,
                   int ld_out, const float* eps_dev, void* stream);
/* The same from and to HOST arrays (n <= batch_size rows): staging, both copies and the stream
 * synchronisation happen inside; what select_action needs per vector-env step in one call. */
int gcrl_agent_act_host(gcrl_agent* a, const float* obs_host, int n, int ld_obs, float* out_host,
                        int ld_out, void* stream);

/* ------------------------------------------------------------------------------------------
 * The acting side: RunningNormalizer on the device (src/utils.py:68-98: float32 batch moments in
 * numpy's order, float64 parallel-variance merge, clip to +-clip_range) and the two fused entry points of one
 * vector-env step.  Rows are float32, `ld` floats apart; `on_device` says where they live.
 * ---------------------------------------------------------------------------------------- */
typedef struct gcrl_normalizer gcrl_normalizer;
gcrl_normalizer* gcrl_normalizer_create(int size, double clip_range, double eps, int device);   /* __init__ :69-73 */
void gcrl_normalizer_destroy(gcrl_normalizer* z);
int gcrl_normalizer_size(const gcrl_normalizer* z);
int gcrl_normalizer_update(gcrl_normalizer* z, const float* x, int n, int ld, int on_device, void* stream);   /* update :75-81 */
/* normalize :95-97, result cast to float32 (what the trainer's torch.from_numpy(..).float() does, src/env.py:189-190) */
int gcrl_normalizer_normalize(gcrl_normalizer* z, const float* x, int n, int ld, int on_device, float* out, int ld_out,
                              int out_on_device, void* stream);
int gcrl_normalizer_get(gcrl_normalizer* z, double* mean_host, double* var_host, double* count_host);   /* save :99-108 */
int gcrl_normalizer_set(gcrl_normalizer* z, const double* mean_host, const double* var_host, double count, double clip_range);
/* The reference's `RunningNormalizer.load` (src/utils.py:108-117) brings mean / var back as FLOAT32 arrays, and everything it
 * computes afterwards — normalize and the merge of update — then runs in float32 (only the count stays a Python float).  `on`
 * != 0 puts the handle in that regime (the statistics set before should be float32 values); pinned by
 * tests/golden/normalizer_loaded.npz.  A created normaliser starts in the float64 regime. */
int gcrl_normalizer_set_float32(gcrl_normalizer* z, int on);
int gcrl_normalizer_is_float32(const gcrl_normalizer* z);
/* The dtype of the rows the REFERENCE's trainer would pass decides numpy's arithmetic.  Observation batches are float64
 * arrays there (the vector env allocates them with the observation space's dtype, which TimeFeatureWrapper declares float64,
 * src/utils.py:156; the values inside are float32-valued, so the float32 rows of this ABI carry them exactly), goal batches
 * float32.  `on` != 0: treat the rows of every later update / normalize / fused entry as float64 — batch moments, merge,
 * subtraction and division in float64 (`self.var * self.count` and sqrt(var) + 1e-8 of a LOADED normaliser stay float32 values),
 * and the first update after a load turns the statistics back into float64 (gcrl_normalizer_is_float32 then reports 0).
 * Pinned by tests/golden/normalizer_f64.npz.  Default off (float32 rows: normalizer.npz, normalizer_loaded.npz). */
int gcrl_normalizer_set_rows_float64(gcrl_normalizer* z, int on);
/* select_action for one vector-env step from RAW host rows (src/env.py:348-355 + src/agent.py:1345-1366 / :253-270 /
 * :641-647): normalize_state_batch with the given normalisers (NULL: that part raw), actor, post-processing —
 * mode 0: clip(tanh(net), -1, 1); 1: clip(tanh(net) + noise, -1, 1), noise = np.random.normal draws [n, A] float64;
 * 2: the network output as it is.  SAC / TQC ignore `mode`: noise = rsample eps (NULL: deterministic tanh(mean)).
 * out_host [n, A] float64.  DDPG's epsilon-random branch (:1348) stays with the caller (shared `random` stream). */
int gcrl_agent_observe_act(gcrl_agent* a, gcrl_normalizer* nz_obs, gcrl_normalizer* nz_dg, const float* obs_host, int obs_dim,
                           const float* dg_host, int goal_dim, int n, const double* noise_host, int mode, double* out_host,
                           void* stream);
/* _process_step for one vector-env step (src/env.py:163-201) from raw host rows: normaliser update with [obs ; next_obs]
 * (when update_stats), state = [normalize(obs) | dg], next_state = [normalize(next_obs) | next_dg] built on the device
 * from the UPDATED statistics, then the n pushes of gcrl_her_push_batch (achieved goal = next_ag, done = dones).
 * Goals are not normalised here; gcrl_her_process_step_g adds the goal normaliser (g_normalize = True, src/env.py:167-175,
 * :222-223): its update from [dg ; next_dg ; ag ; next_ag] (ag_host = the state's achieved goals, only for these statistics), then
 * the goal columns of both states and the pushed achieved goal normalised by the UPDATED goal statistics.  Returns ring rows appended. */
int64_t gcrl_her_process_step(gcrl_her* h, gcrl_normalizer* nz_obs, int update_stats, const float* obs_host,
                              const float* next_obs_host, int obs_dim, const float* dg_host, const float* next_dg_host,
                              const float* next_ag_host, const float* actions_host, const float* rewards_host,
                              const uint8_t* dones_host, int env0, int n, void* stream);
int64_t gcrl_her_process_step_g(gcrl_her* h, gcrl_normalizer* nz_obs, int update_stats, gcrl_normalizer* nz_dg, int update_goal_stats,
                                const float* obs_host, const float* next_obs_host, int obs_dim, const float* dg_host,
                                const float* next_dg_host, const float* ag_host, const float* next_ag_host, const float* actions_host,
                                const float* rewards_host, const uint8_t* dones_host, int env0, int n, void* stream);

/* ------------------------------------------------------------------------------------------
 * Stand-alone ops exposed for tests / reuse.
 * ---------------------------------------------------------------------------------------- */
/* Row-wise sort of `width` (<= 64) fp32 values per row in one wavefront (bitonic network via
 * cross-lane swaps), drop the `drop` largest, mean of the rest: the truncation of
 * src/agent.py:919-921 / :972-974 generalised to BASELINE.json's 25x2-atom shape.
 * in_dev [rows, width] -> sorted_dev [rows, width] (optional) , mean_dev [rows]. */
int gcrl_sort_truncate_mean(const float* in_dev, int64_t rows, int width, int drop,
                            float* sorted_dev, float* mean_dev, void* stream);

/* The batched fp32-MFMA GEMM every Linear forward/backward of the engine runs on, as a single
 * problem: C[M,N] = act(A.B + bias) with element strides (A(m,k) = A[m*a_rs + k*a_cs],
 * B(k,n) = B[k*b_rs + n*b_cs], C row stride c_rs); act: 0 none, 1 LeakyReLU(0.01), 2 ReLU,
 * 3 tanh.  shape: 0 auto, 1 = 16x16 tile per workgroup with K split over its 4 waves,
 * 2 = 16x16 per wave, 3 = 32x32 per wave, 4 = LDS-tiled 64x64 per workgroup, 5 = K == 1 only: the outer product as a
 * streaming kernel (16-byte aligned B / C / bias, N and c_rs multiples of 4, b_cs == 1; else GCRL_ERR_ARG) — tests sweep all. */
int gcrl_gemm_f32(const float* a_dev, int64_t a_rs, int64_t a_cs, const float* b_dev, int64_t b_rs,
                  int64_t b_cs, float* c_dev, int64_t c_rs, const float* bias_dev, int M, int N,
                  int K, int act, int shape, void* stream);

/* The weight / bias gradient of a Linear layer as the engine computes it at batch >= 1024 (csrc/gemm_tiled.h): dW [out, in] =
 * G^T X and db [out] = column sums of G, for G [batch, ldg] (first `out` columns) and X [batch, ldx] (first `in` columns), on
 * the LDS-tiled form with the reduction over the batch split over `ksplit` workgroups per 64x64 tile (1: no split) — partial
 * tiles through a scratch array, a ticket per tile, the last arriver sums them in index order (deterministic).  Stand-alone
 * for tests: allocates and frees its scratch per call (the engine keeps it per layer).  `sumsq_dev`, if not null, receives the
 * sum of squares of dW | db (one float), as the fused global-norm clip consumes it.  Reference: the backward of
 * `nn.Linear` (src/model.py:18,57,63). */
int gcrl_gemm_dw_split_f32(const float* g_dev, int64_t ldg, const float* x_dev, int64_t ldx, float* dw_dev, float* db_dev,
                           int out, int in, int batch, int ksplit, float* sumsq_dev, void* stream);

/* nn.BatchNorm1d in training mode followed by ReLU, as the SAC / TQC actors run it per hidden layer
 * (src/model.py:106-108: Linear -> BatchNorm1d -> ReLU; eps 1e-5, momentum 0.1, running variance unbiased), forward and
 * backward as single problems.  All pointers are device memory, row-major [B, H]; H must be a multiple of 4 and every
 * pointer 16-byte aligned.  `scratch_dev`: 2 * ceil(B/64) * H floats.
 *   fwd: z -> h = relu(gamma * xhat + beta); saves xhat [B,H] and invstd [H]; updates running_mean / running_var in place.
 *   bwd: dh (gradient w.r.t. h), xhat / invstd from the forward -> dz [B,H], dgamma [H], dbeta [H]. */
int gcrl_bn_relu_fwd_f32(const float* z_dev, int B, int H, const float* gamma_dev, const float* beta_dev, float* h_dev,
                         float* xhat_dev, float* invstd_dev, float* running_mean_dev, float* running_var_dev,
                         float* scratch_dev, void* stream);
int gcrl_bn_relu_bwd_f32(const float* dh_dev, const float* xhat_dev, const float* invstd_dev, const float* gamma_dev,
                         const float* beta_dev, int B, int H, float* dz_dev, float* dgamma_dev, float* dbeta_dev,
                         float* scratch_dev, void* stream);

/* [Linear -> BatchNorm1d(train) -> ReLU] (src/model.py:104-108) as ONE launch per direction (csrc/bn_slab.hip: a workgroup
 * owns 16 columns over all rows, so the batch statistics never leave it): what the SAC / TQC actors run per hidden layer at
 * B <= 512, H % 16 == 0 (anything else: GCRL_ERR_ARG; the engine then uses the GEMM + BatchNorm launches above).
 *   fwd: x [B, ldx] (first K columns), w [H, K], bias / gamma / beta [H] -> h = relu(bn(x w^T + bias)) [B, H]; xhat [B, H] and
 *        invstd [H] (either may be null); bstat [2][H] = the batch mean and the biased batch variance (the running statistics
 *        are updated from these by the step's tanh-Gaussian launch).
 *   bwd: dh = g_up [B, ldg] (first K_up columns) . w_up [K_up, H]  — the input gradient of the consuming Linear layer —
 *        -> dz written over xhat_dz [B, H] (xhat on entry), dgamma [H], dbeta [H].  ldg % 4 == 0, g_up 16-byte aligned.
 *   row_split > 1 (and B > 128): the rows of a slab split over ceil(B / 128) workgroups that exchange their column partials once
 *        per launch through agent-scope memory and a bounded wait (what the engine uses for its K >= 128 layers); <= 1: one
 *        workgroup per slab holds every row. */
int gcrl_bn_linear_slab_fwd_f32(const float* x_dev, int64_t ldx, const float* w_dev, const float* bias_dev,
                                const float* gamma_dev, const float* beta_dev, int B, int H, int K, float* h_dev,
                                float* xhat_dev, float* invstd_dev, float* bstat_dev, int row_split, void* stream);
int gcrl_bn_linear_slab_bwd_f32(const float* g_up_dev, int64_t ldg, int K_up, const float* w_up_dev, float* xhat_dz_dev,
                                const float* invstd_dev, const float* gamma_dev, const float* beta_dev, int B, int H,
                                float* dgamma_dev, float* dbeta_dev, int row_split, void* stream);

/* The device-RNG mode's Gaussian: out[i] = hash_normal(seed, ctr0 + i), the counter-hash Box-Muller normal that stands in
 * for torch.randn_like (src/agent.py:175, TD3 target smoothing) and Normal.rsample's eps (src/model.py:134) whenever an
 * update is not given injected noise.  Not torch's stream; restated in oracle/device_rng_oracle.py and tested against it. */
int gcrl_hash_normal_fill(uint64_t seed, uint64_t ctr0, int64_t n, float* out_dev, void* stream);

/* hipEvent helpers so a Python caller can time the engine's own stream (torch.cuda.Event only
 * sees torch's current stream). */
void* gcrl_event_create(void);
void gcrl_event_destroy(void* ev);
int gcrl_event_record(void* ev, void* stream);
int gcrl_event_elapsed_ms(void* ev_start, void* ev_stop, float* ms_out); /* syncs on stop */
int gcrl_stream_synchronize(void* stream);
/* plain device memory for callers without torch */
void* gcrl_malloc(size_t bytes);
void gcrl_free(void* p);
int gcrl_memcpy_h2d(void* dst_dev, const void* src_host, size_t bytes, void* stream);
int gcrl_memcpy_d2h(void* dst_host, const void* src_dev, size_t bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GCRL_H */
