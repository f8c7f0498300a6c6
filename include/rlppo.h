/*
 * rlppo.h -- C ABI of librlppo.so, the MI355X (gfx950) hot path of rlgym-ppo.
 *
 * The reference (AechPro/rlgym-ppo v1.3.13) is pure Python and has no FFI; its boundary for this path is
 * the Python class API.  The host-side mirror of that API lives in
 * rlgym_ppo_amd/ and calls the entry points below through ctypes.  Every entry point names the reference
 * code it replaces (paths relative to the reference tree).
 *
 * Conventions
 *   - extern "C", plain pointers and sizes only.  All `const float*` / `float*` / `int64_t*` data arguments
 *     are DEVICE pointers borrowed from the caller (torch tensors' data_ptr()), except where a parameter is
 *     explicitly marked HOST.  The library never allocates or frees device memory: scratch space is a
 *     caller-provided workspace sized by the matching *_workspace_bytes() query.
 *   - `stream` is a hipStream_t passed as void*.  All device work is enqueued asynchronously on it; no entry
 *     point synchronises the device, so every call is legal inside a hipGraph capture.
 *   - Return value: 0 = OK; non-zero = error (RLPPO_ERR_* or a hipError_t).  rlppo_last_error() returns a
 *     thread-local description of the most recent failure.  No C++ exception crosses the ABI.
 *   - A network is described by `dims[0..n_layers]` = {d_in, h_1, ..., h_L, d_out} (logical sizes, exactly the
 *     nn.Linear sizes the reference builds: discrete_policy.py:22-31, value_estimator.py:19-28) and by its
 *     parameters in `torch.nn.utils.parameters_to_vector` order: W_0[out][in], b_0, W_1, b_1, ...
 *     ("flat arena").  Kernels consume a PACKED copy (tile-padded W, W^T and b) built by rlppo_net_pack().
 *   - Row-major everywhere.  Observation matrices have a leading dimension `ld` (floats) >= d_in; rows that the
 *     kernels read directly must have ld a multiple of 32 with zero fill beyond d_in (rlppo_padded_width()).
 */
#ifndef RLPPO_H
#define RLPPO_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* The export table: the library is built with -fvisibility=hidden, and exactly the functions declared between this push and
 * the matching pop have default visibility (`nm -D` shows them and nothing else of the library's own; tests/test_abi_and_layout.py). */
#pragma GCC visibility push(default)

#define RLPPO_ABI_VERSION 6
#define RLPPO_MAX_LAYERS 16

#define RLPPO_OK 0
#define RLPPO_ERR_ARG 1001        /* bad argument / unsupported shape */
#define RLPPO_ERR_WORKSPACE 1002  /* workspace too small */
#define RLPPO_ERR_COLLECT_TIMEOUT 1003  /* rlppo_collector_collect: no worker message for a minute */
#define RLPPO_ERR_INTERRUPTED 1004     /* rlppo_collector_collect: a signal arrived; n_collected holds the progress, call again with resume = 1 */

/* policy head codes == the reference's `policy_type` (ppo_learner.py:34-50) */
#define RLPPO_HEAD_DISCRETE 0
#define RLPPO_HEAD_MULTIDISCRETE 1
#define RLPPO_HEAD_GAUSSIAN 2

/* precision of a call (rlppo_act_opts.precision, rlppo_minibatch_args.precision): 0 = the process default set with
 * rlppo_set_inference_precision / rlppo_set_update_precision, else 1 + that setter's mode */
#define RLPPO_PRECISION_DEFAULT 0
#define RLPPO_PRECISION_FP32 1
#define RLPPO_PRECISION_BF16 2
#define RLPPO_PRECISION_X3 3   /* update only: fp32 with split-bf16 hidden products */

int rlppo_abi_version(void);
const char *rlppo_last_error(void);
/* [r6] 16 hex digits: SHA-256 over the sources this library was built from (csrc/Makefile).  Evidence replayed from profiles/
 * (bench.py's `roofline.traffic`: PMC bytes per launch, measured by a builder-run rocprofv3 pass) is stamped with the id of the
 * library it was measured on and reported only while that is the library in use. */
const char *rlppo_build_id(void);

/* ---------------------------------------------------------------------------------------------- layout */

/* Width (floats) to which an input feature dimension is padded for the kernels (multiple of 32). */
int64_t rlppo_padded_width(int64_t d);
/* Width to which a layer OUTPUT dimension is padded (32, 64, 96, 128 or a multiple of 128). */
int64_t rlppo_padded_out(int64_t d);
/* Number of floats in the packed copy of a network. */
int64_t rlppo_packed_floats(const int32_t *dims, int32_t n_layers);
/* Number of floats in the flat arena of a network (sum of out*in + out). */
int64_t rlppo_flat_floats(const int32_t *dims, int32_t n_layers);

/* Build the packed copy (padded W, W^T, b per layer) from the flat arena.  Called after every optimiser step.
 * Replaces nothing in the reference (it is the price of tile-aligned kernels). */
int rlppo_net_pack(void *stream, const int32_t *dims, int32_t n_layers, const float *flat, float *packed);

/* Copy/convert observation rows into the padded device layout: dst[r][0:d] = (float)src[r][0:d], zero fill to
 * ld_dst.  `src_is_f64` selects a float64 source (learner.py:347 builds a float64 value-net input).
 * Optional standardisation with the reference's SCALAR statistics (quirk Q5, batched_agent_manager.py:313-315):
 * if standardize != 0, dst = clip((src - mean0) / std0, -5, 5). */
int rlppo_pad_rows(void *stream, const void *src, int32_t src_is_f64, int64_t n, int64_t d, int64_t ld_src,
                   float *dst, int64_t ld_dst, int32_t standardize, float mean0, float std0);
/* The same copy with per-feature statistics (device vectors of d floats): dst = clip((src - mean[c]) / std[c], -5, 5).
 * Not the reference's behaviour (it applies the scalars of feature 0 to every feature, batched_agent_manager.py:230-235,
 * 313-315): the corrected form SURVEY.md section 8(f) row 4 asks for, selected by per_feature_obs_standardization=True. */
int rlppo_pad_rows_per_feature(void *stream, const void *src, int32_t src_is_f64, int64_t n, int64_t d, int64_t ld_src,
                               float *dst, int64_t ld_dst, const float *mean, const float *stdv);

/* ------------------------------------------------------------------------------------ rollout inference */

/* [r5] Per-call options of the rollout entry points below (their last parameter; NULL = all defaults).
 *   precision: the inference precision of THIS call -- RLPPO_PRECISION_DEFAULT (what rlppo_set_inference_precision chose for the
 *     process), RLPPO_PRECISION_FP32 or RLPPO_PRECISION_BF16 (operands rounded to bf16, fp32 accumulate): two policies of one
 *     process may act in different precisions.
 *   done_words / done_value: completion without a stream synchronisation.  done_words (optional) points at rlppo_act_done_words(n)
 *     uint32 in HOST-VISIBLE memory (pinned / page-locked, coherent); word b receives done_value -- stored with release semantics
 *     at system scope -- once every output of rows 16 b .. 16 b + 15 is visible to the host (the one-launch kernel of
 *     rlppo_discrete_act / _step stores it itself, workgroup by workgroup; the layer chains append one tiny launch that stores them
 *     all).  A host whose outputs live in pinned memory clears the words, makes the call and polls them (rlppo_host_wait_words)
 *     instead of hipStreamSynchronize: batched_agent_manager.py:202-204 calls get_action on 8-80 observations per environment
 *     step, where the synchronisation was a third of the call.
 *   noise_ctl (rlppo_discrete_step's one-launch kernel only; needs done_words, done_value < 2^31): the noise arrives WHILE the
 *     kernel runs -- the bit-exact Exp(1) draw of the reference's CPU stream (5-11 us at 8-80 rows) hides behind the launch
 *     latency and the layers.  noise_ctl points at 32 uint32 the host can write and the kernel reads past its caches (pinned
 *     memory, or better a host window, below): [0] the call's sequence (never the previous call's), [1] live rows -- both written
 *     BEFORE the call -- [2] the sequence of the noise that is complete in noise_q, [3..] statistics the kernel writes
 *     ([3] polls / [4] 10 ns ticks the first wave spent on its noise after the last layer; [8..11] 100 MHz time stamps of the last
 *     workgroup: start, observations staged, layers done, sampled).  noise_q
 *     is [n][n_actions] floats in such memory too; the host fills it AFTER the launch and then stores the call's sequence into
 *     word 2 (rlppo_host_push does both in order).  The waves that have nothing to multiply in the head layer look at word 2 when
 *     that layer starts and bring the numbers in; if the host was not done by then every wave waits for the word after the last
 *     layer.  Rows >= live rows are not sampled (their outputs are untouched).  Between the launch and word 2 the host must not
 *     wait for the GPU (the kernel waits for the host).  A workgroup that has waited 20 ms gives up: it stores
 *     done_value | 0x80000000 (rlppo_host_wait_words returns 2), its outputs are void; the caller completes word 2 and makes the
 *     call again. */
typedef struct rlppo_act_opts {
    int32_t precision;
    uint32_t done_value;
    uint32_t *done_words;
    uint32_t *noise_ctl;
} rlppo_act_opts;
int64_t rlppo_act_done_words(int64_t n);
/* HOST: spins until words[0..count) all hold `value` (acquire loads) or timeout_us has passed; 0 = all there, 1 = timed out,
 * 2 = a word holds value | 0x80000000 (the kernel gave up: noise_ctl above). */
int rlppo_host_wait_words(const uint32_t *words, int64_t count, uint32_t value, int64_t timeout_us);
/* [r5] Host window: DEVICE memory (fine-grained, zeroed) that the host writes directly through the PCIe aperture -- a posted write
 * of a small call's observations costs the host 1.5-2.5 us, where the kernel's own reads of pinned host memory cost it 11-20 us
 * (a GPU-initiated PCIe read is a round trip of microseconds and ~30 ns per 64 bytes behind it).  Write-only for the host
 * (its reads are uncached: 65 us per 3 KB).  Fails (RLPPO_ERR_ARG) on a device that does not expose its memory (no large BAR).
 * rlppo_host_push: HOST: memcpy(dst, src, bytes), a store fence, then -- if flag is not NULL -- *flag = value and another fence:
 * nothing that follows (the doorbell of a launch, the flag) can overtake the bytes. */
int rlppo_host_window_alloc(size_t bytes, void **ptr);
int rlppo_host_window_free(void *ptr);
/* HOST: flushes the host data path of the window's device (a store to the HDP flush register HIP maps into the process): every
 * host write that has left the CPU is in device memory on return.  ~1-2.5 us.  Ordering without it: host writes reach the device in
 * program order (posted PCIe writes behind a store fence), so a flag word written after the bytes it announces arrives after them
 * (the late noise), and a launch's doorbell arrives after the bytes staged before it, with the packet fetch and the dispatch (>= 3 us)
 * between the doorbell and the kernel's first read.  ActGraph issues the flush between staging and launch (by the book: 0.9 us of a
 * 39 us call); behind the launch it is free. */
int rlppo_host_window_flush(const void *window);
int rlppo_host_push(void *dst, const void *src, size_t bytes, uint32_t *flag, uint32_t value);
/* HOST: what precedes the launch of a small rollout call, in one call: ctl[0] = sequence, ctl[1] = live_rows (ctl may be NULL),
 * memcpy(obs_dst, obs_src, obs_bytes), one store fence. */
int rlppo_host_stage_call(uint32_t *ctl, uint32_t sequence, uint32_t live_rows, void *obs_dst, const void *obs_src, size_t obs_bytes);
/* ... with the observations as ROWS of a padded image: row r (row_bytes, src_stride apart) goes to dst + r * dst_stride.  A host window
 * that only ever receives the first row_bytes of its rows stays zero beyond them: it IS the zero-padded input the forward entry points
 * read (ld = dst_stride / 4), and the layer chain of a small call needs no rlppo_pad_rows launch. */
int rlppo_host_stage_rows(uint32_t *ctl, uint32_t sequence, uint32_t live_rows, void *dst, size_t dst_stride, const void *src,
                          size_t src_stride, size_t row_bytes, int64_t rows);

/* Bytes of workspace needed by the forward entry points for `n` rows. */
size_t rlppo_forward_workspace_bytes(const int32_t *dims, int32_t n_layers, int64_t n);

/* MLP forward: out[n][ld_out] = Linear_L(relu(...relu(Linear_0(obs)))) (+ tanh if out_tanh).
 * Replaces nn.Sequential.forward of value_estimator.py:30-36 / the body of *_policy.get_output. */
int rlppo_mlp_forward(void *stream, const int32_t *dims, int32_t n_layers, const float *packed,
                      const float *obs, int64_t ld_obs, int64_t n, int32_t out_tanh,
                      float *out, int64_t ld_out, void *workspace, size_t ws_bytes, const rlppo_act_opts *opts);

/* DiscreteFF.get_action (discrete_policy.py:44-62): forward, softmax, clamp(1e-11,1), action =
 * argmax_a(p_a / q_a) with the caller's Exp(1) noise q[n][n_actions] (== torch.multinomial(p,1,True), SURVEY
 * 8(a1); first index wins ties), logp = log(p_action).  actions: int64[n]; logp: float[n];
 * probs_out (optional, may be NULL): float[n][n_actions]. */
int rlppo_discrete_act(void *stream, const int32_t *dims, int32_t n_layers, const float *packed,
                       const float *obs, int64_t ld_obs, int64_t n, const float *noise_q,
                       int64_t *actions, float *logp, float *probs_out, void *workspace, size_t ws_bytes, const rlppo_act_opts *opts);

/* The whole rollout step of the discrete policy [r3] (discrete_policy.py:35-62 behind batched_agent_manager.py:202-204,303-315):
 * raw observations as the environment hands them over ([n][ld_obs] fp32 or fp64, d = dims[0] features) -> standardise
 * (0: no; 1: the reference's scalars of feature 0, clip((x - mean0) / std0, -5, 5); 2: per-feature vectors mean_v / std_v) and
 * zero-pad -> MLP -> softmax -> clamp -> argmax(p / q) -> log p.  One launch when the network has the form csrc/fused_act.hip
 * covers (equal hidden widths of 64 / 128 / 256, <= 128 actions, fp32 inference precision), else rlppo_pad_rows + the chain of
 * rlppo_discrete_act: the same numbers, bit for bit.  Outputs: actions (int64; may be pinned host memory, like logp: the kernel
 * stores straight into it), actions_f32 (optional: the index as float, the experience buffer's encoding), logp, rows_out
 * (optional: the padded, standardised rows [n][ld_rows_out] -- the policy-input rows a device-resident rollout stores).
 * workspace: rlppo_discrete_step_workspace_bytes(dims, n_layers, n). */
size_t rlppo_discrete_step_workspace_bytes(const int32_t *dims, int32_t n_layers, int64_t n);
/* 1: rlppo_discrete_step(dims, n rows, opts) runs as the one-launch kernel under the current switches; 0: as the chain; -1: bad argument */
int rlppo_discrete_step_one_launch(const int32_t *dims, int32_t n_layers, int64_t n, const rlppo_act_opts *opts);
int rlppo_discrete_step(void *stream, const int32_t *dims, int32_t n_layers, const float *packed, const void *obs, int32_t obs_is_f64,
                        int64_t ld_obs, int64_t n, int32_t standardize, float mean0, float std0, const float *mean_v,
                        const float *std_v, const float *noise_q, int64_t *actions, float *actions_f32, float *logp, float *rows_out,
                        int64_t ld_rows_out, void *workspace, size_t ws_bytes, const rlppo_act_opts *opts);

/* DiscreteFF.get_output (discrete_policy.py:34-42) and the deterministic branch of get_action (:52-57).
 * probs_out (optional): float[n][ld_probs] = softmax of the head (clamp_probs = 0, get_output) or clamp(softmax, 1e-11, 1)
 * (clamp_probs = 1, what get_action works on).  flat_argmax (optional): int64[1] = numpy's argmax over the FLATTENED clamped
 * [n, n_actions] array -- the reference's deterministic action (one index for the whole batch, first occurrence of the maximum;
 * n * n_actions < 2^32).  At least one of the two outputs must be given.  Workspace: rlppo_forward_workspace_bytes. */
int rlppo_discrete_probs(void *stream, const int32_t *dims, int32_t n_layers, const float *packed,
                         const float *obs, int64_t ld_obs, int64_t n, int32_t clamp_probs, float *probs_out,
                         int64_t ld_probs, int64_t *flat_argmax, void *workspace, size_t ws_bytes, const rlppo_act_opts *opts);

/* The selection step alone on caller-supplied probabilities p[n][ld_p] (used by tests to show index equality
 * is exact given identical probs, and by the multi-discrete head with n*8 rows of 3). */
int rlppo_categorical_select(void *stream, const float *probs, int64_t ld_p, int64_t n, int32_t n_cat,
                             const float *noise_q, int64_t *actions, float *logp);

/* ContinuousPolicy.get_action (continuous_policy.py:75-98): tanh head, std = y*var_m + var_b
 * (torch_functions.py:30-33), action = clamp(mean + std*eps, -1, 1) with the caller's N(0,1) noise
 * eps[n][k], logp = sum_k logpdf(action) with the reference's 4-term formula (continuous_policy.py:54-63).
 * dims[n_layers] == 2k.  actions: float[n][k]; logp: float[n]. */
int rlppo_gaussian_act(void *stream, const int32_t *dims, int32_t n_layers, const float *packed,
                       const float *obs, int64_t ld_obs, int64_t n, const float *noise_eps, float var_m, float var_b,
                       float *actions, float *logp, void *workspace, size_t ws_bytes, const rlppo_act_opts *opts);

/* MultiDiscreteFF.get_action (multi_discrete_policy.py:46-74, torch_functions.py:93-122): 21 logits -> 8
 * categoricals (5x3 + 3x2); noise_q[n*8][3] as Categorical.sample draws it; actions int64[n][8]; logp[n]. */
int rlppo_multidiscrete_act(void *stream, const int32_t *dims, int32_t n_layers, const float *packed,
                            const float *obs, int64_t ld_obs, int64_t n, const float *noise_q,
                            int64_t *actions, float *logp, void *workspace, size_t ws_bytes, const rlppo_act_opts *opts);

/* ---------------------------------------------------------------------------------------------- GAE */

size_t rlppo_gae_workspace_bytes(int64_t n);

/* compute_gae (util/torch_functions.py:36-78) as two segmented reverse affine scans.
 * rews, dones, truncated: float[n]; values: float[n+1] (V of every state + V(next_state of the last step),
 * learner.py:347-352).  return_std: the running return std (learner.py:356); pass NaN for `None`
 * (no reward scaling).  Outputs float[n]: value_targets = V + adv, advantages, returns.
 * Arithmetic: reward scaling in float32 (as the reference's float32/float32 division), recurrences in
 * float64, outputs rounded once to float32 -- the reference's behaviour under its pinned NumPy < 2.
 * One launch (chunk scans + decoupled look-back; per-launch record tags, so launches on one workspace never interfere); inside
 * a stream capture the stateless two-launch form is used instead.  A look-back wait is bounded: if it ever times out (not
 * expected: a chunk only waits on workgroups dispatched before it) the affected outputs are NaN and word 1 of the workspace
 * (uint32) counts the event -- never a silent wrong result, never a hang. */
int rlppo_gae(void *stream, const float *rews, const float *dones, const float *truncated, const float *values,
              int64_t n, double gamma, double lmbda, float return_std,
              float *value_targets, float *advantages, float *returns, void *workspace, size_t ws_bytes);

/* -------------------------------------------------------------------------------------- PPO update */

size_t rlppo_minibatch_workspace_bytes(const int32_t *pol_dims, int32_t pol_layers, const int32_t *val_dims,
                                       int32_t val_layers, int64_t mb);
size_t rlppo_minibatch_workspace_bytes_for(const int32_t *pol_dims, int32_t pol_layers, const int32_t *val_dims, int32_t val_layers,
                                           int64_t mb, int32_t precision);

typedef struct rlppo_minibatch_args {
    int32_t head;                 /* RLPPO_HEAD_* */
    int32_t pol_layers;
    int32_t val_layers;
    int32_t act_dim;              /* floats per action row in `actions` (1, 8, k) */
    int32_t slot;                 /* 0..RLPPO_MAX_SLOTS-1: minibatches given different slots (and different workspaces) may run
                                     concurrently on library-owned streams; rlppo_ppo_join() makes `stream` wait for all of them */
    int32_t precision;            /* [r5] update precision of THIS call: RLPPO_PRECISION_DEFAULT (0: what rlppo_set_update_precision
                                     chose for the process) or 1 + mode (RLPPO_PRECISION_FP32 / _BF16 / _X3): two learners of one
                                     process may train in different precisions; size the workspace with
                                     rlppo_minibatch_workspace_bytes_for(..., precision) */
    const int32_t *pol_dims;      /* HOST */
    const int32_t *val_dims;      /* HOST */
    const float *pol_packed;
    const float *val_packed;
    const float *pol_packed_r;    /* bf16 update precision only (else NULL): rlppo_net_pack_bf16's images of both networks; in the
                                   * split-bf16 precision (mode 2) pol_wb16 / val_wb16 hold rlppo_net_pack_x3's planes instead */
    const float *val_packed_r;
    const void *pol_wb16;
    const void *val_wb16;
    float *pol_grad;              /* flat arena, accumulated (+=) */
    float *val_grad;
    /* device-resident experience (ExperienceBuffer, experience_buffer.py:42-50), gathered by `idx` */
    const float *states;          /* [N][ld_states], padded rows */
    int64_t ld_states;
    int64_t n_rows;               /* N, the rows the experience arrays hold (ring_cap when they are a ring).  With it known the
                                     first layer's four launches fetch row idx[r] straight from `states` (the gather of
                                     experience_buffer.py:82-87 fused into their load stage); 0 = unknown: the rows
                                     are gathered into the workspace by a pass of their own first */
    const float *actions;         /* [N][act_dim], float-encoded (experience_buffer.py:72) */
    const float *old_logp;        /* [N] */
    const float *targets;         /* [N]  value targets ("values" in the buffer) */
    const float *advantages;      /* [N] */
    const int64_t *idx;           /* [mb] LOGICAL rows of this minibatch (a slice of the epoch's permutation) */
    int64_t mb;                   /* rows in this minibatch */
    int64_t ring_base, ring_cap;  /* the experience arrays as a ring (the FIFO of experience_buffer.py:18-37 without the
                                     torch.cat re-allocation): logical row i is physical row (i + ring_base) mod ring_cap;
                                     ring_cap == 0: plain arrays */
    float clip_range;
    float ent_coef;
    float mb_ratio;               /* mini_batch_size / batch_size (ppo_learner.py:175) */
    float var_m, var_b;           /* gaussian head only */
    double *stats;                /* [RLPPO_N_STATS] device accumulators, += */
    void *workspace;
    size_t ws_bytes;
} rlppo_minibatch_args;

#define RLPPO_STAT_ENTROPY 0     /* += entropy of this minibatch (mean over rows)            ppo_learner.py:184 */
#define RLPPO_STAT_KL 1          /* += mean((ratio-1) - log ratio)                            ppo_learner.py:161-162 */
#define RLPPO_STAT_VLOSS 2       /* += mse(v, target)                                         ppo_learner.py:182 */
#define RLPPO_STAT_CLIPFRAC 3    /* += mean(|ratio-1| > clip)                                 ppo_learner.py:165-169 */
#define RLPPO_STAT_PLOSS 4       /* += -mean(min(ratio*A, clamp(ratio)*A))  (diagnostic)      ppo_learner.py:172-174 */
#define RLPPO_STAT_GNORM2_POL 5  /* written by rlppo_clip_adam: squared grad norm, policy */
#define RLPPO_STAT_GNORM2_VAL 6
#define RLPPO_STAT_PASSES 7      /* host side: number of rlppo_ppo_minibatch passes behind the sums (all-reduced with them) */
#define RLPPO_N_STATS 8

#define RLPPO_MAX_SLOTS 8

/* One minibatch of PPOLearner.learn (ppo_learner.py:134-185): value forward, policy forward,
 * get_backprop_data, clipped surrogate + entropy + value losses, both backward passes; gradients are ADDED into
 * pol_grad / val_grad (the reference accumulates over the minibatches of a batch, ppo_learner.py:179-180). */
int rlppo_ppo_minibatch(void *stream, const rlppo_minibatch_args *args);

/* Orders `stream` after every minibatch enqueued through non-zero slots since the last join (call it before the
 * gradient all-reduce / rlppo_clip_adam).  Independent minibatches of one batch only meet in the gradient arena
 * (one add per element and launch), so they may overlap; each launch is short (50-100 us at 65,536 rows), and overlapping
 * chains fill the CUs that one chain's ramp-up and tail leave idle. */
int rlppo_ppo_join(void *stream);

/* clip_grad_norm_(max_norm) + torch.optim.Adam.step() on one flat arena (ppo_learner.py:187-193).
 * Hyper-parameters are doubles (python floats in the reference); `step` is the 1-based Adam step count of
 * THIS update; the bias corrections lr/(1-beta1^t), sqrt(1-beta2^t) are formed in double and cast to fp32 where
 * they meet tensor data, as torch/optim/adam.py does.  gnorm2: device double that receives the squared gradient
 * norm before clipping.  grads are scaled in place by the clip coefficient, as clip_grad_norm_ does. */
int rlppo_clip_adam(void *stream, float *params, float *grads, float *exp_avg, float *exp_avg_sq, int64_t n,
                    double max_norm, double lr, double beta1, double beta2, double eps, int64_t step,
                    double *gnorm2);

/* The optimiser tail of one batch for BOTH networks in ONE launch (ppo_learner.py:187-193: clip_grad_norm_ x2,
 * value_optimizer.step(), policy_optimizer.step()) + the re-packing rlppo_net_pack would do + the zero_grad of the next batch
 * (ppo_learner.py:135-136): squared norms of both gradient arenas, then clip + Adam on both, every updated parameter written
 * straight into its W / W^T / b slots of `packed` (which must already hold a full rlppo_net_pack image: the zero padding is not
 * rewritten), gradients left ZERO.  Element for element the arithmetic of rlppo_clip_adam + rlppo_net_pack.  Used by
 * PPOLearner.learn; FusedAdam.step() (the public optimiser API, gradients scaled in place) stays on rlppo_clip_adam.
 * sync_ws: RLPPO_OPT_SYNC_BYTES of device memory, 16-byte aligned, owned by the caller and ZEROED ONCE when it is allocated
 * (never again: every completed call leaves it armed for the next): the state of
 * the grid barrier between the norm and the update (arrival counter, generation word, one partial-sum slot per workgroup, added
 * in a fixed order: the norm is bit-reproducible), and at byte offset 8 a uint32 that counts barrier waits that gave up.
 * Giving up is all or nothing: NO element of either network is then touched by that call (parameters, moments, gradients and
 * `packed` stay as they were), the block is dead -- every later call on it skips its update at once and counts itself --
 * until the caller has zeroed it again; callers read the word back with their report and can repeat the step with
 * sync_ws == NULL.  The grid is clamped to what the device keeps resident (a device too small for one workgroup per network
 * takes the three-operation form by itself).  Not shareable between concurrent calls.  sync_ws == NULL selects the
 * three-operation form (fill, norms, update). */
#define RLPPO_OPT_SYNC_BYTES 16384
typedef struct rlppo_opt_net {
    const int32_t *dims;  /* layer widths, n_layers + 1 entries */
    int32_t n_layers;
    float *params, *grads, *exp_avg, *exp_avg_sq; /* flat arenas, rlppo_flat_floats entries */
    float *packed;        /* rlppo_packed_floats entries */
    double *gnorm2;       /* receives the squared gradient norm before clipping */
    double max_norm, lr, beta1, beta2, eps;
    int64_t step;         /* 1-based Adam step count of THIS update */
} rlppo_opt_net;
int rlppo_clip_adam_pack2(void *stream, const rlppo_opt_net *a, const rlppo_opt_net *b, void *sync_ws);

/* [r6] The tail of PPOLearner.learn in ONE launch (ppo_learner.py:213-234: the update magnitudes ||theta_before - theta_after||_2 of
 * both networks, the report means' sums read out, one device -> host synchronisation): round 5 spent ten dependent eager launches
 * and a blocking copy here, 0.1 ms of a 3 ms learn() at the reference's own batch size.  The kernel
 *   - adds add_passes to stats[RLPPO_STAT_PASSES], copies the RLPPO_N_STATS accumulators to out[0 .. RLPPO_N_STATS) and ZEROES them
 *     (the next learn() starts from clean sums without a fill of its own);
 *   - out[RLPPO_N_STATS], out[RLPPO_N_STATS + 1] <- sqrt(sum (before_i - now_i)^2) of the policy / the critic: float32 differences as
 *     the reference forms them, squares and sums in double, partial sums added in a fixed order (bit-reproducible);
 *   - out[RLPPO_N_STATS + 2] <- *timeout_word (the give-up counter of rlppo_clip_adam_pack2's sync block; 0 when NULL),
 *     out[RLPPO_N_STATS + 3] <- *extra (a device double the caller wants back with the report -- the give-up count summed over the
 *     ranks of a data-parallel run; out[RLPPO_N_STATS + 2] again when NULL);
 *   - then stores done_value into *done_word with release semantics at system scope (rlppo_host_wait_words polls it).
 * `out` and `done_word` are HOST-VISIBLE (pinned) memory; ws: RLPPO_REPORT_WS_BYTES of device memory owned by the caller, zeroed
 * once when it is allocated (arrival counter + one partial-sum slot per workgroup; every call leaves it armed for the next). */
#define RLPPO_REPORT_WS_BYTES 4096
#define RLPPO_REPORT_OUT_DOUBLES (RLPPO_N_STATS + 4)
typedef struct rlppo_report_args {
    const float *pol_before, *pol_now;
    int64_t n_pol;
    const float *val_before, *val_now;
    int64_t n_val;
    double *stats;
    double add_passes;
    const uint32_t *timeout_word;
    const double *extra;
    double *out;
    uint32_t *done_word;
    uint32_t done_value;
    void *ws;
} rlppo_report_args;
int rlppo_learn_report(void *stream, const rlppo_report_args *args);

/* ---------------------------------------------------------------------------- process-mode collection (host only) */

/* [r6] The learner-side loop of the reference's process-per-environment collection (rlgym_ppo/batched_agents/
 * batched_agent_manager.py:126-350, batched_trajectory.py:58-105) as host C++ behind the SAME wire format (UDP headers of three magic
 * floats + one slab per worker of a shared float32 array: rlgym_ppo_amd/batched_agents/comm_consts.py): wait for the ready sockets,
 * read the datagram, parse the slab, advance the observation statistics (Welford, in the state's dtype and the reference's operation
 * order), standardise, keep the episode-reward bookkeeping, bank the timestep into the environment's trajectory, and lay the
 * trajectories out agent by agent at the end of a collect_timesteps call (last step force-marked truncated unless done: quirk Q4).
 * The policy call stays with the host; one inference is  _ready -> [policy.get_action] -> _send -> _collect.  All pointers HOST.
 *   create: socket_fds[i] = the learner's UDP socket of worker i (owned by the caller), peer_ports[i] = that worker's port on
 *     127.0.0.1, shm_base + i * shm_floats_per_worker = its slab.
 *   set_obs: current_obs[worker] <- rows x obs_dim floats (a reset state, as received: not standardised), ready != 0: the worker
 *     waits for actions.
 *   ready: the stacked observations of the workers waiting for actions (at most cap_rows rows) -> n_rows.
 *   send: actions[n_rows][act_width] float32 and log_probs[n_rows] for exactly those rows: recorded, and sent to the workers.
 *   collect: blocks until messages worth >= min_obs agent-steps have been banked (a message counts its prev_n_agents) -> n_collected;
 *     a signal makes it return RLPPO_ERR_INTERRUPTED with the progress so far (the host's handlers run; resume = 1 continues the wait);
 *     standardize 0 = off, 1 = (x - mean[0]) / std[0] clipped to +-5 (the reference's scalars, quirk Q5), 2 = per feature;
 *     stats_mean / stats_var / stats_count / steps_since_increment: the WelfordRunningStat's state (float32 arrays; float64 when
 *     stats_f64 -- statistics restored from JSON -- and then mean / std are doubles too and the standardisation is formed in float64
 *     and rounded once, as numpy does) and the manager's cadence counter, advanced in place every steps_per_increment-th message with
 *     the RAW rows.
 *   finish: flushes every trajectory -> the sizes _emit needs; emit: states / next_states [n][obs_dim], actions [n][act_width],
 *     log_probs [n] float32, rewards / dones / truncated [n] float64 (the dtypes the reference's lists become), and every message's
 *     metrics record (values flat; 9 ints per record: rank, dimensions).
 *   average_reward: get (set == 0) / set the manager's running average (is_none: it has not seen an episode end yet). */
int rlppo_collector_create(int32_t n_workers, const int32_t *socket_fds, const int32_t *peer_ports, const float *shm_base,
                           int64_t shm_floats_per_worker, int32_t obs_dim, void **handle);
int rlppo_collector_destroy(void *handle);
int rlppo_collector_set_obs(void *handle, int32_t worker, const float *obs, int32_t rows, int32_t ready);
int rlppo_collector_ready(void *handle, float *obs_out, int64_t cap_rows, int64_t *n_rows);
int rlppo_collector_send(void *handle, const float *actions, int32_t act_width, const float *log_probs);
int rlppo_collector_collect(void *handle, int64_t min_obs, int32_t resume, int32_t standardize, const void *mean, const void *stdv, void *stats_mean,
                            void *stats_var, int64_t *stats_count, int32_t stats_f64, int64_t steps_per_increment, int64_t *steps_since_increment,
                            int64_t *n_collected);
int rlppo_collector_finish(void *handle, int64_t *n_steps, int32_t *act_width, int64_t *n_metrics, int64_t *metrics_floats);
int rlppo_collector_emit(void *handle, float *states, float *actions, float *log_probs, double *rewards, float *next_states, double *dones,
                         double *truncated, float *metrics_values, int32_t *metrics_shapes);
int rlppo_collector_average_reward(void *handle, int32_t set, double *value, int32_t *is_none);

/* ------------------------------------------------------------------------------------- data-parallel exchange */

/* The one exchange step of the path: an in-place sum over the ranks of the flat
 * [grad_policy | grad_value] arena, after a rank's last backward of a batch and before clip_grad_norm_ / Adam, which then
 * run replicated (ppo_learner.py:187-193 sees the reduced gradients).  The reference is single-device and has no
 * counterpart.  RCCL is loaded at run time (rlppo_comm_set_library(path) to name a particular librccl.so, e.g. PyTorch's
 * own copy; default: the loader's "librccl.so").  One communicator per process (one process per GPU): rank 0 obtains
 * the 128-byte id with rlppo_comm_unique_id and hands it to the other ranks by any host channel; every rank then calls
 * rlppo_comm_init on the thread whose current device is its GPU.  rlppo_allreduce enqueues ncclAllReduce(sum) on the
 * caller's stream (fp32, or fp64 for the report statistics) and returns without synchronising.  Errors: 2000 + ncclResult_t.
 * rlgym_ppo_amd/dp.py uses these when RLPPO_RCCL_DIRECT=1 and torch.distributed's all_reduce otherwise. */
#define RLPPO_COMM_ID_BYTES 128
int rlppo_comm_set_library(const char *path);
int rlppo_comm_unique_id(void *id128);
int rlppo_comm_init(int32_t rank, int32_t world, const void *id128);
int rlppo_allreduce(void *stream, void *buf, int64_t n, int32_t is_f64);
int rlppo_comm_destroy(void);

/* ----------------------------------------------------------------------------------------- host helpers */

/* numpy.random.RandomState legacy stream (MT19937 + masked rejection + reverse Fisher-Yates): the index stream
 * of ExperienceBuffer.get_all_batches_shuffled (experience_buffer.py:52,97-98).  HOST code, HOST pointers.
 * state: 625 uint32 (624 words of key + position), as numpy's get_state() exposes it. */
int rlppo_mt19937_seed(uint32_t *state625, uint32_t seed);
int rlppo_mt19937_permutation(uint32_t *state625, int64_t n, int64_t *out);
/* The same permutation in two phases (same stream consumption, same result).  draw_targets: the serial phase -- advances
 * the generator exactly as rlppo_mt19937_permutation(n) does and records the n-1 swap targets of numpy's reverse
 * Fisher-Yates loop (mtrand.pyx _shuffle_raw) in draw order: targets[t] = j_i for i = n-1-t.  apply_swap_targets: applies
 * those swaps to arange(n); no generator state, so it can run on another thread while the next epoch is being drawn
 * (the shuffle of experience_buffer.py:97-98 is the serial host work of an epoch once 8 ranks share the GPU work). */
int rlppo_mt19937_draw_targets(uint32_t *state625, int64_t n, uint32_t *targets);
int rlppo_apply_swap_targets(int64_t n, const uint32_t *targets, int64_t *out);

/* torch.empty(n).exponential_(lambda) of PyTorch's CPU generator, bit for bit: the Exp(1) noise torch.multinomial(probs, 1,
 * True) / Categorical.sample() consume in the reference's rollout (discrete_policy.py:59), i.e. the stream
 * that makes a seeded run pick the reference's action indices.  HOST code, HOST pointers.  `state`: the generator's serialised
 * state as torch.get_rng_state() returns it (5056 bytes), advanced in place exactly as torch advances it (2 n MT19937 words).
 * The serial MT19937 phase is vectorised and the double-precision log1p transform runs on `threads` threads: 4096 x 90 values
 * in ~0.5 ms instead of the ~4.7 ms of torch's serial kernel. */
int rlppo_torch_cpu_exponential(void *state, int64_t state_bytes, int64_t n, double lambda, float *out, int32_t threads);
/* The same draw in two calls, for callers that pipeline consecutive draws (the stream phase of the next draw needs only the state
 * this one's stream phase leaves behind): _words fills words[2 n + 8] with the tempered MT19937 words and advances `state` exactly as
 * the one-call form does; rlppo_exponential_from_words turns them into the identical n float32 values (no generator involved: any
 * thread). */
int rlppo_torch_cpu_exponential_words(void *state, int64_t state_bytes, int64_t n, uint32_t *words);
int rlppo_exponential_from_words(const uint32_t *words, int64_t n, double lambda, float *out);
/* One draw as one call that chains itself to its predecessor on another thread: link blocks are { int32 ready; padding to 64 bytes;
 * the generator state (state_bytes) }.  The call waits (spinning, in C) until `link_in` -- the block its predecessor publishes as soon
 * as that call's STREAM phase is done -- is ready (link_in == NULL: starts from `state`, which is not modified), runs its stream phase,
 * publishes `link_out` (zero its first word beforehand) and then transforms into `out`; `words` is scratch of 2 n + 8 uint32.  The state
 * after the draw is link_out's.  Values and states are those of rlppo_torch_cpu_exponential. */
#define RLPPO_EXP_LINK_HEADER 64
int rlppo_torch_cpu_exponential_chained(const void *state, int64_t state_bytes, int64_t n, double lambda, float *out, uint32_t *words,
                                        void *link_in, void *link_out);
/* [r4] A run of chained draws on the calling thread: of a burst of `count` consecutive draws of n values each, the draws first,
 * first + step, ... (`step` threads share a burst, one call each, first = 0 .. step - 1).  Draw i writes out + i * out_stride floats
 * and publishes the link block links + i * link_stride bytes, laid out as above with one more word: { int32 ready; int32 done
 * (1: the values are complete; -1: failed or cancelled); padding to 64 bytes; state }; zero the first 8 bytes of every block
 * beforehand.  Draw 0 starts from `state` (not modified), or from `link_in0` -- the block an earlier chained call publishes -- when
 * that is not NULL.  `cancel`: caller-owned int32 read before every draw; non-zero ends the run (remaining draws marked failed).
 * Values and states are those of `count` consecutive rlppo_torch_cpu_exponential calls.  (engine.HostExponential.prefetch: the
 * next rollout's noise, drawn while PPOLearner.learn runs -- learner.py:257-270, discrete_policy.py:59.) */
int rlppo_torch_cpu_exponential_burst(const void *state, void *link_in0, int64_t state_bytes, int64_t n, double lambda, float *out,
                                      int64_t out_stride, void *links, int64_t link_stride, int32_t first, int32_t step, int32_t count,
                                      const int32_t *cancel);

/* dst[r][0..width) = src[idx[r]][0..width), fp32 rows, 16 bytes per thread (width, ld_src multiples of 4; dst rows are
 * `width` floats apart).  The minibatch gather of experience_buffer.py:82-87 (used inside rlppo_ppo_minibatch) and the
 * step-major -> trajectory-major flatten of a device-resident rollout (batched_agent_manager.py:154-172 builds the same
 * order with Python lists). */
int rlppo_gather_rows(void *stream, const float *src, int64_t ld_src, const int64_t *idx, float *dst, int32_t width, int64_t n);

/* WelfordRunningStat.increment(samples, n) (running_stats.py:28-46) on the device, bit-exact with the reference's sample-by-
 * sample update: mean[d], m2[d] (= running_mean, running_variance) are updated in place with the n rows of `samples` (fp32,
 * row stride ld floats); `count` is the number of samples already absorbed (the caller adds n afterwards).  The state is
 * float32 as the class constructs it, or float64 (state_is_f64 != 0) as WelfordRunningStat.from_json leaves it after a
 * checkpoint load (running_stats.py:121-125: np.asarray of Python floats) -- the reference then updates in float64. */
int rlppo_welford_increment(void *stream, const float *samples, int64_t ld, int64_t n, int32_t d, void *mean, void *m2,
                            int64_t count, int32_t state_is_f64);
/* WelfordRunningStat.increment_from_serialized_other (running_stats.py:71-98): combine the running statistics of another
 * instance (other_mean[d], other_m2[d] fp32 -- the reference casts the serialised list to float32 -- and other_count) into
 * mean / m2 in place, in the reference's operation order; the caller sets count += other_count.  SURVEY.md 8(f) row 4. */
int rlppo_welford_merge(void *stream, int32_t d, void *mean, void *m2, int64_t count, const float *other_mean,
                        const float *other_m2, int64_t other_count, int32_t state_is_f64);

/* Precision of the ROLLOUT forward passes (rlppo_mlp_forward, rlppo_*_act): 0 = fp32 (default; the parity mode),
 * 1 = activations and master weights rounded to bf16 as MFMA operands, fp32 accumulation / bias / activation
 * (BASELINE configs[4] "bf16 fwd / fp32 master weights").  The update has its own switch (rlppo_set_update_precision).  The reference
 * has no such mode (it is fp32 throughout): outputs then agree with an fp32 forward to ~1e-2, not 1e-5. */
int rlppo_set_inference_precision(int32_t mode);

/* Precision of the PPO UPDATE (rlppo_ppo_minibatch): 0 = fp32 (default; the parity mode: fp32 losses/grads within 1e-5 of the
 * reference), 2 = fp32 with split-bf16 hidden products (below), 1 = BASELINE configs[4] "bf16 fwd / fp32 master weights" = mixed-precision training as torch writes it:
 *   - every forward product of both networks multiplies bf16-rounded operands (activations and weights, round-to-nearest-even)
 *     on the bf16 MFMA pipe and accumulates in fp32; bias and activation in fp32; the hidden activations are stored as bf16;
 *   - the gradient with respect to each hidden activation is therefore a bf16 tensor too (rounded once, after the fp32
 *     accumulation), and every backward product -- dX = dY . r(W), dW = dY^T . r(X) -- multiplies bf16 values on the bf16 MFMA
 *     pipe with fp32 accumulation;
 *   - the loss and its gradient with respect to the network outputs, dW / db accumulation, clip and Adam are fp32 and the master
 *     weights are the fp32 arena (the gradient of a weight is never rounded).
 * It is torch.autograd of F.linear(h.bfloat16().float(), r(W), b) per layer, r = rounding of a master weight with an identity
 * backward.  The reference has no such mode; the test suite restates it on the CPU (tests/test_gpu_cfg5.py).  Changing the mode
 * changes rlppo_minibatch_workspace_bytes.
 * rlppo_net_pack_bf16: the rounded images the mode needs, rebuilt after every optimiser step -- packed_r: the packed layout
 * (rlppo_packed_floats) holding the rounded weights as fp32; wb16: rlppo_wb16_elems bf16 values, the W[Pout][Pin] blocks of all
 * layers followed by the W^T[Pin][Pout] blocks. */
int rlppo_set_update_precision(int32_t mode);
int rlppo_get_update_precision(void);
int64_t rlppo_wb16_elems(const int32_t *dims, int32_t n_layers);
int rlppo_net_pack_bf16(void *stream, const int32_t *dims, int32_t n_layers, const float *flat, float *packed_r, void *wb16);
/* [r4] mode 2 (OPT-IN, not the default and not what bench.py's headline runs): the update stays fp32 in memory and in meaning --
 * activations, gradients, losses, dW, clip, Adam exactly as mode 0 -- but the hidden-layer FORWARD and dX products (widths that are
 * multiples of 256, contraction a multiple of 32, layers >= 1) run on the bf16 MFMA pipe from three-piece operands: every fp32
 * value x = x_h + x_m + x_l (bf16 pieces, rounded to nearest), six piece products per element pair, the five small ones summed
 * apart from the accumulator (csrc/gemm_split.hip).  gfx950's fp32-input MFMA has 1/16 of the bf16 rate; this form is 1.35 x
 * faster per launch and, against float64, MORE accurate than the fp32 MFMA's fmaf chain (0.43-0.46 x its error at K = 256).  The
 * gradients then differ from mode 0's by fp32 rounding noise only (tests/test_gpu_kernels.py, tests/test_gpu_learner.py hold
 * mode 2 to the same float64 gates as mode 0) -- for FINITE operands of ordinary magnitude: a +-inf, a NaN or a finite
 * |x| >= 3.396e38 (it rounds to a bf16 infinity) makes x - bf16(x) an inf - inf, so every output of that ROW is NaN where the fp32 MFMA
 * returns +-inf (or a finite product); pieces of |x| below ~1e-33 fall under bf16's normal range and lose low bits (an error relative
 * to 1e-38, not to the data).  Clipped observations and ReLU activations never get there; test_gemm_nt_x3_special_values
 * (tests/test_gpu_kernels.py) pins this behaviour.  Needs, per network, the image rlppo_net_pack_x3 derives from the PACKED copy
 * after every optimiser step: rlppo_x3_elems bf16 values (the stage-major planes [K / 32][3][N / 16][1 KiB block] of W for every
 * covered forward and of W^T for every covered dX; inside a 16-row block the 16-byte chunk (row r, k quarter q) at chunk
 * 4 r + (q ^ (-(r / 4) & 3)): the image the kernels' LDS fragment reads are bank-conflict-free on), handed over in rlppo_minibatch_args.pol_wb16 / val_wb16.  Policy and critic run as two
 * chains in this mode (no paired launches, the critic's head as its own matrix-vector launch). */
int64_t rlppo_x3_elems(const int32_t *dims, int32_t n_layers);
int rlppo_net_pack_x3(void *stream, const int32_t *dims, int32_t n_layers, const float *packed, void *planes);
/* single-kernel entry points of that precision (tests, bench.py): planes[C / 32][3][R / 16][1 KiB block, chunk order as above] <- the pieces of S[R][C] (row stride ld; R % 16 == 0);
 * C[M][N] = relu(A[M][K] . W^T + bias) with bits <- [C > 0] (mode 0) or (A . B^T) masked by bits (mode 1); N % 256 == 0, K % 32 == 0,
 * bits as rlppo_dbg_gemm_nt_bits_bytes(M, N). */
int rlppo_dbg_pack_x3(void *stream, const float *S, int64_t ld, int32_t R, int32_t C, void *planes);
int rlppo_dbg_gemm_nt_x3(void *stream, const float *A, int64_t lda, const void *planes, const float *bias, float *C, int64_t ldc, int64_t M,
                         int32_t N, int32_t K, int32_t mode, void *bits);

/* ------------------------------------------------------------------------------------------ diagnostics */
/* A/B switches for measurements and tests (also RLPPO_TUNE="key=value,..." in the Python host).  Defaults in brackets.
 *   1 GAE algorithm [1 single-pass look-back | 0 two launches]
 *   4 policy / critic chains of rlppo_ppo_minibatch on two streams [1]
 *  21 GAE look-back spin limit [-1 = default 2^20 | 0 = every wait times out at once (tests)]
 *  22 GAE grid [0 = at most the resident capacity, workgroups loop over chunks beyond it | 1 = one workgroup per chunk always]
 *  23 bf16 update precision, form of the hidden-layer forward / dX / dW products [2 = 256 x 256 tiles walked by persistent workgroups
 *     (forward / dX; default) | 1 = 256 x 256, one workgroup per tile | 0 = 128 x 128]
 *  24 rlppo_torch_cpu_exponential transform [1 = AVX2 logarithm certified element by element against float32 rounding, libm for the
 *     rest (default) | 0 = libm for every element]
 *  26 minibatch gather [1 = fused into the first layer's GEMM launches through a row table from 262,144 rows per pass, a gather pass of its
 *     own below (default; [r5]: measured, csrc/api.hip) | 2 = fused at every size | 0 = always a gather pass of its own]
 *  27 rlppo_discrete_act / rlppo_discrete_step [1 = one fused launch where the network has that form (default) | 0 = layer chain]
 *  29 policy + critic layers of equal widths as ONE launch [1 = from 262,144 rows per pass (default) | 0 = never | 2 = always]
 *  31 paired pass: the critic's output-layer backward waits for the policy's loss kernel [1 (default) | 0 = both chains free-running]
 *  32 a one-output critic head is computed in the last hidden layer's forward epilogue [1 (default) | 0 = its own matrix-vector launch]
 *  33 paired launches: the two products' tiles interleave in the grid (a row tile's workgroups of both networks back to back on one
 *     XCD) [1 (default) | 0 = the second product stacked behind the first]
 *  34 rlppo_clip_adam_pack2 grid-barrier spin limit [-1 = default 2^22 | 0 = a waiter gives up at once (tests)]
 *  35 rlppo_clip_adam_pack2 test hook [0 | 1 = workgroup (0, 0) never arrives at the barrier: a grid that is not co-resident]
 *  36 split-bf16 products (update precision 2) [1 = persistent workgroups for large launches (default) | 0 = one workgroup per tile]
 *  37 [r5] the weight-gradient products of a pass [1 = ONE grouped launch + ONE reduction after the chains have joined (default; fp32 and
 *     split-bf16 precisions) | 0 = one launch + reduction per layer inside the chains]
 *  38 [r5] workgroups of a grouped weight-gradient launch [0 = two per CU (default) | n]
 *  40 [r5] forward layers of up to 1024 rows (the layer chain of a small rollout call) [1 = one wave per 16 x 16 output block, operands
 *     straight from L2 (default) | 0 = the 128-row tiles of the large kernel]: bit-identical outputs
 *  41 [r6] test hook of rlppo_host_window_alloc [0 = as the device is (default) | 1 = answer like a device that does not expose its
 *     memory to the host (no large BAR: the call fails, the host falls back to pinned memory) | 2 = windows are registered without a
 *     flush register (rlppo_host_window_flush then does nothing)] */
int rlppo_dbg_set(int32_t key, int32_t value);
/* Counter bumped by every call that changes which kernels later launches select (rlppo_dbg_set, rlppo_set_*_precision): a
 * host that caches captured graphs of library calls keys them on it (rlgym_ppo_amd/ppo/_mlp.py::ActGraph). */
int64_t rlppo_selection_epoch(void);
const int64_t *rlppo_selection_epoch_ptr(void);   /* the counter itself (a host that reads it per call) */
/* Which form calls took so far in this process (tests assert that the kernel they mean to pin is the one that ran): 0 = rollout steps
 * served by the one-launch kernel (rlppo_discrete_act / rlppo_discrete_step), 1 = by the layer chain, 2 = rlppo_ppo_minibatch passes,
 * 3 = of them with paired policy + critic launches, 4 = of them with the gather fused into the first layer, 5 = of them with the
 * grouped weight-gradient launch; -1 for an unknown key. */
int64_t rlppo_dbg_counter(int32_t key);
/* Single-kernel entry points used by tests/ and bench.py to check / time each GEMM flavour in isolation.
 * epilogue: 0 bias, 1 bias+relu, 2 bias+tanh, 3 relu-mask (mask_src > 0).  Shapes as in csrc/gemm.hip. */
int rlppo_dbg_gemm_nt(void *stream, const float *A, int64_t lda, const float *B, int64_t ldb, const float *bias,
                      const float *mask_src, int64_t ld_mask, float *C, int64_t ldc, int64_t M, int32_t N, int32_t K,
                      int32_t epilogue);
/* The bitmask form of the hidden-layer forward (epilogue 1: C = relu(A.B^T + bias), bits <- [C > 0], 8 bytes per lane and
 * 128 x 128 tile) and of the masked dX product (epilogue 3: C = (A.B^T) masked by bits).  N % 128 == 0. */
size_t rlppo_dbg_gemm_nt_bits_bytes(int64_t M, int32_t N);
int rlppo_dbg_gemm_nt_bits(void *stream, const float *A, int64_t lda, const float *B, int64_t ldb, const float *bias, float *C,
                           int64_t ldc, int64_t M, int32_t N, int32_t K, int32_t epilogue, void *bits);
/* The forward product of the bf16 update precision: A[M][K] and W[N][K] bf16 in memory, fp32 accumulate.  hidden != 0:
 * C (fp32, may be NULL) and Cb (bf16, may be NULL) receive relu(A.W^T + bias) rounded to bf16, bits (may be NULL) the ReLU
 * bitmask; N % 128 == 0.  hidden == 0: C = A.W^T + bias (epilogue 0) or tanh of it (epilogue 2), fp32.  K % 64 == 0. */
int rlppo_dbg_gemm_nt_b16(void *stream, const void *A, int64_t lda, const void *W, int64_t ldw, const float *bias, float *C,
                          int64_t ldc, void *Cb, int64_t ldcb, int64_t M, int32_t N, int32_t K, int32_t epilogue, int32_t hidden,
                          void *bits);
/* hidden == 2 is the backward product of that precision: Cb[M][N] (bf16) = round_bf16(A . W^T) masked by the ReLU bitmask `bits`
 * the hidden forward of the same M x N geometry wrote (epilogue 3, bias and C NULL; N % 128 == 0, K % 64 == 0).
 * rlppo_dbg_gemm_tn_b16: the weight-gradient product of that precision, dW[out][in] += dY^T . X, db[out] += colsum(dY), both
 * operands bf16 [M][ld] with pout / pin (multiples of 128) columns of which out / in are meaningful; workspace as rlppo_dbg_gemm_tn. */
int rlppo_dbg_gemm_tn_b16(void *stream, const void *dY, int64_t ldy, const void *X, int64_t ldx, float *dW, float *db, int32_t pout,
                          int32_t pin, int32_t out, int32_t in, int64_t M, void *ws, size_t ws_bytes);
/* The output layer's two backward products in the bf16 update precision for a head of out <= 32 outputs (gemv.hip): dY[M][ldy]
 * fp32 (the loss gradient), W[out][ldw] fp32 (bf16-rounded values), hb[M][ldh] the last hidden activation as bf16, bits its ReLU
 * bitmask (rlppo_dbg_gemm_nt_bits_bytes(M, kp)): dxb[M][kp] (bf16) = round_bf16(dY . W) masked; dW[out][in] += dY^T . hb,
 * db[out] += colsum(dY) with a fixed summation order.  kp: padded hidden width, a power of two and a multiple of 128. */
size_t rlppo_dbg_thin_head_workspace_bytes(int32_t out, int32_t kp, int64_t M);
int rlppo_dbg_thin_head_b16(void *stream, const float *dY, int64_t ldy, int32_t out, const float *W, int64_t ldw, const void *bits,
                            const void *hb, int64_t ldh, void *dxb, float *dW, float *db, int32_t in, int32_t kp, int64_t M, void *ws,
                            size_t ws_bytes);
/* dW[out][in] += dY^T . X, db[out] += colsum(dY) through partial tiles in `ws` (rlppo_dbg_gemm_tn_workspace_bytes) and a
 * fixed-order reduction: the form rlppo_ppo_minibatch uses. */
size_t rlppo_dbg_gemm_tn_workspace_bytes(int32_t out, int32_t in, int64_t M);
int rlppo_dbg_gemm_tn(void *stream, const float *dY, int64_t ldy, int32_t ny_valid, const float *X, int64_t ldx,
                      int32_t kx_valid, float *dW, float *db, int32_t out, int32_t in, int64_t M, void *ws, size_t ws_bytes);

/* [r5] Every weight-gradient product of a pass as ONE launch + ONE fixed-order reduction (the form rlppo_ppo_minibatch uses from
 * round 5 on, rlppo_dbg_set(37)): product i is dW_i[out][in] += dY_i^T . X_i, db_i[out] += colsum(dY_i) over the same M rows (db may be
 * NULL); rowtab != NULL: sample m of product i is X_i[rowtab[m]] of a src_rows-row matrix (the fused minibatch gather of
 * experience_buffer.py:82-87).  The row splits of each product are sized so that every workgroup of the grid multiplies the same
 * number of MFMA blocks; the result depends on the SET of shapes and M, not on the order of the products. */
typedef struct rlppo_tn_product {
    const float *dY;
    int64_t ldy;
    int32_t ny_valid;
    const float *X;
    int64_t ldx;
    int32_t kx_valid;
    float *dW, *db;
    int32_t out, in;
    const uint32_t *rowtab;
    int64_t src_rows;
} rlppo_tn_product;
size_t rlppo_dbg_gemm_tn_group_workspace_bytes(const rlppo_tn_product *p, int32_t n, int64_t M);
int rlppo_dbg_gemm_tn_group(void *stream, const rlppo_tn_product *p, int32_t n, int64_t M, void *ws, size_t ws_bytes);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif /* RLPPO_H */
