/* ns2vc_hip.h — C ABI of libns2vc_hip.so, the MI355X (gfx950) denoiser engine.
 *
 * Drop-in boundary for ONE path of adelacvg/NS2VC (branch vc-v2): the latent
 * diffusion denoiser `UNet1DConditionModel.forward` driven by the DPM-Solver++ /
 * UniPC sampling loop.  The reference is pure Python/PyTorch, so there is no
 * reference FFI to mirror; each entry point below names the reference call site
 * (file:line under the reference tree) whose work it replaces.  INTEGRATION.md
 * shows the ctypes binding a maintainer adds on the reference side.
 *
 * Conventions
 *   - every function returns 0 on success, non-zero on failure; the message is
 *     available from ns2vc_last_error() (thread-local, valid until the next call);
 *   - all tensor pointers are DEVICE pointers owned by the caller unless the
 *     parameter is documented as "host or device" (copied with hipMemcpyDefault);
 *   - `stream` is a hipStream_t passed as void*; no call blocks the host except
 *     load/finalize/prepare (setup) and the explicitly synchronous helpers;
 *   - API tensors use the reference's layouts: latent/content (B, C, T) "NCT"
 *     fp32 contiguous, prompt (B, Lp, 256) fp32 contiguous, mask (B, Lp) uint8
 *     (1 = keep), timesteps (B) fp32 (fractional values are legal,
 *     sampler/dpm_solver.py:278).
 */
#ifndef NS2VC_HIP_H
#define NS2VC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NS2VC_ABI_VERSION 7
#define NS2VC_MAX_LEVELS 8
#define NS2VC_NCOEF 12 /* floats per row of the solver table, ns2vc_amd/schedule.py:COEF_COLUMNS */

typedef struct ns2vc_unet ns2vc_unet; /* opaque engine handle */

/* Mirrors the ctor kwargs NS2VC passes (model.py:391-400) plus the defaults of
 * unet1d/unet_1d_condition.py:151-203 that shape the network. */
typedef struct ns2vc_unet_cfg {
  int32_t latent_channels;    /* 100: out_channels and the x part of in_channels  */
  int32_t content_channels;   /* 256: in_channels - latent_channels               */
  int32_t n_levels;           /* 4                                                */
  int32_t block_out_channels[NS2VC_MAX_LEVELS]; /* 128,256,384,512                */
  int32_t norm_num_groups;    /* 8                                                */
  int32_t cross_attention_dim;/* 256                                              */
  int32_t heads;              /* 8  (the reference's `attention_head_dim`)        */
  int32_t layers_per_block;   /* 2                                                */
  int32_t pool_heads;         /* 64 (addition_embed_type_num_heads)               */
} ns2vc_unet_cfg;

/* Operand precision = what the MFMAs read (activations as GEMM / attention operands, packed weights).  In every mode the
 * residual stream, GroupNorm / LayerNorm statistics, softmax state, accumulators and the solver state are fp32.
 * Measured end-to-end error of the predicted latent vs the reference fp32 CPU path (tests/test_engine_gpu.py):
 *   F32 1e-6, F16 8e-4 (inside the 1e-3 parity gate: the default 16-bit mode), BF16 6.5e-3 (same speed as F16). */
enum { NS2VC_PREC_F32 = 0,  /* fp32 operands, exact-fp32 MFMA (v_mfma_f32_32x32x2_f32) */
       NS2VC_PREC_BF16 = 1, /* bf16 operands (v_mfma_f32_32x32x16_bf16) */
       NS2VC_PREC_F16 = 2   /* fp16 operands (v_mfma_f32_32x32x16_f16); stores saturate at +-65504 */ };

/* ---- library ----------------------------------------------------------------------- */
int ns2vc_abi_version(void);
const char* ns2vc_last_error(void);
int ns2vc_device_count(int* out_count);
int ns2vc_set_device(int device);               /* one process per GPU: call with LOCAL_RANK */
int ns2vc_device_name(char* buf, int buflen);
/* ABI v5.  1 when workgroup ids that differ by a multiple of 8 run on one XCD of the current device (the dispatcher's round robin;
 * probed once per device with HW_REG_XCC_ID), 0 when they do not, -1 when the probe could not run.  ns2vc_gemm_args.gnp_sync -- rows
 * exchanged between such workgroups through the L2 they share -- is only FAST where this is 1; the engine's `gn_coop` option
 * defaults to it and cannot be switched on elsewhere.  ABI v6: correctness no longer rests on the probe -- the siblings of a row block
 * count their arrivals per XCC id (HW_REG_XCC_ID) and only trust each other's rows when all of them ran on one XCD; otherwise every
 * workgroup builds all its rows itself (a CU-masked stream, another partition mode), counted in ns2vc_unet_gn_coop_alone.  (The
 * dispatcher's round robin does not start at XCD 0 for every launch -- profiles/r05_placement_probe.txt -- so there is no fixed
 * "XCD of workgroup id i" to rely on, only "ids 8 apart share one".) */
int ns2vc_device_xcd_round_robin(int* out);

/* ---- engine lifetime (replaces UNet1DConditionModel.__init__, unet_1d_condition.py:151-607) */
int ns2vc_unet_create(const ns2vc_unet_cfg* cfg, ns2vc_unet** out);
int ns2vc_unet_destroy(ns2vc_unet* h);

/* One call per state-dict tensor, keyed by the reference's parameter name
 * (load_state_dict, inference/infer_tool.py:24-29; model.py:819-829).  `data` is a
 * host or device pointer to contiguous fp32 of the given shape. */
int ns2vc_unet_load_weight(ns2vc_unet* h, const char* key, const void* data, const int64_t* shape, int ndim);
/* Checks that all tensors arrived, packs them for the chosen precision and uploads. */
int ns2vc_unet_finalize_weights(ns2vc_unet* h, int precision);
int ns2vc_unet_num_missing_weights(ns2vc_unet* h, char* first_missing, int buflen);

/* Plan options, by name; call before ns2vc_unet_prepare (an existing plan / workspace is dropped):
 *   "ln_linear" 1|0  LayerNorm by linearity (default 1) vs explicit normalisation passes
 *   "fold_ff"   1|0  ff.net.2 folded into proj_out at pack time (default 1) vs two launches
 *   "fuse_ffn"  1|0  GEGLU feed-forward + proj_out in ONE launch where eligible (16-bit precisions, dim <= 256; needs
 *                    ln_linear and fold_ff; default 1) vs the GEGLU GEMM + the folded GEMM
 * The environment variables NS2VC_LN_LINEAR / NS2VC_FOLD_FF / NS2VC_FUSE_FFN set the defaults at ns2vc_unet_create. */
int ns2vc_unet_set_option(ns2vc_unet* h, const char* name, int value);
/* LayerNorm-by-linearity health: the largest |mean| / std over every LayerNorm input row seen since the last read-out
 * (or since prepare).  The 16-bit modes round the raw row before centring, so their error on a row grows ~linearly
 * with this ratio (1 at ratio <~ 1; use "ln_linear" 0 when it is >> 10); in fp32 the variance E[x^2] - mean^2 loses
 * ~ratio^2 ulps.  The read-and-reset is enqueued on `stream` (the stream the evaluations ran on), never on the legacy
 * stream:  ns2vc_unet_ln_ratio      = enqueue + synchronise THAT stream + return the value;
 *          ns2vc_unet_ln_ratio_post = enqueue only (no host wait);
 *          ns2vc_unet_ln_ratio_poll = non-blocking: *out_ready = 1 and the value of the last post once it has completed. */
int ns2vc_unet_ln_ratio(ns2vc_unet* h, float* out_ratio, void* stream);
int ns2vc_unet_ln_ratio_post(ns2vc_unet* h, void* stream);
int ns2vc_unet_ln_ratio_poll(ns2vc_unet* h, float* out_ratio, int* out_ready);

/* Allocate workspace and build the launch plan for a (batch, frames, prompt frames) shape. */
int ns2vc_unet_prepare(ns2vc_unet* h, int B, int T, int Lp);
int ns2vc_unet_workspace_bytes(ns2vc_unet* h, size_t* out);

/* Step-invariant work, once per utterance batch: add_embedding(prompt)
 * (unet_1d_condition.py:869-870), the 32 cross-attention to_k/to_v(prompt)
 * (attention_processor.py:1019-1020), the content half of conv_in over cat([x, content])
 * (model.py:409, unet_1d_condition.py:943) and the mask -> bias conversion (:816-818).
 * content (B, content_channels, T); prompt (B, Lp, cross_dim); mask (B, Lp) or NULL. */
int ns2vc_unet_set_condition(ns2vc_unet* h, const float* content_bct, const float* prompt_blc, const uint8_t* mask_bl, void* stream);
/* The two halves of set_condition, for callers whose content and prompt change at different rates (the drop-in
 * nn.Module receives cat([x, content]) anew on every solver step, model.py:409, while the prompt tensor persists):
 * set_content = content half of conv_in only; set_prompt = cross-attention K/V, add_embedding (+ set_mask);
 * set_mask = the (B, Lp) keep-mask -> additive bias conversion only (NULL = no mask), for a mask tensor that is rebuilt
 * per call (model.py:412) over an unchanged prompt. */
int ns2vc_unet_set_content(ns2vc_unet* h, const float* content_bct, void* stream);
int ns2vc_unet_set_prompt(ns2vc_unet* h, const float* prompt_blc, const uint8_t* mask_bl, void* stream);
int ns2vc_unet_set_mask(ns2vc_unet* h, const uint8_t* mask_bl, void* stream);

/* One denoiser evaluation = Diffusion_Encoder.forward (model.py:403-415) ->
 * UNet1DConditionModel.forward (unet_1d_condition.py:743-1037) for the condition set above.
 * x (B, latent_channels, T); t (B) fp32; out (B, latent_channels, T). */
int ns2vc_unet_forward(ns2vc_unet* h, const float* x_bct, const float* t_b, float* out_bct, void* stream);

/* ---- sampling loop (replaces DPM_Solver.sample sampler/dpm_solver.py:1171-1213 and
 * UniPC.sample sampler/uni_pc.py:606-658 incl. the x_start wrapper :271-292/:170-191).
 * `coef` = host array [steps][NS2VC_NCOEF] built by ns2vc_amd.schedule.build_table. */
int ns2vc_sampler_load(ns2vc_unet* h, int steps, const float* coef_host);
/* x (B, latent_channels, T): x_T in, sample out.  use_graph != 0 replays one captured
 * hipGraph per step (no host sync inside the loop). */
int ns2vc_sampler_run(ns2vc_unet* h, float* x_inout_bct, int use_graph, void* stream);
/* The same loop in parts (ns2vc_sampler_run = begin + steps(all) + end), so that a second engine can take over in
 * mid-loop.  Use: MIXED PRECISION -- the last evaluations of a loop dominate the error of the sampled latent (the final
 * second-order update of DPM-Solver++(2M), dpm_solver.py:796-831, extrapolates over a large log-SNR step), so a 16-bit
 * engine runs steps [0, N-k) and an fp32 engine, prepared for the same shape / condition / table, runs the last k:
 *   begin(h16, x_T); steps(h16, N-k); handoff(h32, h16); steps(h32, k); end(h32, x_out).
 * handoff copies the fp32 solver state (x_e, x_bar, d1, m_prev, loop position) on `stream`. */
int ns2vc_sampler_begin(ns2vc_unet* h, const float* x_T_bct, void* stream);
int ns2vc_sampler_steps(ns2vc_unet* h, int n_steps, int use_graph, void* stream);
int ns2vc_sampler_end(ns2vc_unet* h, float* x_out_bct, void* stream);
int ns2vc_sampler_handoff(ns2vc_unet* dst, ns2vc_unet* src, void* stream);   /* both engines must hold the SAME table (compared by hash) */
/* x_e of a loop in progress (the point the next evaluation is taken at) without ending the loop: what a precision self-check
 * evaluates two engines on (ns2vc_amd.pipeline.Denoiser) */
int ns2vc_sampler_peek(ns2vc_unet* h, float* x_out_bct, void* stream);
/* workgroups of the attention launches of this engine whose optimistic pass (no per-tile maximum; r3) had to be repeated by the
 * exact pass since the last reset -- each of them paid the kernel twice.  Synchronises `stream`. */
int ns2vc_unet_attn_fallbacks(ns2vc_unet* h, unsigned long long* count, int reset, void* stream);
/* ABI v5.  Workgroups of the cooperative GroupNorm prologue (ns2vc_gemm_args.gnp_sync, option `gn_coop`) that waited in vain for a
 * sibling since the plan was built (or the last reset) and built all their rows themselves: a performance counter -- the results do
 * not depend on it.  Synchronises `stream`. */
int ns2vc_unet_gn_coop_alone(ns2vc_unet* h, unsigned long long* count, int reset, void* stream);

/* ---- introspection for tests / profiling -------------------------------------------- */
int ns2vc_unet_set_debug(ns2vc_unet* h, int enable);  /* keep a copy of every block output; drops the plan: call before prepare() */
int ns2vc_unet_num_taps(ns2vc_unet* h);
int ns2vc_unet_tap_info(ns2vc_unet* h, int idx, char* name, int buflen, int* rows, int* cols);
int ns2vc_unet_tap_read(ns2vc_unet* h, int idx, float* host_dst);   /* synchronous, [rows][cols] channels-last */
int ns2vc_unet_num_launches(ns2vc_unet* h, int* per_forward, int* per_condition);
/* which: 0 = per-step forward plan, 1 = condition plan.  kind: 0 other, 1 implicit GEMM, 2 attention,
 * 3 norm statistics, 4 copy.  flops / bytes: algorithmic work of that launch. */
int ns2vc_unet_op_info(ns2vc_unet* h, int which, int idx, char* name, int buflen, int* kind, double* flops, double* bytes);
/* Per-launch timing of the per-step plan: each launch repeated `reps` times between one hipEvent pair on
 * `stream`; ms[i] = average milliseconds of launch i (n_ms >= launches). Leaves garbage in the workspace. Synchronous. */
int ns2vc_unet_profile_forward(ns2vc_unet* h, float* ms, int n_ms, int reps, void* stream);

/* ---- raw device helpers so tests/bench can drive the ABI without torch ------------------ */
int ns2vc_dev_malloc(void** out, size_t bytes);
int ns2vc_dev_free(void* p);
int ns2vc_memcpy_h2d(void* dst, const void* src, size_t bytes);
int ns2vc_memcpy_d2h(void* dst, const void* src, size_t bytes);
int ns2vc_dev_sync(void);
int ns2vc_stream_create(void** out);
int ns2vc_stream_destroy(void* stream);
int ns2vc_stream_sync(void* stream);
/* ABI v6: a stream whose kernels only run on the CUs whose bit is set (hipExtStreamCreateWithCUMask: bit i % 32 of word i / 32 = CU i of the
 * device's enumeration, XCD-interleaved on MI355X).  For partitioning the chip between the denoiser and the PyTorch stages of a pipeline
 * (ns2vc_amd/pipeline.py) and for the placement tests of the cooperative GroupNorm prologue. */
int ns2vc_stream_create_cu_mask(void** out, const uint32_t* mask_words, int n_words);
int ns2vc_event_create(void** out);
int ns2vc_event_destroy(void* ev);
int ns2vc_event_record(void* ev, void* stream);
int ns2vc_event_elapsed_ms(void* start, void* stop, float* ms);  /* synchronises on `stop` */

/* ---- kernel-level entry points (unit-tested one by one through this ABI) -------------- */
typedef struct ns2vc_gemm_args {  /* implicit GEMM: conv1d k3/k1 (stride 1, stride 2, nearest-up), linear */
  /* "operand" tensors are stored in the engine's MFMA operand type: fp32 (precision 0), bf16 (1) or fp16 (2) */
  const void* a0; const void* a1; /* channels-last operand-typed sources; a1 = second half of a no-copy concat or NULL */
  int32_t lda0, lda1, c0, c1;
  int32_t B, Tin, Tout, M;        /* M = B*Tout output rows */
  int32_t taps, tmode;            /* taps 1|3; tmode 0 same, 1 stride-2, 2 nearest-upsample-then-conv */
  const void* w; int32_t K;       /* packed weights [N][K] from ns2vc_pack_weight */
  int32_t N;
  const float* bias;
  const float* res; int32_t ldres;/* fp32 residual added in the epilogue, or NULL */
  int32_t geglu;
  float* out_f32; int32_t ldo_f32;/* fp32 result (residual stream), or NULL */
  void* out_op; int32_t ldo_op;   /* operand-typed copy of the result (feeds the next GEMM / attention), or NULL */
  /* optional GroupNorm statistics of the result, accumulated by the epilogue: int64 fixed point
   * [B][N/16][2] = (sum * 2^28, sum of squares * 2^16) per batch item and 16-channel block; must be zeroed
   * by the caller; needs Tout >= 32, geglu == 0 */
  long long* stats;
  /* optional second K segment (a fused 1x1 conv on another operand tensor, e.g. the resnet shortcut):
   * K = taps*(c0+c1) + c2, out += A2[m, :] * W[:, taps*(c0+c1):]; same row mapping, centre tap */
  const void* a2; int32_t lda2, c2;
  /* LayerNorm by linearity (attention.py:83,102,118 without a normalisation pass):
   *   producer: rowstats != NULL -> the epilogue stores (sum, sum of squares) of every RESULT row per 64-column slice:
   *             fp32 [M][N/64][2], one writer per slot (nothing to zero; geglu == 0, N % 128 == 0);
   *   consumer: ln_stats != NULL -> the A rows are the RAW x of LayerNorm(x) (gamma/beta folded into w/bias) and the
   *             epilogue applies  out = rstd[m] * (acc[m][n] - mean[m] * ln_wsum[n]) + bias[n]  with mean/rstd over
   *             ln_dim channels (128..512, multiple of 128) from the producer's [M][ln_dim/64][2] pairs; ln_wsum[n] = sum_k w[n][k]
   *             of the packed, rounded weights (ns2vc_weight_rowsum). */
  float* rowstats;
  const float* ln_stats; const float* ln_wsum; float ln_eps; int32_t ln_dim;
  unsigned* ln_health;            /* optional (consumer): atomicMax of the bits of |mean| * rstd over the rows, or NULL */
  /* optional GroupNorm-apply prologue (ABI v4; replaces a separate ns2vc_k_groupnorm launch in front of this GEMM, the
   * `group_norm` (+ time scale/shift) (+ SiLU) of resnet.py:606-629 / transformer_1d.py:268): when gnp_x != NULL every
   * workgroup FIRST writes act(GroupNorm(gnp_x)) for exactly the rows its tile will read (its output rows, plus one row
   * either side for taps == 3) into a0 -- same arithmetic as ns2vc_k_groupnorm, bit-identical rows -- and then runs as
   * usual, reading them back through its own L2.  a0 must be writable; c1 == 0, tmode == 0, Tin == Tout, N % 128 == 0,
   * c0 <= 1024, (c0 / gnp_G) % 16 == 0, gnp_G <= 8, Tin >= 66.  gnp_x fp32 [B*Tin][gnp_ldx]; gnp_stats = int64
   * [B][c0/16][2] as left by a producer's `stats`; gnp_gamma / gnp_beta [c0]; gnp_temb (or NULL) points at row 0 of the
   * per-item (scale | shift) pairs: scale at [b*gnp_ldtemb + c], shift at [b*gnp_ldtemb + c0 + c]. */
  const float* gnp_x; int32_t gnp_ldx;
  const long long* gnp_stats; const float* gnp_gamma; const float* gnp_beta;
  const float* gnp_temb; int32_t gnp_ldtemb;
  float gnp_eps; int32_t gnp_G, gnp_silu;
  /* ABI v5 / v6, optional: [ceil(M / 64)] 64-bit arrival words (8-byte aligned), owned by this call site and ZERO when a launch starts (the
   * engine keeps them in the statistics pool, cleared at the start of every forward).  With them (and 2 .. 15 column tiles) the column tiles
   * of a row block -- dispatched as neighbours on one XCD -- build a share of the block's rows each and wait for the others' (bounded),
   * instead of each building all of them.  Same values either way.  ABI v6: the placement is checked, not assumed -- an arrival adds 1 to
   * bits 0-15 and 1 to the 4-bit count of ITS XCC id (bits 16 + 4 * HW_REG_XCC_ID ..), and a workgroup uses its siblings' rows only when
   * all arrivals it sees carry its own XCC id; otherwise (and when it waits in vain) it builds every row itself (gnp_alone).  Bit 62 set:
   * "do not wait" (tests).  Needs a0 128-byte aligned and whole 128-byte lines per row (lda0 * operand size % 128 == 0). */
  unsigned* gnp_sync;
  unsigned* gnp_alone;            /* optional: += 1 per workgroup that waited in vain and built every row itself */
  /* ABI v5, optional: the normalised input is the channel concat of TWO tensors (resnet.py:591 on torch.cat([h, skip]) in the up
   * blocks): the last gnp_c1 of the c0 channels come from gnp_x1 (fp32 [B*Tin][gnp_ldx1], statistics gnp_stats1 [B][gnp_c1/16][2]),
   * the first c0 - gnp_c1 from gnp_x / gnp_stats; gamma / beta / temb index the concatenated channels; groups may straddle the
   * two.  c0 <= 1024, gnp_c1 % 16 == 0.  gnp_raw (or NULL): the un-normalised rows in the operand type, same layout as a0 --
   * the copy a 1x1 shortcut conv reads later. */
  const float* gnp_x1; int32_t gnp_ldx1, gnp_c1;
  const long long* gnp_stats1;
  void* gnp_raw;
  /* ABI v6: kernel choice.  0 = automatic (k = 3 / stride-1 convolutions with Tin >= 66 run on the tap-sharing kernel, convts.hip: the
   * activation chunk of the three taps is loaded once; same result to fp32 rounding, another summation order over K), 1 = never that kernel. */
  int32_t algo;
  /* ABI v6, optional (k = 3 / stride-1 launches only): the same weights as `w` in the tile-major layout of ns2vc_pack_conv3_tiled.  The
   * tap-sharing kernel then reads every (64-column group, step) weight block as 8 KB of consecutive bytes instead of 64 row segments
   * K * 2 bytes apart (-1.1 % of the step: the tiles come from beyond L2 every step); `w` is still required (other kernels, fallbacks). */
  const void* w_tiled;
  /* ABI v7.  algo also takes 2: the tap-sharing kernel with the materialising GroupNorm prologue; with algo = 0 and gnp_x set it MAY normalise
   * inside its K loop instead (nothing is written to a0 then; same values).
   * ABI v7, optional (k = 3 / stride-1 launches of the tap-sharing kernel with ONE column tile, i.e. N == 128: the denoiser's conv_out): the
   * solver update of the sampling loop (sampler/uni_pc.py:471-588, sampler/dpm_solver.py:433-442, 547-580 as restated in
   * ns2vc_amd/schedule.py) applied to the result in the epilogue instead of by a separate ns2vc_k_solver_update launch: with x0 = the
   * conv's result (bias added), row *sol_step of the coefficient table sol_coef [steps][sol_ncoef] and the solver state
   * (sol_xe, sol_xbar, sol_d1, sol_mprev: fp32 [M][sol_ld], updated in place; sol_xe_op: the operand-typed copy of xe the next
   * evaluation's conv_in reads) exactly the arithmetic of ns2vc_k_solver_update, element for element.  out_f32 / out_op may be NULL. */
  const float* sol_coef; const int32_t* sol_step; int32_t sol_ncoef;
  float* sol_xe; void* sol_xe_op; float* sol_xbar; float* sol_d1; float* sol_mprev; int32_t sol_ld;
  /* ABI v7: column tile of the tap-sharing kernel for this launch: 0 = the library's heuristic for the current device, 64 | 128 = fixed.  The engine
   * decides per plan (from ITS device's CU count) and passes the choice along, so that two engines on different devices -- or a debug hook called
   * between plan build and launch -- cannot change a launch's tiling relative to what its plan assumed. */
  int32_t conv_bn;
  /* ABI v7, 16-bit precisions: hi + lo operand pairs (x = hi(x) + lo(x) with hi = the operand type's rounding of x, lo = the rounding of the rest).  A GEMM
   * over [hi | lo | hi] activation columns against [hi(w) | hi(w) | lo(w)] weight columns (the caller packs them so) gives x * w to ~2^-21 relative instead of
   * 2^-11 for 3x the K -- the engine's `split_io` option uses it for conv_in and conv_out (23 % of the fp16 forward error's energy, profiles/r06_error_budget.txt).
   *   gnp_pair = 1: the GroupNorm prologue normalises c0 / 2 channels and writes the pair: hi at a0[r][c], lo at a0[r][c0 / 2 + c]; the launch may then
   *                  carry a1 = a0, c1 = c0 / 2 (the hi plane once more) -- the one case where c1 != 0 goes with gnp_x.  Tap-sharing kernel with the materialising
   *                  prologue only (its own instantiation; other kernels refuse the flag).  Without gnp_x a pair is just a wider operand tensor: any kernel takes it.
   *   sol_op_pair = 1: sol_xe_op rows hold such a pair, [hi(sol_ld) | lo(sol_ld)], row stride 2 * sol_ld (what ns2vc_k_solver_update writes for the engine). */
  int32_t gnp_pair, sol_op_pair;
} ns2vc_gemm_args;

typedef struct ns2vc_attn_args {
  const void* q; const void* k; const void* v; /* operand-typed rows; head h lives at columns [h*hd, (h+1)*hd) */
  int32_t ldq, ldk, ldv;
  int32_t B, H, Lq, Lk;
  const float* bias;              /* additive [B][Lk] or NULL */
  float scale;
  void* out; int32_t ldo;         /* operand-typed */
  int32_t pv_fp8;                 /* 16-bit precisions only: 1 = the PV product on the fp8 MFMA (V and the probabilities rounded to OCP e4m3) */
  int32_t exact_only;             /* 1 = skip the optimistic pass (no per-tile maximum) and take the exact pass directly */
  unsigned* fallbacks;            /* device counter or NULL: += 1 per workgroup whose optimistic pass had to be repeated exactly */
} ns2vc_attn_args;

/* Fused feed-forward + proj_out of one transformer block (attention.py:178-203 GEGLU feed-forward, transformer_1d.py:287-295),
 * 16-bit precisions, dim 128 | 256, T >= 64:
 *   out = [Wpo W2 | Wpo] [GEGLU(LayerNorm(y) W1^T + b1) | y] + (Wpo b2 + bpo) + res
 * `yn` = the RAW rows y in the operand type with their LayerNorm-by-linearity statistics `ln_stats` (as a producer GEMM's
 * `rowstats` leaves them); `wstream` from ns2vc_pack_ffn; `consts` [8*dim][2] = (sum_k of the rounded packed W1 row, folded
 * bias) in packed row order (per 32 hidden units: 32 value rows, then their 32 gate rows); `bias2` [dim]. */
typedef struct ns2vc_ffn_args {
  const void* yn; int32_t ldy;
  const float* ln_stats; float ln_eps;
  const void* wstream; const float* consts; const float* bias2;
  const float* res; int32_t ldres;          /* fp32 block residual x */
  float* out_f32; int32_t ldo_f32;          /* fp32 result, or NULL */
  void* out_op; int32_t ldo_op;             /* operand-typed copy, or NULL */
  long long* stats;                         /* optional GroupNorm statistics of the result, as in ns2vc_gemm_args */
  int32_t B, T, M, dim;                     /* M = B*T rows */
  unsigned* ln_health;                      /* optional, as in ns2vc_gemm_args */
  /* optional pre-stage (attn2.to_out + residual, attention_processor.py:1040-1050): y = pre_a Wo^T + pre_bias + pre_res is
   * computed inside the kernel and never stored; yn / ln_stats are ignored.  wstream must then come from ns2vc_pack_ffn_pre. */
  const void* pre_a; int32_t pre_lda;       /* attention output rows, operand-typed [M][pre_lda] */
  const float* pre_bias;                    /* [dim] */
  const float* pre_res; int32_t pre_ldres;  /* fp32 residual stream before the attention [M][pre_ldres] */
  /* ABI v7, optional, with the pre-stage only: the prompt cross-attention itself (F.scaled_dot_product_attention of attn2,
   * attention_processor.py:1032 with the mask bias of unet_1d_condition.py:816-818) computed INSIDE the kernel for the workgroup's 64 tokens -- the
   * k | v of the prompt are hoisted per utterance, so attn2's SDPA is token-local work: one wave per head (8 heads), two passes over the Lk keys (row
   * maximum, then exp2 / P V with the probabilities rounded to the operand type), the result written straight into the token panel; pre_a is ignored
   * and no attention-output tensor exists.  att_q = attn2.to_q rows [M][att_ldq]; att_kv = ns2vc_k_xattn_pack's image of this layer's k | v rows (MFMA
   * fragments in consumption order: every load of the kernel is 1 KB of consecutive bytes per wave); att_bias = additive mask bias [B][att_Lk] or NULL;
   * att_scale = 1/sqrt(dim/8).  Token blocks are then cut per batch item (ceil(T / 64) workgroups per item). */
  const void* att_q; int32_t att_ldq;
  const void* att_kv; const float* att_bias; float att_scale; int32_t att_Lk;
} ns2vc_ffn_args;
/* w1_packed [8*dim][dim]: LayerNorm-folded ff.net.0 rows in the packed (32 value | 32 gate) order; w2f [dim][5*dim] =
 * [Wpo W2 | Wpo]; both fp32 host, row-major.  Returns the device tile stream the kernel consumes. */
int ns2vc_pack_ffn(const float* w1_packed_host, const float* w2f_host, int dim, int precision, void** out_stream_dev);
/* the same with the pre-stage matrix w0 [dim][dim] (attn2.to_out) in front */
int ns2vc_pack_ffn_pre(const float* w1_packed_host, const float* w2f_host, const float* w0_host, int dim, int precision, void** out_stream_dev);
int ns2vc_k_ffn(const ns2vc_ffn_args* a, int precision, void* stream);
/* ABI v7: the k | v image ns2vc_ffn_args.att_kv reads.  k, v = operand-typed rows [B*Lk][ldk / ldv] (head h at columns h*hd ..), 8 heads, hd = 16 | 32.
 * out, per (batch item, head, tile of 32 keys): hd/16 K fragments and 2 V^T fragments of 1 KB each (lane l of a wave reads bytes 16 l ..): K fragment s, lane
 * (key = l % 32, half = l / 32) = k[key][16 s + 8 half .. + 7]; V^T fragment j, lane (d = l % 32, half) = the values v[key(slot)][d] of slots 16 j + 8 half .. + 7,
 * where inside every group of 16 slots key bits 2 and 3 are swapped (the order in which the score MFMA leaves a lane's probabilities); hd 16: row d = 16 is
 * all ones (the softmax denominator rides the P V product), rows above zero; keys beyond Lk zero.  ns2vc_xattn_pack_bytes = size of `out`. */
size_t ns2vc_xattn_pack_bytes(int B, int Lk, int hd);
int ns2vc_k_xattn_pack(const void* k, int ldk, const void* v, int ldv, int B, int Lk, int hd, void* out, int precision, void* stream);

/* r5: token-stationary GEGLU projection (csrc/geglu.hip; 16-bit precisions, dim 384): h = (n W1v^T + b1v) * gelu(n W1g^T + b1g),
 * n = LayerNorm(y) by linearity -- BasicTransformerBlock.ff.net.0 (reference unet1d/attention.py:178-203, GEGLU 206-301) of the blocks
 * whose hidden tensor does not fit the fused ns2vc_k_ffn.  128 tokens stay in registers, four workgroups per token block sweep a quarter of
 * the hidden units each; out_op [M][ldo] (>= 4 dim columns) operand-typed.  yn = the RAW operand copy of y [M][ldy]; ln_stats as in
 * ns2vc_gemm_args.ln_stats ([M][dim/64] (sum, sumsq)); wstream / consts from ns2vc_pack_geglu. */
typedef struct ns2vc_geglu_args {
  const void* yn; int32_t ldy;
  const float* ln_stats; float ln_eps;
  const void* wstream; const float* consts;
  void* out_op; int32_t ldo;
  int32_t M, dim;
  unsigned* ln_health;                      /* optional, as in ns2vc_gemm_args */
} ns2vc_geglu_args;
/* w1_packed [8*dim][dim], bias1_packed [8*dim] (or NULL): LayerNorm-folded ff.net.0 rows / bias in the packed (32 value | 32 gate) order,
 * fp32 host.  Returns the device tile stream and the device constants ((rowsum of the rounded row, bias) per stream row). */
int ns2vc_pack_geglu(const float* w1_packed_host, const float* bias1_packed_host, int dim, int precision, void** out_stream_dev, float** out_consts_dev);
int ns2vc_k_geglu(const ns2vc_geglu_args* a, int precision, void* stream);
/* the same packing on the host only (no GPU needed; tests): stream_out [8*dim*dim] operand-typed bit patterns in consumption order
 * ([quarter][unit block][K tile][128 rows][64 k], rows of every 32-unit group permuted, 16-byte chunks XOR-swizzled), consts_out [8*dim][2] */
int ns2vc_pack_geglu_host(const float* w1_packed_host, const float* bias1_packed_host, int dim, int precision, uint16_t* stream_out, float* consts_out);

/* Two token-local GEMMs with a LayerNorm in between, in one launch (csrc/rowchain.hip; 16-bit precisions, dim 128 / 256):
 *   y = A W1^T + bias1 (+ res)  -> out1_f32 (optional);   z = LayerNorm(y) W2'^T + b2'  -> out2_op   (gamma/beta folded into W2' / b2')
 * Replaces Transformer2DModel.proj_in + BasicTransformerBlock.norm1 + attn1.to_q|to_k|to_v (transformer_1d.py:270-279,
 * attention.py:130-140; n2 = 3 dim) and attn1.to_out + residual + norm2 + attn2.to_q (attention_processor.py:1040-1050,
 * attention.py:141-160; n2 = dim).  a_op [M][lda] operand rows; wstream from ns2vc_pack_rowchain; consts2 [n2][2] =
 * (sum_k of the rounded W2' row, folded bias).  res may alias out1_f32 element for element. */
typedef struct ns2vc_rowchain_args {
  const void* a_op; int32_t lda;
  const void* wstream; const float* bias1; const float* consts2;
  const float* res; int32_t ldres;          /* optional fp32 residual added to y */
  float* out1_f32; int32_t ldo1;            /* fp32 y, or NULL */
  void* out2_op; int32_t ldo2;              /* operand-typed z [M][ldo2] */
  float ln_eps; int32_t M, dim, n2;
  unsigned* ln_health;                      /* optional, as in ns2vc_gemm_args */
  /* optional GroupNorm prologue: A = GroupNorm(gn_x) (affine, no activation) built inside the kernel; a_op is ignored.
   * gn_x fp32 [M][ldx]; gn_stats = the producer's per-(batch item, 16-channel block) int64 (sum, sumsq) as left by
   * ns2vc_gemm_args.stats; gn_gamma / gn_beta [dim]; T = rows per batch item (>= 64); G = groups (dim/G a multiple of 16) */
  const float* gn_x; int32_t ldx;
  const long long* gn_stats; const float* gn_gamma; const float* gn_beta;
  float gn_eps; int32_t T, G;
  /* r4: N-sliced stage 2 (dim 384): `slices` = 2 -> two workgroups per token block, each repeats stage 1 and computes one slice of the stage-2
   * rows (5 + 4 of the 9 row blocks of q|k|v, 2 + 1 of the 3 of to_q); wstream / consts2 from ns2vc_pack_rowchain_sliced.  0 / 1 = one workgroup. */
  int32_t slices;
} ns2vc_rowchain_args;
/* w1 [dim][dim], w2 [n2][dim]: fp32 host, row-major.  Returns the device tile stream the kernel consumes. */
int ns2vc_pack_rowchain(const float* w1_host, const float* w2_host, int dim, int n2, int precision, void** out_stream_dev);
int ns2vc_pack_rowchain_sliced(const float* w1_host, const float* w2_host, int dim, int n2, int slices, int precision, void** out_stream_dev);
int ns2vc_k_rowchain(const ns2vc_rowchain_args* a, int precision, void* stream);
int ns2vc_debug_set_geglu_min_rows(int rows); /* tests / tuning: row count from which the planner picks ns2vc_k_geglu over the GEMM (default 4608; < 0 restores it); takes effect at the next plan build */
int ns2vc_debug_set_attn_keys(int keys); /* tests / tuning: 128 selects the 128-key K/V tile kernels (16-bit precisions, hd 16 / 32); 0 or 64 = the default 64-key tiles */
int ns2vc_debug_set_attn_optimistic(int on); /* tests: 0 = every attention workgroup takes the exact (per-tile maximum) pass only; 1 = default */
int ns2vc_debug_set_rowchain_tokens(int nt); /* tests: force 64-token (1) / 128-token (2, dim 128 only) workgroups; 0 = heuristic */

/* operand-typed conversions for tests: fp32 host [n] -> device operand buffer and back */
int ns2vc_to_operand(const float* host, size_t n, int precision, void** out_dev);
int ns2vc_from_operand(const void* dev, size_t n, int precision, float* host);
/* the host-side rounding used for weight packing (round to nearest even; fp16 overflow -> inf): host in, host out, no GPU needed */
int ns2vc_round_to_operand(const float* host_in, size_t n, int precision, float* host_out);
int ns2vc_pack_weight(const float* rows_host, int N, int K, int precision, void** out_dev); /* [N][K] fp32 host -> device, engine dtype */
int ns2vc_k_gemm(const ns2vc_gemm_args* a, int precision, void* stream);
/* ABI v6: rows [N][3 * ctot + c2] fp32 host (k = tap * ctot + c, then the c2 columns of a fused 1x1 segment) -> device, engine dtype, N padded to 128,
 * tile-major: [N / 64][steps][64 rows][128 B] in the tap-sharing kernel's step order, pre-swizzled (csrc/convts.hip pack_conv3_tiled). */
int ns2vc_pack_conv3_tiled(const float* rows_host, int N, int ctot, int c2, int precision, void** out_dev);
int ns2vc_weight_rowsum(const float* rows_host, int N, int K, int precision, float** out_dev); /* [N] fp32: sum_k round_to_operand(rows[n][k]) */
int ns2vc_debug_set_gemm_trace(void* dev_u64_blocks_x8); /* tuning: per-workgroup s_memtime stamps of the next GEMM launches; NULL = off */
int ns2vc_debug_poison(unsigned pattern, int lds_bytes, void* stream); /* test tool: leave `pattern` in every CU's LDS (first lds_bytes) and in vector registers, as a foreign kernel would */
/* ABI v6 (diagnostic): where the 256-thread blocks of an n_blocks launch on `stream` ran: out_host[2 i] = XCC id (HW_REG_XCC_ID),
 * out_host[2 i + 1] = HW_REG_HW_ID of block i; `spin` > 0 keeps blocks resident for a while so the grid spreads over the CUs the stream may use. */
int ns2vc_debug_placement(void* stream, int n_blocks, int spin, uint32_t* out_host);
int ns2vc_debug_set_gemm_tile(int bm, int bn, int stages); /* force the GEMM tile: stages 2..4 = 4-wave kernel ring depth, 12|13 = 8-wave K-split kernel ring 2|3; 0,0,0 = heuristic; (-1,0,0) = heuristic without the loader/consumer tiles, (-2,0,0) = with them again; ABI v6: (128, 64|128, 54|58) = the tap-sharing conv kernel with 4|8 loader waves (k = 3 launches only), (-3,0,0) = never that kernel, (-4,0,nl) = heuristic with it again, nl loader waves (0 = default) */
int ns2vc_k_attention(const ns2vc_attn_args* a, int head_dim, int precision, void* stream);
/* GroupNorm (+ optional resnet time scale/shift, + optional SiLU) of a (possibly concatenated) fp32 tensor,
 * written as an operand tensor [B*T][c0+c1]; raw_op (optional) receives the un-normalised concat. Synchronous. */
int ns2vc_k_groupnorm(const float* a0, int lda0, int c0, const float* a1, int lda1, int c1, int B, int T, int G, float eps,
                      const float* gamma, const float* beta, const float* temb, int ldtemb, int temb_off, int silu,
                      void* out_op, void* raw_op, int precision, void* stream);
/* the engine's form of the same launch: statistics = the int64 fixed-point sums a producer GEMM's epilogue left
 * (ns2vc_gemm_args.stats, [B][c0/16][2]); one source, asynchronous on `stream`.  What ns2vc_gemm_args.gnp_* reproduces bit for bit. */
int ns2vc_k_groupnorm_stats(const float* a0, int lda0, int c0, const long long* stats0, int B, int T, int G, float eps, const float* gamma,
                            const float* beta, const float* temb, int ldtemb, int temb_off, int silu, void* out_op, int precision, void* stream);
/* ABI v5: the same for the channel concat of two tensors (the up blocks' torch.cat([h, skip])), each with its own statistics, plus the
 * optional un-normalised operand copy.  What ns2vc_gemm_args.gnp_x1 / gnp_raw reproduce bit for bit. */
int ns2vc_k_groupnorm_stats2(const float* a0, int lda0, int c0, const long long* stats0, const float* a1, int lda1, int c1, const long long* stats1,
                             int B, int T, int G, float eps, const float* gamma, const float* beta, const float* temb, int ldtemb, int temb_off,
                             int silu, void* out_op, void* raw_op, int precision, void* stream);
/* LayerNorm without affine (gamma/beta are folded into the consumer's weights): fp32 rows -> operand rows */
int ns2vc_k_layernorm_apply(const float* x, int ldx, int M, int C, float eps, void* out_op, int precision, void* stream);
int ns2vc_k_nct_to_btc(const float* src, int C, int T, int B, float* dst, int ldd, int cpad, void* stream);
int ns2vc_k_btc_to_nct(const float* src, int lds, int C, int T, int B, float* dst, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* NS2VC_HIP_H */
