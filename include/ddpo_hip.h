/* ddpo_hip.h — C ABI of libddpo_hip.so: the MI355X (gfx950) kernels behind the DDPO hot path.
 *
 * The reference (jannerm/ddpo) is pure Python/JAX and has no FFI boundary of its own; every entry point
 * below names the reference call site (file:line under /root/reference) whose arithmetic it replaces.
 * Conventions: all pointers are DEVICE pointers owned by the caller unless marked "host"; tensors are
 * fp32 unless stated; `stream` is a hipStream_t passed as void*; functions never allocate, never
 * synchronise and never throw; they return 0 on success, DDPO_EINVAL (-1) for a bad argument and
 * DDPO_ELAUNCH (-2) if the launch was rejected.  Activations are NHWC ("pixel-major") inside the U-Net,
 * latents/trajectories are NCHW at rest exactly as the reference stores them.
 */
#ifndef DDPO_HIP_H
#define DDPO_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define DDPO_OK 0
#define DDPO_EINVAL (-1)
#define DDPO_ELAUNCH (-2)

#define DDPO_PRED_EPSILON 0
#define DDPO_PRED_V 1
#define DDPO_PRED_SAMPLE 2

int ddpo_abi_version(void);
size_t ddpo_sizeof_gemm_desc(void);     /* for binding self-checks (ctypes / cffi struct mirrors) */
size_t ddpo_sizeof_ddim_consts(void);
/* ABI v10.  Cumulative HOST-side launch counts of the bf16-MFMA GEMM / conv template per tile class since the library was loaded (out_host:
 * n >= 8 counters): [0] 256x320 "tall", [1] 128x320 "wide", [2] 128x128, [3] 128x64, [4] launches that fell back to the pointer-addressed
 * generic loader, [5] split-K reduce passes, [6] launches of the f16mx (NPASS = 4) instantiation (also counted under their tile), [7] 0.
 * Introspection only (which instantiation a layer geometry is routed to is otherwise invisible to the caller); kernels replayed from a
 * captured HIP graph are counted once, at capture. */
int ddpo_gemm_tile_launch_counts(unsigned long long* out_host, int n);

/* ---- PRNG: jax.random (Threefry-2x32) -------------------------------------------------------------
 * jax.random.split / PRNGKey bookkeeping, pipeline/policy_gradient.py:51,201,244-245 and
 * ddpo/diffusers_patch/pipeline_flax_stable_diffusion.py:196,232,252.  Host-side, integer-exact. */
int ddpo_threefry_bits_host(uint32_t k0, uint32_t k1, int64_t n, uint32_t* out_host);
/* jax.random.normal(key, shape, float32) flattened to n elements:
 * pipeline_flax_stable_diffusion.py:197 (initial latents), scheduling_ddim_flax.py:347 (step noise).
 * bits_out (optional, may be NULL) receives the raw uint32 words for bit-exact checks. */
int ddpo_threefry_normal(uint32_t k0, uint32_t k1, float* out, uint32_t* bits_out, int64_t n, void* stream);

/* ---- DDIM scheduler ------------------------------------------------------------------------------
 * Scheduler constants shared by both DDIM kernels: alphas_cumprod is the device table ᾱ[0..T_train),
 * final_alpha_cumprod / step_ratio / eta / pred_type as in scheduling_ddim_flax.py:154-158,279-281,325. */
typedef struct {
  const float* alphas_cumprod;   /* device, num_train_timesteps floats */
  int num_train_timesteps;
  int step_ratio;                /* num_train_timesteps / num_inference_steps */
  float final_alpha_cumprod;
  float eta;
  int pred_type;                 /* DDPO_PRED_* */
} ddpo_ddim_consts;

/* Sampling-mode FlaxDDIMScheduler.step fused with the CFG combine:
 * pipeline_flax_stable_diffusion.py:226-235 + scheduling_ddim_flax.py:279-359.
 *   eps = eps_u + g (eps_c - eps_u);  x_next = mu(eps, x, t) + sigma z;  logp[b] = mean_chw N(x_next; mu, sigma_c)
 * eps_u/eps_c/x/z/x_next: (B, chw) contiguous; ts: (B,) int32 device; logp: (B,). */
int ddpo_ddim_step_fwd(const float* eps_u, const float* eps_c, const float* x, const float* z,
                       const int32_t* ts, float guidance_scale, const ddpo_ddim_consts* c,
                       float* x_next, float* logp, int B, int chw, void* stream);

/* Scoring-mode step + PPO-clip loss + its gradient w.r.t. the two U-Net outputs:
 * ddpo/training/policy_gradient.py:95-125 (forward), jax.grad of it (:138-139) down to d eps_c / d eps_u.
 * per_sample: (B,4) = {log_prob, ratio, max(unclipped,clipped), clipped?}; info: 3 floats
 * {approx_kl, clipfrac, loss} (:132-134).  If train_cfg == 0, eps_u/d_eps_u may be NULL. */
int ddpo_ddim_logprob_ppo_fwd_bwd(const float* eps_c, const float* eps_u, const float* x,
                                  const float* x_next, const int32_t* ts, const float* old_logp,
                                  const float* advantages, float guidance_scale, float clip_range,
                                  int train_cfg, const ddpo_ddim_consts* c, float* d_eps_c,
                                  float* d_eps_u, float* per_sample, float* info, int B, int chw,
                                  void* stream);
/* The same over k micro-batches at once: rows [j*group, (j+1)*group) form micro-batch j (B % group == 0).  Each
 * micro-batch's loss is the mean over ITS `group` rows, so the gradients written are exactly those k separate calls
 * would write (AccumulatingTrainState sums them anyway, ddpo/training/policy_gradient.py:32-48), and
 * info is (B/group, 3): one {approx_kl, clipfrac, loss} per micro-batch.  group == B is the call above. */
int ddpo_ddim_logprob_ppo_fwd_bwd_grouped(const float* eps_c, const float* eps_u, const float* x,
                                          const float* x_next, const int32_t* ts, const float* old_logp,
                                          const float* advantages, float guidance_scale, float clip_range,
                                          int train_cfg, const ddpo_ddim_consts* c, float* d_eps_c,
                                          float* d_eps_u, float* per_sample, float* info, int B, int group,
                                          int chw, void* stream);

/* ---- optimizer: optax.chain(clip_by_global_norm, adamw(mu_dtype=bf16)) + AccumulatingTrainState ----
 * pipeline/policy_gradient.py:130-150; ddpo/training/policy_gradient.py:32-48. */
/* RWR baseline (reward-weighted regression, /root/reference/ddpo/training/diffusion.py:19-90).
 * ddpo_rwr_noisy_latents: moments (B,h,w,2C) NHWC = the stored VAE posterior (mean | logvar), e1 (B,h,w,C) NHWC and noise (B,C,h,w)
 *   NCHW standard normals, ts (B) int32 -> latents = (mean + exp(clip(logvar,-30,20)/2) e1) * scale and
 *   noisy = sqrt(acp[t]) latents + sqrt(1 - acp[t]) noise, both (B,C,h,w) NCHW (FlaxDDPMScheduler.add_noise).
 * ddpo_rwr_mse_fwd_bwd: eps_c / eps_u / noise (B, chw); noise_pred = eps_u + g (eps_c - eps_u) if train_cfg else eps_c;
 *   per_sample (B,2) = {mean_chw (noise - noise_pred)^2, w_b * that}; *loss = sum_b w_b loss_b with w_b = weights[b], or 1/B when
 *   weights == NULL (the batch mean); d_eps_c / d_eps_u = d loss / d eps (closed form).  One workgroup per sample, no atomics. */
int ddpo_rwr_noisy_latents(const float* moments, const float* e1, const float* noise, const int32_t* ts,
                           const float* alphas_cumprod, int num_train_timesteps, float scale, float* latents, float* noisy,
                           int B, int C, int hw, void* stream);
int ddpo_rwr_mse_fwd_bwd(const float* eps_c, const float* eps_u, const float* noise, const float* weights, float guidance_scale,
                         int train_cfg, float* d_eps_c, float* d_eps_u, float* per_sample, float* loss, int B, int chw,
                         void* stream);
/* out_sq (device double, must be zeroed by the caller or zero_first=1) += sum g^2 */
int ddpo_grad_sqnorm(const float* g, int64_t n, double* out_sq, int zero_first, void* stream);
/* One update over flat buffers.  g holds the SUM of accumulated grads; inv_n_acc = 1/(n_acc+1);
 * sqnorm_of_sum = device double holding sum(g^2) of that SUM (the kernel applies inv_n_acc itself).
 * mu is bf16 (uint16), nu fp32.  step_t is the new count (>=1).  If zero_grad, g is cleared. */
int ddpo_adamw_bf16mu_step(float* p, float* g, uint16_t* mu, float* nu, int64_t n,
                           const double* sqnorm_of_sum, double inv_n_acc, double lr, double b1, double b2,
                           double eps, double weight_decay, double max_grad_norm, int step_t,
                           int mu_decay_in_bf16, int zero_grad, void* stream);

/* ---- U-Net / VAE building blocks (diffusers FlaxUNet2DConditionModel.apply / FlaxAutoencoderKL.decode;
 *      call sites pipeline_flax_stable_diffusion.py:219-224, ddpo/training/policy_gradient.py:87-102,
 *      pipeline/policy_gradient.py:174-182) --------------------------------------------------------- */

/* GroupNorm(+SiLU) over NHWC x:(B,HW,C) with row stride ldx/ldy (floats).
 * ws: 16-byte aligned scratch of ddpo_groupnorm_ws_bytes(B,HW,C,G) bytes (per-chunk group sums).
 * stats: 16-byte aligned OUTPUT of ddpo_groupnorm_stats_floats(B,C,G) floats = per-(b,c) affine {rstd*gamma,
 * beta-mean*rstd*gamma} followed by per-(b,g) {mean, rstd}; the backward pass reads it.
 * Reductions run in a fixed order: results are bit-reproducible. */
size_t ddpo_groupnorm_ws_bytes(int B, int HW, int C, int G);
size_t ddpo_groupnorm_stats_floats(int B, int C, int G);
int ddpo_groupnorm_fwd(const float* x, int ldx, float* y, int ldy, const float* gamma, const float* beta,
                       int B, int HW, int C, int G, float eps, int fuse_silu, void* ws, float* stats, void* stream);
/* Same, but the result is written as bf16 hi / lo planes (B*HW, ldy) — hi = bf16(y), lo = bf16(y - hi) — the activation
 * operand format of ddpo_gemm_conv_fwd_bf16_planes (the fp32 tensor is not materialised; same bytes).
 * ACTIVATION-PLANE LAYOUTS (ABI v6; every function that writes or reads activation planes takes the same convention through its
 * plane row stride ldy / ld_out / ld_planes / lda / ld_src / ld_w):  ld > 0: row-major (rows, ld);  ld == 0: k-blocked
 * (C / 32, rows, 32), C % 32 == 0 — the 32 channels of one k-tile of consecutive rows are consecutive memory, so every 1 KiB
 * LDS-DMA piece of the plane-fed GEMM (16 consecutive rows x 64 B) is 8 full cache lines instead of 16 half lines.  `rows` is the
 * row count of the tensor the planes describe: B*HW here, `rows` / M elsewhere, B*H*W source pixels for a convolution's input.
 * PLANE FORMAT (ABI v9): bit 1 of `fuse_silu` here (values 2 / 3) and bit 1 of `kblocked` in ddpo_layernorm_fwd_planes select the f16mx
 * format instead of bf16 hi / lo: y_hi = f16(y), y_lo = the interleaved e5m2 chunks of ddpo_split_planes_f16mx (C, ldy % 32 == 0) —
 * the operand of ddpo_gemm_conv_fwd_f16mx_planes, bit for bit what ddpo_split_planes_f16mx would make of the fp32 result. */
int ddpo_groupnorm_fwd_planes(const float* x, int ldx, uint16_t* y_hi, uint16_t* y_lo, int ldy, const float* gamma,
                              const float* beta, int B, int HW, int C, int G, float eps, int fuse_silu, void* ws,
                              float* stats, void* stream);
/* Backward of y = act(GroupNorm(x)): dx (+= dx_add if given), dgamma/dbeta accumulated atomically (they live in the
 * flat gradient buffer).  ws: ddpo_groupnorm_bwd_ws_bytes(B,HW,C,G) bytes. */
size_t ddpo_groupnorm_bwd_ws_bytes(int B, int HW, int C, int G);
int ddpo_groupnorm_bwd(const float* x, int ldx, const float* dy, int lddy, const float* stats, const float* gamma,
                       int B, int HW, int C, int G, int fuse_silu, const float* dx_add, int ld_add, float* dx, int lddx,
                       float* dgamma, float* dbeta, void* ws, void* stream);
/* LayerNorm over the last dim: x,y:(rows,C) contiguous. */
int ddpo_layernorm_fwd(const float* x, float* y, const float* gamma, const float* beta, int rows, int C,
                       float eps, void* stream);
int ddpo_layernorm_fwd_planes(const float* x, uint16_t* y_hi, uint16_t* y_lo, const float* gamma, const float* beta,
                              int rows, int C, float eps, int kblocked, void* stream);      /* planes (rows, C), or k-blocked (C/32, rows, 32) */
/* dx = LayerNorm backward (+ dx_add if given; statistics recomputed from x); dgamma/dbeta += (two-stage reduction
 * through ws = ddpo_layernorm_bwd_ws_bytes(rows, C) bytes of 16-byte aligned scratch). */
size_t ddpo_layernorm_bwd_ws_bytes(int rows, int C);
int ddpo_layernorm_bwd(const float* x, const float* dy, const float* gamma, int rows, int C, float eps,
                       const float* dx_add, float* dx, float* dgamma, float* dbeta, void* ws, void* stream);

/* Implicit-GEMM convolution / dense GEMM on the exact-fp32 MFMA datapath (v_mfma_f32_32x32x2_f32).
 *   out[m][n] = alpha * sum_k A(m,k) * W[k][n] (+ bias[n]) (+ rowbias[m / rows_per_batch][n]) (+ residual[m][n])
 * conv mode: m = (b, oy, ox), k = (ky, kx, ci); src is NHWC with pixel stride ld_src; optional nearest-2x
 * upsampling (upsample=1) of the source folded into the gather (FlaxUpsample2D), stride 1|2, pad 0|1, ksize 1|3.
 * dense mode (ksize==0): A = src (M,K) with row stride ld_src.
 * W is (K, N) row-major (Flax HWIO / (in,out) layout) unless w_trans, then (N, K). */
typedef struct {
  const float* src; int ld_src;
  const float* w; int w_trans;
  const float* bias;              /* (N) or NULL */
  const float* rowbias; int rows_per_batch; int ld_rowbias;   /* (Bt, N): time-embedding add, or NULL */
  const float* residual; int ld_res;                          /* (M, N) or NULL */
  float* out; int ld_out;
  float alpha;
  int M, N, K;
  /* conv geometry (ksize==0 => dense) */
  int ksize, stride, pad, upsample;
  int B, H, W, Cin;               /* source dims (before upsample) */
  int OH, OW;
  /* backward-only fields (zero for the forward pass) */
  int w_dgrad;                    /* with w_trans: w is the FORWARD HWIO kernel (taps, N, Cin); taps are flipped -> data gradient */
  int ld_w;                       /* wgrad: row stride of dY (the `w` operand) */
  int splits;                     /* wgrad: split of the reduction over M (0 = auto) */
  int accumulate;                 /* wgrad: atomically add into out instead of storing */
  /* fused output stage (ddpo_gemm_conv_fwd_bf16 only; 0 everywhere else) */
  int epilogue;                   /* 0: out = alpha*acc + bias + rowbias + residual (N columns)
                                     1: GEGLU.  The N columns of W / bias come as interleaved 32-column blocks
                                        [a_0 | gate_0 | a_1 | gate_1 ...]; out has N/2 columns,
                                        out[:, 32q + c] = (acc_a + bias_a) * gelu_tanh(acc_gate + bias_gate);
                                        needs N % 128 == 0, K % 32 == 0, no rowbias / residual, alpha == 1
                                     2: GEGLU on the 256 x 320 tile (ABI v13; ddpo_gemm_conv_fwd_bf16_planes only).  The columns of W / bias come
                                        as 320-column blocks [a (160) | gate (160)]: block t holds the value and gate columns of the output
                                        columns 160 t .. 160 t + 159; same formula, bit-identical results (aux_out included); needs N % 320 == 0 */
  /* plane-emitting output stage (ddpo_gemm_conv_fwd_bf16 / _planes on the buffer-addressed kernels only; NULL elsewhere).
   * When out_hi != NULL the final value v of every output element is ALSO written as bf16 hi / lo planes
   * (hi = bf16(v), lo = bf16(v - hi): the operand format of ddpo_gemm_conv_fwd_bf16_planes, bit for bit what the
   * fp32-fed loader would split v into), rows of ld_planes elements (% 4 == 0, planes 8-byte aligned).  `out` may then be
   * NULL (planes only).  Needs the vector output stage (N, ld_out, ld_res, ld_rowbias % 4 == 0, 16-byte aligned
   * pointers); DDPO_EINVAL otherwise.  With epilogue == 1 / 2 the planes hold the N/2 GEGLU outputs. */
  uint16_t* out_hi; uint16_t* out_lo; int ld_planes;
  /* layout of the forward weight planes handed to ddpo_gemm_conv_fwd_bf16 / _planes (ABI v6):
   *   0: row-major (N, ldw) bf16, k contiguous per output column (ddpo_pack_weights_bf16);
   *   1: k-blocked (ceil(Kp / 32), N, 32) bf16 (ddpo_pack_weights_bf16_kblocked): the 32 k of one k-tile of all N columns are ONE
   *      contiguous block, so a 1 KiB LDS-DMA piece of the weight operand (16 columns x 64 B) is 1 KiB of consecutive memory =
   *      8 full 128-byte cache lines instead of 16 half lines at the row stride — the weight stream of the k-loop measured
   *      1.25-1.4x faster per CU (tools/native/dma_bench, profiles/r03_dma_bench.log).  `ldw` is ignored.  Not with w_dgrad. */
  int w_layout;
  /* f16mx datapath (ABI v7; ddpo_gemm_conv_fwd_f16mx_planes only, zero elsewhere): one E8M0 scale byte per output column of the
   * 8-bit weight plane (ddpo_pack_weights_f16mx), and the FORMAT of the planes the output stage emits into out_hi / out_lo:
   *   0: bf16 hi / lo (above);  1: f16mx — out_hi = f16 plane, out_lo = per 32-column block the 64 bytes
   *      [h8 c0-15 | l8 c0-15 | h8 c16-31 | l8 c16-31], h8 = e5m2(h), l8 = e5m2(l * 2^11) (needs the emitted column count % 32 == 0).  Same geometry and ld_planes convention as the bf16 planes. */
  const uint8_t* w_scale;
  int planes_fmt;
  /* weight gradients on the bf16x3 kernels (ABI v8; NULL elsewhere): when set and dY (`w`) is fp32, the kernel also accumulates the column sums of
   * dY — the BIAS gradient of the layer, sum_m dY[m][n] — into colsum[n] (fp32 atomic adds, like dW) from the registers that stage dY anyway,
   * instead of a separate ddpo_colsum_accum launch re-reading dY.  Ignored (returns DDPO_EINVAL) with a plane dY operand. */
  float* colsum;
  /* epilogue == 1 (GEGLU) only (ABI v8; NULL elsewhere): also store the PRE-activation x W + b, (M, N) fp32 with row stride N in the ORIGINAL
   * column order [a (N/2) | gate (N/2)] (the interleaving of the packed weights undone) — what the GEGLU backward needs, so the training forward
   * can use the fused launch too. */
  float* aux_out;
} ddpo_gemm_desc;
int ddpo_gemm_conv_fwd(const ddpo_gemm_desc* d, void* stream);
/* Data gradients reuse ddpo_gemm_conv_fwd: src = dY, w = forward kernel with w_trans=1, w_dgrad=1, and for the
 * gradient of a stride-2 convolution upsample=2 ("zero-insert" source: only even virtual coordinates exist).
 * Weight gradient (jax.grad w.r.t. conv / dense kernels, ddpo/training/policy_gradient.py:138-139):
 *   out[k][n] (+)= alpha * sum_m A(m,k) * dY[m][n],  A as in the forward pass (src = forward input),
 *   w = dY (M,N) with row stride ld_w, out = dW (K,N) with row stride ld_out.  With splits != 1 the partial sums
 *   are combined with fp32 atomic adds (out must hold zeros or the running gradient accumulation). */
int ddpo_gemm_conv_wgrad(const ddpo_gemm_desc* d, void* stream);

/* Same contraction on the bf16 MFMA datapath (v_mfma_f32_32x32x16_bf16) with fp32 operands emulated by a bf16 split
 * (x = hi + lo): npass = 3 computes a_lo*b_hi + a_hi*b_lo + a_hi*b_hi (XLA "bf16_3x"/HIGH, ~1e-5 relative),
 * npass = 1 computes a_hi*b_hi (XLA's TPU DEFAULT precision — what the reference ran with); fp32 accumulation.
 * Activations (d->src) stay fp32 and are split on the fly; the weights are given as pre-split bf16 planes made by
 * ddpo_pack_weights_bf16: forward order (N, ldw=Kp) for the forward pass, the original (K, N) order with
 * d->w_dgrad = 1 for data gradients.  d->w / d->w_trans are ignored.  Requires Cin % 8 == 0 (K % 8 == 0 if dense). */
int ddpo_gemm_conv_fwd_bf16(const ddpo_gemm_desc* d, const uint16_t* w_hi, const uint16_t* w_lo, int ldw, int npass,
                            void* ws, size_t ws_bytes, void* stream);
/* Plane-fed variant: the activation operand is given ALREADY split into bf16 hi / lo planes a_hi / a_lo,
 * (rows, lda) bf16 with lda in elements (% 8 == 0), rows = B*H*W source pixels (conv) or M (dense), as written by
 * ddpo_split_planes_bf16 or by the plane-emitting output stage of a normalisation kernel.  d->src / d->ld_src / d->w are
 * ignored.  Both operands reach LDS by LDS-DMA (no register staging, no split in the loader).  Same tiles, k order and
 * MFMA passes as ddpo_gemm_conv_fwd_bf16 on the fp32 tensor the planes were split from: bit-identical results.
 * Requires Cin % 32 == 0 (K % 32 == 0 if dense), no w_dgrad; returns DDPO_EINVAL otherwise (use the fp32-fed entry).
 * ABI v14: a_lo == NULL AND w_lo == NULL selects SINGLE-PASS bf16 (a_hi * w_hi only, fp32 accumulation: npass = 1 of
 * ddpo_gemm_conv_fwd_bf16 — XLA's TPU default precision, BASELINE configs[4]'s dtype) on the same LDS-DMA loaders: one plane per operand,
 * 256 x 320 tiles on a four-stage LDS ring with a counted vmcnt where their grid fills the chip.  hi = bf16(x) is the operand the fp32-fed
 * single-pass kernel forms in its loader: bit-identical to it.  Exactly one of the two lo pointers NULL, or epilogue == 2, is DDPO_EINVAL. */
int ddpo_gemm_conv_fwd_bf16_planes(const ddpo_gemm_desc* d, const uint16_t* a_hi, const uint16_t* a_lo, int lda,
                                   const uint16_t* w_hi, const uint16_t* w_lo, int ldw, void* ws, size_t ws_bytes,
                                   void* stream);
/* f16mx forward datapath (plane-fed, ABI v7).  a*b ~= a_h*b_h + a_h8*b_l8 + a_l8*b_h8: per 32x32 accumulator block and 32-wide
 * k-tile two v_mfma_f32_32x32x16_f16 (the f16 x f16 term) and ONE v_mfma_scale_f32_32x32x64_f8f6f4 carrying both cross terms,
 * against six bf16 MFMAs of the bf16x3 datapath — 2/3 of its matrix-pipe time on the same LDS-DMA operand stream (the planes have
 * the geometry of the bf16 hi / lo planes; profiles/r03_ktime_mx_ablation.log).  The dropped term is a_l*b_l (~2^-22 relative) and
 * the 8-bit rounding of the factors of the cross terms (~2^-15): ~7e-5 relative on a whole SD-1.5 U-Net forward against
 * 2e-5 for bf16x3 and ~2e-3 for single-pass bf16.  Operands:
 *   a16 / a8  activation planes written by ddpo_split_planes_f16mx or a plane-emitting producer (planes_fmt = 1), (rows, lda) or
 *             k-blocked (lda == 0) like the bf16 planes;
 *   w16 / w8 / d->w_scale  weight planes of ddpo_pack_weights_f16mx (k-blocked only: d->w_layout must be 1).
 * Tiles, split-K, output stage (incl. epilogue = 1 and plane emission in either format) as ddpo_gemm_conv_fwd_bf16_planes;
 * every tile class accumulates in the same order (bit-identical results for one layer whatever the batch).  Forward only:
 * data / weight gradients stay on bf16x3. 
 * ABI v14: a8 == NULL AND w8 == NULL (d->w_scale unused) runs the operator WITHOUT its cross terms — a_h * w_h on the f16 MFMA only, on the
 * single-plane kernels (four-stage ring on the 256 x 320 tile): one f16 pass per product, 7e-4 instead of 4e-5 on a U-Net forward and outside the 1e-3 gradient contract at full size.  Opt-in
 * (DDPO_MX_CROSS=0); exactly one of the two NULL is DDPO_EINVAL. */
int ddpo_gemm_conv_fwd_f16mx_planes(const ddpo_gemm_desc* d, const uint16_t* a16, const uint16_t* a8, int lda,
                                    const uint16_t* w16, const uint16_t* w8, void* ws, size_t ws_bytes, void* stream);
/* fp32 W (K, N) -> f16mx weight planes, k-blocked: w16 (ceil(K/32), N, 32) f16 = f16(w); w8 (ceil(K/32), N, 64) bytes =
 * [e4m3(l * 2^11 / s_n) x 32 | e4m3(h / s_n) x 32] with h = f16(w), l = w - h and s_n = 2^(scale[n] - 127) the power of two that
 * puts the column's largest |w| in [128, 256); scale (N) bytes.  Zero padded in k. */
int ddpo_pack_weights_f16mx(const float* w, int K, int N, uint16_t* w16, uint16_t* w8, uint8_t* scale, void* stream);
/* fp32 activations (rows, cols), row stride ldx -> f16mx activation planes (what the plane-emitting producers write); ld_out as
 * ddpo_split_planes_bf16 (0 = k-blocked); cols % 32 == 0. */
int ddpo_split_planes_f16mx(const float* x, int ldx, uint16_t* p16, uint16_t* p8, int ld_out, int64_t rows, int cols, void* stream);
/* fp32 W (K, N) -> bf16 hi / lo planes in the k-blocked forward layout (ceil(K / 32), N, 32), zero padded in k (w_layout = 1). */
int ddpo_pack_weights_bf16_kblocked(const float* w, int K, int N, uint16_t* fwd_hi, uint16_t* fwd_lo, void* stream);
/* ABI v13.  The DATA-GRADIENT operand of the layer with forward kernel w (taps, Cin, Cout) (HWIO flattened; dense layer: taps = 1), packed straight from
 * w into the same k-blocked layout: W'[tap' * Cout + co][ci] = w[taps - 1 - tap'][ci][co], i.e. (ceil(taps * Cout / 32), Cin, 32) planes — what
 * ddpo_gemm_conv_fwd_bf16[_planes] reads to compute dX as a forward contraction of dY (/root/reference/ddpo/training/policy_gradient.py:104-139,
 * the jax.grad of the U-Net call).  Same values as packing the flipped / transposed copy of w with ddpo_pack_weights_bf16_kblocked. */
int ddpo_pack_weights_bf16_kblocked_dgrad(const float* w, int taps, int Cin, int Cout, uint16_t* hi, uint16_t* lo, void* stream);
/* x:(rows, cols) fp32, row stride ldx -> hi / lo bf16 planes (rows, ld_out): hi = bf16(x), lo = bf16(x - hi). */
int ddpo_split_planes_bf16(const float* x, int ldx, uint16_t* hi, uint16_t* lo, int ld_out, int64_t rows, int cols,
                           void* stream);
/* ws (optional, 16-byte aligned): scratch for a deterministic split-K of launches whose tile grid under-fills the chip
 * (partials + fixed-order reduce with the fused epilogue); pass NULL/0 to disable. */
/* Weight gradient on the bf16x3 MFMA datapath (same arguments as ddpo_gemm_conv_wgrad; always accumulates with fp32
 * atomics).  Dense, or convolutions with pad = ksize/2, stride 1 or 2, optionally over a nearest-2x upsampled input
 * (returns DDPO_EINVAL otherwise - callers fall back to ddpo_gemm_conv_wgrad). */
int ddpo_gemm_conv_wgrad_bf16x3(const ddpo_gemm_desc* d, void* stream);
/* The same weight gradient with one or both operands given as bf16 hi / lo planes (the format of
 * ddpo_gemm_conv_fwd_bf16_planes): a_hi / a_lo replace d->src (the forward input; rows of d->ld_src ELEMENTS), dy_hi / dy_lo
 * replace d->w (dY; rows of d->ld_w elements).  A NULL pair means that operand is fp32 in the descriptor.  Same values reach
 * the MFMAs as in the fp32-fed form (the planes ARE its split), so the result differs only by the atomics' summation order. */
int ddpo_gemm_conv_wgrad_bf16x3_planes(const ddpo_gemm_desc* d, const uint16_t* a_hi, const uint16_t* a_lo,
                                       const uint16_t* dy_hi, const uint16_t* dy_lo, void* stream);
/* fp32 W (K,N) -> bf16 hi/lo planes: fwd_* (N, Kp) k-contiguous (Kp = K rounded up to 8, zero padded) and, if
 * bwd_hi != NULL, bwd_* (K, N).  Call after every optimizer update (weights only change there). */
int ddpo_pack_weights_bf16(const float* w, int K, int N, int Kp, uint16_t* fwd_hi, uint16_t* fwd_lo,
                           uint16_t* bwd_hi, uint16_t* bwd_lo, void* stream);

/* Fused multi-head attention, softmax(q k^T * scale) v, flash-style on fp32 MFMA (16x16x4).
 * q:(B,Nq,·) k,v:(B,Nk,·) o:(B,Nq,·): head h occupies columns [h*d,(h+1)*d) of each row; ld* = row strides.
 * lse (optional, (B,heads,Nq)): log2-domain logsumexp of the scaled scores, consumed by the backward pass. */
int ddpo_attention_fwd(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv,
                       float* o, int ldo, float* lse, int B, int heads, int Nq, int Nk, int d, float scale,
                       void* stream);
/* Same contract on the bf16 MFMA datapath: Q, K, V and the probabilities are split into bf16 hi + lo and every
 * product takes three passes (fp32 accumulate, ~1e-5 relative).  d in {8, 16, 40, 64, 80}.
 * ws (optional, 16-byte aligned, >= ddpo_attention_fwd_bf16x3_ws_bytes): when given and Nk >= 256, K and V are split /
 * transposed once per (batch, head) into per-tile LDS images that the attention kernel streams; without it (or for
 * short key sequences, where the size query returns 0) every query tile stages K / V itself.  Results are identical. */
size_t ddpo_attention_fwd_bf16x3_ws_bytes(int B, int heads, int Nk, int d);
/* For keys / values that stay constant over many attention calls (the text context of the cross-attention layers over the DDIM
 * steps of a sampling call, /root/reference/ddpo/diffusers_patch/pipeline_flax_stable_diffusion.py:219-224 re-projects and
 * re-reads it every step): pack K / V ONCE into the per-tile images (any Nk, `images` 16-byte aligned, >=
 * ddpo_attention_kv_images_bytes) and run every attention from them.  Same kernels as above: identical results. */
size_t ddpo_attention_kv_images_bytes(int B, int heads, int Nk, int d);
int ddpo_attention_pack_kv_bf16x3(const float* k, int ldk, const float* v, int ldv, void* images, size_t images_bytes,
                                  int B, int heads, int Nk, int d, void* stream);
int ddpo_attention_fwd_bf16x3_images(const float* q, int ldq, const void* images, size_t images_bytes, float* o, int ldo,
                                     float* lse, int B, int heads, int Nq, int Nk, int d, float scale, void* stream);
int ddpo_attention_fwd_bf16x3(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv,
                              float* o, int ldo, float* lse, int B, int heads, int Nq, int Nk, int d, float scale,
                              void* ws, size_t ws_bytes, void* stream);
/* Attention backward with probability recomputation (no N x N tensor is ever materialised):
 * dvec (B,heads,Nq) scratch = rowsum(dO * O); dq/dk/dv have the layout of q/k/v with contiguous rows of heads*d. */
int ddpo_attention_bwd(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, const float* o,
                       const float* d_o, const float* lse, float* dvec, float* dq, float* dk, float* dv,
                       int B, int heads, int Nq, int Nk, int d, float scale, void* stream);

/* Attention backward on the bf16x3 datapath (same arguments; d in {8, 16, 40, 64, 80}): every product on three bf16 passes. */
int ddpo_attention_bwd_bf16x3(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, const float* o,
                              const float* d_o, const float* lse, float* dvec, float* dq, float* dk, float* dv,
                              int B, int heads, int Nq, int Nk, int d, float scale, void* stream);

/* ---- `f16p` attention operators (ABI v11): the attention of the f16mx datapath.  Same arguments and layouts as the bf16x3 functions of the
 * same name; the images of ddpo_attention_pack_kv_f16p are only valid for ddpo_attention_fwd_f16p_images (same byte size,
 * ddpo_attention_kv_images_bytes).  Arithmetic: the scores stay bf16 hi + lo on three MFMA passes; in the second product the probabilities are
 * ONE f16 term p = exp2(s - m + 14) (round to nearest even) against V split into f16 hi + lo — two passes — and the softmax denominator is the
 * sum of the same rounded probabilities (a row of ones in V^T where the head dim leaves a spare MFMA row), so O is an exact convex combination
 * of the values with weights perturbed by <= 2^-12 (csrc/attention_bf16.hip; 1.42 -> 1.29 ms on 4096^2 keys, d = 40, batch 16).
 * Replaces nn.dot_product_attention of diffusers' FlaxAttentionBlock inside the U-Net call sites of
 * ddpo/diffusers_patch/pipeline_flax_stable_diffusion.py:219-224 and ddpo/training/policy_gradient.py:87-102. */
int ddpo_attention_fwd_f16p(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv,
                            float* o, int ldo, float* lse, int B, int heads, int Nq, int Nk, int d, float scale,
                            void* ws, size_t ws_bytes, void* stream);
int ddpo_attention_pack_kv_f16p(const float* k, int ldk, const float* v, int ldv, void* images, size_t images_bytes, int B, int heads,
                                int Nk, int d, void* stream);
int ddpo_attention_fwd_f16p_images(const float* q, int ldq, const void* images, size_t images_bytes, float* o, int ldo,
                                   float* lse, int B, int heads, int Nq, int Nk, int d, float scale, void* stream);
/* Backward of the same operator.  `dvec` is a scratch of B * heads * (Nq + 1) floats: rowsum(dO * O) followed by one word per (batch, head)
 * slab — the float bits of max |dO| over the slab, from which the kernels take the power of two that brings dO into the f16 range whatever the
 * loss scale.  Scores: bf16 hi + lo, three passes; dP = dO V^T, dV = P^T dO, dK = dS^T Q, dQ = dS K: a single f16 term (dO, P, dS) against an
 * f16 hi + lo split, two passes (csrc/attention_bwd_bf16.hip). */
int ddpo_attention_bwd_f16p(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, const float* o,
                            const float* d_o, const float* lse, float* dvec, float* dq, float* dk, float* dv,
                            int B, int heads, int Nq, int Nk, int d, float scale, void* stream);


/* ---- Plane-emitting attention forwards (ABI v12).  Same arithmetic and arguments as the function of the same name without `_po`, except that
 * the normalised output leaves as bf16 hi / lo PLANES (o_hi / o_lo: (B * Nq, heads * d) bf16 each, row stride ld_planes elements, 8-byte
 * aligned; ld_planes == 0: k-blocked (heads * d / 32, B * Nq, 32), heads * d % 32 == 0) holding exactly the split hi = bf16(x), lo = bf16(x - hi)
 * of the fp32 value the plain function writes — the activation format of ddpo_gemm_conv_fwd_bf16_planes, so the to_out projection behind the
 * attention (diffusers FlaxAttentionBlock: proj_attn / to_out_0) reads planes by LDS-DMA and no fp32 tensor is written or re-split: in the
 * sampling forward at the 64x64 level that projection drops from 103 to 62 us per launch (profiles/r04_timeline_sampling_step.txt).
 * lse may be NULL.  Sampling path only (the training forward keeps the fp32 output its backward reads). */
int ddpo_attention_fwd_bf16x3_po(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, uint16_t* o_hi, uint16_t* o_lo,
                                 int ld_planes, float* lse, int B, int heads, int Nq, int Nk, int d, float scale, void* ws, size_t ws_bytes,
                                 void* stream);
int ddpo_attention_fwd_f16p_po(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, uint16_t* o_hi, uint16_t* o_lo,
                               int ld_planes, float* lse, int B, int heads, int Nq, int Nk, int d, float scale, void* ws, size_t ws_bytes,
                               void* stream);
int ddpo_attention_fwd_bf16x3_images_po(const float* q, int ldq, const void* images, size_t images_bytes, uint16_t* o_hi, uint16_t* o_lo,
                                        int ld_planes, float* lse, int B, int heads, int Nq, int Nk, int d, float scale, void* stream);
int ddpo_attention_fwd_f16p_images_po(const float* q, int ldq, const void* images, size_t images_bytes, uint16_t* o_hi, uint16_t* o_lo,
                                      int ld_planes, float* lse, int B, int heads, int Nq, int Nk, int d, float scale, void* stream);


/* Small element-wise pieces. */
int ddpo_geglu_fwd(const float* x, float* y, int64_t rows, int F, void* stream);      /* y = x[:, :F] * gelu_tanh(x[:, F:]) */
int ddpo_silu_fwd(const float* x, float* y, int64_t n, void* stream);
/* Reward-model pieces (aesthetic_fn, /root/reference/ddpo/training/callbacks.py:60-95: CLIP ViT-L/14 image tower + LAION MLP run on
 * this library's GEMM / LayerNorm / attention kernels): quick-GELU y = x * sigmoid(1.702 x) (x, y 16-byte aligned), and the row-wise
 * L2 normalisation of the image features (:80-82). */
int ddpo_quick_gelu_fwd(const float* x, float* y, int64_t n, void* stream);
int ddpo_l2_normalize_rows(const float* x, float* y, int rows, int cols, void* stream);
int ddpo_timestep_embedding(const int32_t* ts, float* out, int B, int dim, void* stream); /* concat([cos, sin]) */
int ddpo_nchw_to_nhwc(const float* x, float* y, int B, int C, int HW, void* stream);
int ddpo_nhwc_to_nchw(const float* x, float* y, int B, int C, int HW, void* stream);
int ddpo_copy_cols(const float* src, int ld_src, float* dst, int ld_dst, int64_t rows, int cols, void* stream);
/* ABI v14.  The inputs of one classifier-free-guidance sampling step, staged in ONE launch into the static input buffers of a captured U-Net
 * graph: s_in[0:n] = s_in[n:2n] = x[0:n] (the reference's jnp.concatenate([latents] * 2), pipeline_flax_stable_diffusion.py:219), and — each
 * optional, both pointers NULL to skip — row_dst[0:row_n] = row_src[0:row_n] (this step's row of the precomputed time-projection table) and
 * ts_dst[0:ts_n] = ts_src[0:ts_n].  n % 4 == 0, row_n % 4 == 0, float pointers 16-byte aligned. */
int ddpo_stage_cfg_inputs(const float* x, float* s_in, int64_t n, const float* row_src, float* row_dst, int row_n,
                          const int32_t* ts_src, int32_t* ts_dst, int ts_n, void* stream);
int ddpo_softmax_rows(float* x, int64_t rows, int cols, float scale, void* stream);    /* in place */
/* backward element-wise pieces */
int ddpo_geglu_bwd(const float* x, const float* dy, float* dx, int64_t rows, int F, void* stream);
int ddpo_silu_bwd(const float* x, const float* dy, float* dx, int64_t n, void* stream);
/* out[seg][n] += sum over the rows of segment seg (rows_per_seg consecutive rows; 0 = one segment) of x[m][n] */
int ddpo_colsum_accum(const float* x, int ldx, int64_t rows, int cols, int rows_per_seg, float* out, void* stream);
/* nearest-2x upsampling backward: y[b,h,w,:] = sum of the 2x2 block of x:(B,2H,2W,C) */
int ddpo_sumpool2x2(const float* x, float* y, int B, int H, int W, int C, void* stream);
int ddpo_add(const float* a, const float* b, float* out, int64_t n, void* stream);
int ddpo_scale_shift_clip(const float* x, float* y, int64_t n, float scale, float shift, float lo, float hi, void* stream);

#ifdef __cplusplus
}
#endif
#endif
