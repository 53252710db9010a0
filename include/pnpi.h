/*
 * pnpi.h -- C ABI of libpnpi.so: the MI355X (gfx950) implementation of PnPInversion's direct-inversion +
 * Prompt-to-Prompt hot path.  Plain pointers and sizes only; no C++ / torch types cross this boundary.
 *
 * The reference (cure-lab/PnPInversion) is pure Python and has no FFI of its own; its "operator API" for this path is the
 * duck-typed pipeline object used by the modules under models/p2p/.  Each entry point below names the reference interface it
 * replaces (file:line in /root/reference).  INTEGRATION.md shows the ctypes binding a maintainer would add.
 *
 * Conventions
 *   - every function returns 0 on success, a negative pnpi_status otherwise; pnpi_last_error() gives the message.
 *   - all tensor arguments are DEVICE pointers to contiguous memory unless the name ends in _host.
 *   - public layouts are the reference's: latents / eps fp32 NCHW, context fp32 [rows,77,768], images uint8 HWC.
 *   - all work is enqueued on the HIP stream given to pnpi_create(); nothing synchronises the device.
 *   - one pnpi_ctx per (process, GPU); a ctx is not thread-safe.
 *   - UNet batch-row convention (models/p2p/p2p_guidance_forward.py:108,170; inversion.py:305,382), per image:
 *       rows_per_image = 4 : [uncond_src, uncond_tgt, cond_src, cond_tgt]   (controllers act on rows 2,3 only)
 *       rows_per_image = 1 : DDIM inversion (cond_src only)
 */
#ifndef PNPI_H
#define PNPI_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pnpi_ctx pnpi_ctx;

typedef enum {
  PNPI_OK = 0,
  PNPI_EINVAL = -1,  /* bad argument */
  PNPI_ESHAPE = -2,  /* unsupported shape / configuration */
  PNPI_EHIP = -3,    /* HIP runtime error */
  PNPI_ESTATE = -4,  /* call sequence error (e.g. weights not loaded) */
  PNPI_ENOMEM = -5
} pnpi_status;

/* Architecture of the Stable-Diffusion-1.x UNet / VAE (diffusers UNet2DConditionModel / AutoencoderKL constructor args,
 * models/edict/my_diffusers/models/unet_2d_condition.py:57-82, vae.py:508-519).  SD-1.x: see pnpi_config_sd1(). */
typedef struct {
  int in_channels, out_channels;   /* 4, 4 */
  int n_blocks;                    /* 4 */
  int block_out_channels[4];       /* 320, 640, 1280, 1280 */
  int block_has_attn[4];           /* 1, 1, 1, 0  (CrossAttnDownBlock2D x3, DownBlock2D) */
  int layers_per_block;            /* 2 */
  int heads;                       /* 8  (diffusers' "attention_head_dim") */
  int cross_dim;                   /* 768 */
  int ctx_len;                     /* 77 */
  int sample_size;                 /* 64 (latent H = W) */
  int norm_groups;                 /* 32 */
  int n_train_timesteps;           /* 1000 */
  int vae_in_channels;             /* 3 */
  int vae_latent_channels;         /* 4 */
  int vae_n_blocks;                /* 4 */
  int vae_block_out_channels[4];   /* 128, 256, 512, 512 */
  int vae_layers_per_block;        /* 2 */
  int vae_norm_groups;             /* 32 */
  /* CLIP text encoder (transformers CLIPTextModel: hidden = cross_dim, positions = ctx_len); clip_layers = 0: not built */
  int clip_layers;                 /* 12 */
  int clip_heads;                  /* 12 */
  int clip_intermediate;           /* 3072 */
  int clip_vocab;                  /* 49408 */
} pnpi_model_config;

void pnpi_config_sd1(pnpi_model_config* cfg);

typedef struct {
  const char* name;   /* diffusers state-dict key prefixed with "unet." or "vae."; transformers CLIPTextModel key (with or
                         without its "text_model." prefix) prefixed with "clip." */
  const void* data;   /* device pointer, contiguous, PyTorch layout ([out,in,kh,kw] / [out,in] / [n]) */
  int dtype;          /* 0 = fp32, 1 = fp16 */
  int ndim;
  int64_t shape[4];
} pnpi_named_tensor;

/* Declarative Prompt-to-Prompt controller (one per image).  Replaces the per-layer Python callback
 * controller(attn, is_cross, place_in_unet) of models/p2p/attention_control.py:44,178-190 for the controller classes
 * AttentionStore (:214), AttentionReplace (:301), AttentionRefine (:317), AttentionReweight (:338) and LocalBlend (:95).
 * All pointers are HOST pointers; tables are copied at the call. */
typedef struct {
  int kind;                        /* 0 = none / AttentionStore (no effect on the output), 1 = Prompt-to-Prompt edit,
                                      2 = MasaCtrl mutual self-attention (models/masactrl/masactrl.py:14-72) */
  int n_alpha_rows;                /* rows of cross_alpha (num_steps + 1 = 51) */
  const float* cross_alpha_host;   /* [n_alpha_rows][77]  get_time_words_attention_alpha, utils/utils.py:117-135 */
  const float* mapper_host;        /* [77][77] source-token w -> target-token j weights (Replace: seq_aligner.py:152-185;
                                      Refine: one-hot of mapper[j]==w, seq_aligner.py:107-118; Reweight alone: identity) */
  const float* alphas_host;        /* [77] Refine blend weights (1 for Replace)            attention_control.py:319-321 */
  const float* equalizer_host;     /* [77] Reweight scales (1 if none)                     attention_control.py:84-92,343 */
  int self_replace_lo, self_replace_hi;  /* num_self_replace = (0, 30)                     attention_control.py:295-297 */
  int self_replace_max_tokens;     /* 1024 (32**2)                                          attention_control.py:259 */
  int lb_enabled;                  /* LocalBlend                                            attention_control.py:95-147 */
  int lb_start;                    /* start_blend = int(0.2 * steps) = 10 */
  float lb_threshold;              /* 0.3 */
  const float* lb_alpha_host;      /* [2][77] alpha_layers one-hot rows (src prompt, tgt prompt) */
  int masa_start_step;             /* kind 2: steps >= start_step (4) ...                   masactrl.py:36,61 */
  int masa_start_layer;            /* ... and transformer blocks >= start_layer (10, execution order 0..15): the target rows'
                                      self-attention reads K and V of the source row of their CFG half (masactrl.py:63-69) */
  const float* lb_sub_alpha_host;  /* LocalBlend substruct_words: [2][77] substruct_layers one-hot rows, or NULL (none).  The blend mask
                                      becomes mask & ~mask_sub, mask_sub = the same maps weighted by these rows, NOT max-pooled, thresholded
                                      at lb_threshold_sub                                    attention_control.py:97-106,114-116,134-143 */
  float lb_threshold_sub;          /* th[1] = 0.3 */
  /* kind 2 with explicit lists (MutualSelfAttentionControl(layer_idx=, step_idx=), masactrl.py:24-37: membership tests at :61).  Both
   * optional; when given they REPLACE the corresponding window above. */
  unsigned masa_layer_mask;        /* bit b set: transformer block b (execution order 0..15) is in layer_idx; 0 = the start_layer window.
                                      Bit 31 = "a list was given": alone it is the EMPTY list (no block matches, control off) */
  int masa_n_steps;                /* length of masa_step_on_host; 0 = the start_step window (an empty step_idx list: one 0 byte) */
  const unsigned char* masa_step_on_host;   /* [masa_n_steps]: 1 where the denoising step is in step_idx (steps past the end: off) */
} pnpi_ctrl_desc;

/* Reconstruction guidance of the proximal-guidance loop (models/p2p/proximal_guidance_forward.py:48-51,60-72 with
 * DDIMSchedulerDev.step's ref_image / recon_lr / recon_mask branch, models/p2p/scheduler_dev.py:68-76): at the steps with
 * (recon_t > 0 and t < recon_t) or (recon_t < 0 and t > -recon_t) the predicted x0 is pulled towards the encoded source image where
 * the (shrunk) CFG difference is not above the proximal threshold:  mask_edit = dilate(|delta| > thr, kernel 2*dilate_mask+1);
 * pred_x0 -= recon_lr * (pred_x0 - ref_image) * (1 - mask_edit).  Only meaningful with prox != 0. */
typedef struct {
  uint32_t struct_size;            /* = sizeof(pnpi_recon_desc), set by the caller: the struct grew a field in round 5 and may grow again; an
                                      entry point answers any other value with PNPI_EINVAL instead of reading past what the caller filled */
  const float* ref_image;          /* device [nimg][4][h][w]: image_enc_latent (0.18215 * VAE posterior mean of the source image) */
  float recon_lr;
  int recon_t;
  int dilate_mask;                 /* radius of the max-pool dilation of mask_edit; 0 = none */
  /* inversion guidance (proximal_guidance_forward.py:73-75): inv_x_stars = the inversion trajectory x*_0 .. x*_nsteps, device fp32
   * [nsteps+1][nimg][4][h][w] (nullable = off); at step i inside the recon_t window the step's result is pulled towards
   * x*_{nsteps-1-i} outside the edit mask with recon_lr.  ref_image may then be NULL (no pred-x0 pull).  pnpi_cfg_ddim_prev (level 1,
   * one step, no step index): inv_x_stars points at THIS step's x*_{t-1}, [nimg][4][h][w].
   * recon_lr: the pred-x0 pull runs for recon_lr > 0 only (scheduler_dev.py:68), the inversion pull for any recon_lr != 0 (the
   * reference applies `latents - recon_lr * (...)` unconditionally inside the window, proximal_guidance_forward.py:75). */
  const float* inv_x_stars;
} pnpi_recon_desc;

typedef struct {
  uint64_t unet_sample_forwards;   /* number of UNet batch rows evaluated (x 803.27 GFLOP each for SD-1.x) */
  uint64_t unet_calls;
  uint64_t vae_encodes, vae_decodes;
  double executed_gemm_flops;      /* 2*M*N*K over every MFMA GEMM/conv launch, padding included */
  double executed_attn_flops;
  uint64_t text_kv_rows;           /* context rows whose 16 cross-attention K / V projections were computed by a precompute (2.95 GFLOP
                                      each for SD-1.x); a forward that reads the cache does NOT execute them: its sample-forward is
                                      803.27 - 2.95 GFLOP */
  uint64_t unet_sample_forwards_cached_kv;   /* of unet_sample_forwards, the rows that read the text K / V cache */
  uint64_t unet_backward_rows;     /* reverse walks (d loss / d context through one UNet row: null-text / null-latent inversion); the
                                      recording forward of each is counted in unet_sample_forwards */
} pnpi_counters;

/* ---- lifetime ------------------------------------------------------------------------------------------------ */
/* replaces StableDiffusionPipeline.from_pretrained(...).to(device)             models/p2p_editor.py:23-25 */
int pnpi_create(pnpi_ctx** out, const pnpi_model_config* cfg, int device, void* hip_stream, int max_unet_rows, int max_vae_images);
/* A further context on the SAME packed weights (same device, same model configuration): own stream, workspaces and caches, the
 * parent's weight arena borrowed read-only instead of copied (several images in flight on one GPU: one 1.9 GB arena in the caches
 * instead of N).  The arena is reference-counted: it is freed by the last pnpi_destroy among the parent and its children, in whatever
 * order they are destroyed.  A failed pnpi_create / pnpi_create_shared leaves a context in *out that only pnpi_last_error and
 * pnpi_destroy accept (call both).  Reloading the parent's weights while children exist is the caller's error
 * (call pnpi_mark_all_loaded on each child afterwards); pnpi_load_weights on a child fails with PNPI_ESTATE.  There is no
 * counterpart in the reference (one pipeline object per process, models/p2p_editor.py:18-25). */
int pnpi_create_shared(pnpi_ctx** out, pnpi_ctx* parent, void* hip_stream, int max_unet_rows, int max_vae_images);
void pnpi_destroy(pnpi_ctx* ctx);
const char* pnpi_last_error(const pnpi_ctx* ctx);
int pnpi_load_weights(pnpi_ctx* ctx, const pnpi_named_tensor* tensors, int n);
int pnpi_missing_weights(const pnpi_ctx* ctx, char* names_out, size_t cap);   /* returns count of unloaded slots */
/* packed fp16/fp32 weight arena, for the one start-up RCCL broadcast */
int pnpi_weight_arena(pnpi_ctx* ctx, void** ptr, size_t* bytes);
/* receiving ranks of that broadcast call this instead of pnpi_load_weights */
int pnpi_mark_all_loaded(pnpi_ctx* ctx);
/* DDIMScheduler tables: alphas_cumprod[n_train] (fp32) and final_alpha_cumprod    models/p2p_editor.py:18-22 */
int pnpi_set_scheduler(pnpi_ctx* ctx, const float* alphas_cumprod_host, int n_train, float final_alpha_cumprod);
int pnpi_get_counters(const pnpi_ctx* ctx, pnpi_counters* out);
int pnpi_reset_counters(pnpi_ctx* ctx);

/* Per-kernel-class timing with HIP events on the ctx stream (used by bench.py for the `roofline` object).
 * Between begin and end every kernel launch of the graph executor is bracketed by two events. */
enum { PNPI_KC_IGEMM128 = 0, PNPI_KC_IGEMM64 = 1, PNPI_KC_IGEMM64_SPLITK = 2, PNPI_KC_ATTN_FLASH = 3, PNPI_KC_ATTN_EDIT = 4,
       PNPI_KC_GROUPNORM = 5, PNPI_KC_LAYERNORM = 6, PNPI_KC_GEGLU = 7, PNPI_KC_SOFTMAX = 8, PNPI_KC_IGEMM_WIDE = 9 /* 128x320, 128x256 tiles */,
       PNPI_KC_COUNT = 10 };
typedef struct {
  uint64_t launches;
  double total_ms;      /* sum of per-launch durations */
  double flops;         /* algorithmic FLOPs (2*M*N*K of the logical problem, head padding excluded for attention) */
  double bytes;         /* algorithmic HBM bytes for the bandwidth-bound classes */
} pnpi_kernel_stats;
int pnpi_profile_begin(pnpi_ctx* ctx);
int pnpi_profile_end(pnpi_ctx* ctx, pnpi_kernel_stats* out /* [PNPI_KC_COUNT] */);

/* Matrix-pipe clock calibration for bench.py's `clock` object (no reference counterpart: it protects the one number the driver times):
 * every SIMD issues `iters` x 16 v_mfma_f32_32x32x16_f16 back to back (2 waves per SIMD, register operands, pseudo-random values), timed
 * with HIP events on the ctx stream; *ghz_out = matrix-pipe cycles / elapsed = the clock the part holds under a pure MFMA load. */
int pnpi_clock_probe(pnpi_ctx* ctx, int iters, float* ghz_out, float* ms_out /* nullable */);

/* ---- level 1: operator boundary (keeps the loops under models/p2p/ usable unmodified) ---------------------------- */
/* model.unet(latents, t, encoder_hidden_states=context)["sample"]    inversion.py:273, p2p_guidance_forward.py:109 */
int pnpi_unet_forward(pnpi_ctx* ctx, const float* latents, int rows, int rows_per_image, int t, const float* context,
                      const pnpi_ctrl_desc* ctrl_host /* nullable, [rows/4] */, int cur_step, float* eps_out);
/* Level-1 fallback for attention controllers the library has no descriptor for: the hooked attention forward of
 * models/p2p/attention_control.py:20-47 executed literally.  While a callback is set, every attention site of pnpi_unet_forward
 * materialises attn = softmax(q k^T * scale) as fp32 [rows*heads][Nq][Nk] (rows outer, heads inner: the reference's
 * reshape_heads_to_batch_dim order) in `attn_buf`, calls cb -- which may rewrite the tensor in place with device work on the
 * library's stream, as `attn = controller(attn, is_cross, place_in_unet)` does -- and then forms out = attn v.  place: 0 down,
 * 1 mid, 2 up; layer: attention site 0..31 in call order.  A non-zero return from cb aborts the forward (PNPI_ESTATE).  Slow by
 * construction (score tensors in HBM, one GEMM pair per (row, head)); the fused descriptor path is the product path.
 * cb = NULL restores the fused path.  attn_buf must hold, for the largest site, rows*heads*Nq*Nk floats + Nq*round_up(Nk, 8) halfs. */
typedef int (*pnpi_attn_callback)(void* user, float* attn, int rows, int heads, int Nq, int Nk, int is_cross, int place, int layer);
int pnpi_set_attention_callback(pnpi_ctx* ctx, pnpi_attn_callback cb, void* user, float* attn_buf, size_t attn_buf_bytes);
/* Cross-attention keys / values of the 16 transformer blocks for `rows` context rows (device fp32 [rows][77][768]); they depend on
 * the text only (CrossAttention.to_k / to_v on encoder_hidden_states, my_diffusers/models/attention.py:230-234, evaluated by the
 * reference inside every one of its 650 UNet calls per image).  pnpi_unet_forward(..., context = NULL, ...) then reads the cache
 * (rows must match); the level-2 loops below precompute it themselves once per loop and drop it at their end (a loop call
 * therefore also invalidates a cache filled here). */
int pnpi_text_kv_precompute(pnpi_ctx* ctx, const float* context, int rows);
/* controller.step_callback -> LocalBlend.__call__ (attention_control.py:108-121,253-256) for level-1 drivers:
 * latents [nimg][2][4][h][w] updated in place, using the maps accumulated by the preceding pnpi_unet_forward calls */
int pnpi_local_blend(pnpi_ctx* ctx, float* latents, int nimg, int step_index);
/* model.vae.encode(x)['latent_dist'].mean  (x fp32 NCHW in [-1,1])                     utils/utils.py:78 */
int pnpi_vae_encode(pnpi_ctx* ctx, const float* x_nchw, int n, int height, int width, float* mean_out);
/* model.vae.decode(z)['sample']                                                        utils/utils.py:61 */
int pnpi_vae_decode(pnpi_ctx* ctx, const float* z_nchw, int n, int lat_h, int lat_w, float* sample_out);
/* image2latent: uint8 HWC -> 0.18215 * mean                                             utils/utils.py:68-80 */
int pnpi_image2latent(pnpi_ctx* ctx, const uint8_t* img_hwc, int n, int height, int width, float* z_out);
/* latent2image: decode(z / 0.18215) -> (x/2+.5).clamp(0,1)*255 -> uint8 HWC              utils/utils.py:58-66 */
int pnpi_latent2image(pnpi_ctx* ctx, const float* z_nchw, int n, int lat_h, int lat_w, uint8_t* img_hwc_out);
/* DirectInversion.next_step (inversion.py:262-270): alpha values are taken from the scheduler table */
int pnpi_ddim_next_step(pnpi_ctx* ctx, const float* eps, int t, int step_ratio, const float* sample, size_t n, float* out);
/* DirectInversion.prev_step (inversion.py:247-260) == DDIMSchedulerDev.step (scheduler_dev.py:38-95), eta = 0 */
int pnpi_ddim_prev_step(pnpi_ctx* ctx, const float* eps, int t, int step_ratio, const float* sample, size_t n, float* out);
/* DDIMSchedulerDev.step with the reconstruction pull          models/p2p/scheduler_dev.py:68-76
 * (kwargs ref_image / recon_lr / recon_mask): pred_x0 -= recon_lr * (pred_x0 - ref_image) [* recon_mask] between the step's two halves.
 * ref_image / recon_mask are device fp32 arrays of the sample's shape (the caller expands them), recon_mask and pred_x0_out nullable. */
int pnpi_ddim_prev_step_recon(pnpi_ctx* ctx, const float* eps, int t, int ratio, const float* sample, size_t n, const float* ref_image,
                              float recon_lr, const float* recon_mask, float* out, float* pred_x0_out);
/* fused CFG + prev_step + direct-inversion offset (inversion.py:383-389; p2p_guidance_forward.py:110-114).
 *   eps [nimg][2R][E]; x [nimg][R][E]; target (nullable) [nimg][E] -> offset_out = (target - prev) * offset_scale (1 on
 *   the paper's path; the not_full / skip_step ablations scale or zero it, inversion.py:491-492,512-515), x_out = prev + offset;
 *   else noise_loss (nullable) [nimg][R][E] added to the first offset_rows rows. */
int pnpi_cfg_ddim_prev(pnpi_ctx* ctx, const float* eps, const float* x, int nimg, int rows_per_img, size_t row_elems,
                       float guidance_scale, int t, int step_ratio, const float* noise_loss, int offset_rows,
                       const float* target, float offset_scale, float* offset_out, float* x_out,
                       const float* prox_threshold /*[nimg] device, nullable*/, int prox /*0 none, 1 l0, 2 l1*/,
                       const pnpi_recon_desc* recon /* nullable; applied when its recon_t window contains t */);
/* threshold of the proximal-guidance step (proximal_guidance_forward.py:41,53): torch.quantile(|eps_c - eps_u|, q) with the
 * default linear interpolation over the rows of each image; eps [nimg][2R][E] -> thr_out [nimg] (device) */
int pnpi_prox_threshold(pnpi_ctx* ctx, const float* eps, int nimg, int rows_per_img, size_t row_elems, float quantile,
                        float* thr_out);

/* model.text_encoder(input_ids)[0] (inversion.py:290-306, p2p_guidance_forward.py:151-164): transformers CLIPTextModel
 * last_hidden_state.  ids: device int32 [n][ctx_len]; out: fp32 [n][ctx_len][cross_dim].  Needs the "clip." weights; the UNet /
 * VAE entry points do not. */
int pnpi_text_encode(pnpi_ctx* ctx, const int32_t* input_ids, int n, float* hidden_out);

/* ---- level 2: loop boundary (whole phases device-resident, no host round trip per step) ------------------------- */
/* DirectInversion.ddim_loop (inversion.py:308-319): latents_out [nsteps+1][nimg][4][h][w]; timesteps_host = scheduler.timesteps */
int pnpi_ddim_invert(pnpi_ctx* ctx, const float* z0, int nimg, const float* ctx_cond /*[nimg][77][768]*/, int nsteps,
                     const int* timesteps_host, float* latents_out);
/* DirectInversion.offset_calculate (inversion.py:375-391): noise_loss_out [nsteps][nimg][2][4][h][w] */
int pnpi_offset_calculate(pnpi_ctx* ctx, const float* ddim_latents /*[nsteps+1][nimg][...]*/, int nimg,
                          const float* context4 /*[nimg][4][77][768]*/, int nsteps, const int* timesteps_host,
                          float guidance_scale, const float* offset_scale_host /*[nsteps], nullable = all 1*/,
                          float* noise_loss_out);
/* DirectInversion.ddim_with_guidance_scale_loop (inversion.py:334-347): DDIM inversion under classifier-free guidance
 * (the directinversion+p2p_guidance_<inv>_<fwd> methods); uncond / cond rows share one launch per step */
int pnpi_ddim_invert_cfg(pnpi_ctx* ctx, const float* z0, int nimg, const float* ctx_uncond /*[nimg][77][768]*/,
                         const float* ctx_cond, float guidance_scale, int nsteps, const int* timesteps_host,
                         float* latents_out);
/* direct_inversion_p2p_guidance_forward (p2p_guidance_forward.py:135-173) incl. controller + LocalBlend:
 * latents_out [nimg][2][4][h][w].  ctrl_host: nullable or [nimg].
 * prox 1 ('l0') / 2 ('l1'): the proximal-guidance step of proximal_guidance_forward.py:39-64 -- the CFG difference is
 * soft-thresholded at quantile `quantile` of its magnitude over the image's two rows (torch.quantile, linear); a
 * quantile <= 0 means the fixed threshold -quantile.  prox 0: plain CFG.  recon (nullable): reconstruction guidance. */
int pnpi_edit_loop(pnpi_ctx* ctx, const float* x_T /*[nimg][4][h][w]*/, int nimg, const float* context4,
                   const float* noise_loss /*[nsteps][nimg][2][...], nullable*/, int offset_rows,
                   const pnpi_ctrl_desc* ctrl_host, int nsteps, const int* timesteps_host, float guidance_scale,
                   int prox, float quantile, const pnpi_recon_desc* recon, float* latents_out);

/* The denoising half of P2PEditor.edit_image_directinversion (p2p_editor.py:99-160) as ONE loop: offset_calculate
 * (inversion.py:375-391) and npass direct_inversion_p2p_guidance_forward passes (p2p_guidance_forward.py:135-173; the
 * reference runs an AttentionStore reconstruction pass and the edit pass) advance in lock step, one UNet launch of
 * (1 + npass) * 4 * nimg rows per timestep.  ctrl_host: nullable or [npass][nimg] (kind 0 = no attention edit).
 * noise_loss_out [nsteps][nimg][2][4][h][w]; latents_out [npass][nimg][2][4][h][w]. */
int pnpi_direct_edit(pnpi_ctx* ctx, const float* ddim_latents /*[nsteps+1][nimg][...]*/, int nimg, const float* context4,
                     int npass, const pnpi_ctrl_desc* ctrl_host, int offset_rows, int nsteps, const int* timesteps_host,
                     float guidance_scale, const float* offset_scale_host /*[nsteps], nullable*/, float* noise_loss_out,
                     float* latents_out);

/* The pruned-equivalent schedule (SURVEY.md 8a Note D; an algebraic reformulation of the same edit, parity-tested against the
 * faithful loops): with the direct-inversion offset the source latent of every step IS x*_{t-1}, so it is assigned from the stored
 * inversion trajectory and only [uncond_tgt, cond_src, cond_tgt] are evaluated -- one 3-row UNet launch per step and image, 200
 * sample-forwards per image instead of 650.  latents_out [nimg][2][4][h][w] = (x*_0, edited latent). */
int pnpi_direct_edit_pruned(pnpi_ctx* ctx, const float* ddim_latents, int nimg, const float* context4,
                            const pnpi_ctrl_desc* ctrl_host /* nullable or [nimg] */, int nsteps, const int* timesteps_host,
                            float guidance_scale, float* latents_out);

/* ---- kernel-level entry points (used by tests/ and bench.py to exercise single kernels) ------------------------- */
int pnpi_op_conv(pnpi_ctx* ctx, const void* x1_nhwc_f16, const void* x2_nhwc_f16, int C1, int C2, int B, int H, int W,
                 int ksize, int stride, int pad, int upsample, int Ho, int Wo, const void* w_f16 /*[N][k*k*(C1+C2)]*/,
                 const float* bias, const void* residual_f16, int N, void* out_nhwc_f16, int force_cfg, int force_split);
/* pnpi_op_conv + the per-(m-tile, channel) (sum, sum of squares) partials its epilogue produces for the consumer GroupNorm
 * (replaces the statistics pass of torch.nn.GroupNorm, my_diffusers/models/resnet.py:296): stats_out [ceil(M/tile_rows)][N][2]
 * fp32 over the stored fp16 values; *tile_rows_out = rows per m-tile, 0 if this launch configuration produced none. */
int pnpi_op_conv_stats(pnpi_ctx* ctx, const void* x1_nhwc_f16, const void* x2_nhwc_f16, int C1, int C2, int B, int H, int W,
                       int ksize, int stride, int pad, int upsample, int Ho, int Wo, const void* w_f16, const float* bias,
                       const void* residual_f16, int N, void* out_nhwc_f16, int force_cfg, int force_split, float* stats_out,
                       int* tile_rows_out);
/* process-wide kernel tuning knobs: variant A/B inside one process (tools/fwd_ab.py, tools/fwd_tune.py) and tests of non-default
 * variants; PNPI_EINVAL for an unknown key.  They are plain process globals read by every context's launches: set them while no other
 * thread is inside a pnpi_* call (the product never calls this; several contexts on several threads -- P2PEditor.edit_stream_in_flight --
 * only READ them).  A non-zero "igemm_vpp" / "igemm_sched" and "igemm_v128" / "igemm_v320" = 11, 12, 15 select ablation instances
 * that exist only in a library built with `python -m pnpinversion_amd.build --ablations` (-DPNPI_ABLATIONS=1 -> csrc/libpnpi_ablations.so,
 * loaded with PNPI_LIBRARY=<path>); the product library REJECTS those values here with PNPI_EINVAL -- and "attn_pipe" = 1 / 2 (round 6: the
 * half-tile software-pipelined forms of the 64-wide flash kernel, measured slower) likewise.  "gn_slab" (0): 1 = a split-K launch whose
 * output goes to a small-map GroupNorm leaves its combine to that kernel (bit-identical, measured slower: profiles/round5_gn_slab_ab.txt).
 * Keys (default): "text_kv" (1) / "temb_cache" (1) per-loop caches; "gn_inline_rows" (0)
 * one-launch GroupNorm below this many rows; "igemm_dma" (1) LDS-DMA kernel family; "igemm_table" (1) measured tile table before the
 * cost model; "igemm_wide" (1) 128x320 / 128x256 tiles; "igemm_deep_rings" (1) deeper LDS rings on sparse launches; "igemm_vt_lds" (1)
 * transposed V^T epilogue through LDS; "igemm_bias_init" (1) bias as the accumulators' initial value; "igemm_sched" (0 / 1) hand-scheduled GEMM main loop (fragment
 * reads behind counted lgkmcnt waits, DMA instructions between the MFMAs); "attn_bwd_flash" (1) attention backward of the null-text path without the score matrices in memory (0: materialised everywhere, 2: flash form with the two-pass dQ even where the forward left its log-sum-exp, 3: flash form for self-attention only); "attn_vt_perm" (1) V^T of the 4096-token self-attention sites stored in the permuted key order the LDS-DMA flash kernel reads with one
 * 16-byte fragment load ("op_attention_vt_perm": pnpi_op_attention is handed such a V^T -- kernel tests); "igemm_res_late" (0) residual
 * added in the store loop; "igemm_force_cfg" (-1) / "igemm_force_split" (0) one tile id / split-K for every launch (sweeps);
 * "igemm_table_near" (1) nearest-row-count table entry for untabled M; "igemm_v128", "igemm_v64", "igemm_v256", "igemm_v320",
 * "igemm_v256n" variant of a tile id; "tile_order" (-1) XCD traversal order; "igemm_vpp" (0) ablations of the 8-wave ping-pong kernel
 * (1 no MFMAs, 2 no DMA, 3 DMA only, 4 MFMAs only -- all but 0 produce garbage, timing only); "igemm_tapin" (0) the ping-pong kernel
 * walks a 3 x 3 convolution's K channel-slab-major; "igemm_pp_only_m" / "_n" / "_k" (0) the table's ping-pong entries apply only to
 * launches with that M / N / K (n < 0: to none; tools/pp_bisect.py); "attn_aug" (1) the 40-wide flash self-attention carries the running
 * maximum and the row sum in the padding column of Q / K / V (0: explicit shift and sum on the VALU; the column is written either way and ignored);"op_attention_aug" (0) pnpi_op_attention is handed K and V whose column 40 holds 1.0
 * and runs that kernel (kernel tests). */
int pnpi_set_tuning(const char* key, int value);
/* Host-only query of the measured tile table launch_igemm consults before its cost model (no device work; used by the CPU tests):
 * returns 1 for an exact {M, N, K, ksize} entry, 2 when the same layer (N, K, ksize) is listed at another row count and the entry
 * nearest in M (at most 4x away) is used, 0 for none; cfg / split / entry_m (each nullable) receive the entry. */
int pnpi_tile_table_lookup(int M, int N, int K, int ksize, int* cfg, int* split, int* entry_m);
int pnpi_op_gemm(pnpi_ctx* ctx, const void* a_f16, int lda, const void* w_f16, int ldw, int M, int N, int K, float alpha,
                 const float* bias, const void* residual_f16, void* out_f16, int ldo, int vt_col0, void* outT,
                 int vt_ld, int vt_f32, int rows_per_batch, int force_cfg, int force_split);
/* out[m][j] = x * gelu(gate) with the N = 2*I weight rows / bias packed as [x(32) | gate(32)] groups (fused GEGLU epilogue) */
int pnpi_op_gemm_geglu(pnpi_ctx* ctx, const void* a_f16, int lda, const void* w_f16, int ldw, int M, int N, int K,
                       const float* bias, void* out_f16, int ldo);
int pnpi_op_groupnorm(pnpi_ctx* ctx, const void* x1, const void* x2, int C1, int C2, int B, int HW, int groups, float eps,
                      const float* gamma, const float* beta, int silu, void* out);
int pnpi_op_layernorm(pnpi_ctx* ctx, const void* x, int M, int C, float eps, const float* gamma, const float* beta, void* out);
int pnpi_op_geglu(pnpi_ctx* ctx, const void* x, int M, int inner, void* out);
int pnpi_op_softmax_rows(pnpi_ctx* ctx, void* x, int M, int N, int ld);
/* Activation-gradient kernels of the null-text / null-latent path (NullInversion.null_optimization's loss.backward(), inversion.py:
 * 196-225) -- groundwork, kernel-level only (no method string reaches them yet).  fp16 tensors in the forward's layouts.
 *   layernorm_bwd: dx of torch.nn.LayerNorm from the saved input;  groupnorm_bwd: dx (dense [B][HW][C1+C2]) of GroupNorm(+SiLU) over the
 *   virtual concat (x1, x2);  geglu_bwd: d(projection) in the interleaved [x(32)|gate(32)] layout;  softmax_bwd_rows: dS = scale * P *
 *   (dP - rowsum(dP * P)) as fp16 rows padded to ld;  accumulate: dst += src;  sumpool2x2: nearest-2x upsample backward;  zero_stuff2 +
 *   repack_dgrad: stride-2 / stride-1 convolution dgrad through the forward kernel (wd[c][k*k-1-tap][n] = w[n][tap][c]);
 *   null_text_loss: mse(prev_step(cfg(eps_u, eps_c)), target) and its gradient w.r.t. eps_u (times grad_scale);  adam_step: torch.optim.Adam
 *   defaults, step k >= 1, gradient times inv_scale. */
int pnpi_op_layernorm_bwd(pnpi_ctx* ctx, const void* x, const void* dy, int M, int C, float eps, const float* gamma, void* dx);
int pnpi_op_groupnorm_bwd(pnpi_ctx* ctx, const void* x1, const void* x2, int C1, int C2, int B, int HW, int groups, float eps,
                          const float* gamma, const float* beta, int silu, const void* dy, void* dx);
int pnpi_op_geglu_bwd(pnpi_ctx* ctx, const void* h, const void* dy, int M, int inner, void* dh);
int pnpi_op_softmax_bwd_rows(pnpi_ctx* ctx, const float* P, const float* dP, int R, int N, int ld, float scale, void* dS);
int pnpi_op_accumulate(pnpi_ctx* ctx, void* dst, const void* src, size_t n);
int pnpi_op_sumpool2x2(pnpi_ctx* ctx, const void* dup, int B, int H, int W, int C, void* dx);
int pnpi_op_zero_stuff2(pnpi_ctx* ctx, const void* dy, int B, int Ho, int Wo, int C, void* out);
int pnpi_op_repack_dgrad(pnpi_ctx* ctx, const void* w, int N, int taps, int Cin, void* wd);
int pnpi_op_null_text_loss(pnpi_ctx* ctx, const float* eps_u, const float* eps_c, const float* x, const float* target, int n, float w,
                           float c_x, float c_e, float grad_scale, void* d_eps_u, float* loss);
int pnpi_op_adam_step(pnpi_ctx* ctx, float* p, float* m, float* v, const float* g, int n, int k, float lr, float inv_scale);
/* Attention backward (CrossAttention.forward of my_diffusers/models/attention.py:217-259 differentiated), materialised per (row, head):
 * q / k / v are [B*N][ld] fp16 views with head h at columns off + h * Dp (dh real columns, pad columns zero), d_o is [B*Nq][ldo] with
 * heads * dh columns; dq / dk / dv receive the gradients in the layout of q / k / v (pad columns untouched).  scratch: device memory
 * of at least pnpi_op_attention_bwd_scratch_bytes(Nq, Nk, dh) bytes. */
int pnpi_op_attention_bwd(pnpi_ctx* ctx, const void* q, int ldq, int q_off, const void* k, int ldk, int k_off, const void* v, int ldv,
                          int v_off, const void* d_o, int ldo, int heads, int Nq, int Nk, int Dp, int dh, float scale, int B, void* dq,
                          void* dk, void* dv, void* scratch, size_t scratch_bytes);
size_t pnpi_op_attention_bwd_scratch_bytes(int Nq, int Nk, int dh);

/* ---- differentiable UNet forward (null-text path; groundwork) ---------------------------------------------------------------
 * pnpi_unet_context_grad: eps = unet(latents [1][4][h][w], t, context [1][77][768]) with every activation recorded, then the reverse
 * walk: d_context_out [77][768] = (d eps / d context)^T d_eps for the given d loss / d eps (fp32 in the layout of eps; multiply it by
 * a power-of-two loss scale -- activation gradients travel in fp16 -- and divide d_context_out by it).  eps_out nullable.
 * pnpi_null_text_optimize: NullInversion.null_optimization (models/p2p/inversion.py:196-225) for one image: ddim_latents
 * [nsteps + 1][4*h*w] (x*_0 first), the "" and the source-prompt embeddings, the denoising timesteps; uncond_out [nsteps][77][768]
 * (the optimised embedding of every step), iters_out [nsteps] (nullable: Adam iterations run per step), losses_out_host
 * [nsteps][num_inner_steps] (nullable, host: the loss of every Adam iteration, -1 where the early stop skipped it).  Context needs
 * max_unet_rows large enough for one row's activations kept without reuse (12 is). */
/* pnpi_edit_loop with per-step unconditional embeddings uncond_steps [nsteps][nimg][77][768] (the output of pnpi_null_text_optimize):
 * p2p_guidance_forward's `uncond_embeddings[i].expand(...)` (p2p_guidance_forward.py:56-57), or with uncond_first_only the single-branch
 * variant (:92).  No direct-inversion offset, no reconstruction guidance (what the null-text method strings use). */
int pnpi_edit_loop_uncond_steps(pnpi_ctx* ctx, const float* x_T, int nimg, const float* context4, const pnpi_ctrl_desc* ctrl_host, int nsteps,
                                const int* timesteps_host, float guidance_scale, int prox, float quantile, const float* uncond_steps,
                                int uncond_first_only, float* latents_out);
/* pnpi_edit_loop_uncond_steps with the reconstruction guidance of pnpi_edit_loop (recon nullable): the edit pass of
 * P2PEditor.edit_image_null_text_inversion_proximal_guidanca(use_reconstruction_guidance=True), models/p2p_editor.py:607-627. */
int pnpi_edit_loop_uncond_steps_recon(pnpi_ctx* ctx, const float* x_T, int nimg, const float* context4, const pnpi_ctrl_desc* ctrl_host, int nsteps,
                                      const int* timesteps_host, float guidance_scale, int prox, float quantile, const float* uncond_steps,
                                      int uncond_first_only, const pnpi_recon_desc* recon, float* latents_out);
int pnpi_unet_context_grad(pnpi_ctx* ctx, const float* latents, int t, const float* context, const float* d_eps, float* eps_out,
                           float* d_context_out);
int pnpi_null_text_optimize(pnpi_ctx* ctx, const float* ddim_latents, const float* ctx_uncond, const float* ctx_cond, int nsteps,
                            const int* timesteps_host, float guidance_scale, int num_inner_steps, float epsilon, float* uncond_out,
                            int* iters_out_host, float* losses_out_host);
/* replaces DirectInversion.null_latent_calculate                               models/p2p/inversion.py:419-460
 * ("ablation_null-latent-inversion+p2p"): the null-text optimisation of the source row's unconditional embedding per step, turned into
 * per-step latent offsets noise_loss_out [nsteps][2][4*h*w] (device) for pnpi_edit_loop.  context4 = [unc_src, unc_tgt, cond_src,
 * cond_tgt]; iters_out_host [nsteps] and losses_out_host [nsteps][num_inner_steps] nullable (host; -1 = iteration not run). */
int pnpi_null_latent_calculate(pnpi_ctx* ctx, const float* ddim_latents, const float* context4, int nsteps, const int* timesteps_host,
                               float guidance_scale, int num_inner_steps, float epsilon, float* noise_loss_out, int* iters_out_host,
                               float* losses_out_host);
int pnpi_op_attention(pnpi_ctx* ctx, const void* q, int ldq, int q_off, const void* k, int ldk, int k_off, const void* vt,
                      int ldv, void* o, int ldo, int heads, int Nq, int Nk, int Dp, int dh, float scale,
                      const int* rows_dev /*[nrows][4]*/, int nrows);
int pnpi_op_cross_edit(pnpi_ctx* ctx, const void* q, int ldq, int q_off, const void* k, int ldk, int k_off, const void* vt,
                       int ldv, void* o, int ldo, int heads, int Nq, int Nk, int Dp, int dh, float scale,
                       const int* pairs_dev, int npairs, const void* mmatT_f16, const float* c1, const float* c2,
                       const float* lb_alpha, float* lb_acc, int lb_slot0, int lb_nslots);
int pnpi_op_local_blend(pnpi_ctx* ctx, const float* lb_acc, int nslots, int map_hw, int lat_hw, int C, float th,
                        float* latents, int nimg);
/* The same with LocalBlend substruct_words: lb_acc holds 4 planes per slot (blend src / tgt, substruct src / tgt); the mask is
 * (blend maps pooled > th) & ~(substruct maps unpooled > th_sub)  (attention_control.py:97-118). */
int pnpi_op_local_blend_sub(pnpi_ctx* ctx, const float* lb_acc, int nslots, int map_hw, int lat_hw, int C, float th, float th_sub,
                            float* latents, int nimg);

#ifdef __cplusplus
}
#endif
#endif /* PNPI_H */
