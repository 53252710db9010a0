// libpnpi: C-ABI entry points + the static SD-1.x UNet / VAE graph executor and the device-resident DI / P2P loops.
// See include/pnpi.h for the reference interface each entry point replaces.
// One translation unit in six files: this one (error helpers, the C ABI: context life cycle, level-1 operators, the level-2 loops, the
// null-text optimisation, kernel-level test hooks) includes, in order,
//   api_weights.inc   weight slots of the packed arena and the model builder
//   api_graph.inc     profiling records, activation tape, op wrappers, ResNet / transformer blocks, unet_fwd
//   api_backward.inc  reverse walk over the tape (gradient w.r.t. the unconditional embedding)
//   api_vae.inc       AutoencoderKL encoder / decoder graph
//   api_ctrl.inc      per-loop text K / V cache, controller descriptor -> device tables
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <functional>
#include <memory>
#include <string>

#include "model.h"

static int g_text_kv = 1;      // tuning "text_kv" = 0: project the text context inside every forward, as the reference does (A/B)
static int g_temb_cache = 1;   // tuning "temb_cache" = 0: three GEMVs per forward

// ---------------------------------------------------------------------------------------------------- error helpers
// explicit status + message
static int fail(pnpi_ctx* c, int status, const char* what) {
  c->err = what;
  return status;
}
// raw launch result: > 0 is a hipError_t, < 0 an argument/shape rejection by a launch wrapper
static int fail_launch(pnpi_ctx* c, int code, const char* what) {
  char buf[640];
  if (code > 0) {
    snprintf(buf, sizeof(buf), "HIP error %d (%s) in %s", code, hipGetErrorString((hipError_t)code), what);
    c->err = buf;
    return PNPI_EHIP;
  }
  snprintf(buf, sizeof(buf), "launch wrapper rejected arguments (code %d) in %s", code, what);
  c->err = buf;
  return PNPI_ESHAPE;
}
// CK: raw launch codes.  CKP: propagate an already-mapped pnpi_status (message already set).
#define CK(x)                                 \
  do {                                        \
    int _r = (x);                             \
    if (_r) return fail_launch(c, _r, #x);    \
  } while (0)
#define CKP(x)                \
  do {                        \
    int _r = (x);             \
    if (_r) return _r;        \
  } while (0)
#define CKH(x)                                     \
  do {                                             \
    hipError_t _e = (x);                           \
    if (_e != hipSuccess) return fail_launch(c, (int)_e, #x); \
  } while (0)

#include "api_weights.inc"
#include "api_graph.inc"
#include "api_backward.inc"
#include "api_vae.inc"
#include "api_ctrl.inc"

// ---------------------------------------------------------------------------------------------------- C ABI
extern "C" {

void pnpi_config_sd1(pnpi_model_config* g) {
  memset(g, 0, sizeof(*g));
  g->in_channels = 4; g->out_channels = 4; g->n_blocks = 4;
  int boc[4] = {320, 640, 1280, 1280}, att[4] = {1, 1, 1, 0}, vb[4] = {128, 256, 512, 512};
  for (int i = 0; i < 4; ++i) { g->block_out_channels[i] = boc[i]; g->block_has_attn[i] = att[i]; g->vae_block_out_channels[i] = vb[i]; }
  g->layers_per_block = 2; g->heads = 8; g->cross_dim = 768; g->ctx_len = 77; g->sample_size = 64; g->norm_groups = 32;
  g->n_train_timesteps = 1000; g->vae_in_channels = 3; g->vae_latent_channels = 4; g->vae_n_blocks = 4;
  g->vae_layers_per_block = 2; g->vae_norm_groups = 32;
  g->clip_layers = 12; g->clip_heads = 12; g->clip_intermediate = 3072; g->clip_vocab = 49408;   /* CLIP ViT-L/14 text model */
}

const char* pnpi_last_error(const pnpi_ctx* c) { return c ? c->err.c_str() : "null ctx"; }

static int validate_config(pnpi_ctx* c) {
  const pnpi_model_config& g = c->cfg;
  if (g.n_blocks < 2 || g.n_blocks > 4 || g.vae_n_blocks < 2 || g.vae_n_blocks > 4) return fail(c, PNPI_ESHAPE, "n_blocks must be 2..4");
  for (int i = 0; i < g.n_blocks; ++i) {
    int ch = g.block_out_channels[i];
    if (ch % g.norm_groups || ch % 8 || ch % g.heads) return fail(c, PNPI_ESHAPE, "block_out_channels must divide by groups, heads and 8");
    int dh = ch / g.heads;
    if (g.block_has_attn[i] && (dh % 4 || dh > 160)) return fail(c, PNPI_ESHAPE, "head dim must be a multiple of 4 and <= 160");
  }
  for (int i = 0; i < g.vae_n_blocks; ++i)
    if (g.vae_block_out_channels[i] % g.vae_norm_groups || g.vae_block_out_channels[i] % 8) return fail(c, PNPI_ESHAPE, "vae channels");
  if (g.cross_dim % 8 || g.ctx_len > 96 || g.in_channels > 8 || g.vae_latent_channels > 4 || g.vae_in_channels != 3)
    return fail(c, PNPI_ESHAPE, "cross_dim/ctx_len/in_channels unsupported");
  if (g.sample_size % (1 << (g.n_blocks - 1))) return fail(c, PNPI_ESHAPE, "sample_size must divide by 2^(n_blocks-1)");
  if (g.block_out_channels[0] % 2) return fail(c, PNPI_ESHAPE, "C0 must be even");
  if (g.clip_layers < 0 || g.clip_layers > 48) return fail(c, PNPI_ESHAPE, "clip_layers out of range");
  if (g.clip_layers > 0) {
    if (g.clip_heads <= 0 || g.cross_dim % g.clip_heads) return fail(c, PNPI_ESHAPE, "cross_dim must divide by clip_heads");
    const int dh = g.cross_dim / g.clip_heads;
    if (dh % 32 || dh > 160) return fail(c, PNPI_ESHAPE, "CLIP head dim must be a multiple of 32 and <= 160");
    if (g.clip_intermediate <= 0 || g.clip_intermediate % 8 || g.clip_vocab <= 0) return fail(c, PNPI_ESHAPE, "clip_intermediate / clip_vocab invalid");
  }
  return 0;
}

static int clip_fwd(pnpi_ctx* c, const int* ids, int n, float* out);

static int create_impl(pnpi_ctx** out, const pnpi_model_config* cfg, int device, void* hip_stream, int max_unet_rows, int max_vae_images, pnpi_ctx* parent);
int pnpi_create(pnpi_ctx** out, const pnpi_model_config* cfg, int device, void* hip_stream, int max_unet_rows, int max_vae_images) {
  if (!out || !cfg) return PNPI_EINVAL;
  return create_impl(out, cfg, device, hip_stream, max_unet_rows, max_vae_images, nullptr);
}
// A further context on the SAME packed weights: the new context borrows the parent's weight arena (read-only from here on) instead of
// holding a copy -- several images in flight on one GPU (own stream, own workspaces, own caches) then share one 1.9 GB arena in the
// Infinity Cache / L2 instead of competing with N copies of it.  The parent must outlive the child and its weights must not be
// reloaded while children exist (a child refuses pnpi_load_weights).
int pnpi_create_shared(pnpi_ctx** out, pnpi_ctx* parent, void* hip_stream, int max_unet_rows, int max_vae_images) {
  if (!out || !parent) return PNPI_EINVAL;
  if (parent->warena_borrowed) return fail(parent, PNPI_ESTATE, "pnpi_create_shared: the parent itself borrows its weights; share from the owner");
  return create_impl(out, &parent->cfg, parent->device, hip_stream, max_unet_rows, max_vae_images, parent);
}
static int create_impl(pnpi_ctx** out, const pnpi_model_config* cfg, int device, void* hip_stream, int max_unet_rows, int max_vae_images, pnpi_ctx* parent) {
  pnpi_ctx* c = new pnpi_ctx();
  *out = c;
  c->cfg = *cfg; c->device = device; c->st = (hipStream_t)hip_stream; c->max_rows = max_unet_rows; c->max_vae = max_vae_images;
  c->dry = true; c->sched_set = false; c->final_alpha = 0.f;
  c->splitk_ws = nullptr; c->gn_partial = nullptr; c->temb_table = nullptr;
  memset(&c->ctr, 0, sizeof(c->ctr));
  CKP(validate_config(c));
  CKH(hipSetDevice(device));
  CK(igemm_init());
  const pnpi_model_config& g = c->cfg;
  // pass 1: measure the weight arena; pass 2: real pointers
  build_model(c);
  const size_t wbytes = align_up(c->warena.peak + 4096, 4096);
  if (parent) {
    if (parent->warena.cap != wbytes) return fail(c, PNPI_ESTATE, "pnpi_create_shared: the parent's weight arena has another size");
    c->warena.base = parent->warena.base; c->warena_borrowed = true;
    c->warena_ref = parent->warena_ref; c->warena_ref->refs.fetch_add(1);      // the arena lives until its last user is destroyed
  } else {
    CKH(hipMalloc((void**)&c->warena.base, wbytes));
    c->warena_ref = new pnpi_ctx::ArenaRef{c->warena.base, {1}};
    CKH(hipMemsetAsync(c->warena.base, 0, wbytes, c->st));
  }
  c->warena.cap = wbytes; c->warena.reset(); c->warena.peak = 0;
  build_model(c);
  if (parent)      // the same deterministic layout: every slot points at the parent's packed tensor and is loaded iff the parent's is
    for (auto& kv : c->slots) { auto it = parent->slots.find(kv.first); kv.second.loaded = it != parent->slots.end() && it->second.loaded; }
  else
    for (const pnpi_ctx::AugBias& ab : c->aug_biases) {      // constants of the arena (not part of any checkpoint): the memset above left zeros
      std::vector<float> hb((size_t)3 * ab.heads * ab.Dp, 0.f);
      for (int part = 1; part < 3; ++part)
        for (int hh = 0; hh < ab.heads; ++hh) hb[(size_t)part * ab.heads * ab.Dp + hh * ab.Dp + ab.dh] = 1.f;
      CKH(hipMemcpyAsync(ab.p, hb.data(), hb.size() * sizeof(float), hipMemcpyHostToDevice, c->st));
      CKH(hipStreamSynchronize(c->st));                        // hb is a temporary
    }
  // small persistent buffers
  const int C0 = g.block_out_channels[0], TE = 4 * C0;
  c->splitk_bytes = ((size_t)96 << 20) + (size_t)(max_unet_rows > 12 ? max_unet_rows - 12 : 0) * ((size_t)8 << 20);   // grows with the rows per launch
  CKH(hipMalloc((void**)&c->splitk_ws, c->splitk_bytes));
  size_t gnp = (size_t)(max_unet_rows > max_vae_images ? max_unet_rows : max_vae_images) * (128 * 64 * 2 + 4096 * 2) * sizeof(float);
  CKH(hipMalloc((void**)&c->gn_partial, gnp));
  CKH(hipMalloc((void**)&c->temb_h, TE * sizeof(float)));
  CKH(hipMalloc((void**)&c->temb_emb, TE * sizeof(float)));
  CKH(hipMalloc((void**)&c->bias_scratch, (size_t)c->unet.temb_total * sizeof(float)));
  c->bias_eff = c->bias_scratch;
  if (max_unet_rows > 0) {
    // per-timestep (conv1 bias + time embedding) table: [n_train][temb_total] fp32 (81 MB for SD-1.x), rows filled on first use
    CKH(hipMalloc((void**)&c->bias_tab, (size_t)g.n_train_timesteps * c->unet.temb_total * sizeof(float)));
    c->bias_valid.assign(g.n_train_timesteps, 0);
    c->tkv.cap = text_kv_bytes(c, max_unet_rows);
    CKH(hipMalloc((void**)&c->tkv.base, c->tkv.cap));
    CKH(hipMemset(c->tkv.base, 0, c->tkv.cap));   // (text_kv_precompute clears each V^T block again: the layout depends on the row count)
  }
  // sinusoidal timestep table, get_timestep_embedding(flip_sin_to_cos=True, downscale_freq_shift=0), fp64 -> fp32
  // (my_diffusers/models/embeddings.py:21-60)
  {
    const int half = C0 / 2;
    std::vector<float> tab((size_t)g.n_train_timesteps * C0);
    for (int t = 0; t < g.n_train_timesteps; ++t)
      for (int i = 0; i < half; ++i) {
        double e = exp(-log(10000.0) * (double)i / (double)half);
        double a = (double)t * e;
        tab[(size_t)t * C0 + i] = (float)cos(a);          // flipped: cos first
        tab[(size_t)t * C0 + half + i] = (float)sin(a);
      }
    CKH(hipMalloc((void**)&c->temb_table, tab.size() * sizeof(float)));
    CKH(hipMemcpy(c->temb_table, tab.data(), tab.size() * sizeof(float), hipMemcpyHostToDevice));
  }
  // controller / loop arena
  {
    const size_t E = (size_t)g.in_channels * g.sample_size * g.sample_size;
    size_t cap = ((size_t)8 << 20) + (size_t)max_unet_rows * E * sizeof(float) * 8 +
                 (size_t)max_unet_rows * (96 * 96 * 2 + 64 * 2 * 96 * 4 * 2 + (size_t)c->unet.lb_nslots * 4 * c->unet.lb_tokens * 4);
    CKH(hipMalloc((void**)&c->ctrl_arena.base, cap));
    c->ctrl_arena.cap = cap;
  }
  {
    c->rows_ident_n = max_unet_rows > 8 ? max_unet_rows : 8;
    std::vector<int> idt((size_t)c->rows_ident_n * 4);
    for (int r = 0; r < c->rows_ident_n; ++r) idt[4 * r] = idt[4 * r + 1] = idt[4 * r + 2] = idt[4 * r + 3] = r;
    CKH(hipMalloc((void**)&c->rows_ident, idt.size() * sizeof(int)));
    CKH(hipMemcpy(c->rows_ident, idt.data(), idt.size() * sizeof(int), hipMemcpyHostToDevice));
  }
  // dry runs to size the activation workspaces
  c->dry = true;
  c->persist = Bump(); c->temp = Bump();
  size_t ppeak = 0, tpeak = 0;
  if (max_unet_rows > 0) {
    int r = unet_fwd(c, nullptr, max_unet_rows, 0, nullptr, false, 0, nullptr);
    if (r) return r;
    ppeak = c->persist.peak; tpeak = c->temp.peak;
  }
  if (!c->clip.layers.empty()) {
    c->persist.reset(); c->temp.reset(); c->persist.peak = 0; c->temp.peak = 0;
    int r = clip_fwd(c, nullptr, c->rows_ident_n, nullptr);
    if (r) return r;
    if (c->persist.peak > ppeak) ppeak = c->persist.peak;
    if (c->temp.peak > tpeak) tpeak = c->temp.peak;
  }
  if (max_vae_images > 0) {
    const int S = g.sample_size, F = 1 << (g.vae_n_blocks - 1);
    c->persist.reset(); c->temp.reset(); c->persist.peak = 0; c->temp.peak = 0;
    (void)palloc(c, (size_t)max_vae_images * S * F * S * F * 8);
    int r = vae_encode_fwd(c, nullptr, max_vae_images, S * F, S * F, nullptr);
    if (r) return r;
    if (c->persist.peak > ppeak) ppeak = c->persist.peak;
    if (c->temp.peak > tpeak) tpeak = c->temp.peak;
    c->persist.reset(); c->temp.reset(); c->persist.peak = 0; c->temp.peak = 0;
    (void)palloc(c, (size_t)max_vae_images * S * S * 8);
    (void)c->persist.alloc((size_t)max_vae_images * 3 * S * F * S * F * sizeof(float));
    r = vae_decode_fwd(c, nullptr, max_vae_images, S, S, nullptr);
    if (r) return r;
    if (c->persist.peak > ppeak) ppeak = c->persist.peak;
    if (c->temp.peak > tpeak) tpeak = c->temp.peak;
  }
  ppeak = align_up(ppeak + (1 << 20), 4096); tpeak = align_up(tpeak + (1 << 20), 4096);
  CKH(hipMalloc((void**)&c->persist.base, ppeak));
  CKH(hipMalloc((void**)&c->temp.base, tpeak));
  c->persist.cap = ppeak; c->temp.cap = tpeak; c->persist.reset(); c->temp.reset();
  c->dry = false;
  memset(&c->ctr, 0, sizeof(c->ctr));
  // identity attention-row table for the controller-free path
  CKP(setup_ctrl(c, nullptr, 0, max_unet_rows > 0 ? max_unet_rows : 1));
  CKH(hipStreamSynchronize(c->st));
  return 0;
}

void pnpi_destroy(pnpi_ctx* c) {
  if (!c) return;
  (void)hipStreamSynchronize(c->st);
  // the weight arena is shared with the contexts pnpi_create_shared made from this one (or borrowed from the one this was made from):
  // whoever is destroyed last frees it -- destroying the owner first does not pull the weights from under a running child
  void* arena = nullptr;
  if (c->warena_ref && c->warena_ref->refs.fetch_sub(1) == 1) { arena = c->warena_ref->base; delete c->warena_ref; }
  void* bufs[] = {arena, c->persist.base, c->temp.base, c->ctrl_arena.base, c->splitk_ws, c->gn_partial, c->gn_bwd_ws,
                  c->temb_table, c->temb_h, c->temb_emb, c->bias_scratch, c->bias_tab, c->tkv.base, c->rows_ident};
  for (void* b : bufs) (void)hipFree(b);
  if (c->tape) {
    (void)hipFree(c->tape->garena.base); (void)hipFree(c->tape->d_ctx); (void)hipFree(c->tape->attn_scratch);
    for (auto& kv : c->tape->wd) (void)hipFree(kv.second);
    delete c->tape;
  }
  delete c;
}

static void invalidate_derived(pnpi_ctx* c) {     // caches of functions of the weights
  std::fill(c->bias_valid.begin(), c->bias_valid.end(), 0);
  c->tkv.rows = 0; c->tkv.use = false;
  if (c->tape) { for (auto& kv : c->tape->wd) (void)hipFree(kv.second); c->tape->wd.clear(); }     // dgrad repacks of the old weights
}

int pnpi_load_weights(pnpi_ctx* c, const pnpi_named_tensor* ts, int n) {
  if (!c || !ts) return PNPI_EINVAL;
  if (c->warena_borrowed) return fail(c, PNPI_ESTATE, "this context borrows its weights (pnpi_create_shared): load them into the owning context");
  invalidate_derived(c);
  for (int i = 0; i < n; ++i) {
    const pnpi_named_tensor& t = ts[i];
    std::string name = t.name;
    // diffusers 0.3-0.10 register the stride-2 conv twice ("conv" and "Conv2d_0"): accept either spelling
    size_t pos = name.find(".downsamplers.0.Conv2d_0.");
    if (pos != std::string::npos) name.replace(pos, strlen(".downsamplers.0.Conv2d_0."), ".downsamplers.0.conv.");
    if (name.compare(0, 16, "clip.text_model.") == 0) name = "clip." + name.substr(16);   // transformers < 5 key spelling
    auto it = c->slots.find(name);
    if (it == c->slots.end()) continue;  // unknown keys are ignored (e.g. buffers)
    Slot& s = it->second;
    size_t numel = 1;
    for (int d = 0; d < t.ndim; ++d) numel *= (size_t)t.shape[d];
    if (s.kind == 0) {
      if (numel != (size_t)s.rows * s.cols * s.taps) { c->err = "shape mismatch for " + name; return PNPI_ESHAPE; }
      CK(launch_repack_matrix(t.data, t.dtype, s.rows, s.cols, s.taps, (half_t*)s.dst, s.dst_ld, s.cin_pad, s.row0, s.dh, s.Dp, c->st,
                              s.ilv_half));
    } else {
      if (numel != (size_t)s.n) { c->err = "shape mismatch for " + name; return PNPI_ESHAPE; }
      CK(launch_repack_vec(t.data, t.dtype, s.n, (float*)s.dst, c->st, s.ilv_half));
    }
    s.loaded = true;
  }
  return 0;
}

int pnpi_missing_weights(const pnpi_ctx* c, char* names_out, size_t cap) {
  int missing = 0;
  size_t used = 0;
  if (names_out && cap) names_out[0] = 0;
  for (auto& kv : c->slots)
    if (!kv.second.loaded && kv.first.compare(0, 5, "clip.") != 0) {   // the text encoder reports through pnpi_text_encode
      ++missing;
      if (names_out && used + kv.first.size() + 2 < cap) {
        memcpy(names_out + used, kv.first.c_str(), kv.first.size());
        used += kv.first.size();
        names_out[used++] = '\n';
        names_out[used] = 0;
      }
    }
  return missing;
}

int pnpi_weight_arena(pnpi_ctx* c, void** ptr, size_t* bytes) {
  if (!c || !ptr || !bytes) return PNPI_EINVAL;
  *ptr = c->warena.base; *bytes = c->warena.cap;
  // after a broadcast every slot of the receiving ranks is populated
  return 0;
}

int pnpi_mark_all_loaded(pnpi_ctx* c) {
  invalidate_derived(c);            // the arena was just overwritten by the broadcast
  for (auto& kv : c->slots) kv.second.loaded = true;
  return 0;
}

int pnpi_set_scheduler(pnpi_ctx* c, const float* ac, int n_train, float final_alpha) {
  if (!c || !ac || n_train != c->cfg.n_train_timesteps) return PNPI_EINVAL;
  c->ac.assign(ac, ac + n_train);
  c->final_alpha = final_alpha;
  c->sched_set = true;
  return 0;
}

int pnpi_get_counters(const pnpi_ctx* c, pnpi_counters* out) { if (!c || !out) return PNPI_EINVAL; *out = c->ctr; return 0; }
int pnpi_reset_counters(pnpi_ctx* c) { if (!c) return PNPI_EINVAL; memset(&c->ctr, 0, sizeof(c->ctr)); return 0; }

// Matrix-pipe clock calibration (bench.py's `clock` object): every SIMD of the chip issues v_mfma_f32_32x32x16_f16 back to back on register
// operands with pseudo-random fp16 values (two waves per SIMD, four independent accumulators each): 32 matrix-pipe cycles per instruction
// (MI355X_MICROARCH.md, "Per-instruction cycle constants"), so cycles / elapsed = the clock the part holds under a pure MFMA load.
__global__ void __launch_bounds__(512, 2) mfma_clock_kernel(int iters, float* sink) {
  const int lane = threadIdx.x & 63;
  unsigned h = (unsigned)(blockIdx.x * 512 + threadIdx.x) * 2654435761u;
  half8 a, b;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13;
    a[j] = (half_t)((float)(h & 0xffff) * (2.f / 65536.f) - 1.f);
    b[j] = (half_t)((float)(h >> 16) * (2.f / 65536.f) - 1.f);
  }
  floatx16 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  for (int t = 0; t < iters; ++t) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = mfma32(a, b, acc[i]);
  }
  float s_ = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) s_ += acc[i][0] + acc[i][15];
  if (s_ == 12345.678f) sink[blockIdx.x * 64 + lane] = s_;      // never true in practice: keeps the accumulators live
}

int pnpi_clock_probe(pnpi_ctx* c, int iters, float* ghz_out, float* ms_out) {
  if (!c || iters <= 0 || iters > (1 << 24) || !ghz_out) return PNPI_EINVAL;
  int dev = 0, cus = 0;
  CKH(hipGetDevice(&dev));
  CKH(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  hipEvent_t e0, e1;
  CKH(hipEventCreate(&e0));
  CKH(hipEventCreate(&e1));
  CKH(hipEventRecord(e0, c->st));
  hipLaunchKernelGGL(mfma_clock_kernel, dim3(cus), dim3(512), 0, c->st, iters, (float*)c->gn_partial);
  CKH(hipGetLastError());
  CKH(hipEventRecord(e1, c->st));
  CKH(hipEventSynchronize(e1));
  float ms = 0.f;
  CKH(hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  // one workgroup of 8 waves per CU = 2 waves per SIMD, 16 MFMAs per wave and iteration, 32 pipe cycles each
  *ghz_out = ms > 0.f ? (float)(2.0 * 16.0 * 32.0 * (double)iters / ((double)ms * 1e6)) : 0.f;
  if (ms_out) *ms_out = ms;
  return 0;
}

int pnpi_profile_begin(pnpi_ctx* c) {
  if (!c) return PNPI_EINVAL;
  CKH(hipStreamSynchronize(c->st));
  c->prof.clear();
  c->prof_on = true;
  return 0;
}
int pnpi_profile_end(pnpi_ctx* c, pnpi_kernel_stats* out) {
  if (!c || !out) return PNPI_EINVAL;
  c->prof_on = false;
  CKH(hipStreamSynchronize(c->st));
  if (const char* path = getenv("PNPI_PROFILE_DUMP")) {   // per-launch records for tools/ (class, M, N, K, ksize, us)
    if (FILE* f = fopen(path, "w")) {
      fprintf(f, "cls,M,N,K,ksize,us,flops,cfg,split,kernel,bytes\n");
      for (ProfRec& r : c->prof) {
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, r.a, r.b);
        char kn[96] = "-";     // the kernel template a rocprofv3 kernel trace shows for this launch
        if (r.geom[0] && r.geom[6] == -1) snprintf(kn, sizeof kn, "igemm_pp_kernel<%d %d %d %d %d>", r.geom[0], r.geom[1], r.geom[2], r.geom[3], r.geom[4]);
        else if (r.geom[0]) snprintf(kn, sizeof kn, "igemm_dma_kernel<%d %d %d %d %d %d %d>", r.geom[0], r.geom[1], r.geom[2], r.geom[3], r.geom[4], r.geom[5], r.geom[6]);
        else if (r.cfg >= 0) snprintf(kn, sizeof kn, "igemm_kernel");
        fprintf(f, "%d,%d,%d,%d,%d,%.3f,%.0f,%d,%d,%s,%.0f\n", r.cls, r.M, r.N, r.K, r.ksize, ms * 1e3, r.flops, r.cfg, r.split, kn, r.bytes);
      }
      fclose(f);
    }
  }
  memset(out, 0, sizeof(pnpi_kernel_stats) * PNPI_KC_COUNT);
  for (ProfRec& r : c->prof) {
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, r.a, r.b);
    if (r.cls >= 0 && r.cls < PNPI_KC_COUNT) {
      out[r.cls].launches += 1; out[r.cls].total_ms += ms; out[r.cls].flops += r.flops; out[r.cls].bytes += r.bytes;
    }
    (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b);
  }
  c->prof.clear();
  return 0;
}

static bool is_clip_slot(const std::string& name) { return name.compare(0, 5, "clip.") == 0; }
static int check_ready(pnpi_ctx* c) {   // UNet / VAE entry points: the text encoder's weights are optional for them
  for (auto& kv : c->slots)
    if (!kv.second.loaded && !is_clip_slot(kv.first)) { c->err = "weights not loaded: " + kv.first; return PNPI_ESTATE; }
  return 0;
}
// Level-2 entry points (whole loops in one call): they drive unet_fwd directly with kernel descriptors, so a host attention callback
// left installed by an earlier level-1 forward would silently replace the descriptor edits (and run during inversion).  Fail loudly.
static int check_loop_ready(pnpi_ctx* c) {
  CKP(check_ready(c));
  if (c->attn_cb) return fail(c, PNPI_ESTATE, "an attention callback is installed (pnpi_set_attention_callback): the loop entry points take kernel "
                                               "descriptors only -- remove the callback first (callback controllers run through pnpi_unet_forward)");
  return 0;
}
static int check_clip_ready(pnpi_ctx* c) {
  if (c->clip.layers.empty()) return fail(c, PNPI_ESTATE, "this context was built without a text encoder (clip_layers = 0)");
  for (auto& kv : c->slots)
    if (!kv.second.loaded && is_clip_slot(kv.first)) { c->err = "weights not loaded: " + kv.first; return PNPI_ESTATE; }
  return 0;
}

int pnpi_set_attention_callback(pnpi_ctx* c, pnpi_attn_callback cb, void* user, float* attn_buf, size_t attn_buf_bytes) {
  if (!c || (cb && (!attn_buf || !attn_buf_bytes))) return PNPI_EINVAL;
  c->attn_cb = cb; c->attn_cb_user = user; c->attn_buf = cb ? attn_buf : nullptr; c->attn_buf_bytes = cb ? attn_buf_bytes : 0;
  return 0;
}

int pnpi_text_kv_precompute(pnpi_ctx* c, const float* context, int rows) {
  if (!c || !context) return PNPI_EINVAL;
  CKP(check_ready(c));
  return text_kv_precompute(c, context, rows);
}

int pnpi_unet_forward(pnpi_ctx* c, const float* latents, int rows, int rows_per_image, int t, const float* context,
                      const pnpi_ctrl_desc* ctrl_host, int cur_step, float* eps_out) {
  if (!c || !latents || !eps_out) return PNPI_EINVAL;
  CKP(check_ready(c));
  if (!context && c->tkv.rows != rows) return fail(c, PNPI_ESTATE, "context is NULL: call pnpi_text_kv_precompute for this row count first");
  if (c->attn_cb && (!context || ctrl_host)) return fail(c, PNPI_EINVAL, "attention callback: pass the context, and no controller descriptor");
  struct UseKV { pnpi_ctx* c; ~UseKV() { c->tkv.use = false; } } guard{c};
  c->tkv.use = context == nullptr;
  bool use_ctrl = false;
  if (ctrl_host) {
    if (rows_per_image != 4 || rows % 4) return fail(c, PNPI_EINVAL, "controllers need rows_per_image == 4");
    if (cur_step == 0 || c->cd.nimg != rows / 4) CKP(setup_ctrl(c, ctrl_host, rows / 4, rows));
    use_ctrl = true;
  } else if (c->cd.any_edit || c->cd.nimg != 0) {
    CKP(setup_ctrl(c, nullptr, 0, c->max_rows));
  }
  return unet_fwd(c, latents, rows, t, context, use_ctrl, cur_step, eps_out);
}

/* LocalBlend step_callback for level-1 drivers (attention_control.py:253-256): latents [nimg][2][4][h][w] in place */
int pnpi_local_blend(pnpi_ctx* c, float* latents, int nimg, int step_index) {
  if (!c || !latents || nimg != c->cd.nimg) return PNPI_EINVAL;
  return apply_local_blend(c, latents, step_index);
}

int pnpi_vae_encode(pnpi_ctx* c, const float* x, int n, int height, int width, float* mean_out) {
  if (!c || !x || !mean_out || n <= 0 || n > c->max_vae) return PNPI_EINVAL;
  CKP(check_ready(c));
  c->persist.reset(); c->temp.reset();
  half_t* xin = palloc(c, (size_t)n * height * width * 8);
  CK(launch_nchw_f32_to_nhwc_f16(x, n, 3, height * width, 8, xin, c->st));
  CKP(vae_encode_fwd(c, xin, n, height, width, mean_out));
  c->ctr.vae_encodes += n;
  if (c->persist.overflow || c->temp.overflow) return fail(c, PNPI_ENOMEM, "workspace overflow (vae encode)");
  return 0;
}

int pnpi_image2latent(pnpi_ctx* c, const uint8_t* img, int n, int height, int width, float* z_out) {
  if (!c || !img || !z_out || n <= 0 || n > c->max_vae) return PNPI_EINVAL;
  CKP(check_ready(c));
  c->persist.reset(); c->temp.reset();
  half_t* xin = palloc(c, (size_t)n * height * width * 8);
  CK(launch_img_u8_to_nhwc(img, n, height * width, 8, xin, c->st));
  CKP(vae_encode_fwd(c, xin, n, height, width, z_out));
  const int F = 1 << (c->cfg.vae_n_blocks - 1);
  size_t ne = (size_t)n * c->cfg.vae_latent_channels * (height / F) * (width / F);
  CK(launch_scale_f32(z_out, ne, 0.18215f, z_out, c->st));
  c->ctr.vae_encodes += n;
  if (c->persist.overflow || c->temp.overflow) return fail(c, PNPI_ENOMEM, "workspace overflow (vae encode)");
  return 0;
}

static int decode_common(pnpi_ctx* c, const float* z, int n, int lh, int lw, float zscale, float* sample_out, uint8_t* u8_out) {
  CKP(check_ready(c));
  c->persist.reset(); c->temp.reset();
  const int L = c->cfg.vae_latent_channels, F = 1 << (c->cfg.vae_n_blocks - 1);
  const float* zin = z;
  if (zscale != 1.0f) {
    float* zs = (float*)c->persist.alloc((size_t)n * L * lh * lw * sizeof(float));
    CK(launch_scale_f32(z, (size_t)n * L * lh * lw, zscale, zs, c->st));
    zin = zs;
  }
  half_t* z16 = palloc(c, (size_t)n * lh * lw * 8);
  CK(launch_nchw_f32_to_nhwc_f16(zin, n, L, lh * lw, 8, z16, c->st));
  float* dst = sample_out;
  if (!dst) dst = (float*)c->persist.alloc((size_t)n * 3 * lh * F * lw * F * sizeof(float));
  CKP(vae_decode_fwd(c, z16, n, lh, lw, dst));
  if (u8_out) CK(launch_dec_to_u8(dst, n, lh * F * lw * F, u8_out, c->st));
  c->ctr.vae_decodes += n;
  if (c->persist.overflow || c->temp.overflow) return fail(c, PNPI_ENOMEM, "workspace overflow (vae decode)");
  return 0;
}

int pnpi_vae_decode(pnpi_ctx* c, const float* z, int n, int lh, int lw, float* sample_out) {
  if (!c || !z || !sample_out || n <= 0 || n > c->max_vae) return PNPI_EINVAL;
  return decode_common(c, z, n, lh, lw, 1.0f, sample_out, nullptr);
}
int pnpi_latent2image(pnpi_ctx* c, const float* z, int n, int lh, int lw, uint8_t* img_out) {
  if (!c || !z || !img_out || n <= 0 || n > c->max_vae) return PNPI_EINVAL;
  // utils/utils.py:60: latents = 1 / 0.18215 * latents  (python float 1/0.18215 rounded to fp32 by the tensor multiply)
  return decode_common(c, z, n, lh, lw, (float)(1.0 / 0.18215), nullptr, img_out);
}

static int alphas_for(pnpi_ctx* c, int t, int ratio, bool forward, float* a_from, float* a_to) {
  if (!c->sched_set) return fail(c, PNPI_ESTATE, "pnpi_set_scheduler not called");
  const int n = c->cfg.n_train_timesteps;
  if (t < 0 || t >= n) return fail(c, PNPI_EINVAL, "timestep out of range");
  if (forward) {  // next_step: (min(t - ratio, 999)) -> t
    int tp = t - ratio; if (tp > n - 1) tp = n - 1;
    *a_from = tp >= 0 ? c->ac[tp] : c->final_alpha;
    *a_to = c->ac[t];
  } else {        // prev_step: t -> t - ratio
    int tp = t - ratio;
    *a_from = c->ac[t];
    *a_to = tp >= 0 ? c->ac[tp] : c->final_alpha;
  }
  return 0;
}

int pnpi_ddim_next_step(pnpi_ctx* c, const float* eps, int t, int ratio, const float* sample, size_t n, float* out) {
  float af, at; CKP(alphas_for(c, t, ratio, true, &af, &at));
  CK(launch_ddim_move(sample, eps, af, at, n, out, c->st));
  return 0;
}
int pnpi_ddim_prev_step(pnpi_ctx* c, const float* eps, int t, int ratio, const float* sample, size_t n, float* out) {
  float af, at; CKP(alphas_for(c, t, ratio, false, &af, &at));
  CK(launch_ddim_move(sample, eps, af, at, n, out, c->st));
  return 0;
}
// DDIMSchedulerDev.step(model_output, t, sample, ref_image=, recon_lr=, recon_mask=) (scheduler_dev.py:38-95 with :68-76): one launch.
// ref / mask (nullable) are full-shape like the sample; pred_x0_out nullable.
int pnpi_ddim_prev_step_recon(pnpi_ctx* c, const float* eps, int t, int ratio, const float* sample, size_t n, const float* ref_image,
                              float recon_lr, const float* recon_mask, float* out, float* pred_x0_out) {
  if (!c || !eps || !sample || !out) return PNPI_EINVAL;
  float af, at; CKP(alphas_for(c, t, ratio, false, &af, &at));
  const bool on = ref_image && recon_lr > 0.f;
  CK(launch_ddim_prev_recon(sample, eps, af, at, on ? ref_image : nullptr, recon_lr, on ? recon_mask : nullptr, n, out, pred_x0_out, c->st));
  return 0;
}
// the recon_t window of proximal_guidance_forward.py:48,60 (and :73 for the inversion pull).  Inside it the pred-x0 pull towards ref_image
// runs for recon_lr > 0 (scheduler_dev.py:68), the pull of the step's result towards x*_{t-1} for any recon_lr != 0 (:75 has no such test)
static bool recon_window(const pnpi_recon_desc* rc, int t) {
  return rc && ((rc->recon_t > 0 && t < rc->recon_t) || (rc->recon_t < 0 && t > -rc->recon_t));
}
static const float* recon_ref_at(const pnpi_recon_desc* rc, int t) { return (recon_window(rc, t) && rc->recon_lr > 0.f) ? rc->ref_image : nullptr; }
static bool recon_inv_at(const pnpi_recon_desc* rc, int t) { return recon_window(rc, t) && rc->inv_x_stars && rc->recon_lr != 0.f; }
static int recon_check(pnpi_ctx* c, const pnpi_recon_desc* rc) {
  if (rc && rc->struct_size != (uint32_t)sizeof(pnpi_recon_desc)) return fail(c, PNPI_EINVAL, "pnpi_recon_desc.struct_size is not sizeof(pnpi_recon_desc) of this library");
  return 0;
}

int pnpi_cfg_ddim_prev(pnpi_ctx* c, const float* eps, const float* x, int nimg, int rpi, size_t row_elems, float gs, int t, int ratio,
                       const float* noise_loss, int offset_rows, const float* target, float offset_scale, float* offset_out,
                       float* x_out, const float* prox_threshold, int prox, const pnpi_recon_desc* recon) {
  if (prox < 0 || prox > 2 || (prox && !prox_threshold)) return fail(c, PNPI_EINVAL, "prox must be 0, or 1 / 2 with a threshold");
  float af, at; CKP(alphas_for(c, t, ratio, false, &af, &at));
  CKP(recon_check(c, recon));
  const float* rref = prox ? recon_ref_at(recon, t) : nullptr;
  const float* rinv = (prox && recon_inv_at(recon, t)) ? recon->inv_x_stars : nullptr;     // level 1: the caller points inv_x_stars at this step's x*_{t-1} [nimg][...]
  const bool rc = rref || rinv;
  const int S = c->cfg.sample_size;
  CK(launch_cfg_ddim_prev(eps, x, nimg, rpi, row_elems, gs, af, at, noise_loss, offset_rows, target, offset_scale, offset_out, x_out, c->st,
                          prox_threshold, prox, rref, rc ? recon->recon_lr : 0.f, rc ? recon->dilate_mask : 0, S, S, rinv));
  return 0;
}

int pnpi_prox_threshold(pnpi_ctx* c, const float* eps, int nimg, int rpi, size_t row_elems, float quantile, float* thr_out) {
  if (!c || !eps || !thr_out || nimg <= 0 || !(quantile > 0.f && quantile <= 1.f)) return PNPI_EINVAL;
  int r = launch_quantile_abs_diff(eps, nimg, rpi, row_elems, quantile, thr_out, c->st);
  if (r == -6) return fail(c, PNPI_ESHAPE, "proximal threshold: more than 32768 elements per image");
  CK(r);
  return 0;
}

// CLIPTextModel.forward -> last_hidden_state (transformers modeling_clip.py: CLIPTextEmbeddings, CLIPEncoderLayer x L with a
// causal mask, final_layer_norm); pre-LN blocks, quick_gelu MLP.  ids: device int32 [n][T]; out fp32 [n][T][H].
static int clip_fwd(pnpi_ctx* c, const int* ids, int n, float* out) {
  const ClipW& t = c->clip;
  const int H = t.H, T = t.T, M = n * T, dh = H / t.heads;
  c->persist.reset(); c->temp.reset();
  half_t* x = palloc(c, (size_t)M * H);
  if (!c->dry) CK(launch_embed_tokens(ids, M, T, H, t.vocab, t.tok, t.pos, x, c->st));
  const int ldv = round_up_i(T, 8);
  for (const ClipLayerW& L : t.layers) {
    const size_t mk = c->temp.mark();
    half_t* h1 = talloc(c, (size_t)M * H);
    if (!c->dry) PROF(PNPI_KC_LAYERNORM, 0.0, 2.0 * M * (double)H * 2.0, launch_layernorm(x, M, H, 1e-5f, L.ln1.g, L.ln1.b, h1, c->st));
    half_t* qk = talloc(c, (size_t)M * 2 * H);
    half_t* vt = talloc(c, (size_t)n * H * ldv);
    {
      VtOut v; v.outT = vt; v.col0 = 2 * H; v.ld = ldv; v.f32 = 0; v.rpb = T;
      CK(op_gemm(c, h1, H, M, H, L.w_qkv, H, 3 * H, L.b_qkv, nullptr, 0, qk, 2 * H, 1.f, &v));
    }
    half_t* ao = talloc(c, (size_t)M * H);
    {
      AttnP a; a.q = qk; a.ldq = 2 * H; a.q_off = 0; a.k = qk; a.ldk = 2 * H; a.k_off = H; a.vt = vt; a.ldv = ldv;
      a.o = ao; a.ldo = H; a.heads = t.heads; a.Nq = T; a.Nk = T; a.Dp = dh; a.dh = dh; a.scale = 1.0f / sqrtf((float)dh);
      a.rows = c->rows_ident; a.nrows = n; a.causal = 1;
      if (!c->dry) PROFD(PNPI_KC_ATTN_FLASH, 4.0 * n * t.heads * (double)T * T * dh, 0.0, T, T, dh, launch_attn_flash(a, c->st));
    }
    half_t* x1 = talloc(c, (size_t)M * H);
    CK(op_gemm(c, ao, H, M, H, L.out.w, H, H, L.out.b, x, H, x1, H));
    half_t* h2 = talloc(c, (size_t)M * H);
    if (!c->dry) PROF(PNPI_KC_LAYERNORM, 0.0, 2.0 * M * (double)H * 2.0, launch_layernorm(x1, M, H, 1e-5f, L.ln2.g, L.ln2.b, h2, c->st));
    half_t* f = talloc(c, (size_t)M * t.I);
    CK(op_gemm(c, h2, H, M, H, L.fc1.w, H, t.I, L.fc1.b, nullptr, 0, f, t.I));
    if (!c->dry) CK(launch_quick_gelu(f, (size_t)M * t.I, c->st));
    CK(op_gemm(c, f, t.I, M, t.I, L.fc2.w, t.I, H, L.fc2.b, x1, H, x, H));    // x <- x1 + mlp (x's old value is dead)
    c->temp.release(mk);
  }
  half_t* y = palloc(c, (size_t)M * H);
  if (!c->dry) {
    PROF(PNPI_KC_LAYERNORM, 0.0, 2.0 * M * (double)H * 2.0, launch_layernorm(x, M, H, 1e-5f, t.final_ln.g, t.final_ln.b, y, c->st));
    CK(launch_f16_to_f32(y, (size_t)M * H, out, c->st));
  }
  return 0;
}

int pnpi_text_encode(pnpi_ctx* c, const int32_t* input_ids, int n, float* hidden_out) {
  if (!c || !input_ids || !hidden_out || n <= 0) return PNPI_EINVAL;
  CKP(check_clip_ready(c));
  if (n > c->rows_ident_n) return fail(c, PNPI_EINVAL, "more prompts than max(max_unet_rows, 8)");
  return clip_fwd(c, (const int*)input_ids, n, hidden_out);
}

// ---- level 2 loops. Scratch for the loops lives at the top of the controller arena (after the controller tables).
// The loops' text context is constant over their steps: project K / V once, then every forward of the loop reads the cache.
struct LoopKV {
  pnpi_ctx* c;
  bool armed = false;
  explicit LoopKV(pnpi_ctx* c_) : c(c_) {}
  int begin(const float* context, int rows) {
    if (!g_text_kv) return 0;
    int r = text_kv_precompute(c, context, rows);
    if (r) return r;
    c->tkv.use = true;
    armed = true;
    return 0;
  }
  // a loop's projections belong to the loop's context: dropped at its end, so that a later pnpi_unet_forward(context = NULL) can
  // never silently read them (it fails and names pnpi_text_kv_precompute instead)
  ~LoopKV() { if (armed) { c->tkv.use = false; c->tkv.rows = 0; } }
};
int pnpi_ddim_invert(pnpi_ctx* c, const float* z0, int nimg, const float* ctx_cond, int nsteps, const int* ts, float* all) {
  if (!c || !z0 || !ctx_cond || !ts || !all || nsteps <= 0) return PNPI_EINVAL;
  CKP(check_loop_ready(c));
  const pnpi_model_config& g = c->cfg;
  const size_t E = (size_t)g.in_channels * g.sample_size * g.sample_size;
  const int ratio = g.n_train_timesteps / nsteps;
  CKP(setup_ctrl(c, nullptr, 0, c->max_rows));
  float* eps = misc_f(c, (size_t)nimg * E);
  CKH(hipMemcpyAsync(all, z0, (size_t)nimg * E * sizeof(float), hipMemcpyDeviceToDevice, c->st));
  LoopKV kv(c);
  CKP(kv.begin(ctx_cond, nimg));
  for (int i = 0; i < nsteps; ++i) {
    const int t = ts[nsteps - i - 1];
    const float* cur = all + (size_t)i * nimg * E;
    int r = unet_fwd(c, cur, nimg, t, ctx_cond, false, 0, eps);
    if (r) return r;
    float af, at; CKP(alphas_for(c, t, ratio, true, &af, &at));
    CK(launch_ddim_move(cur, eps, af, at, (size_t)nimg * E, all + (size_t)(i + 1) * nimg * E, c->st));
  }
  return 0;
}

static int upload_ints(pnpi_ctx* c, const std::vector<int>& v, int** dst);

/* DirectInversion.ddim_with_guidance_scale_loop (inversion.py:334-347): inversion under classifier-free guidance.  The reference
 * makes two B=1 UNet calls per step (uncond, cond); here they are the two rows of one launch. */
int pnpi_ddim_invert_cfg(pnpi_ctx* c, const float* z0, int nimg, const float* ctx_uncond, const float* ctx_cond, float gs, int nsteps,
                         const int* ts, float* all) {
  if (!c || !z0 || !ctx_uncond || !ctx_cond || !ts || !all || nsteps <= 0) return PNPI_EINVAL;
  CKP(check_loop_ready(c));
  const pnpi_model_config& g = c->cfg;
  const size_t E = (size_t)g.in_channels * g.sample_size * g.sample_size, CE = (size_t)g.ctx_len * g.cross_dim;
  const int ratio = g.n_train_timesteps / nsteps, rows = 2 * nimg;
  if (rows > c->max_rows) return fail(c, PNPI_EINVAL, "nimg * 2 exceeds max_unet_rows");
  CKP(setup_ctrl(c, nullptr, 0, c->max_rows));
  float* eps = misc_f(c, (size_t)rows * E);
  float* in = misc_f(c, (size_t)rows * E);
  float* ctx2 = misc_f(c, (size_t)rows * CE);
  std::vector<int> inmap(rows);
  for (int i = 0; i < nimg; ++i) { inmap[2 * i] = i; inmap[2 * i + 1] = i; }
  int* d_inmap;
  CKP(upload_ints(c, inmap, &d_inmap));
  for (int i = 0; i < nimg; ++i) {   // rows [img][uncond, cond]
    CKH(hipMemcpyAsync(ctx2 + (size_t)(2 * i) * CE, ctx_uncond + (size_t)i * CE, CE * sizeof(float), hipMemcpyDeviceToDevice, c->st));
    CKH(hipMemcpyAsync(ctx2 + (size_t)(2 * i + 1) * CE, ctx_cond + (size_t)i * CE, CE * sizeof(float), hipMemcpyDeviceToDevice, c->st));
  }
  CKH(hipMemcpyAsync(all, z0, (size_t)nimg * E * sizeof(float), hipMemcpyDeviceToDevice, c->st));
  LoopKV kv(c);
  CKP(kv.begin(ctx2, rows));
  for (int i = 0; i < nsteps; ++i) {
    const int t = ts[nsteps - i - 1];
    const float* cur = all + (size_t)i * nimg * E;
    CK(launch_gather_rows_f32(cur, d_inmap, rows, E, in, c->st));
    int r = unet_fwd(c, in, rows, t, ctx2, false, 0, eps);
    if (r) return r;
    float af, at; CKP(alphas_for(c, t, ratio, true, &af, &at));
    // noise = eps_u + gs * (eps_c - eps_u); next_step (the same fused kernel as the denoising direction, other alphas)
    CK(launch_cfg_ddim_prev(eps, cur, nimg, 1, E, gs, af, at, nullptr, 0, nullptr, 1.f, nullptr, all + (size_t)(i + 1) * nimg * E, c->st));
  }
  if (c->ctrl_arena.overflow) return fail(c, PNPI_ENOMEM, "loop arena overflow");
  return 0;
}

static int upload_ints(pnpi_ctx* c, const std::vector<int>& v, int** dst) {
  *dst = (int*)c->ctrl_arena.alloc(v.size() * sizeof(int));
  return upload(c, *dst, v.data(), v.size() * sizeof(int));
}

int pnpi_offset_calculate(pnpi_ctx* c, const float* lat_all, int nimg, const float* context4, int nsteps, const int* ts, float gs,
                          const float* offset_scale_host, float* noise_loss_out) {
  if (!c || !lat_all || !context4 || !ts || !noise_loss_out || nsteps <= 0) return PNPI_EINVAL;
  CKP(check_loop_ready(c));
  const pnpi_model_config& g = c->cfg;
  const size_t E = (size_t)g.in_channels * g.sample_size * g.sample_size;
  const int ratio = g.n_train_timesteps / nsteps, rows = 4 * nimg;
  if (rows > c->max_rows) return fail(c, PNPI_EINVAL, "nimg * 4 exceeds max_unet_rows");
  CKP(setup_ctrl(c, nullptr, 0, c->max_rows));
  float* cur = misc_f(c, (size_t)nimg * 2 * E);
  float* in = misc_f(c, (size_t)rows * E);
  float* eps = misc_f(c, (size_t)rows * E);
  std::vector<int> expand(nimg * 2), inmap(rows);
  for (int i = 0; i < nimg; ++i) { expand[2 * i] = i; expand[2 * i + 1] = i; for (int k = 0; k < 4; ++k) inmap[4 * i + k] = 2 * i + (k & 1); }
  int *d_expand, *d_inmap;
  CKP(upload_ints(c, expand, &d_expand));
  CKP(upload_ints(c, inmap, &d_inmap));
  CK(launch_gather_rows_f32(lat_all + (size_t)nsteps * nimg * E, d_expand, nimg * 2, E, cur, c->st));
  LoopKV kv(c);
  CKP(kv.begin(context4, rows));
  for (int i = 0; i < nsteps; ++i) {
    const int t = ts[i];
    CK(launch_gather_rows_f32(cur, d_inmap, rows, E, in, c->st));
    int r = unet_fwd(c, in, rows, t, context4, false, 0, eps);
    if (r) return r;
    float af, at; CKP(alphas_for(c, t, ratio, false, &af, &at));
    const float* target = lat_all + (size_t)(nsteps - i - 1) * nimg * E;
    CK(launch_cfg_ddim_prev(eps, cur, nimg, 2, E, gs, af, at, nullptr, 0, target, offset_scale_host ? offset_scale_host[i] : 1.f,
                            noise_loss_out + (size_t)i * nimg * 2 * E, cur, c->st));
  }
  return 0;
}

// uncond_steps (nullable): [nsteps][nimg][77][768] per-step unconditional embeddings (null-text inversion).  p2p_guidance_forward uses the
// step's embedding for every unconditional row of the image (p2p_guidance_forward.py:56-57); uncond_first_only = the single-branch variant
// (:92: the first row only).  The text K / V are then projected once per STEP instead of once per loop.
static int edit_loop_impl(pnpi_ctx* c, const float* x_T, int nimg, const float* context4, const float* noise_loss, int offset_rows,
                          const pnpi_ctrl_desc* ctrl_host, int nsteps, const int* ts, float gs, int prox, float quantile,
                          const pnpi_recon_desc* recon, float* latents_out, const float* uncond_steps, int uncond_first_only) {
  if (!c || !x_T || !context4 || !ts || !latents_out || nsteps <= 0) return PNPI_EINVAL;
  CKP(recon_check(c, recon));
  CKP(check_loop_ready(c));
  const pnpi_model_config& g = c->cfg;
  const size_t E = (size_t)g.in_channels * g.sample_size * g.sample_size;
  const int ratio = g.n_train_timesteps / nsteps, rows = 4 * nimg;
  if (rows > c->max_rows) return fail(c, PNPI_EINVAL, "nimg * 4 exceeds max_unet_rows");
  if (ctrl_host) CKP(setup_ctrl(c, ctrl_host, nimg, rows));
  else CKP(setup_ctrl(c, nullptr, 0, c->max_rows));
  const bool use_ctrl = ctrl_host != nullptr;
  if (prox < 0 || prox > 2) return fail(c, PNPI_EINVAL, "prox must be 0 (none), 1 (l0) or 2 (l1)");
  float* lat = misc_f(c, (size_t)nimg * 2 * E);
  float* in = misc_f(c, (size_t)rows * E);
  float* eps = misc_f(c, (size_t)rows * E);
  float* thr = misc_f(c, (size_t)nimg);
  std::vector<int> expand(nimg * 2), inmap(rows);
  for (int i = 0; i < nimg; ++i) { expand[2 * i] = i; expand[2 * i + 1] = i; for (int k = 0; k < 4; ++k) inmap[4 * i + k] = 2 * i + (k & 1); }
  int *d_expand, *d_inmap;
  CKP(upload_ints(c, expand, &d_expand));
  CKP(upload_ints(c, inmap, &d_inmap));
  CK(launch_gather_rows_f32(x_T, d_expand, nimg * 2, E, lat, c->st));
  if (prox && !(quantile > 0.f)) CK(launch_fill_f32(thr, nimg, -quantile, c->st));   // negative quantile = fixed threshold (:43-44)
  const size_t CE = (size_t)g.ctx_len * g.cross_dim;
  float* ctx_step = nullptr;
  if (uncond_steps) {
    ctx_step = misc_f(c, (size_t)rows * CE);
    CKH(hipMemcpyAsync(ctx_step, context4, (size_t)rows * CE * sizeof(float), hipMemcpyDeviceToDevice, c->st));
  }
  LoopKV kv(c);
  if (!uncond_steps) CKP(kv.begin(context4, rows));
  for (int i = 0; i < nsteps; ++i) {
    const int t = ts[i];
    const float* ctx_i = context4;
    if (uncond_steps) {
      for (int im = 0; im < nimg; ++im) {
        const float* u = uncond_steps + ((size_t)i * nimg + im) * CE;
        CKH(hipMemcpyAsync(ctx_step + (size_t)(4 * im) * CE, u, CE * sizeof(float), hipMemcpyDeviceToDevice, c->st));
        if (!uncond_first_only) CKH(hipMemcpyAsync(ctx_step + (size_t)(4 * im + 1) * CE, u, CE * sizeof(float), hipMemcpyDeviceToDevice, c->st));
      }
      ctx_i = ctx_step;
      CKP(kv.begin(ctx_i, rows));
    }
    CK(launch_gather_rows_f32(lat, d_inmap, rows, E, in, c->st));
    int r = unet_fwd(c, in, rows, t, ctx_i, use_ctrl, i, eps);
    if (r) return r;
    float af, at; CKP(alphas_for(c, t, ratio, false, &af, &at));
    const float* nl = noise_loss ? noise_loss + (size_t)i * nimg * 2 * E : nullptr;
    if (prox && quantile > 0.f) CK(launch_quantile_abs_diff(eps, nimg, 2, E, quantile, thr, c->st));
    const float* rref = prox ? recon_ref_at(recon, t) : nullptr;
    // inversion guidance: x_stars[len(x_stars) - i - 2] (proximal_guidance_forward.py:75), one latent per image for both of its rows
    const float* inv = (prox && recon_inv_at(recon, t)) ? recon->inv_x_stars + (size_t)(nsteps - 1 - i) * nimg * E : nullptr;
    const bool rc = rref || inv;
    CK(launch_cfg_ddim_prev(eps, lat, nimg, 2, E, gs, af, at, nl, offset_rows, nullptr, 1.f, nullptr, lat, c->st, prox ? thr : nullptr, prox,
                            rref, rc ? recon->recon_lr : 0.f, rc ? recon->dilate_mask : 0, g.sample_size, g.sample_size, inv));
    if (use_ctrl) CKP(apply_local_blend(c, lat, i));
  }
  CKH(hipMemcpyAsync(latents_out, lat, (size_t)nimg * 2 * E * sizeof(float), hipMemcpyDeviceToDevice, c->st));
  if (c->ctrl_arena.overflow) return fail(c, PNPI_ENOMEM, "loop arena overflow");
  return 0;
}
int pnpi_edit_loop(pnpi_ctx* c, const float* x_T, int nimg, const float* context4, const float* noise_loss, int offset_rows,
                   const pnpi_ctrl_desc* ctrl_host, int nsteps, const int* ts, float gs, int prox, float quantile,
                   const pnpi_recon_desc* recon, float* latents_out) {
  return edit_loop_impl(c, x_T, nimg, context4, noise_loss, offset_rows, ctrl_host, nsteps, ts, gs, prox, quantile, recon, latents_out, nullptr, 0);
}
int pnpi_edit_loop_uncond_steps(pnpi_ctx* c, const float* x_T, int nimg, const float* context4, const pnpi_ctrl_desc* ctrl_host, int nsteps,
                                const int* ts, float gs, int prox, float quantile, const float* uncond_steps, int uncond_first_only,
                                float* latents_out) {
  if (!uncond_steps) return PNPI_EINVAL;
  return edit_loop_impl(c, x_T, nimg, context4, nullptr, 1, ctrl_host, nsteps, ts, gs, prox, quantile, nullptr, latents_out, uncond_steps,
                        uncond_first_only);
}

// the same with reconstruction guidance (null-text-inversion+proximal-guidance, use_reconstruction_guidance=True: p2p_editor.py:620-627)
int pnpi_edit_loop_uncond_steps_recon(pnpi_ctx* c, const float* x_T, int nimg, const float* context4, const pnpi_ctrl_desc* ctrl_host, int nsteps,
                                      const int* ts, float gs, int prox, float quantile, const float* uncond_steps, int uncond_first_only,
                                      const pnpi_recon_desc* recon, float* latents_out) {
  if (!uncond_steps) return PNPI_EINVAL;
  return edit_loop_impl(c, x_T, nimg, context4, nullptr, 1, ctrl_host, nsteps, ts, gs, prox, quantile, recon, latents_out, uncond_steps,
                        uncond_first_only);
}

/* offset_calculate + npass guidance-forward passes of P2PEditor.edit_image_directinversion (p2p_editor.py:99-160) advanced in
 * lock step: every pass walks the same timesteps and pass p's step i needs only noise_loss[i], which the offset pass produces
 * at the same step -- so one UNet launch per step serves all (1 + npass) * 4 * nimg rows. */
int pnpi_direct_edit(pnpi_ctx* c, const float* lat_all, int nimg, const float* context4, int npass, const pnpi_ctrl_desc* ctrl_host,
                     int offset_rows, int nsteps, const int* ts, float gs, const float* offset_scale_host, float* noise_loss_out,
                     float* latents_out) {
  if (!c || !lat_all || !context4 || !ts || !noise_loss_out || !latents_out || nsteps <= 0 || npass <= 0 || nimg <= 0) return PNPI_EINVAL;
  CKP(check_loop_ready(c));
  const pnpi_model_config& g = c->cfg;
  const size_t E = (size_t)g.in_channels * g.sample_size * g.sample_size;
  const size_t CE = (size_t)g.ctx_len * g.cross_dim;
  const int ratio = g.n_train_timesteps / nsteps, NI = (1 + npass) * nimg, rows = 4 * NI;
  if (rows > c->max_rows) return fail(c, PNPI_EINVAL, "(1 + npass) * nimg * 4 exceeds max_unet_rows");
  std::vector<pnpi_ctrl_desc> cds(NI);
  memset(cds.data(), 0, cds.size() * sizeof(pnpi_ctrl_desc));      // the offset pass (pseudo-images 0..nimg-1) runs no controller
  if (ctrl_host) for (int i = 0; i < npass * nimg; ++i) cds[nimg + i] = ctrl_host[i];
  CKP(setup_ctrl(c, cds.data(), NI, rows));
  float* lat = misc_f(c, (size_t)NI * 2 * E);
  float* in = misc_f(c, (size_t)rows * E);
  float* eps = misc_f(c, (size_t)rows * E);
  float* ctxrep = misc_f(c, (size_t)rows * CE);
  std::vector<int> expand(NI * 2), inmap(rows), ctxmap(rows);
  for (int q = 0; q < NI; ++q) {
    const int img = q % nimg;
    expand[2 * q] = img; expand[2 * q + 1] = img;
    for (int k = 0; k < 4; ++k) { inmap[4 * q + k] = 2 * q + (k & 1); ctxmap[4 * q + k] = 4 * img + k; }
  }
  int *d_expand, *d_inmap, *d_ctxmap;
  CKP(upload_ints(c, expand, &d_expand));
  CKP(upload_ints(c, inmap, &d_inmap));
  CKP(upload_ints(c, ctxmap, &d_ctxmap));
  if (c->ctrl_arena.overflow) return fail(c, PNPI_ENOMEM, "loop arena overflow");
  CK(launch_gather_rows_f32(lat_all + (size_t)nsteps * nimg * E, d_expand, NI * 2, E, lat, c->st));
  CK(launch_gather_rows_f32(context4, d_ctxmap, rows, CE, ctxrep, c->st));
  LoopKV kv(c);
  CKP(kv.begin(ctxrep, rows));
  for (int i = 0; i < nsteps; ++i) {
    const int t = ts[i];
    CK(launch_gather_rows_f32(lat, d_inmap, rows, E, in, c->st));
    int r = unet_fwd(c, in, rows, t, ctxrep, true, i, eps);
    if (r) return r;
    float af, at; CKP(alphas_for(c, t, ratio, false, &af, &at));
    const float* target = lat_all + (size_t)(nsteps - i - 1) * nimg * E;
    float* nl = noise_loss_out + (size_t)i * nimg * 2 * E;
    CK(launch_cfg_ddim_prev(eps, lat, nimg, 2, E, gs, af, at, nullptr, 0, target, offset_scale_host ? offset_scale_host[i] : 1.f, nl, lat, c->st));
    for (int p = 1; p <= npass; ++p) {
      float* lp = lat + (size_t)p * nimg * 2 * E;
      CK(launch_cfg_ddim_prev(eps + (size_t)p * nimg * 4 * E, lp, nimg, 2, E, gs, af, at, nl, offset_rows, nullptr, 1.f, nullptr, lp, c->st));
    }
    CKP(apply_local_blend(c, lat, i));
  }
  CKH(hipMemcpyAsync(latents_out, lat + (size_t)nimg * 2 * E, (size_t)npass * nimg * 2 * E * sizeof(float), hipMemcpyDeviceToDevice, c->st));
  return 0;
}

/* The pruned-equivalent schedule of SURVEY.md Note D (algebra, not approximation): in direct-inversion mode the source latent after
 * every step is prev + (x*_{t-1} - prev) == x*_{t-1}, and no controller ever touches the unconditional rows or the conditional
 * source row's output.  So the offset pass and the reconstruction pass are redundant, the source latent can be ASSIGNED from the
 * stored trajectory, and the unconditional-source row is dead: one 3-row launch per step and image
 * [uncond_tgt, cond_src (attention maps only), cond_tgt] instead of 12.  200 sample-forwards per image instead of 650.
 * context4 rows as everywhere: [unc_src, unc_tgt, cond_src, cond_tgt] per image (row 0 is not used). */
int pnpi_direct_edit_pruned(pnpi_ctx* c, const float* lat_all, int nimg, const float* context4, const pnpi_ctrl_desc* ctrl_host,
                            int nsteps, const int* ts, float gs, float* latents_out) {
  if (!c || !lat_all || !context4 || !ts || !latents_out || nsteps <= 0 || nimg <= 0) return PNPI_EINVAL;
  CKP(check_loop_ready(c));
  const pnpi_model_config& g = c->cfg;
  const size_t E = (size_t)g.in_channels * g.sample_size * g.sample_size, CE = (size_t)g.ctx_len * g.cross_dim;
  const int ratio = g.n_train_timesteps / nsteps, rows = 3 * nimg;
  if (rows > c->max_rows) return fail(c, PNPI_EINVAL, "3 * nimg exceeds max_unet_rows");
  std::vector<pnpi_ctrl_desc> none(nimg);
  memset(none.data(), 0, none.size() * sizeof(pnpi_ctrl_desc));
  CKP(setup_ctrl(c, ctrl_host ? ctrl_host : none.data(), nimg, rows, 3, 1, 2));
  float* lat = misc_f(c, (size_t)nimg * 2 * E);      // [img][src, tgt]
  float* in = misc_f(c, (size_t)rows * E);
  float* eps = misc_f(c, (size_t)rows * E);
  float* eps2 = misc_f(c, (size_t)nimg * 2 * E);     // [img][unc_tgt, cond_tgt]
  float* xt = misc_f(c, (size_t)nimg * E);
  float* ctx3 = misc_f(c, (size_t)rows * CE);
  std::vector<int> expand(nimg * 2), inmap(rows), ctxmap(rows), epsmap(nimg * 2), tgtmap(nimg);
  for (int i = 0; i < nimg; ++i) {
    expand[2 * i] = i; expand[2 * i + 1] = i;
    inmap[3 * i] = 2 * i + 1; inmap[3 * i + 1] = 2 * i; inmap[3 * i + 2] = 2 * i + 1;
    ctxmap[3 * i] = 4 * i + 1; ctxmap[3 * i + 1] = 4 * i + 2; ctxmap[3 * i + 2] = 4 * i + 3;
    epsmap[2 * i] = 3 * i; epsmap[2 * i + 1] = 3 * i + 2;
    tgtmap[i] = 2 * i + 1;
  }
  int *d_expand, *d_inmap, *d_ctxmap, *d_epsmap, *d_tgtmap;
  CKP(upload_ints(c, expand, &d_expand)); CKP(upload_ints(c, inmap, &d_inmap)); CKP(upload_ints(c, ctxmap, &d_ctxmap));
  CKP(upload_ints(c, epsmap, &d_epsmap)); CKP(upload_ints(c, tgtmap, &d_tgtmap));
  if (c->ctrl_arena.overflow) return fail(c, PNPI_ENOMEM, "loop arena overflow");
  CK(launch_gather_rows_f32(lat_all + (size_t)nsteps * nimg * E, d_expand, nimg * 2, E, lat, c->st));      // both rows start from x*_T
  CK(launch_gather_rows_f32(context4, d_ctxmap, rows, CE, ctx3, c->st));
  LoopKV kv(c);
  CKP(kv.begin(ctx3, rows));
  for (int i = 0; i < nsteps; ++i) {
    const int t = ts[i];
    CK(launch_gather_rows_f32(lat, d_inmap, rows, E, in, c->st));
    int r = unet_fwd(c, in, rows, t, ctx3, true, i, eps);
    if (r) return r;
    float af, at; CKP(alphas_for(c, t, ratio, false, &af, &at));
    CK(launch_gather_rows_f32(eps, d_epsmap, nimg * 2, E, eps2, c->st));
    CK(launch_gather_rows_f32(lat, d_tgtmap, nimg, E, xt, c->st));
    CK(launch_cfg_ddim_prev(eps2, xt, nimg, 1, E, gs, af, at, nullptr, 0, nullptr, 1.f, nullptr, xt, c->st));
    // source latent := x*_{t-1} (assigned, not reconstructed); target latent := the step's result
    const float* target = lat_all + (size_t)(nsteps - i - 1) * nimg * E;
    CKH(hipMemcpy2DAsync(lat, 2 * E * sizeof(float), target, E * sizeof(float), E * sizeof(float), nimg, hipMemcpyDeviceToDevice, c->st));
    CKH(hipMemcpy2DAsync(lat + E, 2 * E * sizeof(float), xt, E * sizeof(float), E * sizeof(float), nimg, hipMemcpyDeviceToDevice, c->st));
    CKP(apply_local_blend(c, lat, i));
  }
  CKH(hipMemcpyAsync(latents_out, lat, (size_t)nimg * 2 * E * sizeof(float), hipMemcpyDeviceToDevice, c->st));
  return 0;
}

// ---------------------------------------------------------------------------------------------------- kernel-level ops
int pnpi_op_conv(pnpi_ctx* c, const void* x1, const void* x2, int C1, int C2, int B, int H, int W, int ksize, int stride, int pad,
                 int ups, int Ho, int Wo, const void* w, const float* bias, const void* res, int N, void* out, int force_cfg,
                 int force_split) {
  GemmP p; gemm_defaults(p);
  p.x1 = (const half_t*)x1; p.x2 = (const half_t*)x2; p.C1 = C1; p.C2 = C2; p.ldx1 = C1; p.ldx2 = C2;
  p.B = B; p.H = H; p.W = W; p.Ho = Ho; p.Wo = Wo; p.ksize = ksize; p.stride = stride; p.pad = pad; p.ups = ups;
  p.K = ksize * ksize * (C1 + C2); p.w = (const half_t*)w; p.ldw = p.K; p.M = B * Ho * Wo; p.N = N;
  p.bias = bias; p.res = (const half_t*)res; p.ldres = N; p.out = (half_t*)out; p.ldo = N;
  CK(launch_igemm(p, c->splitk_ws, c->splitk_bytes, c->st, force_cfg, force_split));
  return 0;
}
/* conv + the per-(m-tile, channel) GroupNorm partial sums its epilogue produces (the statistics fusion of resnet_fwd): stats_out
 * [ceil(M / tile_rows)][N][2] fp32 = (sum, sum of squares) of the stored fp16 values; *tile_rows_out = rows per m-tile, 0 when the
 * launch configuration produced no statistics (split-K, non-DMA shapes). */
int pnpi_op_conv_stats(pnpi_ctx* c, const void* x1, const void* x2, int C1, int C2, int B, int H, int W, int ksize, int stride, int pad,
                       int ups, int Ho, int Wo, const void* w, const float* bias, const void* res, int N, void* out, int force_cfg,
                       int force_split, float* stats_out, int* tile_rows_out) {
  if (!c || !stats_out || !tile_rows_out) return PNPI_EINVAL;
  GemmP p; gemm_defaults(p);
  p.x1 = (const half_t*)x1; p.x2 = (const half_t*)x2; p.C1 = C1; p.C2 = C2; p.ldx1 = C1; p.ldx2 = C2;
  p.B = B; p.H = H; p.W = W; p.Ho = Ho; p.Wo = Wo; p.ksize = ksize; p.stride = stride; p.pad = pad; p.ups = ups;
  p.K = ksize * ksize * (C1 + C2); p.w = (const half_t*)w; p.ldw = p.K; p.M = B * Ho * Wo; p.N = N;
  p.bias = bias; p.res = (const half_t*)res; p.ldres = N; p.out = (half_t*)out; p.ldo = N; p.stats = stats_out;
  CK(launch_igemm(p, c->splitk_ws, c->splitk_bytes, c->st, force_cfg, force_split, nullptr, tile_rows_out));
  return 0;
}
/* process-wide kernel tuning knobs (tile-variant A/B inside one process, tests of non-default variants) */
int pnpi_tile_table_lookup(int M, int N, int K, int ksize, int* cfg, int* split, int* entry_m) {
  return igemm_table_lookup(M, N, K, ksize, cfg, split, entry_m);
}
int pnpi_set_tuning(const char* key, int value) {
  if (!key) return PNPI_EINVAL;
  if (!strcmp(key, "text_kv")) { g_text_kv = value; return 0; }
  if (!strcmp(key, "temb_cache")) { g_temb_cache = value; return 0; }
  if (!strcmp(key, "gn_inline_rows")) { norm_set_tuning_gn_inline_rows(value); return 0; }
  if (!strcmp(key, "attn_vt_perm")) { g_vt_perm = value; return 0; }
  if (!strcmp(key, "attn_aug")) { g_attn_aug = value; return 0; }
  if (!strcmp(key, "attn_pipe")) return attn_set_tuning_pipe(value) == 0 ? 0 : PNPI_EINVAL;
  if (!strcmp(key, "gn_slab")) { g_gn_slab = value; return 0; }
  if (!strcmp(key, "op_attention_aug")) { g_op_attention_aug = value; return 0; }
  if (!strcmp(key, "op_attention_vt_perm")) { g_op_attention_vt_perm = value; return 0; }
  if (!strcmp(key, "attn_bwd_flash")) { g_attn_bwd_flash = value; return 0; }
  return igemm_set_tuning(key, value) == 0 ? 0 : PNPI_EINVAL;
}
int pnpi_op_gemm(pnpi_ctx* c, const void* a, int lda, const void* w, int ldw, int M, int N, int K, float alpha, const float* bias,
                 const void* res, void* out, int ldo, int vt_col0, void* outT, int vt_ld, int vt_f32, int rpb, int force_cfg,
                 int force_split) {
  GemmP p; gemm_defaults(p);
  p.x1 = (const half_t*)a; p.C1 = K; p.ldx1 = lda; p.B = 1; p.H = 1; p.W = M; p.Ho = 1; p.Wo = M; p.ksize = 1;
  p.w = (const half_t*)w; p.ldw = ldw; p.M = M; p.N = N; p.K = K; p.alpha = alpha; p.bias = bias; p.res = (const half_t*)res;
  p.ldres = N; p.out = (half_t*)out; p.ldo = ldo;
  if (outT) { p.outT = outT; p.vt_col0 = vt_col0; p.vt_ld = vt_ld; p.vt_f32 = vt_f32; p.rows_per_batch = rpb; }
  CK(launch_igemm(p, c->splitk_ws, c->splitk_bytes, c->st, force_cfg, force_split));
  return 0;
}
int pnpi_op_gemm_geglu(pnpi_ctx* c, const void* a, int lda, const void* w, int ldw, int M, int N, int K, const float* bias, void* out,
                       int ldo) {
  GemmP p; gemm_defaults(p);
  p.x1 = (const half_t*)a; p.C1 = K; p.ldx1 = lda; p.B = 1; p.H = 1; p.W = M; p.Ho = 1; p.Wo = M; p.ksize = 1;
  p.w = (const half_t*)w; p.ldw = ldw; p.M = M; p.N = N; p.K = K; p.bias = bias; p.out = (half_t*)out; p.ldo = ldo; p.geglu = 1;
  CK(launch_igemm(p, c->splitk_ws, c->splitk_bytes, c->st));
  return 0;
}
int pnpi_op_groupnorm(pnpi_ctx* c, const void* x1, const void* x2, int C1, int C2, int B, int HW, int G, float eps, const float* gamma,
                      const float* beta, int silu, void* out) {
  CK(launch_groupnorm((const half_t*)x1, (const half_t*)x2, C1, C2, B, HW, G, eps, gamma, beta, silu, (half_t*)out, c->gn_partial, c->st));
  return 0;
}
int pnpi_op_layernorm(pnpi_ctx* c, const void* x, int M, int C, float eps, const float* gamma, const float* beta, void* out) {
  CK(launch_layernorm((const half_t*)x, M, C, eps, gamma, beta, (half_t*)out, c->st));
  return 0;
}
int pnpi_op_geglu(pnpi_ctx* c, const void* x, int M, int inner, void* out) {
  CK(launch_geglu((const half_t*)x, M, inner, (half_t*)out, c->st));
  return 0;
}
int pnpi_op_softmax_rows(pnpi_ctx* c, void* x, int M, int N, int ld) {
  CK(launch_softmax_rows((half_t*)x, M, N, ld, c->st));
  return 0;
}
// ---- differentiable UNet forward (null-text path groundwork)
static int tape_ensure(pnpi_ctx* c) {
  if (c->tape) return 0;
  const pnpi_model_config& g = c->cfg;
  // The activation arenas were sized at create for max_unet_rows rows of the plain forward (block temporaries released in stack order).
  // A recording forward of ONE row keeps every temporary: measure it with a dry run and grow the arenas if that is more (set-up time
  // only -- nothing is allocated in the optimisation loop).  New buffers are allocated BEFORE the old ones are freed and the context's
  // state changes only after every allocation has succeeded: a failed hipMalloc leaves the context as it was (and without a tape).
  std::unique_ptr<Tape> T(new Tape());
  CKH(hipStreamSynchronize(c->st));
  size_t pp, tp;
  {
    const Bump sp = c->persist, stmp = c->temp;
    Tape* const prev = c->tape;
    c->persist = Bump(); c->temp = Bump();
    c->tape = T.get();
    c->dry = true; T->rec = true;
    const bool kv = c->tkv.use; c->tkv.use = false;
    const int r = unet_fwd(c, nullptr, 1, 0, nullptr, false, 0, nullptr);
    c->dry = false; T->rec = false; c->tkv.use = kv;
    pp = align_up(c->persist.peak + (1 << 20), 4096); tp = align_up(c->temp.peak + (1 << 20), 4096);
    c->persist = sp; c->temp = stmp; c->tape = prev;
    if (r) return r;
  }
  // gradients + dgrad scratch of one UNet row: about 2.8x the recorded activations at SD-1.x width (1.1 of 0.4 GB); 6x with a 64 MB floor
  const size_t gcap = align_up(std::max((size_t)64 << 20, 6 * (pp + tp)), 4096);
  char *gbase = nullptr, *nper = nullptr, *ntmp = nullptr;
  float* dctx = nullptr;
  auto undo = [&]() { if (gbase) (void)hipFree(gbase); if (nper) (void)hipFree(nper); if (ntmp) (void)hipFree(ntmp); if (dctx) (void)hipFree(dctx); };
  hipError_t e = hipMalloc((void**)&gbase, gcap);
  if (e == hipSuccess) e = hipMalloc((void**)&dctx, (size_t)g.ctx_len * g.cross_dim * sizeof(float));
  if (e == hipSuccess && pp > c->persist.cap) e = hipMalloc((void**)&nper, pp);
  if (e == hipSuccess && tp > c->temp.cap) e = hipMalloc((void**)&ntmp, tp);
  if (e != hipSuccess) { undo(); const std::string msg = std::string("null-text tape: ") + hipGetErrorString(e); return fail(c, PNPI_EHIP, msg.c_str()); }
  if (nper) { (void)hipFree(c->persist.base); c->persist.base = nper; c->persist.cap = pp; }
  if (ntmp) { (void)hipFree(c->temp.base); c->temp.base = ntmp; c->temp.cap = tp; }
  c->persist.reset(); c->temp.reset(); c->persist.overflow = false; c->temp.overflow = false;
  T->garena.base = gbase; T->garena.cap = gcap; T->d_ctx = dctx;
  c->tape = T.release();
  return 0;
}
// eps = UNet(latents, t, context) for ONE row, and d_context = (d loss / d eps)^T (d eps / d context) for the given d loss / d eps
// (fp32, the layout of eps; pre-multiplied by the caller's power-of-two loss scale -- activations' gradients travel in fp16).
int pnpi_unet_context_grad(pnpi_ctx* c, const float* latents, int t, const float* context, const float* d_eps, float* eps_out, float* d_context_out) {
  if (!c || !latents || !context || !d_eps || !d_context_out) return PNPI_EINVAL;
  CKP(check_loop_ready(c));
  CKP(tape_ensure(c));
  Tape& T = *c->tape;
  const pnpi_model_config& g = c->cfg;
  const size_t E = (size_t)g.in_channels * g.sample_size * g.sample_size, CE = (size_t)g.ctx_len * g.cross_dim;
  CKP(setup_ctrl(c, nullptr, 0, c->max_rows));
  float* eps = eps_out ? eps_out : misc_f(c, E);
  T.ops.clear(); T.grads.clear(); T.garena.reset(); T.garena.overflow = false;
  c->tkv.use = false;
  T.rec = true;
  int r = unet_fwd(c, latents, 1, t, context, false, 0, eps);
  T.rec = false;
  if (r) return r;
  half_t* d_out = tape_galloc(c, (size_t)g.sample_size * g.sample_size * 8);
  if (!d_out) return fail(c, PNPI_ENOMEM, "gradient arena overflow");
  CK(launch_nchw_f32_to_nhwc_f16(d_eps, 1, g.in_channels, g.sample_size * g.sample_size, 8, d_out, c->st));
  CKH(hipMemsetAsync(T.d_ctx, 0, CE * sizeof(float), c->st));
  CKP(tape_backward(c, d_out));
  CKH(hipMemcpyAsync(d_context_out, T.d_ctx, CE * sizeof(float), hipMemcpyDeviceToDevice, c->st));
  return 0;
}

// The Adam loop both optimisations share (inversion.py:203-218 and :430-447): eps_c = UNet(lat, t, ctx_cond) once, then up to
// num_inner_steps x {recording forward with the current embedding `unc`, loss = mse(prev_step(CFG), target) and its gradient, reverse
// walk to the embedding, Adam (torch.optim.Adam defaults, state fresh per DDIM step)}; the loss is read back for the reference's
// early-stop test `loss < epsilon + i * 2e-5`.  eps2 = [eps_u | eps_c] (2E floats).  losses_host (nullable): [num_inner_steps].
struct NullOptBufs { float *eps2, *d_eps, *am, *av, *loss_d; };
static int null_inner_loop(pnpi_ctx* c, const NullOptBufs& b, const float* lat, int t, int i, float* unc, const float* ctx_cond,
                           const float* target, float guidance_scale, float a_t, float a_p, int num_inner_steps, float epsilon,
                           int* its_out, float* losses_host) {
  const pnpi_model_config& g = c->cfg;
  const size_t E = (size_t)g.in_channels * g.sample_size * g.sample_size, CE = (size_t)g.ctx_len * g.cross_dim;
  const float scale = 4096.f;                                   // loss scale of the fp16 activation gradients (removed before Adam)
  const double sa_t = sqrt((double)a_t), sb_t = sqrt(1.0 - a_t), sa_p = sqrt((double)a_p), sb_p = sqrt(1.0 - a_p);
  const float c_x = (float)(sa_p / sa_t), c_e = (float)(sb_p - sa_p * sb_t / sa_t);       // rec = c_x x + c_e eps
  const float lr = (float)(1e-2 * (1.0 - i / 100.0));
  int its = 0;
  c->tkv.use = false;
  int r = unet_fwd(c, lat, 1, t, ctx_cond, false, 0, b.eps2 + E);
  if (r) return r;
  CKH(hipMemsetAsync(b.am, 0, CE * sizeof(float), c->st));
  CKH(hipMemsetAsync(b.av, 0, CE * sizeof(float), c->st));
  for (int j = 0; j < num_inner_steps; ++j) {
    // forward with the tape recording; the loss head needs eps_u first, so forward and backward are two calls of the tape machinery
    Tape& T = *c->tape;
    T.ops.clear(); T.grads.clear(); T.garena.reset(); T.garena.overflow = false;
    T.rec = true;
    r = unet_fwd(c, lat, 1, t, unc, false, 0, b.eps2);
    T.rec = false;
    if (r) return r;
    CK(launch_null_text_loss(b.eps2, b.eps2 + E, lat, target, (int)E, guidance_scale, c_x, c_e, scale, b.d_eps, b.loss_d, c->st));
    half_t* d_out = tape_galloc(c, (size_t)g.sample_size * g.sample_size * 8);
    if (!d_out) return fail(c, PNPI_ENOMEM, "gradient arena overflow");
    CK(launch_nchw_f32_to_nhwc_f16(b.d_eps, 1, g.in_channels, g.sample_size * g.sample_size, 8, d_out, c->st));
    CKH(hipMemsetAsync(T.d_ctx, 0, CE * sizeof(float), c->st));
    CKP(tape_backward(c, d_out));
    CK(launch_adam_step(unc, b.am, b.av, T.d_ctx, (int)CE, j + 1, lr, 1.f / scale, c->st));
    float loss_h = 0.f;
    CKH(hipMemcpyAsync(&loss_h, b.loss_d, sizeof(float), hipMemcpyDeviceToHost, c->st));
    CKH(hipStreamSynchronize(c->st));
    if (losses_host) losses_host[j] = loss_h;
    its = j + 1;
    c->ctr.unet_backward_rows += 1;
    if (loss_h < epsilon + i * 2e-5f) break;
  }
  *its_out = its;
  return 0;
}
static int null_bufs(pnpi_ctx* c, NullOptBufs& b) {
  const pnpi_model_config& g = c->cfg;
  const size_t E = (size_t)g.in_channels * g.sample_size * g.sample_size, CE = (size_t)g.ctx_len * g.cross_dim;
  b.eps2 = misc_f(c, 2 * E); b.d_eps = misc_f(c, E); b.am = misc_f(c, CE); b.av = misc_f(c, CE); b.loss_d = misc_f(c, 1);
  return 0;
}

// NullInversion.null_optimization (models/p2p/inversion.py:196-225) for one image, device resident.  ddim_latents [nsteps + 1][E] (the
// inversion trajectory, x*_0 first), ctx_uncond / ctx_cond [77][768]; uncond_out [nsteps][77][768] receives the optimised embedding of
// every step; then the CFG step with the optimised embedding moves the latent on.  losses_out (nullable, host): [nsteps][num_inner_steps]
// loss of every Adam iteration (-1 for iterations the early stop skipped).
int pnpi_null_text_optimize(pnpi_ctx* c, const float* ddim_latents, const float* ctx_uncond, const float* ctx_cond, int nsteps,
                            const int* ts, float guidance_scale, int num_inner_steps, float epsilon, float* uncond_out, int* iters_out,
                            float* losses_out) {
  if (!c || !ddim_latents || !ctx_uncond || !ctx_cond || !ts || !uncond_out || nsteps <= 0 || num_inner_steps < 0) return PNPI_EINVAL;
  CKP(check_loop_ready(c));
  CKP(tape_ensure(c));
  const pnpi_model_config& g = c->cfg;
  const size_t E = (size_t)g.in_channels * g.sample_size * g.sample_size, CE = (size_t)g.ctx_len * g.cross_dim;
  const int ratio = g.n_train_timesteps / nsteps;
  CKP(setup_ctrl(c, nullptr, 0, c->max_rows));
  NullOptBufs b; null_bufs(c, b);
  float* lat = misc_f(c, E);
  float* unc = misc_f(c, CE);
  if (c->ctrl_arena.overflow) return fail(c, PNPI_ENOMEM, "loop arena overflow");
  if (losses_out) for (int k = 0; k < nsteps * num_inner_steps; ++k) losses_out[k] = -1.f;
  CKH(hipMemcpyAsync(unc, ctx_uncond, CE * sizeof(float), hipMemcpyDeviceToDevice, c->st));
  CKH(hipMemcpyAsync(lat, ddim_latents + (size_t)nsteps * E, E * sizeof(float), hipMemcpyDeviceToDevice, c->st));
  for (int i = 0; i < nsteps; ++i) {
    const int t = ts[i];
    float a_t, a_p; CKP(alphas_for(c, t, ratio, false, &a_t, &a_p));
    const float* target = ddim_latents + (size_t)(nsteps - i - 1) * E;
    int its = 0;
    if (num_inner_steps > 0)
      CKP(null_inner_loop(c, b, lat, t, i, unc, ctx_cond, target, guidance_scale, a_t, a_p, num_inner_steps, epsilon, &its,
                          losses_out ? losses_out + (size_t)i * num_inner_steps : nullptr));
    if (iters_out) iters_out[i] = its;
    CKH(hipMemcpyAsync(uncond_out + (size_t)i * CE, unc, CE * sizeof(float), hipMemcpyDeviceToDevice, c->st));
    // latent_cur = prev_step(CFG(eps(unc), eps(cond)))   (get_noise_pred with the optimised embedding, inversion.py:221-224)
    c->tkv.use = false;
    int r = unet_fwd(c, lat, 1, t, unc, false, 0, b.eps2);
    if (r) return r;
    if (num_inner_steps == 0) { r = unet_fwd(c, lat, 1, t, ctx_cond, false, 0, b.eps2 + E); if (r) return r; }
    CK(launch_cfg_ddim_prev(b.eps2, lat, 1, 1, E, guidance_scale, a_t, a_p, nullptr, 0, nullptr, 1.f, nullptr, lat, c->st));
  }
  return 0;
}

// DirectInversion.null_latent_calculate (models/p2p/inversion.py:419-460, "ablation_null-latent-inversion+p2p") for one (source, target)
// prompt pair.  context4 rows = [unc_src, unc_tgt, cond_src, cond_tgt].  Per step: the unconditional embeddings are optimised as in
// null-text inversion -- the reference's loss reads the SOURCE row only (:441), so the target row's embedding has a zero gradient, Adam
// leaves it where it is, and only the source row needs the recording forward / reverse walk; its conditional prediction is constant over
// the iterations -- then the step's effect becomes a latent offset for both rows:
//   noise_loss[i] = prev_step(CFG with the optimised embeddings) - prev_step(CFG with the ORIGINAL ones),  latent_cur = plain + noise_loss[i]
// (:449-459).  The two 4-row forwards run with rows ordered [unc_src, cond_src, unc_tgt, cond_tgt] (row results do not depend on the
// order) so that the step kernel sees them as two one-row images.  noise_loss_out [nsteps][2][E].
int pnpi_null_latent_calculate(pnpi_ctx* c, const float* ddim_latents, const float* context4, int nsteps, const int* ts, float guidance_scale,
                               int num_inner_steps, float epsilon, float* noise_loss_out, int* iters_out, float* losses_out) {
  if (!c || !ddim_latents || !context4 || !ts || !noise_loss_out || nsteps <= 0 || num_inner_steps < 0) return PNPI_EINVAL;
  CKP(check_loop_ready(c));
  if (c->max_rows < 4) return fail(c, PNPI_EINVAL, "null-latent inversion needs max_unet_rows >= 4");
  CKP(tape_ensure(c));
  const pnpi_model_config& g = c->cfg;
  const size_t E = (size_t)g.in_channels * g.sample_size * g.sample_size, CE = (size_t)g.ctx_len * g.cross_dim;
  const int ratio = g.n_train_timesteps / nsteps;
  CKP(setup_ctrl(c, nullptr, 0, c->max_rows));
  NullOptBufs b; null_bufs(c, b);
  float* cur = misc_f(c, 2 * E);          // latent_cur [src, tgt]
  float* in4 = misc_f(c, 4 * E);          // [src, src, tgt, tgt]
  float* eps4 = misc_f(c, 4 * E);
  float* opt = misc_f(c, 2 * E);
  float* unc = misc_f(c, 2 * CE);         // the embeddings being optimised [src, tgt] (warm-started from step to step)
  float* ctx4 = misc_f(c, 4 * CE);        // [unc_src, cond_src, unc_tgt, cond_tgt] of the forward at hand
  if (c->ctrl_arena.overflow) return fail(c, PNPI_ENOMEM, "loop arena overflow");
  if (losses_out) for (int k = 0; k < nsteps * num_inner_steps; ++k) losses_out[k] = -1.f;
  const float* cond = context4 + 2 * CE;
  auto d2d = [&](float* d, const float* s, size_t n) { return hipMemcpyAsync(d, s, n * sizeof(float), hipMemcpyDeviceToDevice, c->st); };
  CKH(d2d(unc, context4, 2 * CE));
  CKH(d2d(cur, ddim_latents + (size_t)nsteps * E, E));
  CKH(d2d(cur + E, ddim_latents + (size_t)nsteps * E, E));
  CKH(d2d(ctx4 + CE, cond, CE));
  CKH(d2d(ctx4 + 3 * CE, cond + CE, CE));
  for (int i = 0; i < nsteps; ++i) {
    const int t = ts[i];
    float a_t, a_p; CKP(alphas_for(c, t, ratio, false, &a_t, &a_p));
    const float* target = ddim_latents + (size_t)(nsteps - i - 1) * E;
    int its = 0;
    if (num_inner_steps > 0)
      CKP(null_inner_loop(c, b, cur, t, i, unc, cond, target, guidance_scale, a_t, a_p, num_inner_steps, epsilon, &its,
                          losses_out ? losses_out + (size_t)i * num_inner_steps : nullptr));
    if (iters_out) iters_out[i] = its;
    CKH(d2d(in4, cur, E)); CKH(d2d(in4 + E, cur, E)); CKH(d2d(in4 + 2 * E, cur + E, E)); CKH(d2d(in4 + 3 * E, cur + E, E));
    c->tkv.use = false;
    // with the optimised embeddings -> opt
    CKH(d2d(ctx4, unc, CE)); CKH(d2d(ctx4 + 2 * CE, unc + CE, CE));
    int r = unet_fwd(c, in4, 4, t, ctx4, false, 0, eps4);
    if (r) return r;
    CK(launch_cfg_ddim_prev(eps4, cur, 2, 1, E, guidance_scale, a_t, a_p, nullptr, 0, nullptr, 1.f, nullptr, opt, c->st));
    // with the original ones -> plain; loss = opt - plain; latent_cur = plain + loss
    CKH(d2d(ctx4, context4, CE)); CKH(d2d(ctx4 + 2 * CE, context4 + CE, CE));
    r = unet_fwd(c, in4, 4, t, ctx4, false, 0, eps4);
    if (r) return r;
    CK(launch_cfg_ddim_prev(eps4, cur, 2, 1, E, guidance_scale, a_t, a_p, nullptr, 0, opt, 1.f, noise_loss_out + (size_t)i * 2 * E, cur, c->st));
  }
  return 0;
}

int pnpi_op_attention_bwd(pnpi_ctx* c, const void* q, int ldq, int q_off, const void* k, int ldk, int k_off, const void* v, int ldv, int v_off,
                          const void* d_o, int ldo, int heads, int Nq, int Nk, int Dp, int dh, float scale, int B, void* dq, void* dk, void* dv,
                          void* scratch, size_t scratch_bytes) {
  if (!c || !q || !k || !v || !d_o || !dq || !dk || !dv) return PNPI_EINVAL;
  CKP(attn_bwd(c, (const half_t*)q, ldq, q_off, (const half_t*)k, ldk, k_off, (const half_t*)v, ldv, v_off, (const half_t*)d_o, ldo,
               heads, Nq, Nk, Dp, dh, scale, B, (half_t*)dq, (half_t*)dk, (half_t*)dv, scratch, scratch_bytes));
  return 0;
}
size_t pnpi_op_attention_bwd_scratch_bytes(int Nq, int Nk, int dh) { return attn_bwd_scratch_bytes(Nq, Nk, dh); }
// ---- activation-gradient kernels (null-text path groundwork; tests/test_gpu_backward.py)
int pnpi_op_layernorm_bwd(pnpi_ctx* c, const void* x, const void* dy, int M, int C, float eps, const float* gamma, void* dx) {
  CK(launch_layernorm_bwd((const half_t*)x, (const half_t*)dy, M, C, eps, gamma, (half_t*)dx, c->st));
  return 0;
}
int pnpi_op_groupnorm_bwd(pnpi_ctx* c, const void* x1, const void* x2, int C1, int C2, int B, int HW, int G, float eps, const float* gamma,
                          const float* beta, int silu, const void* dy, void* dx) {
  const int C = C1 + C2;
  if (!(C & 7) && !(C1 & 7) && G <= 64) {      // the product path: three chip-wide phases (here with one dense destination)
    float* ws = nullptr;
    CKP(gn_bwd_workspace(c, B, HW, G, &ws));
    CK(launch_groupnorm_bwd2((const half_t*)x1, (const half_t*)x2, C1, C2, B, HW, G, eps, gamma, beta, silu, (const half_t*)dy,
                             GnbOut{(half_t*)dx, C, 0}, GnbOut{(half_t*)dx + C1, C, 0}, ws, c->st));
    return 0;
  }
  CK(launch_groupnorm_bwd((const half_t*)x1, (const half_t*)x2, C1, C2, B, HW, G, eps, gamma, beta, silu, (const half_t*)dy, (half_t*)dx, c->st));
  return 0;
}
int pnpi_op_geglu_bwd(pnpi_ctx* c, const void* h, const void* dy, int M, int inner, void* dh) {
  CK(launch_geglu_bwd((const half_t*)h, (const half_t*)dy, M, inner, (half_t*)dh, c->st));
  return 0;
}
int pnpi_op_softmax_bwd_rows(pnpi_ctx* c, const float* P, const float* dP, int R, int N, int ld, float scale, void* dS) {
  CK(launch_softmax_bwd_rows(P, dP, (size_t)R, N, ld, scale, (half_t*)dS, c->st));
  return 0;
}
int pnpi_op_accumulate(pnpi_ctx* c, void* dst, const void* src, size_t n) {
  CK(launch_accumulate_f16((half_t*)dst, (const half_t*)src, n, c->st));
  return 0;
}
int pnpi_op_sumpool2x2(pnpi_ctx* c, const void* dup, int B, int H, int W, int C, void* dx) {
  CK(launch_sumpool2x2((const half_t*)dup, B, H, W, C, (half_t*)dx, c->st));
  return 0;
}
int pnpi_op_zero_stuff2(pnpi_ctx* c, const void* dy, int B, int Ho, int Wo, int C, void* out) {
  CK(launch_zero_stuff2((const half_t*)dy, B, Ho, Wo, C, (half_t*)out, c->st));
  return 0;
}
int pnpi_op_repack_dgrad(pnpi_ctx* c, const void* w, int N, int taps, int Cin, void* wd) {
  CK(launch_repack_dgrad((const half_t*)w, N, N, taps, Cin, (half_t*)wd, c->st));
  return 0;
}
int pnpi_op_null_text_loss(pnpi_ctx* c, const float* eps_u, const float* eps_c, const float* x, const float* target, int n, float w, float c_x,
                           float c_e, float grad_scale, void* d_eps_u, float* loss) {
  CK(launch_null_text_loss(eps_u, eps_c, x, target, n, w, c_x, c_e, grad_scale, (float*)d_eps_u, loss, c->st));
  return 0;
}
int pnpi_op_adam_step(pnpi_ctx* c, float* p, float* m, float* v, const float* g, int n, int k, float lr, float inv_scale) {
  CK(launch_adam_step(p, m, v, g, n, k, lr, inv_scale, c->st));
  return 0;
}
int pnpi_op_attention(pnpi_ctx* c, const void* q, int ldq, int q_off, const void* k, int ldk, int k_off, const void* vt, int ldv,
                      void* o, int ldo, int heads, int Nq, int Nk, int Dp, int dh, float scale, const int* rows_dev, int nrows) {
  AttnP a; a.q = (const half_t*)q; a.ldq = ldq; a.q_off = q_off; a.k = (const half_t*)k; a.ldk = ldk; a.k_off = k_off;
  a.vt = (const half_t*)vt; a.ldv = ldv; a.o = (half_t*)o; a.ldo = ldo; a.heads = heads; a.Nq = Nq; a.Nk = Nk; a.Dp = Dp; a.dh = dh;
  a.scale = scale; a.rows = rows_dev; a.nrows = nrows;
  a.vt_perm = (g_op_attention_vt_perm && attn_flash_uses_dma64(Dp, Nk, 0)) ? 1 : 0;
  a.aug = g_op_attention_aug;
  CK(launch_attn_flash(a, c->st));
  return 0;
}
int pnpi_op_cross_edit(pnpi_ctx* c, const void* q, int ldq, int q_off, const void* k, int ldk, int k_off, const void* vt, int ldv,
                       void* o, int ldo, int heads, int Nq, int Nk, int Dp, int dh, float scale, const int* pairs_dev, int npairs,
                       const void* mmatT, const float* c1, const float* c2, const float* lb_alpha, float* lb_acc, int lb_slot0,
                       int lb_nslots) {
  CrossEditP e; e.q = (const half_t*)q; e.ldq = ldq; e.q_off = q_off; e.k = (const half_t*)k; e.ldk = ldk; e.k_off = k_off;
  e.vt = (const half_t*)vt; e.ldv = ldv; e.o = (half_t*)o; e.ldo = ldo; e.heads = heads; e.Nq = Nq; e.Nk = Nk; e.Dp = Dp; e.dh = dh;
  e.scale = scale; e.pairs = pairs_dev; e.npairs = npairs; e.mmatT = (const half_t*)mmatT; e.c1 = c1; e.c2 = c2;
  e.lb_alpha = lb_alpha; e.lb_acc = lb_acc; e.lb_slot0 = lb_slot0; e.lb_nslots = lb_nslots; e.write_src = 1;
  CK(launch_attn_cross_edit(e, c->st));
  return 0;
}
int pnpi_op_local_blend(pnpi_ctx* c, const float* lb_acc, int nslots, int map_hw, int lat_hw, int C, float th, float* latents, int nimg) {
  CK(launch_local_blend(lb_acc, nslots, map_hw, lat_hw, C, th, latents, nimg, c->st));
  return 0;
}
int pnpi_op_local_blend_sub(pnpi_ctx* c, const float* lb_acc, int nslots, int map_hw, int lat_hw, int C, float th, float th_sub,
                            float* latents, int nimg) {
  CK(launch_local_blend(lb_acc, nslots, map_hw, lat_hw, C, th, latents, nimg, c->st, 4, th_sub));
  return 0;
}

}  // extern "C"
