// libpnpi: C-ABI entry points + the static SD-1.x UNet / VAE graph executor and the device-resident DI / P2P loops.
// See include/pnpi.h for the reference interface each entry point replaces.
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

// ---------------------------------------------------------------------------------------------------- weight slots
static void reg_mat(pnpi_ctx* c, const std::string& name, half_t* dst, int rows, int cols, int taps, int dst_ld, int cin_pad,
                    int row0 = 0, int dh = 0, int Dp = 0) {
  Slot s; s.kind = 0; s.dst = dst; s.rows = rows; s.cols = cols; s.taps = taps; s.dst_ld = dst_ld; s.cin_pad = cin_pad;
  s.row0 = row0; s.dh = dh; s.Dp = Dp; s.n = 0; s.ilv_half = 0; s.loaded = false;
  c->slots[name] = s;
}
static void reg_vec(pnpi_ctx* c, const std::string& name, float* dst, int n) {
  Slot s; s.kind = 1; s.dst = dst; s.rows = 0; s.cols = 0; s.taps = 0; s.dst_ld = 0; s.cin_pad = 0; s.row0 = 0; s.dh = 0; s.Dp = 0;
  s.n = n; s.ilv_half = 0; s.loaded = false;
  c->slots[name] = s;
}
static half_t* walloc_h(pnpi_ctx* c, size_t n) { return (half_t*)c->warena.alloc(n * sizeof(half_t)); }
static float* walloc_f(pnpi_ctx* c, size_t n) { return (float*)c->warena.alloc(n * sizeof(float)); }

static ConvW make_conv(pnpi_ctx* c, const std::string& pre, int cin, int cout, int k, int cout_alloc = 0, float* bias_dst = nullptr) {
  ConvW w; w.cin = cin; w.cin_pad = round_up_i(cin, 8); w.cout = cout; w.k = k;
  int ra = cout_alloc > cout ? cout_alloc : cout;
  w.w = walloc_h(c, (size_t)ra * k * k * w.cin_pad);
  w.b = bias_dst ? bias_dst : walloc_f(c, ra);
  reg_mat(c, pre + ".weight", w.w, cout, cin, k * k, k * k * w.cin_pad, w.cin_pad);
  reg_vec(c, pre + ".bias", w.b, cout);
  return w;
}
static LinW make_lin(pnpi_ctx* c, const std::string& pre, int in, int out) {
  LinW l; l.in = in; l.out = out;
  l.w = walloc_h(c, (size_t)out * in);
  l.b = walloc_f(c, out);
  reg_mat(c, pre + ".weight", l.w, out, in, 1, in, in);
  reg_vec(c, pre + ".bias", l.b, out);
  return l;
}
static NormW make_norm(pnpi_ctx* c, const std::string& pre, int ch) {
  NormW n; n.c = ch; n.g = walloc_f(c, ch); n.b = walloc_f(c, ch);
  reg_vec(c, pre + ".weight", n.g, ch);
  reg_vec(c, pre + ".bias", n.b, ch);
  return n;
}
static ResnetW make_resnet(pnpi_ctx* c, const std::string& pre, int cin, int cout, bool temb) {
  ResnetW r; r.cin = cin; r.cout = cout;
  r.n1 = make_norm(c, pre + ".norm1", cin);
  if (temb) {
    UNetW& u = c->unet;
    r.temb_off = u.temb_total;
    r.c1 = make_conv(c, pre + ".conv1", cin, cout, 3, 0, u.conv1_b + r.temb_off);
    int te = 4 * c->cfg.block_out_channels[0];
    reg_mat(c, pre + ".time_emb_proj.weight", u.temb_w, cout, te, 1, te, te, r.temb_off);
    reg_vec(c, pre + ".time_emb_proj.bias", u.temb_b + r.temb_off, cout);
    u.temb_total += cout;
  } else {
    r.temb_off = -1;
    r.c1 = make_conv(c, pre + ".conv1", cin, cout, 3);
  }
  r.n2 = make_norm(c, pre + ".norm2", cout);
  r.c2 = make_conv(c, pre + ".conv2", cout, cout, 3);
  r.has_sc = cin != cout;
  if (r.has_sc) r.sc = make_conv(c, pre + ".conv_shortcut", cin, cout, 1);
  return r;
}
static TransformerW make_transformer(pnpi_ctx* c, const std::string& pre, int C, int place) {
  TransformerW t; t.C = C; t.heads = c->cfg.heads; t.dh = C / t.heads; t.Dp = round_up_i(t.dh, 32); t.place = place; t.lb_slot0 = -1;
  const int hd = t.heads * t.Dp, X = c->cfg.cross_dim;
  t.gn = make_norm(c, pre + ".norm", C);
  t.proj_in = make_conv(c, pre + ".proj_in", C, C, 1);
  const std::string tb = pre + ".transformer_blocks.0";
  t.ln1 = make_norm(c, tb + ".norm1", C);
  t.ln2 = make_norm(c, tb + ".norm2", C);
  t.ln3 = make_norm(c, tb + ".norm3", C);
  t.w_qkv = walloc_h(c, (size_t)3 * hd * C);
  reg_mat(c, tb + ".attn1.to_q.weight", t.w_qkv, C, C, 1, C, C, 0, t.dh, t.Dp);
  reg_mat(c, tb + ".attn1.to_k.weight", t.w_qkv, C, C, 1, C, C, hd, t.dh, t.Dp);
  reg_mat(c, tb + ".attn1.to_v.weight", t.w_qkv, C, C, 1, C, C, 2 * hd, t.dh, t.Dp);
  t.b_qkv_aug = nullptr;
  if (t.Dp == 64 && t.dh == 40) {      // SD-1.x's 64 x 64 level: one padding column of K and V carries a constant 1 (see TransformerW)
    t.b_qkv_aug = walloc_f(c, (size_t)3 * hd);
    c->aug_biases.push_back({t.b_qkv_aug, t.heads, t.Dp, t.dh});
  }
  t.o1 = make_lin(c, tb + ".attn1.to_out.0", C, C);
  t.w_q2 = walloc_h(c, (size_t)hd * C);
  reg_mat(c, tb + ".attn2.to_q.weight", t.w_q2, C, C, 1, C, C, 0, t.dh, t.Dp);
  t.w_kv2 = walloc_h(c, (size_t)2 * hd * X);
  reg_mat(c, tb + ".attn2.to_k.weight", t.w_kv2, C, X, 1, X, X, 0, t.dh, t.Dp);
  reg_mat(c, tb + ".attn2.to_v.weight", t.w_kv2, C, X, 1, X, X, hd, t.dh, t.Dp);
  t.o2 = make_lin(c, tb + ".attn2.to_out.0", C, C);
  t.ff1 = make_lin(c, tb + ".ff.net.0.proj", C, 8 * C);
  // GEGLU fused into this GEMM's epilogue: x rows and gate rows are interleaved in groups of 32 (attention.py:331-333)
  c->slots[tb + ".ff.net.0.proj.weight"].ilv_half = 4 * C;
  c->slots[tb + ".ff.net.0.proj.bias"].ilv_half = 4 * C;
  t.ff2 = make_lin(c, tb + ".ff.net.2", 4 * C, C);
  t.proj_out = make_conv(c, pre + ".proj_out", C, C, 1);
  return t;
}
static VaeAttnW make_vae_attn(pnpi_ctx* c, const std::string& pre, int C) {
  VaeAttnW a; a.C = C;
  a.gn = make_norm(c, pre + ".group_norm", C);
  a.w_qkv = walloc_h(c, (size_t)3 * C * C);
  a.b_qkv = walloc_f(c, 3 * C);
  reg_mat(c, pre + ".query.weight", a.w_qkv, C, C, 1, C, C, 0);
  reg_mat(c, pre + ".key.weight", a.w_qkv, C, C, 1, C, C, C);
  reg_mat(c, pre + ".value.weight", a.w_qkv, C, C, 1, C, C, 2 * C);
  reg_vec(c, pre + ".query.bias", a.b_qkv, C);
  reg_vec(c, pre + ".key.bias", a.b_qkv + C, C);
  reg_vec(c, pre + ".value.bias", a.b_qkv + 2 * C, C);
  a.proj = make_lin(c, pre + ".proj_attn", C, C);
  return a;
}

static int temb_total_channels(const pnpi_model_config& g) {
  int n = g.n_blocks, total = 0;
  for (int i = 0; i < n; ++i) total += g.layers_per_block * g.block_out_channels[i];
  total += 2 * g.block_out_channels[n - 1];
  for (int i = 0; i < n; ++i) total += (g.layers_per_block + 1) * g.block_out_channels[n - 1 - i];
  return total;
}

static void build_model(pnpi_ctx* c) {
  const pnpi_model_config& g = c->cfg;
  c->slots.clear();
  c->aug_biases.clear();
  UNetW& u = c->unet;
  u = UNetW();
  const int n = g.n_blocks, C0 = g.block_out_channels[0], TE = 4 * C0;
  const int sumC = temb_total_channels(g);
  u.temb_w = walloc_h(c, (size_t)sumC * TE);
  u.temb_b = walloc_f(c, sumC);
  u.conv1_b = walloc_f(c, sumC);
  u.temb_total = 0;
  u.conv_in = make_conv(c, "unet.conv_in", g.in_channels, C0, 3);
  u.t1 = make_lin(c, "unet.time_embedding.linear_1", C0, TE);
  u.t2 = make_lin(c, "unet.time_embedding.linear_2", TE, TE);
  std::vector<TransformerW*> down_stored, up_stored;
  int out_ch = C0;
  u.down_res.resize(n); u.down_attn.resize(n);
  for (int i = 0; i < n; ++i) {
    int in_ch = out_ch; out_ch = g.block_out_channels[i];
    for (int j = 0; j < g.layers_per_block; ++j) {
      std::string pre = "unet.down_blocks." + std::to_string(i);
      u.down_res[i].push_back(make_resnet(c, pre + ".resnets." + std::to_string(j), j == 0 ? in_ch : out_ch, out_ch, true));
      if (g.block_has_attn[i]) u.down_attn[i].push_back(make_transformer(c, pre + ".attentions." + std::to_string(j), out_ch, 0));
    }
    if (i != n - 1) u.down_samp.push_back(make_conv(c, "unet.down_blocks." + std::to_string(i) + ".downsamplers.0.conv", out_ch, out_ch, 3));
  }
  const int Cl = g.block_out_channels[n - 1];
  u.mid_res[0] = make_resnet(c, "unet.mid_block.resnets.0", Cl, Cl, true);
  u.mid_attn = make_transformer(c, "unet.mid_block.attentions.0", Cl, 1);
  u.mid_res[1] = make_resnet(c, "unet.mid_block.resnets.1", Cl, Cl, true);
  u.up_res.resize(n); u.up_attn.resize(n);
  out_ch = Cl;
  for (int i = 0; i < n; ++i) {
    int prev_out = out_ch; out_ch = g.block_out_channels[n - 1 - i];
    int in_ch = g.block_out_channels[n - 1 - (i + 1 < n ? i + 1 : n - 1)];
    for (int j = 0; j <= g.layers_per_block; ++j) {
      int skip_ch = (j == g.layers_per_block) ? in_ch : out_ch;
      int res_in = (j == 0) ? prev_out : out_ch;
      std::string pre = "unet.up_blocks." + std::to_string(i);
      u.up_res[i].push_back(make_resnet(c, pre + ".resnets." + std::to_string(j), res_in + skip_ch, out_ch, true));
      if (g.block_has_attn[n - 1 - i]) u.up_attn[i].push_back(make_transformer(c, pre + ".attentions." + std::to_string(j), out_ch, 2));
    }
    if (i != n - 1) u.up_samp.push_back(make_conv(c, "unet.up_blocks." + std::to_string(i) + ".upsamplers.0.conv", out_ch, out_ch, 3));
  }
  u.norm_out = make_norm(c, "unet.conv_norm_out", C0);
  u.conv_out = make_conv(c, "unet.conv_out", C0, g.out_channels, 3);

  // LocalBlend layers: attention_store["down_cross"][2:4] + ["up_cross"][:3] over the stored (<= 32^2 token) cross maps
  // (models/p2p/attention_control.py:112,223)
  std::vector<std::pair<TransformerW*, int>> dstored, ustored;
  for (int i = 0; i < n; ++i) {
    int tok = (g.sample_size >> i) * (g.sample_size >> i);
    for (auto& t : u.down_attn[i]) if (tok <= 1024) dstored.push_back({&t, tok});
  }
  for (int i = 0; i < n; ++i) {
    int s = g.sample_size >> (n - 1 - i);
    for (auto& t : u.up_attn[i]) if (s * s <= 1024) ustored.push_back({&t, s * s});
  }
  u.lb_nslots = 0; u.lb_tokens = 0;
  if (dstored.size() >= 4 && ustored.size() >= 3) {
    std::vector<std::pair<TransformerW*, int>> lb = {dstored[2], dstored[3], ustored[0], ustored[1], ustored[2]};
    bool same = true;
    for (auto& e : lb) same = same && e.second == lb[0].second;
    int side = (int)lroundf(sqrtf((float)lb[0].second));
    if (same && side * side == lb[0].second) {
      for (size_t k = 0; k < lb.size(); ++k) lb[k].first->lb_slot0 = (int)k * g.heads;
      u.lb_nslots = 5 * g.heads;
      u.lb_tokens = lb[0].second;
    }
  }

  // ---- VAE
  VaeW& v = c->vae;
  v = VaeW();
  const int vn = g.vae_n_blocks, L = g.vae_latent_channels;
  const int* vb = g.vae_block_out_channels;
  v.e_conv_in = make_conv(c, "vae.encoder.conv_in", g.vae_in_channels, vb[0], 3);
  v.e_res.resize(vn);
  int o = vb[0];
  for (int i = 0; i < vn; ++i) {
    int in_ch = o; o = vb[i];
    for (int j = 0; j < g.vae_layers_per_block; ++j)
      v.e_res[i].push_back(make_resnet(c, "vae.encoder.down_blocks." + std::to_string(i) + ".resnets." + std::to_string(j), j == 0 ? in_ch : o, o, false));
    if (i != vn - 1) v.e_down.push_back(make_conv(c, "vae.encoder.down_blocks." + std::to_string(i) + ".downsamplers.0.conv", o, o, 3));
  }
  const int Vl = vb[vn - 1];
  v.e_mid[0] = make_resnet(c, "vae.encoder.mid_block.resnets.0", Vl, Vl, false);
  v.e_attn = make_vae_attn(c, "vae.encoder.mid_block.attentions.0", Vl);
  v.e_mid[1] = make_resnet(c, "vae.encoder.mid_block.resnets.1", Vl, Vl, false);
  v.e_norm_out = make_norm(c, "vae.encoder.conv_norm_out", Vl);
  v.e_conv_out = make_conv(c, "vae.encoder.conv_out", Vl, 2 * L, 3, 8);
  v.quant = make_conv(c, "vae.quant_conv", 2 * L, 2 * L, 1, 8);
  v.post_quant = make_conv(c, "vae.post_quant_conv", L, L, 1, 8);
  v.d_conv_in = make_conv(c, "vae.decoder.conv_in", L, Vl, 3);
  v.d_mid[0] = make_resnet(c, "vae.decoder.mid_block.resnets.0", Vl, Vl, false);
  v.d_attn = make_vae_attn(c, "vae.decoder.mid_block.attentions.0", Vl);
  v.d_mid[1] = make_resnet(c, "vae.decoder.mid_block.resnets.1", Vl, Vl, false);
  v.d_res.resize(vn);
  o = Vl;
  for (int i = 0; i < vn; ++i) {
    int prev = o; o = vb[vn - 1 - i];
    for (int j = 0; j <= g.vae_layers_per_block; ++j)
      v.d_res[i].push_back(make_resnet(c, "vae.decoder.up_blocks." + std::to_string(i) + ".resnets." + std::to_string(j), j == 0 ? prev : o, o, false));
    if (i != vn - 1) v.d_up.push_back(make_conv(c, "vae.decoder.up_blocks." + std::to_string(i) + ".upsamplers.0.conv", o, o, 3));
  }
  v.d_norm_out = make_norm(c, "vae.decoder.conv_norm_out", vb[0]);
  v.d_conv_out = make_conv(c, "vae.decoder.conv_out", vb[0], g.vae_in_channels, 3);

  // ---- CLIP text encoder (transformers CLIPTextModel; keys without the optional "text_model." prefix)
  ClipW& t = c->clip;
  t = ClipW();
  if (g.clip_layers > 0) {
    t.H = g.cross_dim; t.heads = g.clip_heads; t.I = g.clip_intermediate; t.vocab = g.clip_vocab; t.T = g.ctx_len;
    const int H = t.H;
    t.tok = walloc_h(c, (size_t)t.vocab * H);
    t.pos = walloc_h(c, (size_t)t.T * H);
    reg_mat(c, "clip.embeddings.token_embedding.weight", t.tok, t.vocab, H, 1, H, H);
    reg_mat(c, "clip.embeddings.position_embedding.weight", t.pos, t.T, H, 1, H, H);
    for (int l = 0; l < g.clip_layers; ++l) {
      const std::string pre = "clip.encoder.layers." + std::to_string(l);
      ClipLayerW L;
      L.ln1 = make_norm(c, pre + ".layer_norm1", H);
      L.ln2 = make_norm(c, pre + ".layer_norm2", H);
      L.w_qkv = walloc_h(c, (size_t)3 * H * H);
      L.b_qkv = walloc_f(c, 3 * H);
      reg_mat(c, pre + ".self_attn.q_proj.weight", L.w_qkv, H, H, 1, H, H, 0);
      reg_mat(c, pre + ".self_attn.k_proj.weight", L.w_qkv, H, H, 1, H, H, H);
      reg_mat(c, pre + ".self_attn.v_proj.weight", L.w_qkv, H, H, 1, H, H, 2 * H);
      reg_vec(c, pre + ".self_attn.q_proj.bias", L.b_qkv, H);
      reg_vec(c, pre + ".self_attn.k_proj.bias", L.b_qkv + H, H);
      reg_vec(c, pre + ".self_attn.v_proj.bias", L.b_qkv + 2 * H, H);
      L.out = make_lin(c, pre + ".self_attn.out_proj", H, H);
      L.fc1 = make_lin(c, pre + ".mlp.fc1", H, t.I);
      L.fc2 = make_lin(c, pre + ".mlp.fc2", t.I, H);
      t.layers.push_back(L);
    }
    t.final_ln = make_norm(c, "clip.final_layer_norm", H);
  }
}

// ---------------------------------------------------------------------------------------------------- profiling
static void prof_open(pnpi_ctx* c, ProfRec& r) {
  (void)hipEventCreate(&r.a); (void)hipEventCreate(&r.b);
  (void)hipEventRecord(r.a, c->st);
}
static void prof_close(pnpi_ctx* c, ProfRec& r, int cls, double flops, double bytes, int M = 0, int N = 0, int K = 0, int ks = 0) {
  (void)hipEventRecord(r.b, c->st);
  r.cls = cls; r.flops = flops; r.bytes = bytes; r.M = M; r.N = N; r.K = K; r.ksize = ks;
  c->prof.push_back(r);
}
#define PROF(cls, flops, bytes, expr) PROFD(cls, flops, bytes, 0, 0, 0, expr)
#define PROFD(cls, flops, bytes, d0, d1, d2, expr)            \
  do {                                                        \
    if (c->prof_on && !c->dry) {                              \
      ProfRec _pr; prof_open(c, _pr);                         \
      int _r = (expr);                                        \
      prof_close(c, _pr, (cls), (flops), (bytes), (d0), (d1), (d2)); \
      if (_r) return fail_launch(c, _r, #expr);               \
    } else {                                                  \
      int _r = (expr);                                        \
      if (_r) return fail_launch(c, _r, #expr);               \
    }                                                         \
  } while (0)

// ---------------------------------------------------------------------------------------------------- activation tape
// A differentiable forward (NullInversion.null_optimization: loss.backward() w.r.t. the unconditional embedding, inversion.py:196-225)
// records every op of the UNet forward below -- through the same wrappers the plain forward uses -- and keeps all activations (the
// arenas stop releasing temporaries; one UNet row is ~0.4 GB).  tape_backward() walks the records in reverse.  Only activation
// gradients exist (no weight gradients); the fused variants that would hide an intermediate are switched off while recording
// (GEGLU in the GEMM epilogue, V^T written by the projection GEMM, the text K/V cache).
enum { TK_CONV = 0, TK_GEMM = 1, TK_GN = 2, TK_LN = 3, TK_ATTN = 4, TK_GEGLU = 5, TK_TRANSPOSE_V = 6 };
struct TapeOp {
  int kind = 0;
  const half_t *x1 = nullptr, *x2 = nullptr, *res = nullptr;
  half_t* out = nullptr;
  int C1 = 0, C2 = 0, B = 0, H = 0, W = 0, Ho = 0, Wo = 0, stride = 1, pad = 0, ups = 0, N = 0;      // conv
  const ConvW* cw = nullptr;
  int lda = 0, M = 0, K = 0, ldw = 0, ldo = 0;                                                        // gemm
  const half_t* w = nullptr;
  float alpha = 1.f;
  const NormW* nw = nullptr;                                                                         // norms
  int G = 0, HW = 0, silu = 0;
  float eps = 0.f;
  const half_t *q = nullptr, *k = nullptr, *v = nullptr;                                             // attention (base tensors)
  const float* lse = nullptr;                                                                        // [B][heads][Nq] log-sum-exp left by the forward kernel
  int ldq = 0, q_off = 0, ldk = 0, k_off = 0, ldv = 0, v_off = 0, heads = 0, Nq = 0, Nk = 0, Dp = 0, dh = 0;
  float scale = 0.f;
};
struct Tape {
  std::vector<TapeOp> ops;
  bool rec = false;                                   // recording (forward in flight)
  const half_t* no_grad_input = nullptr;              // the network input: its consumers' dgrad is skipped
  const half_t* ctx16 = nullptr;                      // the text context of this forward: gradients into it are summed in fp32
  Bump garena;                                        // gradient buffers + scratch of one backward
  std::unordered_map<const void*, half_t*> grads;     // activation pointer -> gradient buffer (same shape)
  std::unordered_map<const void*, half_t*> wd;        // weight pointer -> dgrad repack (device, built once)
  float* d_ctx = nullptr;                             // [rows * ctx_len * cross_dim] fp32
  void* attn_scratch = nullptr; size_t attn_scratch_bytes = 0;
};
static int g_attn_aug = 1;           // tuning "attn_aug" = 0: the 64-wide flash kernel ignores the augmented column (A/B)
static int g_vt_perm = 1;            // tuning "attn_vt_perm": V^T of the 4096-token self-attention sites in the permuted key order (A/B)
static int g_attn_bwd_flash = 1;     // tuning "attn_bwd_flash": self-attention backward without the [N][N] matrices in memory (0: the materialised form everywhere)
static int g_op_attention_aug = 0;       // tuning "op_attention_aug": pnpi_op_attention is handed K / V^T with 1.0 in column / row dh (kernel tests)
static int g_op_attention_vt_perm = 0;   // tuning "op_attention_vt_perm": pnpi_op_attention is handed a permuted V^T (kernel tests)
static inline bool taping(pnpi_ctx* c) { return c->tape && c->tape->rec && !c->dry; }
// a recording forward (or the dry run that sizes the arenas for one) keeps every activation and takes the plain-layout transformer block
static inline bool keep_acts(pnpi_ctx* c) { return c->tape && c->tape->rec; }

// ---------------------------------------------------------------------------------------------------- op wrappers
// Per-channel GroupNorm partial sums attached to an activation by the GEMM that produced it ([tiles][C][2], `rows` per tile).
struct Stats { const float* p = nullptr; int rows = 0; };
struct StatsReq { float* buf = nullptr; int rows = 0; };   // in: buffer; out: rows per tile actually produced (0 = none)

static half_t* talloc(pnpi_ctx* c, size_t n) { return (half_t*)c->temp.alloc(n * sizeof(half_t)); }
static half_t* palloc(pnpi_ctx* c, size_t n) { return (half_t*)c->persist.alloc(n * sizeof(half_t)); }

static int op_gn(pnpi_ctx* c, const half_t* x1, const half_t* x2, int C1, int C2, int B, int HW, const NormW& nw, int G, float eps,
                 int silu, half_t* out, Stats s1 = Stats(), Stats s2 = Stats()) {
  if (c->dry) return 0;
  if (taping(c)) {
    TapeOp o; o.kind = TK_GN; o.x1 = x1; o.x2 = x2; o.C1 = C1; o.C2 = C2; o.B = B; o.HW = HW; o.nw = &nw; o.G = G; o.eps = eps; o.silu = silu; o.out = out;
    c->tape->ops.push_back(o);
  }
  const bool ok1 = s1.p && s1.rows > 0 && HW % s1.rows == 0 && HW / s1.rows <= 256;
  const bool ok2 = !x2 || (s2.p && s2.rows > 0 && HW % s2.rows == 0 && HW / s2.rows <= 256);
  static const bool gn_nofuse = getenv("PNPI_GN_NOFUSE") != nullptr;   // ablation: always recompute the statistics
  if (ok1 && ok2 && !gn_nofuse) {
    PROFD(PNPI_KC_GROUPNORM, 0.0, 2.0 * B * HW * (double)(C1 + C2) * 2.0, B * HW, C1 + C2, 1,
          launch_groupnorm_fused(x1, x2, C1, C2, B, HW, G, eps, nw.g, nw.b, silu, out, s1.p, HW / s1.rows, s2.p,
                                 x2 ? HW / s2.rows : 1, c->gn_partial, c->st));
    return 0;
  }
  PROFD(PNPI_KC_GROUPNORM, 0.0, 3.0 * B * HW * (double)(C1 + C2) * 2.0, B * HW, C1 + C2, 0,
       launch_groupnorm(x1, x2, C1, C2, B, HW, G, eps, nw.g, nw.b, silu, out, c->gn_partial, c->st));
  return 0;
}

struct VtOut { void* outT = nullptr; int col0 = 1 << 30; int ld = 0; int f32 = 0; int rpb = 1; int perm16 = 0; };


static int igemm_prof(pnpi_ctx* c, const GemmP& p, double alg_flops, StatsReq* sr = nullptr) {
  int srows = 0, r;
  if (c->prof_on) {
    ProfRec pr; prof_open(c, pr);
    int used = 0;
    r = launch_igemm(p, c->splitk_ws, c->splitk_bytes, c->st, -1, 0, &used, &srows);
    // algorithmic HBM bytes: the input tensor(s) once, the weight once, the output once (+ the residual it adds)
    const double in_rows = (double)p.B * p.H * p.W;
    const double alg_bytes = 2.0 * (in_rows * (p.C1 + p.C2) + (double)p.N * p.K + (double)p.M * (p.geglu ? p.N / 2 : p.N) * (p.res ? 2.0 : 1.0));
    igemm_last_launch(&pr.cfg, &pr.split, pr.geom);
    prof_close(c, pr, used, alg_flops, alg_bytes, p.M, p.N, p.K, p.ksize);
  } else {
    r = launch_igemm(p, c->splitk_ws, c->splitk_bytes, c->st, -1, 0, nullptr, &srows);
  }
  if (sr) sr->rows = srows;
  return r;
}
static float* stats_alloc(pnpi_ctx* c, int M, int N) {   // worst case: 64-row tiles
  return (float*)c->persist.alloc((size_t)((M + 63) / 64) * N * 2 * sizeof(float));
}

static int op_conv(pnpi_ctx* c, const half_t* x1, int C1, const half_t* x2, int C2, int B, int H, int W, const ConvW& w, int stride,
                   int pad, int ups, const float* bias, const half_t* res, half_t* out, int Ho, int Wo, int N = -1,
                   const VtOut* vt = nullptr, StatsReq* sr = nullptr) {
  GemmP p; gemm_defaults(p);
  int C1p = C2 ? C1 : w.cin_pad;  // single-source inputs are stored with the padded channel count
  p.x1 = x1; p.x2 = x2; p.C1 = C1p; p.C2 = C2; p.ldx1 = C1p; p.ldx2 = C2;
  p.B = B; p.H = H; p.W = W; p.Ho = Ho; p.Wo = Wo; p.ksize = w.k; p.stride = stride; p.pad = pad; p.ups = ups;
  p.K = w.k * w.k * (C1p + C2); p.w = w.w; p.ldw = p.K;
  p.M = B * Ho * Wo; p.N = N > 0 ? N : w.cout;
  p.bias = bias; p.res = res; p.ldres = p.N; p.out = out; p.ldo = p.N;
  if (vt) { p.outT = vt->outT; p.vt_col0 = vt->col0; p.vt_ld = vt->ld; p.vt_f32 = vt->f32; p.rows_per_batch = vt->rpb; }
  c->ctr.executed_gemm_flops += 2.0 * p.M * p.N * p.K;
  if (sr) { sr->buf = stats_alloc(c, p.M, p.N); sr->rows = 0; p.stats = sr->buf; }
  if (c->dry) return 0;
  if (taping(c)) {
    TapeOp o; o.kind = TK_CONV; o.x1 = x1; o.x2 = x2; o.C1 = C1p; o.C2 = C2; o.B = B; o.H = H; o.W = W; o.Ho = Ho; o.Wo = Wo; o.stride = stride; o.pad = pad;
    o.ups = ups; o.N = p.N; o.cw = &w; o.res = res; o.out = out;
    c->tape->ops.push_back(o);
  }
  return igemm_prof(c, p, 2.0 * p.M * (double)w.cout * w.k * w.k * w.cin, sr);
}

struct GemmBatch { int n = 1; long sa = 0, sw = 0, sout = 0, soutT = 0; };   // n problems: strides of a / w / out in elements, of vt->outT in bytes
static int op_gemm(pnpi_ctx* c, const half_t* a, int lda, int M, int K, const half_t* w, int ldw, int N, const float* bias,
                   const half_t* res, int ldres, half_t* out, int ldo, float alpha = 1.f, const VtOut* vt = nullptr,
                   double alg_flops = -1.0, int geglu = 0, const GemmBatch* gb = nullptr) {
  GemmP p; gemm_defaults(p);
  if (gb && gb->n > 1) { p.nbatch = gb->n; p.sx1 = gb->sa; p.sw = gb->sw; p.sout = gb->sout; p.soutT = gb->soutT; }
  p.x1 = a; p.C1 = K; p.ldx1 = lda; p.B = 1; p.H = 1; p.W = M; p.Ho = 1; p.Wo = M; p.ksize = 1;
  p.w = w; p.ldw = ldw; p.M = M; p.N = N; p.K = K; p.bias = bias; p.res = res; p.ldres = ldres; p.alpha = alpha;
  p.out = out; p.ldo = ldo; p.geglu = geglu;
  if (vt) { p.outT = vt->outT; p.vt_col0 = vt->col0; p.vt_ld = vt->ld; p.vt_f32 = vt->f32; p.rows_per_batch = vt->rpb; p.vt_perm16 = vt->perm16; }
  c->ctr.executed_gemm_flops += 2.0 * M * N * K * p.nbatch;
  if (c->dry) return 0;
  if (taping(c) && out) {        // plain row-major outputs only: the recording forward uses no transposed / fused-GEGLU epilogue
    TapeOp o; o.kind = TK_GEMM; o.x1 = a; o.lda = lda; o.M = M; o.K = K; o.w = w; o.ldw = ldw; o.N = N; o.res = res; o.out = out; o.ldo = ldo; o.alpha = alpha;
    c->tape->ops.push_back(o);
  }
  return igemm_prof(c, p, alg_flops >= 0 ? alg_flops : 2.0 * M * (double)N * K * p.nbatch);
}

// ResnetBlock2D.forward (my_diffusers/models/resnet.py:331-365); x2 = skip tensor concatenated on the channel axis
static int resnet_fwd(pnpi_ctx* c, const ResnetW& r, const half_t* x1, int C1, const half_t* x2, int C2, int B, int H, int W, int G,
                      float eps, half_t* out, Stats s1 = Stats(), Stats s2 = Stats(), Stats* so = nullptr) {
  const size_t mk = c->temp.mark();
  const int HW = H * W;
  const size_t M = (size_t)B * HW;
  half_t* t1 = talloc(c, M * r.cin);
  CK(op_gn(c, x1, x2, C1, C2, B, HW, r.n1, G, eps, 1, t1, s1, s2));
  half_t* t2 = talloc(c, M * r.cout);
  const float* b1 = r.temb_off >= 0 ? c->bias_eff + r.temb_off : r.c1.b;
  StatsReq q1, q2;
  CK(op_conv(c, t1, r.cin, nullptr, 0, B, H, W, r.c1, 1, 1, 0, b1, nullptr, t2, H, W, -1, nullptr, &q1));
  half_t* t3 = talloc(c, M * r.cout);
  Stats st2; st2.p = q1.buf; st2.rows = q1.rows;
  CK(op_gn(c, t2, nullptr, r.cout, 0, B, HW, r.n2, G, eps, 1, t3, st2));
  const half_t* sc = x1;
  if (r.has_sc) {
    half_t* s = talloc(c, M * r.cout);
    CK(op_conv(c, x1, C1, x2, C2, B, H, W, r.sc, 1, 0, 0, r.sc.b, nullptr, s, H, W));
    sc = s;
  }
  CK(op_conv(c, t3, r.cout, nullptr, 0, B, H, W, r.c2, 1, 1, 0, r.c2.b, sc, out, H, W, -1, nullptr, &q2));
  if (so) { so->p = q2.buf; so->rows = q2.rows; }
  if (!keep_acts(c)) c->temp.release(mk);      // a recording forward keeps every activation for the backward pass
  return 0;
}

// The hooked attention forward of models/p2p/attention_control.py:20-47, literally: sim = q k^T * scale -> softmax -> controller(attn,
// is_cross, place) -> attn v, with the probabilities materialised in fp32 for a host callback (level-1 fallback for controllers
// without a descriptor).  One GEMM pair per (row, head); rows are not redirected (the callback does the editing).
static int attn_materialized(pnpi_ctx* c, const half_t* q, int ldq, int q_off, const half_t* k, int ldk, int k_off, const half_t* vt, int ldv,
                             half_t* o, int ldo, int heads, int Nq, int Nk, int Dp, int dh, float scale, int B, int is_cross, int place,
                             int layer) {
  const int ldp = round_up_i(Nk, 8);
  const size_t need = align_up((size_t)B * heads * Nq * Nk * sizeof(float), 256);
  // the fp16 copy of one (row, head) probability block (the MFMA operand of P V) lives behind the fp32 tensor in the caller's buffer
  if (!c->attn_buf || need + (size_t)Nq * ldp * sizeof(half_t) > c->attn_buf_bytes)
    return fail(c, PNPI_ENOMEM, "attention callback buffer too small for this site (rows*heads*Nq*Nk floats + Nq*Nk halfs)");
  half_t* p16 = reinterpret_cast<half_t*>(reinterpret_cast<char*>(c->attn_buf) + need);
  for (int b = 0; b < B; ++b)
    for (int h = 0; h < heads; ++h) {
      float* S = c->attn_buf + ((size_t)b * heads + h) * Nq * Nk;
      VtOut v; v.outT = S; v.col0 = 0; v.ld = Nk; v.f32 = 1; v.rpb = Nk;      // outT[q][key] = scale * sum_d K[key][d] Q[q][d]
      CK(op_gemm(c, k + (size_t)b * Nk * ldk + k_off + h * Dp, ldk, Nk, Dp, q + (size_t)b * Nq * ldq + q_off + h * Dp, ldq, Nq, nullptr,
                 nullptr, 0, nullptr, Nq, scale, &v));
    }
  CK(launch_softmax_rows_f32(c->attn_buf, (size_t)B * heads * Nq, Nk, c->st));
  CKH(hipStreamSynchronize(c->st));                 // the callback is host code: it sees finished probabilities
  if (c->attn_cb(c->attn_cb_user, c->attn_buf, B, heads, Nq, Nk, is_cross, place, layer)) return fail(c, PNPI_ESTATE, "attention callback failed");
  for (int b = 0; b < B; ++b)
    for (int h = 0; h < heads; ++h) {
      const float* S = c->attn_buf + ((size_t)b * heads + h) * Nq * Nk;
      CK(launch_f32_rows_to_f16_padded(S, Nq, Nk, ldp, p16, c->st));
      CK(op_gemm(c, p16, ldp, Nq, ldp, vt + ((size_t)b * heads + h) * Dp * (size_t)ldv, ldv, dh, nullptr, nullptr, 0,
                 o + (size_t)b * Nq * ldo + h * dh, ldo));
    }
  return 0;
}

// SpatialTransformer + BasicTransformerBlock (my_diffusers/models/attention.py:140-200) with the hooked attention of
// models/p2p/attention_control.py:20-47 and the controller semantics of :178-190, :269-282 fused into the kernels.
static int transformer_fwd(pnpi_ctx* c, const TransformerW& t, const half_t* x, int B, int H, int W, const half_t* ctx16,
                           bool use_ctrl, int cur_step, half_t* out, Stats sx = Stats(), Stats* so = nullptr) {
  const pnpi_model_config& g = c->cfg;
  const size_t mk = c->temp.mark();
  const int block_index = c->tf_index++;    // transformer blocks in execution order (down 0.., mid, up ..15)
  const int C = t.C, N = H * W, M = B * N, hd = t.heads * t.Dp, X = g.cross_dim, T = g.ctx_len;
  const float scale = 1.0f / sqrtf((float)t.dh);
  CtrlDev& cd = c->cd;
  const bool edit = use_ctrl && cd.any_edit;

  half_t* g0 = talloc(c, (size_t)M * C);
  CK(op_gn(c, x, nullptr, C, 0, B, N, t.gn, g.norm_groups, 1e-6f, 0, g0, sx));
  half_t* hs = talloc(c, (size_t)M * C);
  CK(op_conv(c, g0, C, nullptr, 0, B, H, W, t.proj_in, 1, 0, 0, t.proj_in.b, nullptr, hs, H, W));

  // ---- self-attention
  half_t* n1 = talloc(c, (size_t)M * C);
  if (!c->dry) PROF(PNPI_KC_LAYERNORM, 0.0, 2.0 * M * (double)C * 2.0, launch_layernorm(hs, M, C, 1e-5f, t.ln1.g, t.ln1.b, n1, c->st));
  half_t* qk = talloc(c, (size_t)M * 2 * hd);
  const int ldv = round_up_i(N, 8);
  half_t* vt = talloc(c, (size_t)B * hd * ldv);
  // call-back path: P V runs over the padded key count, so the pad columns of V^T must be zeros -- cleared BEFORE the projection fills
  // the real columns (only the 2 x 2 level of the narrow test configurations has a token count that is not a multiple of 8)
  if (c->attn_cb && !c->dry && ldv != N) CKH(hipMemsetAsync(vt, 0, (size_t)B * hd * ldv * sizeof(half_t), c->st));
  // the 4096-token sites run the 64-wide LDS-DMA kernel, which reads V^T in the permuted key order (one ds_read_b128 per P V fragment):
  // the projection's epilogue writes it that way (not under a host callback: the materialised path reads plain V^T)
  const int vperm = (g_vt_perm && !c->attn_cb && N % 16 == 0 && attn_flash_uses_dma64(t.Dp, N, 0)) ? 1 : 0;
  {
    VtOut v; v.outT = vt; v.col0 = 2 * hd; v.ld = ldv; v.f32 = 0; v.rpb = N; v.perm16 = vperm;
    CK(op_gemm(c, n1, C, M, C, t.w_qkv, C, 3 * hd, t.b_qkv_aug, nullptr, 0, qk, 2 * hd, 1.f, &v, 2.0 * M * 3.0 * C * C));
  }
  half_t* ao = talloc(c, (size_t)M * C);
  {
    AttnP a; a.q = qk; a.ldq = 2 * hd; a.q_off = 0; a.k = qk; a.ldk = 2 * hd; a.k_off = hd; a.vt = vt; a.ldv = ldv; a.vt_perm = vperm;
    a.aug = (t.b_qkv_aug && g_attn_aug) ? 1 : 0;      // K / V column dh hold 1.0 (the projection above added b_qkv_aug)
    a.o = ao; a.ldo = C; a.heads = t.heads; a.Nq = N; a.Nk = N; a.Dp = t.Dp; a.dh = t.dh; a.scale = scale;
    const bool rep = edit && cur_step >= cd.self_lo && cur_step < cd.self_hi && N <= cd.self_max_tokens;
    const bool masa_step = cd.masa_step_list ? (cur_step >= 0 && cur_step < (int)cd.masa_step_on.size() && cd.masa_step_on[cur_step]) : cur_step >= cd.masa_start_step;
    const bool masa_layer = cd.masa_layer_mask ? ((cd.masa_layer_mask >> block_index) & 1u) != 0 && block_index < 31 : block_index >= cd.masa_start_layer;
    const bool masa = use_ctrl && cd.masa_any && masa_step && masa_layer;
    a.rows = rep ? cd.rows_rep : (masa ? cd.rows_masa : cd.rows_id); a.nrows = B;
    c->ctr.executed_attn_flops += 4.0 * B * t.heads * (double)N * N * t.Dp;
    if (c->attn_cb && !c->dry) {
      CKP(attn_materialized(c, qk, 2 * hd, 0, qk, 2 * hd, hd, vt, ldv, ao, C, t.heads, N, N, t.Dp, t.dh, scale, B, 0, t.place, 2 * block_index));
    } else if (!c->dry) PROFD(PNPI_KC_ATTN_FLASH, 4.0 * B * t.heads * (double)N * N * t.dh, 0.0, N, N, t.Dp, launch_attn_flash(a, c->st));
  }
  half_t* hs1 = talloc(c, (size_t)M * C);
  CK(op_gemm(c, ao, C, M, C, t.o1.w, C, C, t.o1.b, hs, C, hs1, C));

  // ---- cross-attention
  half_t* n2 = talloc(c, (size_t)M * C);
  if (!c->dry) PROF(PNPI_KC_LAYERNORM, 0.0, 2.0 * M * (double)C * 2.0, launch_layernorm(hs1, M, C, 1e-5f, t.ln2.g, t.ln2.b, n2, c->st));
  half_t* q2 = talloc(c, (size_t)M * hd);
  CK(op_gemm(c, n2, C, M, C, t.w_q2, C, hd, nullptr, nullptr, 0, q2, hd, 1.f, nullptr, 2.0 * M * (double)C * C));
  const int ldv2 = round_up_i(T, 8);
  half_t *k2, *vt2;
  if (c->tkv.use) {     // projected once per loop by text_kv_precompute
    k2 = c->tkv.k[block_index]; vt2 = c->tkv.vt[block_index];
  } else {
    k2 = talloc(c, (size_t)B * T * hd);
    vt2 = talloc(c, (size_t)B * hd * ldv2);
    // call-back path: the P V product runs over the padded key count, so the pad columns of V^T must be zeros (not stale arena bytes)
    if (c->attn_cb && !c->dry && ldv2 != T) CKH(hipMemsetAsync(vt2, 0, (size_t)B * hd * ldv2 * sizeof(half_t), c->st));
    VtOut v; v.outT = vt2; v.col0 = hd; v.ld = ldv2; v.f32 = 0; v.rpb = T;
    CK(op_gemm(c, ctx16, X, B * T, X, t.w_kv2, X, 2 * hd, nullptr, nullptr, 0, k2, hd, 1.f, &v, 2.0 * B * T * 2.0 * C * X));
  }
  half_t* ao2 = talloc(c, (size_t)M * C);
  {
    AttnP a; a.q = q2; a.ldq = hd; a.q_off = 0; a.k = k2; a.ldk = hd; a.k_off = 0; a.vt = vt2; a.ldv = ldv2;
    a.o = ao2; a.ldo = C; a.heads = t.heads; a.Nq = N; a.Nk = T; a.Dp = t.Dp; a.dh = t.dh; a.scale = scale;
    a.rows = edit ? cd.rows_plain : cd.rows_id; a.nrows = edit ? cd.n_plain : B;
    c->ctr.executed_attn_flops += 4.0 * B * t.heads * (double)N * 96 * t.Dp;
    if (c->attn_cb && !c->dry) {
      CKP(attn_materialized(c, q2, hd, 0, k2, hd, 0, vt2, ldv2, ao2, C, t.heads, N, T, t.Dp, t.dh, scale, B, 1, t.place, 2 * block_index + 1));
    } else if (!c->dry) PROFD(PNPI_KC_ATTN_FLASH, 4.0 * a.nrows * t.heads * (double)N * T * t.dh, 0.0, N, T, t.Dp, launch_attn_flash(a, c->st));
    if (edit && !c->dry && !c->attn_cb) {
      CrossEditP e; e.q = q2; e.ldq = hd; e.q_off = 0; e.k = k2; e.ldk = hd; e.k_off = 0; e.vt = vt2; e.ldv = ldv2;
      e.o = ao2; e.ldo = C; e.heads = t.heads; e.Nq = N; e.Nk = T; e.Dp = t.Dp; e.dh = t.dh; e.scale = scale;
      e.pairs = cd.pairs; e.npairs = cd.npairs; e.mmatT = cd.mmatT;
      int srow = cur_step < cd.n_alpha_rows ? cur_step : cd.n_alpha_rows - 1;
      e.c1 = cd.coef + ((size_t)srow * 2 + 0) * cd.npairs * 96;
      e.c2 = cd.coef + ((size_t)srow * 2 + 1) * cd.npairs * 96;
      const bool lb = cd.lb_any && t.lb_slot0 >= 0 && N == c->unet.lb_tokens;
      e.lb_alpha = lb ? cd.lb_alpha : nullptr;
      e.lb_acc = lb ? cd.lb_acc : nullptr;
      e.lb_slot0 = t.lb_slot0; e.lb_nslots = c->unet.lb_nslots; e.lb_planes = cd.lb_planes; e.write_src = 0;
      PROF(PNPI_KC_ATTN_EDIT, 4.0 * 2 * cd.npairs * t.heads * (double)N * T * t.dh, 0.0, launch_attn_cross_edit(e, c->st));
    }
  }
  half_t* hs2 = talloc(c, (size_t)M * C);
  CK(op_gemm(c, ao2, C, M, C, t.o2.w, C, C, t.o2.b, hs1, C, hs2, C));

  // ---- GEGLU feed-forward
  half_t* n3 = talloc(c, (size_t)M * C);
  if (!c->dry) PROF(PNPI_KC_LAYERNORM, 0.0, 2.0 * M * (double)C * 2.0, launch_layernorm(hs2, M, C, 1e-5f, t.ln3.g, t.ln3.b, n3, c->st));
  half_t* f2 = talloc(c, (size_t)M * 4 * C);
  if (C % 64 == 0) {
    // ff1 GEMM with the GEGLU product in its epilogue: [M][8C] never exists in memory
    CK(op_gemm(c, n3, C, M, C, t.ff1.w, C, 8 * C, t.ff1.b, nullptr, 0, f2, 4 * C, 1.f, nullptr, -1.0, 1));
  } else {
    // narrow test configurations: the interleaved projection is materialised and combined by a small kernel
    half_t* f1 = talloc(c, (size_t)M * 8 * C);
    CK(op_gemm(c, n3, C, M, C, t.ff1.w, C, 8 * C, t.ff1.b, nullptr, 0, f1, 8 * C));
    if (!c->dry) PROF(PNPI_KC_GEGLU, 0.0, 12.0 * M * (double)C * 2.0, launch_geglu(f1, M, 4 * C, f2, c->st));
  }
  half_t* hs3 = talloc(c, (size_t)M * C);
  CK(op_gemm(c, f2, 4 * C, M, 4 * C, t.ff2.w, 4 * C, C, t.ff2.b, hs2, C, hs3, C));
  StatsReq qo;
  CK(op_conv(c, hs3, C, nullptr, 0, B, H, W, t.proj_out, 1, 0, 0, t.proj_out.b, x, out, H, W, -1, nullptr, &qo));
  if (so) { so->p = qo.buf; so->rows = qo.rows; }
  c->temp.release(mk);
  return 0;
}

// The same block while a tape is recording (null-text path: one prompt row, no controller): every intermediate the backward needs is
// kept in its plain layout -- q | k | v as one row-major projection output (V^T for the flash kernel is a transposed COPY), the text
// K / V projected inside the forward from the context (their gradient is the point of the exercise), GEGLU unfused.
static int op_ln(pnpi_ctx* c, const half_t* x, int M, int C, const NormW& nw, half_t* out) {
  if (c->dry) return 0;
  if (taping(c)) { TapeOp o; o.kind = TK_LN; o.x1 = x; o.M = M; o.C1 = C; o.nw = &nw; o.eps = 1e-5f; o.out = out; c->tape->ops.push_back(o); }
  PROF(PNPI_KC_LAYERNORM, 0.0, 2.0 * M * (double)C * 2.0, launch_layernorm(x, M, C, 1e-5f, nw.g, nw.b, out, c->st));
  return 0;
}
static int transformer_fwd_tape(pnpi_ctx* c, const TransformerW& t, const half_t* x, int B, int H, int W, const half_t* ctx16, half_t* out) {
  const pnpi_model_config& g = c->cfg;
  c->tf_index++;
  const int C = t.C, N = H * W, M = B * N, hd = t.heads * t.Dp, X = g.cross_dim, T = g.ctx_len;
  const float scale = 1.0f / sqrtf((float)t.dh);
  half_t* g0 = talloc(c, (size_t)M * C);
  CK(op_gn(c, x, nullptr, C, 0, B, N, t.gn, g.norm_groups, 1e-6f, 0, g0));
  half_t* hs = talloc(c, (size_t)M * C);
  CK(op_conv(c, g0, C, nullptr, 0, B, H, W, t.proj_in, 1, 0, 0, t.proj_in.b, nullptr, hs, H, W));
  // ---- self-attention
  half_t* n1 = talloc(c, (size_t)M * C);
  CK(op_ln(c, hs, M, C, t.ln1, n1));
  half_t* qkv = talloc(c, (size_t)M * 3 * hd);
  CK(op_gemm(c, n1, C, M, C, t.w_qkv, C, 3 * hd, nullptr, nullptr, 0, qkv, 3 * hd));
  const int ldv = round_up_i(N, 8);
  half_t* vt = talloc(c, (size_t)B * hd * ldv);
  half_t* ao = talloc(c, (size_t)M * C);
  float* lse1 = (float*)talloc(c, (size_t)B * t.heads * N * 2);       // per-query log-sum-exp for the backward kernel
  if (!c->dry) {
    for (int b = 0; b < B; ++b) CK(launch_transpose_f16(qkv + (size_t)b * N * 3 * hd + 2 * hd, 3 * hd, N, hd, vt + (size_t)b * hd * ldv, ldv, c->st));
    AttnP a; a.q = qkv; a.ldq = 3 * hd; a.q_off = 0; a.k = qkv; a.ldk = 3 * hd; a.k_off = hd; a.vt = vt; a.ldv = ldv;
    a.o = ao; a.ldo = C; a.heads = t.heads; a.Nq = N; a.Nk = N; a.Dp = t.Dp; a.dh = t.dh; a.scale = scale; a.rows = c->cd.rows_id; a.nrows = B;
    a.lse = lse1;
    CK(launch_attn_flash(a, c->st));
    c->ctr.executed_attn_flops += 4.0 * B * t.heads * (double)N * N * t.Dp;
    if (taping(c)) {
      TapeOp o; o.kind = TK_ATTN; o.q = qkv; o.ldq = 3 * hd; o.q_off = 0; o.k = qkv; o.ldk = 3 * hd; o.k_off = hd; o.v = qkv; o.ldv = 3 * hd; o.v_off = 2 * hd;
      o.out = ao; o.ldo = C; o.heads = t.heads; o.Nq = N; o.Nk = N; o.Dp = t.Dp; o.dh = t.dh; o.scale = scale; o.B = B; o.lse = lse1;
      c->tape->ops.push_back(o);
    }
  }
  half_t* hs1 = talloc(c, (size_t)M * C);
  CK(op_gemm(c, ao, C, M, C, t.o1.w, C, C, t.o1.b, hs, C, hs1, C));
  // ---- cross-attention
  half_t* n2 = talloc(c, (size_t)M * C);
  CK(op_ln(c, hs1, M, C, t.ln2, n2));
  half_t* q2 = talloc(c, (size_t)M * hd);
  CK(op_gemm(c, n2, C, M, C, t.w_q2, C, hd, nullptr, nullptr, 0, q2, hd));
  half_t* kv2 = talloc(c, (size_t)B * T * 2 * hd);
  CK(op_gemm(c, ctx16, X, B * T, X, t.w_kv2, X, 2 * hd, nullptr, nullptr, 0, kv2, 2 * hd));
  const int ldv2 = round_up_i(T, 8);
  half_t* vt2 = talloc(c, (size_t)B * hd * ldv2);
  half_t* ao2 = talloc(c, (size_t)M * C);
  float* lse2 = (float*)talloc(c, (size_t)B * t.heads * N * 2);
  if (!c->dry) {
    for (int b = 0; b < B; ++b) CK(launch_transpose_f16(kv2 + (size_t)b * T * 2 * hd + hd, 2 * hd, T, hd, vt2 + (size_t)b * hd * ldv2, ldv2, c->st));
    AttnP a; a.q = q2; a.ldq = hd; a.q_off = 0; a.k = kv2; a.ldk = 2 * hd; a.k_off = 0; a.vt = vt2; a.ldv = ldv2;
    a.o = ao2; a.ldo = C; a.heads = t.heads; a.Nq = N; a.Nk = T; a.Dp = t.Dp; a.dh = t.dh; a.scale = scale; a.rows = c->cd.rows_id; a.nrows = B;
    a.lse = lse2;
    CK(launch_attn_flash(a, c->st));
    c->ctr.executed_attn_flops += 4.0 * B * t.heads * (double)N * 96 * t.Dp;
    if (taping(c)) {
      TapeOp o; o.kind = TK_ATTN; o.q = q2; o.ldq = hd; o.q_off = 0; o.k = kv2; o.ldk = 2 * hd; o.k_off = 0; o.v = kv2; o.ldv = 2 * hd; o.v_off = hd;
      o.out = ao2; o.ldo = C; o.heads = t.heads; o.Nq = N; o.Nk = T; o.Dp = t.Dp; o.dh = t.dh; o.scale = scale; o.B = B; o.lse = lse2;
      c->tape->ops.push_back(o);
    }
  }
  half_t* hs2 = talloc(c, (size_t)M * C);
  CK(op_gemm(c, ao2, C, M, C, t.o2.w, C, C, t.o2.b, hs1, C, hs2, C));
  // ---- GEGLU feed-forward (unfused: the gate pre-activation is needed for its derivative)
  half_t* n3 = talloc(c, (size_t)M * C);
  CK(op_ln(c, hs2, M, C, t.ln3, n3));
  half_t* f1 = talloc(c, (size_t)M * 8 * C);
  CK(op_gemm(c, n3, C, M, C, t.ff1.w, C, 8 * C, t.ff1.b, nullptr, 0, f1, 8 * C));
  half_t* f2 = talloc(c, (size_t)M * 4 * C);
  if (!c->dry) {
    CK(launch_geglu(f1, M, 4 * C, f2, c->st));
    if (taping(c)) { TapeOp o; o.kind = TK_GEGLU; o.x1 = f1; o.out = f2; o.M = M; o.N = 4 * C; c->tape->ops.push_back(o); }
  }
  half_t* hs3 = talloc(c, (size_t)M * C);
  CK(op_gemm(c, f2, 4 * C, M, 4 * C, t.ff2.w, 4 * C, C, t.ff2.b, hs2, C, hs3, C));
  CK(op_conv(c, hs3, C, nullptr, 0, B, H, W, t.proj_out, 1, 0, 0, t.proj_out.b, x, out, H, W));
  return 0;
}

// UNet2DConditionModel.forward (my_diffusers/models/unet_2d_condition.py:189-273)
static int unet_fwd(pnpi_ctx* c, const float* latents, int rows, int t, const float* context, bool use_ctrl, int cur_step,
                    float* eps_out) {
  const pnpi_model_config& g = c->cfg;
  const UNetW& u = c->unet;
  if (rows <= 0 || rows > c->max_rows) return fail(c, PNPI_EINVAL, "unet rows out of range (max_unet_rows)");
  if (t < 0 || t >= g.n_train_timesteps) return fail(c, PNPI_EINVAL, "timestep out of range");
  c->persist.reset(); c->temp.reset();
  c->tf_index = 0;
  const int S = g.sample_size, n = g.n_blocks, C0 = g.block_out_channels[0], TE = 4 * C0, G = g.norm_groups;
  const int B = rows;
  const float eps = 1e-5f;
  half_t* x0 = palloc(c, (size_t)B * S * S * 8);
  half_t* ctx16 = nullptr;
  if (c->tkv.use && !c->dry) {
    if (c->tkv.rows != B) return fail(c, PNPI_ESTATE, "text K/V cache holds a different row count than this forward");
  } else {
    ctx16 = palloc(c, (size_t)B * g.ctx_len * g.cross_dim);
  }
  if (taping(c)) { c->tape->no_grad_input = x0; c->tape->ctx16 = ctx16; }
  if (!c->dry) {
    CK(launch_nchw_f32_to_nhwc_f16(latents, B, g.in_channels, S * S, 8, x0, c->st));
    if (ctx16) {
      if (!context) return fail(c, PNPI_EINVAL, "context is NULL and no text K/V cache is active");
      CK(launch_f32_to_f16(context, (size_t)B * g.ctx_len * g.cross_dim, ctx16, c->st));
    }
    // conv1 bias + time embedding of every ResNet for this timestep (TimestepEmbedding + the 22 time_emb_proj linears as three
    // GEMVs): a function of t and the weights only -> computed on the first forward at t, then read from the table
    if (c->bias_tab && g_temb_cache) {
      float* row = c->bias_tab + (size_t)t * u.temb_total;
      if (!c->bias_valid[t]) {
        CK(launch_gemv(c->temb_table + (size_t)t * C0, C0, u.t1.w, TE, u.t1.b, nullptr, 0, c->temb_h, c->st));
        CK(launch_gemv(c->temb_h, TE, u.t2.w, TE, u.t2.b, nullptr, 1, c->temb_emb, c->st));
        CK(launch_gemv(c->temb_emb, TE, u.temb_w, u.temb_total, u.temb_b, u.conv1_b, 1, row, c->st));
        c->bias_valid[t] = 1;
      }
      c->bias_eff = row;
    } else {
      CK(launch_gemv(c->temb_table + (size_t)t * C0, C0, u.t1.w, TE, u.t1.b, nullptr, 0, c->temb_h, c->st));
      CK(launch_gemv(c->temb_h, TE, u.t2.w, TE, u.t2.b, nullptr, 1, c->temb_emb, c->st));
      CK(launch_gemv(c->temb_emb, TE, u.temb_w, u.temb_total, u.temb_b, u.conv1_b, 1, c->bias_scratch, c->st));
      c->bias_eff = c->bias_scratch;
    }
  }
  struct Act { half_t* p; int C, H; Stats s; };
  std::vector<Act> skips;
  int H = S;
  half_t* h = palloc(c, (size_t)B * H * H * C0);
  Stats hs_;   // GroupNorm partial sums travelling with h
  {
    StatsReq q;
    CK(op_conv(c, x0, 8, nullptr, 0, B, H, H, u.conv_in, 1, 1, 0, u.conv_in.b, nullptr, h, H, H, -1, nullptr, &q));
    hs_.p = q.buf; hs_.rows = q.rows;
  }
  int ch = C0;
  skips.push_back({h, ch, H, hs_});
  for (int i = 0; i < n; ++i) {
    const int oc = g.block_out_channels[i];
    for (int j = 0; j < g.layers_per_block; ++j) {
      half_t* o = palloc(c, (size_t)B * H * H * oc);
      Stats ns;
      CKP(resnet_fwd(c, u.down_res[i][j], h, ch, nullptr, 0, B, H, H, G, eps, o, hs_, Stats(), &ns));
      h = o; ch = oc; hs_ = ns;
      if (g.block_has_attn[i]) {
        half_t* o2 = palloc(c, (size_t)B * H * H * oc);
        if (keep_acts(c)) { CKP(transformer_fwd_tape(c, u.down_attn[i][j], h, B, H, H, ctx16, o2)); ns = Stats(); }
        else CKP(transformer_fwd(c, u.down_attn[i][j], h, B, H, H, ctx16, use_ctrl, cur_step, o2, hs_, &ns));
        h = o2; hs_ = ns;
      }
      skips.push_back({h, ch, H, hs_});
    }
    if (i != n - 1) {
      const int Ho = H / 2;
      half_t* o = palloc(c, (size_t)B * Ho * Ho * oc);
      StatsReq q;
      CK(op_conv(c, h, ch, nullptr, 0, B, H, H, u.down_samp[i], 2, 1, 0, u.down_samp[i].b, nullptr, o, Ho, Ho, -1, nullptr, &q));
      h = o; H = Ho; hs_.p = q.buf; hs_.rows = q.rows;
      skips.push_back({h, ch, H, hs_});
    }
  }
  {
    half_t* o = palloc(c, (size_t)B * H * H * ch);
    Stats n1, n2, n3;
    CKP(resnet_fwd(c, u.mid_res[0], h, ch, nullptr, 0, B, H, H, G, eps, o, hs_, Stats(), &n1));
    half_t* o2 = palloc(c, (size_t)B * H * H * ch);
    if (keep_acts(c)) { CKP(transformer_fwd_tape(c, u.mid_attn, o, B, H, H, ctx16, o2)); n2 = Stats(); }
    else CKP(transformer_fwd(c, u.mid_attn, o, B, H, H, ctx16, use_ctrl, cur_step, o2, n1, &n2));
    half_t* o3 = palloc(c, (size_t)B * H * H * ch);
    CKP(resnet_fwd(c, u.mid_res[1], o2, ch, nullptr, 0, B, H, H, G, eps, o3, n2, Stats(), &n3));
    h = o3; hs_ = n3;
  }
  for (int i = 0; i < n; ++i) {
    const int oc = g.block_out_channels[n - 1 - i];
    for (int j = 0; j <= g.layers_per_block; ++j) {
      Act s = skips.back(); skips.pop_back();
      half_t* o = palloc(c, (size_t)B * H * H * oc);
      Stats ns;
      CKP(resnet_fwd(c, u.up_res[i][j], h, ch, s.p, s.C, B, H, H, G, eps, o, hs_, s.s, &ns));
      h = o; ch = oc; hs_ = ns;
      if (g.block_has_attn[n - 1 - i]) {
        half_t* o2 = palloc(c, (size_t)B * H * H * oc);
        if (keep_acts(c)) { CKP(transformer_fwd_tape(c, u.up_attn[i][j], h, B, H, H, ctx16, o2)); ns = Stats(); }
        else CKP(transformer_fwd(c, u.up_attn[i][j], h, B, H, H, ctx16, use_ctrl, cur_step, o2, hs_, &ns));
        h = o2; hs_ = ns;
      }
    }
    if (i != n - 1) {
      const int Ho = H * 2;
      half_t* o = palloc(c, (size_t)B * Ho * Ho * oc);
      StatsReq q;
      CK(op_conv(c, h, ch, nullptr, 0, B, H, H, u.up_samp[i], 1, 1, 1, u.up_samp[i].b, nullptr, o, Ho, Ho, -1, nullptr, &q));
      h = o; H = Ho; hs_.p = q.buf; hs_.rows = q.rows;
    }
  }
  half_t* gno = palloc(c, (size_t)B * H * H * ch);
  CK(op_gn(c, h, nullptr, ch, 0, B, H * H, u.norm_out, G, eps, 1, gno, hs_));
  {
    VtOut v; v.outT = eps_out; v.col0 = 0; v.ld = H * H; v.f32 = 1; v.rpb = H * H;
    CK(op_conv(c, gno, ch, nullptr, 0, B, H, H, u.conv_out, 1, 1, 0, u.conv_out.b, nullptr, nullptr, H, H, -1, &v));
  }
  c->ctr.unet_calls += c->dry ? 0 : 1;
  c->ctr.unet_sample_forwards += c->dry ? 0 : rows;
  c->ctr.unet_sample_forwards_cached_kv += (c->dry || !c->tkv.use) ? 0 : rows;
  if (c->persist.overflow || c->temp.overflow) return fail(c, PNPI_ENOMEM, "workspace overflow");
  return 0;
}

// Attention backward, materialised per (row, head) like the call-back path's forward (first version: generic GEMM launches and
// transposes; a fused flash backward replaces it later).  q / k: [B*N][ld] views with the head's columns at off + h * Dp (pad columns
// dh .. Dp are zero, as the forward guarantees); v: plain [B*Nk][ldvp] with the same head layout; d_o: [B*Nq][ldo], heads * dh wide.
// Outputs dq / dk / dv in the layout of q / k / v (their pad columns are left untouched: clear the buffers first).
//   S = scale q k^T, P = softmax(S);  dV = P^T dO;  dP = dO V^T;  dS = scale P (dP - rowsum(dP P));  dQ = dS K;  dK = dS^T Q.
// scratch: Nq * Nk * 8 (S / P and dP, fp32) + 2 * Nq * ldp * 2 (P, dS as fp16) + 2 * Nk * ldq8 * 2 (their transposes) + dh * (ldp + 2 * ldq8) * 2
// bytes (K^T, Q^T, dO^T), reused for every (row, head).
static size_t attn_bwd_scratch_bytes(int Nq, int Nk, int dh) {
  const size_t ldp = round_up_i(Nk, 8), ldq8 = round_up_i(Nq, 8);
  return align_up((size_t)Nq * Nk * 4, 256) * 2 + align_up((size_t)Nq * ldp * 2, 256) * 2 + align_up((size_t)Nk * ldq8 * 2, 256) * 2 +
         align_up((size_t)dh * ldp * 2, 256) + align_up((size_t)dh * ldq8 * 2, 256) * 2;
}
static int attn_bwd_materialized(pnpi_ctx* c, const half_t* q, int ldq, int q_off, const half_t* k, int ldk, int k_off, const half_t* v, int ldvp,
                                 int v_off, const half_t* d_o, int ldo, int heads, int Nq, int Nk, int Dp, int dh, float scale, int B,
                                 half_t* dq, half_t* dk, half_t* dv, void* scratch, size_t scratch_bytes) {
  const size_t per = attn_bwd_scratch_bytes(Nq, Nk, dh);
  if (!scratch || scratch_bytes < per) return fail(c, PNPI_ENOMEM, "attention backward scratch too small");
  if ((dh & 7) || (ldq & 7) || (ldk & 7) || (ldvp & 7) || (ldo & 7)) return fail(c, PNPI_ESHAPE, "attention backward: extents must be multiples of 8");
  const int ldp = round_up_i(Nk, 8), ldq8 = round_up_i(Nq, 8);
  // The heads of a row are independent problems of one shape: every step below is ONE launch over a group of `nb` heads (as many as
  // the scratch holds; all of them with the tape's own scratch) -- a batched GEMM (grid z = head) or a row-wise kernel over nb * Nq rows.
  // Head by head the same work was 13 launches of a few microseconds each per head: 3 300 launches per reverse walk.
  const int nb_max = (int)std::min<size_t>((size_t)heads, scratch_bytes / per);
  const size_t szS = align_up((size_t)Nq * Nk * 4, 256), szP = align_up((size_t)Nq * ldp * 2, 256), szT = align_up((size_t)Nk * ldq8 * 2, 256),
               szK = align_up((size_t)dh * ldp * 2, 256), szQ = align_up((size_t)dh * ldq8 * 2, 256);
  for (int b = 0; b < B; ++b)
    for (int h0 = 0; h0 < heads; h0 += nb_max) {
      const int nb = std::min(nb_max, heads - h0);
      char* sp = (char*)scratch;
      auto take = [&](size_t bytes_each) { char* r = sp; sp += bytes_each * nb; return r; };     // [nb] consecutive per-head buffers
      float* S = (float*)take(szS);
      float* dP = (float*)take(szS);
      half_t* P16 = (half_t*)take(szP);
      half_t* dS16 = (half_t*)take(szP);
      half_t* PT = (half_t*)take(szT);
      half_t* dST = (half_t*)take(szT);
      half_t* Kt = (half_t*)take(szK);
      half_t* Qt = (half_t*)take(szQ);
      half_t* dOt = (half_t*)take(szQ);
      const long eS = (long)(szS / 4), eP = (long)(szP / 2), eT = (long)(szT / 2), eK = (long)(szK / 2), eQ = (long)(szQ / 2);   // per-head strides in elements
      const half_t* qh = q + (size_t)b * Nq * ldq + q_off + h0 * Dp;
      const half_t* kh = k + (size_t)b * Nk * ldk + k_off + h0 * Dp;
      const half_t* vh = v + (size_t)b * Nk * ldvp + v_off + h0 * Dp;
      const half_t* doh = d_o + (size_t)b * Nq * ldo + h0 * dh;
      GemmBatch gb; gb.n = nb;
      VtOut vs; vs.outT = S; vs.col0 = 0; vs.ld = Nk; vs.f32 = 1; vs.rpb = Nk;            // S[q][key] = scale * sum_d k[key][d] q[q][d]
      gb.sa = Dp; gb.sw = Dp; gb.sout = 0; gb.soutT = (long)szS;
      CK(op_gemm(c, kh, ldk, Nk, dh, qh, ldq, Nq, nullptr, nullptr, 0, nullptr, Nq, scale, &vs, -1.0, 0, &gb));
      VtOut vp; vp.outT = dP; vp.col0 = 0; vp.ld = Nk; vp.f32 = 1; vp.rpb = Nk;          // dP[q][key] = sum_d v[key][d] dO[q][d]
      gb.sa = Dp; gb.sw = dh;
      CK(op_gemm(c, vh, ldvp, Nk, dh, doh, ldo, Nq, nullptr, nullptr, 0, nullptr, Nq, 1.f, &vp, -1.0, 0, &gb));
      if (szS == (size_t)Nq * Nk * 4 && szP == (size_t)Nq * ldp * 2) {
        // the per-head buffers are dense: the row-wise kernels take all nb * Nq rows at once
        CK(launch_softmax_rows_f32(S, (size_t)nb * Nq, Nk, c->st));
        CK(launch_f32_rows_to_f16_padded(S, (size_t)nb * Nq, Nk, ldp, P16, c->st));
        CK(launch_softmax_bwd_rows(S, dP, (size_t)nb * Nq, Nk, ldp, scale, dS16, c->st));
      } else {
        for (int i = 0; i < nb; ++i) {
          CK(launch_softmax_rows_f32(S + i * eS, (size_t)Nq, Nk, c->st));
          CK(launch_f32_rows_to_f16_padded(S + i * eS, (size_t)Nq, Nk, ldp, P16 + i * eP, c->st));
          CK(launch_softmax_bwd_rows(S + i * eS, dP + i * eS, (size_t)Nq, Nk, ldp, scale, dS16 + i * eP, c->st));
        }
      }
      CK(launch_transpose_f16(P16, ldp, Nq, Nk, PT, ldq8, c->st, nb, eP, eT));
      CK(launch_transpose_f16(dS16, ldp, Nq, Nk, dST, ldq8, c->st, nb, eP, eT));
      CK(launch_transpose_f16(kh, ldk, Nk, dh, Kt, ldp, c->st, nb, Dp, eK));
      CK(launch_transpose_f16(qh, ldq, Nq, dh, Qt, ldq8, c->st, nb, Dp, eQ));
      CK(launch_transpose_f16(doh, ldo, Nq, dh, dOt, ldq8, c->st, nb, dh, eQ));
      gb.soutT = 0; gb.sout = Dp;
      gb.sa = eP; gb.sw = eK;
      CK(op_gemm(c, dS16, ldp, Nq, ldp, Kt, ldp, dh, nullptr, nullptr, 0, dq + (size_t)b * Nq * ldq + q_off + h0 * Dp, ldq, 1.f, nullptr, -1.0, 0, &gb));     // dQ = dS K
      gb.sa = eT; gb.sw = eQ;
      CK(op_gemm(c, dST, ldq8, Nk, ldq8, Qt, ldq8, dh, nullptr, nullptr, 0, dk + (size_t)b * Nk * ldk + k_off + h0 * Dp, ldk, 1.f, nullptr, -1.0, 0, &gb));   // dK = dS^T Q
      CK(op_gemm(c, PT, ldq8, Nk, ldq8, dOt, ldq8, dh, nullptr, nullptr, 0, dv + (size_t)b * Nk * ldvp + v_off + h0 * Dp, ldvp, 1.f, nullptr, -1.0, 0, &gb)); // dV = P^T dO
    }
  return 0;
}
// Attention backward in flash form (attn.hip: attn_bwd_flash_kernel): one launch each for dQ (which also leaves D, and the per-query
// log-sum-exp unless the forward did), dK and dV.  Workspace per batch row: 2 * heads * N floats (+ the fp32 partial sums of a split
// cross-attention walk) -- at the 64 x 64 level 260 KB against the 8.6 GB-per-8-heads of the score matrices.
// cross-attention (77 keys): dK / dV have one 128-row key tile per head, so the query walk is split over workgroups (fp32 partial sums,
// added in a fixed order by a small reduce launch)
static int attn_bwd_flash_nsplit(int Nq, int Nk) {
  if (Nk > 128 || Nq < 256) return 1;
  const int n = Nq / 128;
  return n > 32 ? 32 : n;
}
static size_t attn_bwd_flash_scratch_bytes(int heads, int Nq, int Nk, int dh) {
  const int ns = attn_bwd_flash_nsplit(Nq, Nk);
  return (size_t)heads * 2 * align_up((size_t)Nq * 4, 256) + (ns > 1 ? align_up((size_t)ns * heads * Nk * dh * 4, 256) : 0);
}
static bool attn_bwd_flash_shape(int Nq, int Nk, int Dp, int dh) {
  // tuning "attn_bwd_flash": 0 = the materialised form everywhere, 1 = flash form for self- and cross-attention (dQ with the forward's
  // log-sum-exp where the tape has it), 2 = the same with the two-pass dQ always, 3 = self-attention only
  const bool self = Nq == Nk, cross = !self && Nk <= 128 && Nk >= 8;
  if (!g_attn_bwd_flash || !(self || (cross && g_attn_bwd_flash != 3))) return false;
  return Nq >= 64 && Nq % 64 == 0 && (Dp == 32 || Dp == 64 || Dp == 96 || Dp == 160) && dh <= Dp && !(dh & 7);
}
static bool attn_bwd_flash_ok(int heads, int Nq, int Nk, int Dp, int dh, size_t scratch_bytes) {
  return attn_bwd_flash_shape(Nq, Nk, Dp, dh) && scratch_bytes >= attn_bwd_flash_scratch_bytes(heads, Nq, Nk, dh);
}
static int attn_bwd_flash(pnpi_ctx* c, const half_t* q, int ldq, int q_off, const half_t* k, int ldk, int k_off, const half_t* v, int ldvp,
                          int v_off, const half_t* d_o, int ldo, int heads, int Nq, int Nk, int Dp, int dh, float scale, int B,
                          half_t* dq, half_t* dk, half_t* dv, void* scratch, const float* lse_fwd, const half_t* o_fwd, int ld_ofwd) {
  if ((ldq & 7) || (ldk & 7) || (ldvp & 7) || (ldo & 7) || (q_off & 7) || (k_off & 7) || (v_off & 7))
    return fail(c, PNPI_ESHAPE, "attention backward: extents must be multiples of 8");
  const size_t szF = align_up((size_t)Nq * 4, 256);
  char* sp = (char*)scratch;
  float* lse = (float*)sp; sp += szF * heads;
  float* dsum = (float*)sp; sp += szF * heads;
  float* part = (float*)sp;
  const int nsplit = attn_bwd_flash_nsplit(Nq, Nk);
  if (szF != (size_t)Nq * 4) return fail(c, PNPI_ESHAPE, "attention backward: Nq * 4 must be a multiple of 256");   // [heads][Nq] dense (Nq >= 64, % 64 == 0 in every caller)
  for (int b = 0; b < B; ++b) {
    const half_t* qh = q + (size_t)b * Nq * ldq + q_off;
    const half_t* kh = k + (size_t)b * Nk * ldk + k_off;
    const half_t* vh = v + (size_t)b * Nk * ldvp + v_off;
    const half_t* doh = d_o + (size_t)b * Nq * ldo;
    const BwdMat mq{qh, Dp, ldq, dh}, mk{kh, Dp, ldk, dh}, mv{vh, Dp, ldvp, dh}, mdo{doh, dh, ldo, dh};
    AttnBwdP a{};
    a.heads = heads; a.scale = scale; a.lse = lse; a.dsum = dsum; a.out_hs = Dp; a.out_w = dh;
    a.b1 = mq; a.b2 = mdo; a.l1 = mk; a.l2 = mv; a.nb = Nq; a.nl = Nk;
    a.out = dq + (size_t)b * Nq * ldq + q_off; a.out_ld = ldq;
    if (lse_fwd && o_fwd && !(ld_ofwd & 7)) {     // the recording forward left the log-sum-exp and O: dQ without its first pass over the keys
      a.lse = const_cast<float*>(lse_fwd) + (size_t)b * heads * Nq;       // read-only in this form
      a.o = o_fwd + (size_t)b * Nq * ld_ofwd; a.o_hs = dh; a.o_ld = ld_ofwd;
    }
    CK(launch_attn_bwd_flash(a, 0, Dp, c->st));
    // tile products of the three launches (dQ: S, dP, dQ -- twice S and dP without the forward's log-sum-exp; dK: S, dP, dK; dV: S, dV)
    c->ctr.executed_attn_flops += 2.0 * heads * (double)Nq * Nk * Dp * ((a.o ? 3 : 5) + 3 + 2);
    a.o = nullptr;
    a.b1 = mk; a.b2 = mv; a.l1 = mq; a.l2 = mdo; a.nb = Nk; a.nl = Nq;
    a.nsplit = nsplit; a.part = nsplit > 1 ? part : nullptr;
    a.out = dk + (size_t)b * Nk * ldk + k_off; a.out_ld = ldk;
    CK(launch_attn_bwd_flash(a, 1, Dp, c->st));
    if (nsplit > 1) CK(launch_attn_bwd_reduce(a, c->st));
    a.out = dv + (size_t)b * Nk * ldvp + v_off; a.out_ld = ldvp;
    CK(launch_attn_bwd_flash(a, 2, Dp, c->st));
    if (nsplit > 1) CK(launch_attn_bwd_reduce(a, c->st));
  }
  return 0;
}
// dq / dk / dv of one attention site: the flash form for self-attention, the materialised form otherwise
static int attn_bwd(pnpi_ctx* c, const half_t* q, int ldq, int q_off, const half_t* k, int ldk, int k_off, const half_t* v, int ldvp, int v_off,
                    const half_t* d_o, int ldo, int heads, int Nq, int Nk, int Dp, int dh, float scale, int B, half_t* dq, half_t* dk, half_t* dv,
                    void* scratch, size_t scratch_bytes, const float* lse_fwd = nullptr, const half_t* o_fwd = nullptr, int ld_ofwd = 0) {
  if (scratch && attn_bwd_flash_ok(heads, Nq, Nk, Dp, dh, scratch_bytes))
    return attn_bwd_flash(c, q, ldq, q_off, k, ldk, k_off, v, ldvp, v_off, d_o, ldo, heads, Nq, Nk, Dp, dh, scale, B, dq, dk, dv, scratch,
                          g_attn_bwd_flash == 2 ? nullptr : lse_fwd, o_fwd, ld_ofwd);      // tuning value 2: always the two-pass dQ
  return attn_bwd_materialized(c, q, ldq, q_off, k, ldk, k_off, v, ldvp, v_off, d_o, ldo, heads, Nq, Nk, Dp, dh, scale, B, dq, dk, dv, scratch, scratch_bytes);
}
static int gn_bwd_workspace(pnpi_ctx* c, int B, int HW, int G, float** out) {
  const size_t need = groupnorm_bwd2_scratch_floats(B, HW, G);
  if (need > c->gn_bwd_ws_floats) {
    CKH(hipStreamSynchronize(c->st));                      // a launch in flight may still read the old buffer
    if (c->gn_bwd_ws) CKH(hipFree(c->gn_bwd_ws));
    c->gn_bwd_ws = nullptr; c->gn_bwd_ws_floats = 0;
    CKH(hipMalloc((void**)&c->gn_bwd_ws, need * sizeof(float)));
    c->gn_bwd_ws_floats = need;
  }
  *out = c->gn_bwd_ws;
  return 0;
}
// ---------------------------------------------------------------------------------------------------- tape backward
static half_t* tape_galloc(pnpi_ctx* c, size_t n_halfs) {
  Tape& T = *c->tape;
  half_t* p = (half_t*)T.garena.alloc(n_halfs * sizeof(half_t));
  return T.garena.overflow ? nullptr : p;
}
// dst = the gradient buffer of activation `key` (n halfs).  First contribution: allocated, `fresh` = true, the producer writes it
// directly.  Later contributions: a scratch buffer is returned and tape_commit() adds it to the existing gradient.
struct GradDst { half_t* p = nullptr; half_t* into = nullptr; bool fresh = false; };
static int tape_target(pnpi_ctx* c, const void* key, size_t n, GradDst& d) {
  Tape& T = *c->tape;
  auto it = T.grads.find(key);
  d.p = tape_galloc(c, n);
  if (!d.p) return fail(c, PNPI_ENOMEM, "gradient arena overflow");
  if (it == T.grads.end()) { T.grads[key] = d.p; d.fresh = true; d.into = d.p; }
  else { d.fresh = false; d.into = it->second; }
  return 0;
}
static int tape_commit(pnpi_ctx* c, const GradDst& d, size_t n) {
  if (d.fresh) return 0;
  CK(launch_accumulate_f16(d.into, d.p, n, c->st));
  return 0;
}
// gradient of `key` += src[r * ld + off + 0 .. C) for r < R (dense destination [R][C])
static int tape_add_strided(pnpi_ctx* c, const void* key, const half_t* src, int ld, int off, size_t R, int C) {
  Tape& T = *c->tape;
  auto it = T.grads.find(key);
  if (it == T.grads.end()) {
    // First contribution from a dense buffer of this backward's own arena (a dgrad result, or the finished gradient of the op's output on
    // its way into a residual branch): the buffer BECOMES the gradient -- no copy.  Safe because the walk is in reverse order: the source
    // is either fresh or the gradient of an activation whose consumers have all been processed, and later contributions are stream-ordered
    // behind this op's own reads of it.
    const char* sb = (const char*)src;
    if (ld == C && off == 0 && !T.garena.overflow && sb >= T.garena.base && sb + R * C * sizeof(half_t) <= T.garena.base + T.garena.cap) {
      T.grads[key] = const_cast<half_t*>(src);
      return 0;
    }
    half_t* p = tape_galloc(c, R * C);
    if (!p) return fail(c, PNPI_ENOMEM, "gradient arena overflow");
    T.grads[key] = p;
    CK(launch_strided_add_f16(p, src, ld, off, R, C, 0, c->st));
  } else {
    CK(launch_strided_add_f16(it->second, src, ld, off, R, C, 1, c->st));
  }
  return 0;
}
// dgrad weights of a forward weight matrix w[N][taps][Cin] (built once per weight, kept on the device): wd[Cin][taps][Npad]
static int tape_wd(pnpi_ctx* c, const half_t* w, int N, int Npad, int taps, int Cin, const half_t** out) {
  Tape& T = *c->tape;
  auto it = T.wd.find(w);
  if (it == T.wd.end()) {
    half_t* p = nullptr;
    CKH(hipMalloc((void**)&p, (size_t)Cin * taps * Npad * sizeof(half_t)));
    CK(launch_repack_dgrad(w, N, Npad, taps, Cin, p, c->st));
    it = T.wd.emplace(w, p).first;
  }
  *out = it->second;
  return 0;
}
static int raw_conv(pnpi_ctx* c, const half_t* x, int Cx, int B, int H, int W, int ks, int pad, const half_t* w, int Nout, half_t* out) {
  GemmP p; gemm_defaults(p);
  p.x1 = x; p.C1 = Cx; p.ldx1 = Cx; p.B = B; p.H = H; p.W = W; p.Ho = H; p.Wo = W; p.ksize = ks; p.stride = 1; p.pad = pad;
  p.K = ks * ks * Cx; p.w = w; p.ldw = p.K; p.M = B * H * W; p.N = Nout; p.out = out; p.ldo = Nout;
  return igemm_prof(c, p, 2.0 * p.M * (double)p.N * p.K);
}

// Reverse walk.  d_out: gradient of the network output as NHWC fp16 with 8 channels (channels 4 .. 7 zero), consumed by conv_out's record
// (the one conv whose forward output is not an fp16 tensor).  Returns with T.d_ctx = d loss / d context (fp32, scaled like d_out).
static int tape_backward(pnpi_ctx* c, const half_t* d_out) {
  Tape& T = *c->tape;
  T.rec = false;
  const pnpi_model_config& g = c->cfg;
  for (size_t oi = T.ops.size(); oi-- > 0;) {
    const TapeOp& o = T.ops[oi];
    const half_t* dy = nullptr;
    if (o.out) {
      auto it = T.grads.find(o.out);
      if (it == T.grads.end()) continue;          // nothing downstream depends on this op's output
      dy = it->second;
    } else if (o.kind == TK_CONV) {
      dy = d_out;
    } else continue;
    switch (o.kind) {
      case TK_CONV: {
        const int taps = o.cw->k * o.cw->k, Cin = o.C1 + o.C2, Ng = o.out ? o.N : 8;
        const size_t Mo = (size_t)o.B * o.Ho * o.Wo;
        if (o.res) CKP(tape_add_strided(c, o.res, dy, Ng, 0, Mo, Ng));
        if (o.x1 == T.no_grad_input) break;
        const half_t* wd = nullptr;
        CKP(tape_wd(c, o.cw->w, o.N, Ng, taps, Cin, &wd));
        const half_t* src = dy;
        int Hs = o.Ho, Ws = o.Wo;
        if (o.stride == 2) {
          half_t* z = tape_galloc(c, (size_t)o.B * 2 * o.Ho * 2 * o.Wo * Ng);
          if (!z) return fail(c, PNPI_ENOMEM, "gradient arena overflow");
          CK(launch_zero_stuff2(dy, o.B, o.Ho, o.Wo, Ng, z, c->st));
          src = z; Hs = 2 * o.Ho; Ws = 2 * o.Wo;
        }
        half_t* dx = tape_galloc(c, (size_t)o.B * Hs * Ws * Cin);          // dense [B][Hs][Ws][C1 + C2]
        if (!dx) return fail(c, PNPI_ENOMEM, "gradient arena overflow");
        CK(raw_conv(c, src, Ng, o.B, Hs, Ws, o.cw->k, o.cw->k == 3 ? 1 : 0, wd, Cin, dx));
        size_t Min = (size_t)o.B * Hs * Ws;
        if (o.ups) {                                                         // the conv read the 2x-upsampled map
          half_t* dd = tape_galloc(c, (size_t)o.B * o.H * o.W * Cin);
          if (!dd) return fail(c, PNPI_ENOMEM, "gradient arena overflow");
          CK(launch_sumpool2x2(dx, o.B, o.H, o.W, Cin, dd, c->st));
          dx = dd; Min = (size_t)o.B * o.H * o.W;
        }
        CKP(tape_add_strided(c, o.x1, dx, Cin, 0, Min, o.C1));
        if (o.C2) CKP(tape_add_strided(c, o.x2, dx, Cin, o.C1, Min, o.C2));
        break;
      }
      case TK_GEMM: {
        if (o.res) CKP(tape_add_strided(c, o.res, dy, o.ldo, 0, (size_t)o.M, o.N));
        if (o.x1 == T.no_grad_input) break;
        const half_t* wt = nullptr;
        CKP(tape_wd(c, o.w, o.N, o.N, 1, o.K, &wt));                         // W^T: [K][N]
        half_t* dx = tape_galloc(c, (size_t)o.M * o.K);
        if (!dx) return fail(c, PNPI_ENOMEM, "gradient arena overflow");
        CK(op_gemm(c, dy, o.ldo, o.M, o.N, wt, o.N, o.K, nullptr, nullptr, 0, dx, o.K, o.alpha));
        if (o.x1 == T.ctx16) CK(launch_add_f16_to_f32(T.d_ctx, dx, (size_t)o.M * o.K, 1.f, c->st));
        else CKP(tape_add_strided(c, o.x1, dx, o.K, 0, (size_t)o.M, o.K));
        break;
      }
      case TK_GN: {
        const int C = o.C1 + o.C2;
        if (!(C & 7) && !(o.C1 & 7) && o.G <= 64 && !(o.C2 && o.x2 == o.x1)) {
          // three chip-wide phases, dx written (or added) straight into the gradient buffers of the concat sources
          const size_t R = (size_t)o.B * o.HW;
          auto dest = [&](const half_t* key, int Cw, GnbOut& out) -> int {
            if (key == T.no_grad_input) { out = GnbOut{nullptr, 0, 0}; return 0; }
            auto it = T.grads.find(key);
            if (it == T.grads.end()) {
              half_t* p = tape_galloc(c, R * Cw);
              if (!p) return fail(c, PNPI_ENOMEM, "gradient arena overflow");
              T.grads[key] = p;
              out = GnbOut{p, Cw, 0};
            } else out = GnbOut{it->second, Cw, 1};
            return 0;
          };
          GnbOut o1{nullptr, 0, 0}, o2{nullptr, 0, 0};
          CKP(dest(o.x1, o.C1, o1));
          if (o.C2) CKP(dest(o.x2, o.C2, o2));
          float* ws = nullptr;
          CKP(gn_bwd_workspace(c, o.B, o.HW, o.G, &ws));
          CK(launch_groupnorm_bwd2(o.x1, o.x2, o.C1, o.C2, o.B, o.HW, o.G, o.eps, o.nw->g, o.nw->b, o.silu, dy, o1, o2, ws, c->st));
          break;
        }
        half_t* dx = tape_galloc(c, (size_t)o.B * o.HW * C);
        if (!dx) return fail(c, PNPI_ENOMEM, "gradient arena overflow");
        CK(launch_groupnorm_bwd(o.x1, o.x2, o.C1, o.C2, o.B, o.HW, o.G, o.eps, o.nw->g, o.nw->b, o.silu, dy, dx, c->st));
        if (o.x1 != T.no_grad_input) CKP(tape_add_strided(c, o.x1, dx, C, 0, (size_t)o.B * o.HW, o.C1));
        if (o.C2) CKP(tape_add_strided(c, o.x2, dx, C, o.C1, (size_t)o.B * o.HW, o.C2));
        break;
      }
      case TK_LN: {
        GradDst d;
        CKP(tape_target(c, o.x1, (size_t)o.M * o.C1, d));
        CK(launch_layernorm_bwd(o.x1, dy, o.M, o.C1, o.eps, o.nw->g, d.p, c->st));
        CKP(tape_commit(c, d, (size_t)o.M * o.C1));
        break;
      }
      case TK_GEGLU: {
        GradDst d;
        CKP(tape_target(c, o.x1, (size_t)o.M * 2 * o.N, d));
        CK(launch_geglu_bwd(o.x1, dy, o.M, o.N, d.p, c->st));
        CKP(tape_commit(c, d, (size_t)o.M * 2 * o.N));
        break;
      }
      case TK_ATTN: {
        // head widths that are not a multiple of 8 (reduced test configurations only): the gradient of the attention output is
        // re-laid out with Dp-wide heads (zero pads) and the whole backward runs on the padded width -- q / k / v pads are zero
        const bool pad_dh = (o.dh & 7) != 0;
        const int dh_eff = pad_dh ? o.Dp : o.dh;
        int ldo_eff = o.ldo;
        if (pad_dh) {
          half_t* dyp = tape_galloc(c, (size_t)o.B * o.Nq * o.heads * o.Dp);
          if (!dyp) return fail(c, PNPI_ENOMEM, "gradient arena overflow");
          CK(launch_pad_heads_f16(dy, (size_t)o.B * o.Nq, o.heads, o.dh, o.Dp, dyp, c->st));
          dy = dyp; ldo_eff = o.heads * o.Dp;
        }
        const size_t need = attn_bwd_flash_shape(o.Nq, o.Nk, o.Dp, dh_eff) ? attn_bwd_flash_scratch_bytes(o.heads, o.Nq, o.Nk, dh_eff)
                                                                           : attn_bwd_scratch_bytes(o.Nq, o.Nk, dh_eff) * (size_t)o.heads;      // every head of a row in one set of launches
        if (need > T.attn_scratch_bytes) {
          if (T.attn_scratch) CKH(hipFree(T.attn_scratch));
          T.attn_scratch = nullptr; T.attn_scratch_bytes = 0;
          CKH(hipMalloc(&T.attn_scratch, need));
          T.attn_scratch_bytes = need;
        }
        // q, k, v are column ranges of at most two projection outputs (self: one tensor; cross: q2 and kv2): their gradients are
        // assembled in buffers of the projections' shapes, zero-initialised (the head pad columns get no gradient)
        auto grad_of = [&](const half_t* base, size_t n, half_t** out) -> int {
          auto it = T.grads.find(base);
          if (it == T.grads.end()) {
            half_t* p = tape_galloc(c, n);
            if (!p) return fail(c, PNPI_ENOMEM, "gradient arena overflow");
            CKH(hipMemsetAsync(p, 0, n * sizeof(half_t), c->st));
            T.grads[base] = p; *out = p;
          } else *out = it->second;
          return 0;
        };
        half_t *gq = nullptr, *gk = nullptr, *gv = nullptr;
        CKP(grad_of(o.q, (size_t)o.B * o.Nq * o.ldq, &gq));
        CKP(grad_of(o.k, (size_t)o.B * o.Nk * o.ldk, &gk));
        CKP(grad_of(o.v, (size_t)o.B * o.Nk * o.ldv, &gv));
        // (the projection outputs have exactly one consumer each -- this attention -- so the kernels may overwrite, not accumulate)
        CKP(attn_bwd(c, o.q, o.ldq, o.q_off, o.k, o.ldk, o.k_off, o.v, o.ldv, o.v_off, dy, ldo_eff, o.heads, o.Nq, o.Nk, o.Dp, dh_eff,
                     o.scale, o.B, gq, gk, gv, T.attn_scratch, T.attn_scratch_bytes, pad_dh ? nullptr : o.lse, o.out, o.ldo));
        break;
      }
      default: break;
    }
  }
  (void)g;
  return T.garena.overflow ? fail(c, PNPI_ENOMEM, "gradient arena overflow") : 0;
}

// AttentionBlock.forward (my_diffusers/models/attention.py:54-92): single head, scores materialised per image (VAE only)
static int vae_attn_fwd(pnpi_ctx* c, const VaeAttnW& a, const half_t* x, int B, int H, int W, half_t* out) {
  const size_t mk = c->temp.mark();
  const int C = a.C, N = H * W, M = B * N;
  half_t* g0 = talloc(c, (size_t)M * C);
  CK(op_gn(c, x, nullptr, C, 0, B, N, a.gn, c->cfg.vae_norm_groups, 1e-6f, 0, g0));
  half_t* qk = talloc(c, (size_t)M * 2 * C);
  const int ldv = round_up_i(N, 8);
  half_t* vt = talloc(c, (size_t)B * C * ldv);
  {
    VtOut v; v.outT = vt; v.col0 = 2 * C; v.ld = ldv; v.f32 = 0; v.rpb = N;
    CK(op_gemm(c, g0, C, M, C, a.w_qkv, C, 3 * C, a.b_qkv, nullptr, 0, qk, 2 * C, 1.f, &v));
  }
  half_t* ao = talloc(c, (size_t)M * C);
  half_t* sc = talloc(c, (size_t)N * ldv);
  const float alpha = 1.0f / sqrtf((float)C);
  for (int b = 0; b < B; ++b) {
    const half_t* qb = qk + (size_t)b * N * 2 * C;
    CK(op_gemm(c, qb, 2 * C, N, C, qb + C, 2 * C, N, nullptr, nullptr, 0, sc, ldv, alpha));
    if (!c->dry) PROF(PNPI_KC_SOFTMAX, 0.0, 2.0 * N * (double)N * 2.0, launch_softmax_rows(sc, N, N, ldv, c->st));
    CK(op_gemm(c, sc, ldv, N, N, vt + (size_t)b * C * ldv, ldv, C, nullptr, nullptr, 0, ao + (size_t)b * N * C, C));
  }
  CK(op_gemm(c, ao, C, M, C, a.proj.w, C, C, a.proj.b, x, C, out, C));
  c->temp.release(mk);
  return 0;
}

// Encoder.forward + quant_conv, posterior mean (my_diffusers/models/vae.py:113-130, 552-560, 329-336). x: NHWC fp16, 8-ch padded.
static int vae_encode_fwd(pnpi_ctx* c, const half_t* x, int B, int H, int W, float* mean_out) {
  const pnpi_model_config& g = c->cfg; const VaeW& v = c->vae;
  const int vn = g.vae_n_blocks, G = g.vae_norm_groups, L = g.vae_latent_channels;
  const float eps = 1e-6f;
  int ch = g.vae_block_out_channels[0];
  half_t* h = palloc(c, (size_t)B * H * W * ch);
  CK(op_conv(c, x, 8, nullptr, 0, B, H, W, v.e_conv_in, 1, 1, 0, v.e_conv_in.b, nullptr, h, H, W));
  for (int i = 0; i < vn; ++i) {
    const int oc = g.vae_block_out_channels[i];
    for (int j = 0; j < g.vae_layers_per_block; ++j) {
      half_t* o = palloc(c, (size_t)B * H * W * oc);
      CKP(resnet_fwd(c, v.e_res[i][j], h, ch, nullptr, 0, B, H, W, G, eps, o));
      h = o; ch = oc;
    }
    if (i != vn - 1) {
      // Downsample2D with padding=0: F.pad(x, (0,1,0,1)) then stride-2 conv (resnet.py:89-95)
      const int Ho = H / 2, Wo = W / 2;
      half_t* o = palloc(c, (size_t)B * Ho * Wo * oc);
      CK(op_conv(c, h, ch, nullptr, 0, B, H, W, v.e_down[i], 2, 0, 0, v.e_down[i].b, nullptr, o, Ho, Wo));
      h = o; H = Ho; W = Wo;
    }
  }
  half_t* m0 = palloc(c, (size_t)B * H * W * ch);
  CKP(resnet_fwd(c, v.e_mid[0], h, ch, nullptr, 0, B, H, W, G, eps, m0));
  half_t* m1 = palloc(c, (size_t)B * H * W * ch);
  CKP(vae_attn_fwd(c, v.e_attn, m0, B, H, W, m1));
  half_t* m2 = palloc(c, (size_t)B * H * W * ch);
  CKP(resnet_fwd(c, v.e_mid[1], m1, ch, nullptr, 0, B, H, W, G, eps, m2));
  half_t* gno = palloc(c, (size_t)B * H * W * ch);
  CK(op_gn(c, m2, nullptr, ch, 0, B, H * W, v.e_norm_out, G, eps, 1, gno));
  half_t* mo = palloc(c, (size_t)B * H * W * 8);
  CK(op_conv(c, gno, ch, nullptr, 0, B, H, W, v.e_conv_out, 1, 1, 0, v.e_conv_out.b, nullptr, mo, H, W, 8));
  {
    VtOut vv; vv.outT = mean_out; vv.col0 = 0; vv.ld = H * W; vv.f32 = 1; vv.rpb = H * W;
    CK(op_conv(c, mo, 8, nullptr, 0, B, H, W, v.quant, 1, 0, 0, v.quant.b, nullptr, nullptr, H, W, L, &vv));
  }
  return 0;
}

// post_quant_conv + Decoder.forward (vae.py:562-566, 191-209). z: NHWC fp16, 8-ch padded. out: fp32 NCHW.
static int vae_decode_fwd(pnpi_ctx* c, const half_t* z, int B, int H, int W, float* out_nchw) {
  const pnpi_model_config& g = c->cfg; const VaeW& v = c->vae;
  const int vn = g.vae_n_blocks, G = g.vae_norm_groups;
  const float eps = 1e-6f;
  half_t* pq = palloc(c, (size_t)B * H * W * 8);
  CK(op_conv(c, z, 8, nullptr, 0, B, H, W, v.post_quant, 1, 0, 0, v.post_quant.b, nullptr, pq, H, W, 8));
  int ch = g.vae_block_out_channels[vn - 1];
  half_t* h = palloc(c, (size_t)B * H * W * ch);
  CK(op_conv(c, pq, 8, nullptr, 0, B, H, W, v.d_conv_in, 1, 1, 0, v.d_conv_in.b, nullptr, h, H, W));
  half_t* m0 = palloc(c, (size_t)B * H * W * ch);
  CKP(resnet_fwd(c, v.d_mid[0], h, ch, nullptr, 0, B, H, W, G, eps, m0));
  half_t* m1 = palloc(c, (size_t)B * H * W * ch);
  CKP(vae_attn_fwd(c, v.d_attn, m0, B, H, W, m1));
  half_t* m2 = palloc(c, (size_t)B * H * W * ch);
  CKP(resnet_fwd(c, v.d_mid[1], m1, ch, nullptr, 0, B, H, W, G, eps, m2));
  h = m2;
  for (int i = 0; i < vn; ++i) {
    const int oc = g.vae_block_out_channels[vn - 1 - i];
    for (int j = 0; j <= g.vae_layers_per_block; ++j) {
      half_t* o = palloc(c, (size_t)B * H * W * oc);
      CKP(resnet_fwd(c, v.d_res[i][j], h, ch, nullptr, 0, B, H, W, G, eps, o));
      h = o; ch = oc;
    }
    if (i != vn - 1) {
      const int Ho = H * 2, Wo = W * 2;
      half_t* o = palloc(c, (size_t)B * Ho * Wo * oc);
      CK(op_conv(c, h, ch, nullptr, 0, B, H, W, v.d_up[i], 1, 1, 1, v.d_up[i].b, nullptr, o, Ho, Wo));
      h = o; H = Ho; W = Wo;
    }
  }
  half_t* gno = palloc(c, (size_t)B * H * W * ch);
  CK(op_gn(c, h, nullptr, ch, 0, B, H * W, v.d_norm_out, G, eps, 1, gno));
  {
    VtOut vv; vv.outT = out_nchw; vv.col0 = 0; vv.ld = H * W; vv.f32 = 1; vv.rpb = H * W;
    CK(op_conv(c, gno, ch, nullptr, 0, B, H, W, v.d_conv_out, 1, 1, 0, v.d_conv_out.b, nullptr, nullptr, H, W, -1, &vv));
  }
  return 0;
}

// ---------------------------------------------------------------------------------------------------- text K / V cache
static void for_each_transformer(pnpi_ctx* c, const std::function<void(const TransformerW&, int)>& fn) {
  UNetW& u = c->unet;
  int idx = 0;
  for (auto& blk : u.down_attn) for (auto& t : blk) fn(t, idx++);
  fn(u.mid_attn, idx++);
  for (auto& blk : u.up_attn) for (auto& t : blk) fn(t, idx++);
}
static size_t text_kv_bytes(pnpi_ctx* c, int rows) {
  const int T = c->cfg.ctx_len, ldv = round_up_i(T, 8);
  size_t total = 0;
  for_each_transformer(c, [&](const TransformerW& t, int) {
    const size_t hd = (size_t)t.heads * t.Dp;
    total += align_up((size_t)rows * T * hd * sizeof(half_t), 256) + align_up((size_t)rows * hd * ldv * sizeof(half_t), 256);
  });
  return total + 4096;
}
// K = context W_k^T, V^T = (context W_v^T)^T for the 16 cross-attention layers, `rows` context rows (fp32 [rows][T][X] on the device)
static int text_kv_precompute(pnpi_ctx* c, const float* context, int rows) {
  const pnpi_model_config& g = c->cfg;
  TextKV& kv = c->tkv;
  kv.rows = 0; kv.use = false;
  if (rows <= 0 || rows > c->max_rows) return fail(c, PNPI_EINVAL, "text K/V precompute: rows out of range (max_unet_rows)");
  if (text_kv_bytes(c, rows) > kv.cap) return fail(c, PNPI_ENOMEM, "text K/V cache arena too small");
  const int T = g.ctx_len, X = g.cross_dim, ldv = round_up_i(T, 8);
  c->persist.reset(); c->temp.reset();
  half_t* ctx16 = palloc(c, (size_t)rows * T * X);
  CK(launch_f32_to_f16(context, (size_t)rows * T * X, ctx16, c->st));
  kv.k.clear(); kv.vt.clear();
  size_t off = 0;
  int rc = 0;
  for_each_transformer(c, [&](const TransformerW& t, int) {
    if (rc) return;
    const int hd = t.heads * t.Dp;
    half_t* k2 = (half_t*)(kv.base + off); off += align_up((size_t)rows * T * hd * sizeof(half_t), 256);
    half_t* vt2 = (half_t*)(kv.base + off); off += align_up((size_t)rows * hd * ldv * sizeof(half_t), 256);
    kv.k.push_back(k2); kv.vt.push_back(vt2);
    // the V^T rows are padded to 8 keys and the blocks sit at row-count-dependent offsets: a pad column of this layout may hold another
    // row count's projection data (or anything), and the call-back path multiplies pad columns by zero probabilities -- 0 * inf = NaN.
    // Clear the block before the projection writes the real columns (as the in-forward talloc path does).
    if (ldv != T && hipMemsetAsync(vt2, 0, (size_t)rows * hd * ldv * sizeof(half_t), c->st) != hipSuccess) { rc = PNPI_EHIP; return; }
    VtOut v; v.outT = vt2; v.col0 = hd; v.ld = ldv; v.f32 = 0; v.rpb = T;
    rc = op_gemm(c, ctx16, X, rows * T, X, t.w_kv2, X, 2 * hd, nullptr, nullptr, 0, k2, hd, 1.f, &v, 2.0 * rows * T * 2.0 * t.C * X);
  });
  if (rc) return fail_launch(c, rc, "text K/V projection");
  kv.rows = rows;
  c->ctr.text_kv_rows += rows;
  return 0;
}

// ---------------------------------------------------------------------------------------------------- controller tables
static float* misc_f(pnpi_ctx* c, size_t n) { return (float*)c->ctrl_arena.alloc(n * sizeof(float)); }

static int upload(pnpi_ctx* c, void* dst, const void* src, size_t bytes) {
  CKH(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->st));
  // the host staging vectors are reused by the caller: make the copy complete before returning
  CKH(hipStreamSynchronize(c->st));
  return 0;
}

// Build the device-side tables for `rows` UNet rows (rows_per_image = 4 when controllers are active).
static int setup_ctrl(pnpi_ctx* c, const pnpi_ctrl_desc* cds, int nimg, int rows, int rpi = 4, int src_off = 2, int tgt_off = 3) {
  CtrlDev& cd = c->cd;
  c->ctrl_arena.reset();
  cd = CtrlDev();
  cd.nimg = nimg;
  std::vector<int> id(rows * 4), rep(rows * 4), plain, pairs;
  for (int r = 0; r < rows; ++r) { id[r * 4] = r; id[r * 4 + 1] = r; id[r * 4 + 2] = r; id[r * 4 + 3] = r; }
  rep = id;
  const int T = c->cfg.ctx_len;
  if (T > 96) return fail(c, PNPI_ESHAPE, "ctx_len > 96 unsupported by the cross-attention edit kernel");
  std::vector<int> edit_img;
  if (cds) {
    if (rows != nimg * rpi) return fail(c, PNPI_EINVAL, "controllers need rows == rows_per_image * nimg");
    for (int i = 0; i < nimg; ++i) if (cds[i].kind == 1) edit_img.push_back(i);
  }
  if (cds) {   // MasaCtrl images (kind 2): rows [unc_src, unc_tgt, cond_src, cond_tgt]; each target row reads its half's source K, V
    std::vector<int> masa = id;
    for (int i = 0; i < nimg; ++i) {
      if (cds[i].kind != 2) continue;
      if (rpi != 4) return fail(c, PNPI_EINVAL, "MasaCtrl controllers need the 4-row layout");
      std::vector<unsigned char> step_on;
      const bool step_list = cds[i].masa_n_steps > 0 && cds[i].masa_step_on_host;
      if (step_list) step_on.assign(cds[i].masa_step_on_host, cds[i].masa_step_on_host + cds[i].masa_n_steps);
      if (cd.masa_any && (cd.masa_start_step != cds[i].masa_start_step || cd.masa_start_layer != cds[i].masa_start_layer ||
                          cd.masa_layer_mask != cds[i].masa_layer_mask || cd.masa_step_list != step_list || cd.masa_step_on != step_on))
        return fail(c, PNPI_EINVAL, "all MasaCtrl controllers of one batch must share their step / layer windows (or lists)");
      cd.masa_any = true; cd.masa_start_step = cds[i].masa_start_step; cd.masa_start_layer = cds[i].masa_start_layer;
      cd.masa_layer_mask = cds[i].masa_layer_mask; cd.masa_step_list = step_list; cd.masa_step_on = step_on;
      for (int half = 0; half < 2; ++half) {
        const int src = i * 4 + 2 * half, tgt = src + 1;
        masa[tgt * 4 + 2] = src; masa[tgt * 4 + 3] = src;
      }
    }
    if (cd.masa_any) {
      cd.rows_masa = (int*)c->ctrl_arena.alloc(masa.size() * sizeof(int));
      CKP(upload(c, cd.rows_masa, masa.data(), masa.size() * sizeof(int)));
    }
  }
  cd.any_edit = !edit_img.empty();
  cd.npairs = (int)edit_img.size();
  std::vector<bool> is_pair_row(rows, false);
  for (int i : edit_img) {
    int src = i * rpi + src_off, tgt = i * rpi + tgt_off;
    rep[tgt * 4 + 1] = src; rep[tgt * 4 + 2] = src;  // q and k of the target row come from the source row
    pairs.push_back(src); pairs.push_back(tgt);
    is_pair_row[tgt] = true;   // only the target row leaves the plain path; the source row stays bit-identical to it
    cd.pair_img.push_back(i);
  }
  for (int r = 0; r < rows; ++r) if (!is_pair_row[r]) { plain.push_back(r); plain.push_back(r); plain.push_back(r); plain.push_back(r); }
  cd.n_plain = (int)plain.size() / 4;
  cd.rows_id = (int*)c->ctrl_arena.alloc(id.size() * sizeof(int));
  cd.rows_rep = (int*)c->ctrl_arena.alloc(rep.size() * sizeof(int));
  CKP(upload(c, cd.rows_id, id.data(), id.size() * sizeof(int)));
  CKP(upload(c, cd.rows_rep, rep.data(), rep.size() * sizeof(int)));
  if (!plain.empty()) {
    cd.rows_plain = (int*)c->ctrl_arena.alloc(plain.size() * sizeof(int));
    CKP(upload(c, cd.rows_plain, plain.data(), plain.size() * sizeof(int)));
  }
  if (!cd.any_edit) return 0;
  cd.pairs = (int*)c->ctrl_arena.alloc(pairs.size() * sizeof(int));
  CKP(upload(c, cd.pairs, pairs.data(), pairs.size() * sizeof(int)));
  const pnpi_ctrl_desc& d0 = cds[edit_img[0]];
  cd.n_alpha_rows = d0.n_alpha_rows;
  cd.self_lo = d0.self_replace_lo; cd.self_hi = d0.self_replace_hi; cd.self_max_tokens = d0.self_replace_max_tokens;
  const int P = cd.npairs;
  std::vector<half_t> mm((size_t)P * 96 * 96, (half_t)0.f);
  // LocalBlend planes: {src, tgt} blend-word selectors, plus {src, tgt} substruct-word selectors when any controller of the batch has them
  int planes = 2;
  for (int pi = 0; pi < P; ++pi)
    if (cds[edit_img[pi]].lb_enabled && cds[edit_img[pi]].lb_sub_alpha_host) planes = 4;
  cd.lb_planes = planes;
  std::vector<float> coef((size_t)cd.n_alpha_rows * 2 * P * 96, 0.f), lba((size_t)P * planes * 96, 0.f);
  for (int pi = 0; pi < P; ++pi) {
    const pnpi_ctrl_desc& d = cds[edit_img[pi]];
    if (d.n_alpha_rows != cd.n_alpha_rows || d.self_replace_lo != cd.self_lo || d.self_replace_hi != cd.self_hi ||
        d.self_replace_max_tokens != cd.self_max_tokens)
      return fail(c, PNPI_EINVAL, "all controllers of one batch must share the step schedule");
    if (!d.cross_alpha_host || !d.mapper_host || !d.alphas_host || !d.equalizer_host)
      return fail(c, PNPI_EINVAL, "controller tables missing");
    for (int w = 0; w < T; ++w)
      for (int j = 0; j < T; ++j) mm[((size_t)pi * 96 + j) * 96 + w] = (half_t)d.mapper_host[w * T + j];
    for (int s = 0; s < cd.n_alpha_rows; ++s)
      for (int j = 0; j < T; ++j) {
        float a = d.cross_alpha_host[s * T + j], eq = d.equalizer_host[j], al = d.alphas_host[j];
        coef[(((size_t)s * 2 + 0) * P + pi) * 96 + j] = a * eq * al;
        coef[(((size_t)s * 2 + 1) * P + pi) * 96 + j] = a * eq * (1.f - al) + (1.f - a);
      }
    cd.lb_enabled.push_back(d.lb_enabled);
    cd.lb_start.push_back(d.lb_start);
    cd.lb_th.push_back(d.lb_threshold);
    // a pair without substruct words in a 4-plane batch: its substruct maps are all zero and never exceed an infinite threshold
    cd.lb_th_sub.push_back(d.lb_enabled && d.lb_sub_alpha_host ? d.lb_threshold_sub : INFINITY);
    if (d.lb_enabled) {
      if (!d.lb_alpha_host) return fail(c, PNPI_EINVAL, "lb_alpha missing");
      if (c->unet.lb_nslots == 0) return fail(c, PNPI_ESHAPE, "LocalBlend needs the five 16x16 cross-attention maps (latent 64x64 layout)");
      cd.lb_any = 1;
      for (int w = 0; w < 2; ++w)
        for (int j = 0; j < T; ++j) {
          lba[((size_t)pi * planes + w) * 96 + j] = d.lb_alpha_host[w * T + j];
          if (d.lb_sub_alpha_host) lba[((size_t)pi * planes + 2 + w) * 96 + j] = d.lb_sub_alpha_host[w * T + j];
        }
    }
  }
  cd.mmatT = (half_t*)c->ctrl_arena.alloc(mm.size() * sizeof(half_t));
  cd.coef = misc_f(c, coef.size());
  CKP(upload(c, cd.mmatT, mm.data(), mm.size() * sizeof(half_t)));
  CKP(upload(c, cd.coef, coef.data(), coef.size() * sizeof(float)));
  if (cd.lb_any) {
    cd.lb_alpha = misc_f(c, lba.size());
    CKP(upload(c, cd.lb_alpha, lba.data(), lba.size() * sizeof(float)));
    size_t nacc = (size_t)P * c->unet.lb_nslots * planes * c->unet.lb_tokens;
    cd.lb_acc = misc_f(c, nacc);
    CKH(hipMemsetAsync(cd.lb_acc, 0, nacc * sizeof(float), c->st));
  }
  if (c->ctrl_arena.overflow) return fail(c, PNPI_ENOMEM, "controller arena overflow");
  return 0;
}

static int apply_local_blend(pnpi_ctx* c, float* latents /*[nimg][2][E]*/, int step_index) {
  CtrlDev& cd = c->cd;
  if (!cd.lb_any) return 0;
  const pnpi_model_config& g = c->cfg;
  const size_t E = (size_t)g.in_channels * g.sample_size * g.sample_size;
  const int mhw = (int)lroundf(sqrtf((float)c->unet.lb_tokens));
  for (int pi = 0; pi < cd.npairs; ++pi) {
    if (!cd.lb_enabled[pi]) continue;
    if (step_index + 1 <= cd.lb_start[pi]) continue;   // LocalBlend.counter > start_blend (attention_control.py:108-110)
    const float* acc = cd.lb_acc + (size_t)pi * c->unet.lb_nslots * cd.lb_planes * c->unet.lb_tokens;
    CK(launch_local_blend(acc, c->unet.lb_nslots, mhw, g.sample_size, g.in_channels, cd.lb_th[pi],
                          latents + (size_t)cd.pair_img[pi] * 2 * E, 1, c->st, cd.lb_planes, cd.lb_th_sub[pi]));
  }
  return 0;
}

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
  } else {
    CKH(hipMalloc((void**)&c->warena.base, wbytes));
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
  void* bufs[] = {c->warena_borrowed ? nullptr : (void*)c->warena.base, c->persist.base, c->temp.base, c->ctrl_arena.base, c->splitk_ws, c->gn_partial, c->gn_bwd_ws,
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
static bool recon_active(const pnpi_recon_desc* rc, int t) {   // proximal_guidance_forward.py:48,60
  return rc && rc->ref_image && rc->recon_lr > 0.f && ((rc->recon_t > 0 && t < rc->recon_t) || (rc->recon_t < 0 && t > -rc->recon_t));
}

int pnpi_cfg_ddim_prev(pnpi_ctx* c, const float* eps, const float* x, int nimg, int rpi, size_t row_elems, float gs, int t, int ratio,
                       const float* noise_loss, int offset_rows, const float* target, float offset_scale, float* offset_out,
                       float* x_out, const float* prox_threshold, int prox, const pnpi_recon_desc* recon) {
  if (prox < 0 || prox > 2 || (prox && !prox_threshold)) return fail(c, PNPI_EINVAL, "prox must be 0, or 1 / 2 with a threshold");
  float af, at; CKP(alphas_for(c, t, ratio, false, &af, &at));
  const bool rc = prox && recon_active(recon, t);
  const int S = c->cfg.sample_size;
  CK(launch_cfg_ddim_prev(eps, x, nimg, rpi, row_elems, gs, af, at, noise_loss, offset_rows, target, offset_scale, offset_out, x_out, c->st,
                          prox_threshold, prox, rc ? recon->ref_image : nullptr, rc ? recon->recon_lr : 0.f, rc ? recon->dilate_mask : 0, S, S));
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
    const bool rc = prox && recon_active(recon, t);
    CK(launch_cfg_ddim_prev(eps, lat, nimg, 2, E, gs, af, at, nl, offset_rows, nullptr, 1.f, nullptr, lat, c->st, prox ? thr : nullptr, prox,
                            rc ? recon->ref_image : nullptr, rc ? recon->recon_lr : 0.f, rc ? recon->dilate_mask : 0, g.sample_size,
                            g.sample_size));
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
