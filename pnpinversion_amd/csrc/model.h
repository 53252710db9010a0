// Host-side model description for libpnpi: weight slots, SD-1.x UNet / VAE graphs, workspace planning.
#pragma once
#include <atomic>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/pnpi.h"
#include "ops.h"

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
static inline int round_up_i(int x, int a) { return (x + a - 1) / a * a; }

// Bump allocator. With base == nullptr it only measures (dry run) and hands out fake, never-dereferenced addresses.
struct Bump {
  char* base = nullptr;
  size_t cap = 0, off = 0, peak = 0;
  bool overflow = false;
  void* alloc(size_t bytes) {
    off = align_up(off, 256);
    size_t at = off;
    off += bytes;
    if (off > peak) peak = off;
    if (base && off > cap) { overflow = true; return base; }
    return base ? (void*)(base + at) : (void*)(uintptr_t)(0x1000 + at);
  }
  size_t mark() const { return off; }
  void release(size_t m) { off = m; }
  void reset() { off = 0; }
};

struct Slot {
  int kind;      // 0 = matrix (fp16), 1 = vector (fp32)
  void* dst;
  int rows, cols, taps, dst_ld, cin_pad, row0, dh, Dp;
  int n;
  int ilv_half;  // > 0: GEGLU x/gate row interleave (groups of 32)
  bool loaded;
};

struct ConvW { half_t* w; float* b; int cin, cin_pad, cout, k; };
struct LinW { half_t* w; float* b; int in, out; };
struct NormW { float* g; float* b; int c; };

struct ResnetW {
  NormW n1, n2;
  ConvW c1, c2, sc;
  bool has_sc;
  int temb_off;   // >= 0: conv1 bias comes from the per-step bias_eff table at this offset; -1: static conv1 bias
  int cin, cout;
};

struct TransformerW {
  int C, heads, dh, Dp;
  NormW gn, ln1, ln2, ln3;
  ConvW proj_in, proj_out;
  half_t* w_qkv;      // [3*heads*Dp][C]   (self-attention, heads zero-padded to Dp)
  float* b_qkv_aug;   // [3*heads*Dp] or nullptr: zeros, but 1.0 at column dh of every K head and every V head -- d = 40 heads padded to 64:
                      // the 64-wide flash kernel takes its max shift and row sum through the MFMAs with it (attn.hip, AUG)
  LinW o1;            // [C][C] + bias
  half_t* w_q2;       // [heads*Dp][C]
  half_t* w_kv2;      // [2*heads*Dp][cross_dim]
  LinW o2;
  LinW ff1;           // [8C][C]
  LinW ff2;           // [C][4C]
  int place;          // 0 down, 1 mid, 2 up
  int lb_slot0;       // first LocalBlend slot of this layer's cross-attention, or -1
};

struct VaeAttnW {
  int C;
  NormW gn;
  half_t* w_qkv; float* b_qkv;   // [3C][C]
  LinW proj;
};

struct UNetW {
  ConvW conv_in, conv_out;
  NormW norm_out;
  LinW t1, t2;
  half_t* temb_w;     // [sumC][4*C0]
  float* temb_b;      // [sumC] time_emb_proj biases
  float* conv1_b;     // [sumC] conv1 biases
  int temb_total;
  std::vector<std::vector<ResnetW>> down_res;
  std::vector<std::vector<TransformerW>> down_attn;
  std::vector<ConvW> down_samp;     // size n_blocks-1
  ResnetW mid_res[2];
  TransformerW mid_attn;
  std::vector<std::vector<ResnetW>> up_res;
  std::vector<std::vector<TransformerW>> up_attn;
  std::vector<ConvW> up_samp;
  int lb_nslots, lb_tokens;         // LocalBlend slots (5 layers x heads) and their token count (256), 0 if unavailable
};

struct VaeW {
  // encoder
  ConvW e_conv_in, e_conv_out;
  std::vector<std::vector<ResnetW>> e_res;
  std::vector<ConvW> e_down;
  ResnetW e_mid[2];
  VaeAttnW e_attn;
  NormW e_norm_out;
  ConvW quant;       // 1x1, 2L -> 2L
  // decoder
  ConvW post_quant;  // 1x1, L -> L (stored as 8 -> 8, zero padded)
  ConvW d_conv_in, d_conv_out;
  ResnetW d_mid[2];
  VaeAttnW d_attn;
  std::vector<std::vector<ResnetW>> d_res;
  std::vector<ConvW> d_up;
  NormW d_norm_out;
};

struct ClipLayerW {
  NormW ln1, ln2;
  half_t* w_qkv; float* b_qkv;   // [3H][H] fused q | k | v (+ biases)
  LinW out, fc1, fc2;
};
struct ClipW {
  int H = 0, heads = 0, I = 0, vocab = 0, T = 0;
  half_t* tok = nullptr;         // [vocab][H]
  half_t* pos = nullptr;         // [T][H]
  std::vector<ClipLayerW> layers;
  NormW final_ln;
};

struct Tensor4 { half_t* p; int B, H, W, C; };

struct CtrlDev {
  bool any_edit = false;
  int nimg = 0, n_alpha_rows = 0;
  int npairs = 0;                 // images with kind == 1
  int lb_any = 0;
  std::vector<int> lb_start;      // per pair
  std::vector<float> lb_th;
  std::vector<float> lb_th_sub;   // per pair; +inf when the pair has no substruct words
  int lb_planes = 2;              // 2, or 4 when any pair of the batch has substruct words
  std::vector<int> lb_enabled;
  std::vector<int> pair_img;      // pair -> image
  int self_lo = 0, self_hi = 0, self_max_tokens = 0;
  // device tables
  int* rows_id = nullptr;         // [rows][4] identity
  int* rows_rep = nullptr;        // [rows][4] self-attention replacement
  int* rows_plain = nullptr;      // rows that take the plain cross-attention path
  int* rows_masa = nullptr;       // [rows][4] MasaCtrl: target rows read K, V of the source row of their CFG half
  bool masa_any = false;
  int masa_start_step = 0, masa_start_layer = 0;
  unsigned masa_layer_mask = 0;                     // bit 31: a layer_idx list was given (bits 0..15 = its blocks); 0 = the start_layer window
  std::vector<unsigned char> masa_step_on;          // step_idx list as a per-step flag array; empty + !masa_step_list = the start_step window
  bool masa_step_list = false;
  int n_plain = 0;
  int* pairs = nullptr;           // [npairs][2]
  half_t* mmatT = nullptr;        // [npairs][96][96]
  float* coef = nullptr;          // [n_alpha_rows][2][npairs][96]  (c1 block, c2 block per step)
  float* lb_alpha = nullptr;      // [npairs][lb_planes][96]
  float* lb_acc = nullptr;        // [npairs][nslots][lb_planes][tokens]
};

// Text K / V cache: the cross-attention keys / values depend only on the text context, which is constant over a denoising loop
// (to_k / to_v of attention.py:230-234 applied to encoder_hidden_states): computed once per loop for the 16 transformer blocks.
struct TextKV {
  std::vector<half_t*> k, vt;     // per transformer block in execution order: [rows][T][hd], [rows][hd][ldv]
  char* base = nullptr; size_t cap = 0;
  int rows = 0;                   // rows the cache holds (0 = invalid)
  bool use = false;               // the forward in flight reads the cache instead of projecting the context
};

struct ProfRec { int cls; double flops, bytes; hipEvent_t a, b; int M, N, K, ksize; int cfg = -1, split = 0; int geom[7] = {0, 0, 0, 0, 0, 0, 0}; };

struct Tape;   // activation tape of a differentiable forward (null-text path; api.hip)

struct pnpi_ctx {
  pnpi_model_config cfg;
  int device;
  hipStream_t st;
  std::string err;
  int max_rows, max_vae;
  bool dry;
  int tf_index = 0;   // transformer block counter of the forward in flight (MasaCtrl start_layer)
  Bump warena, persist, temp, ctrl_arena;
  struct AugBias { float* p; int heads, Dp, dh; };
  std::vector<AugBias> aug_biases;                  // the b_qkv_aug vectors of this build (filled after the arena exists)
  bool warena_borrowed = false;                     // pnpi_create_shared: warena.base is the parent's (never written here)
  struct ArenaRef { void* base; std::atomic<int> refs; };
  ArenaRef* warena_ref = nullptr;                   // shared by the owner and every context that borrows the arena: the last pnpi_destroy frees it
  float* splitk_ws; size_t splitk_bytes;
  float* gn_partial;
  float* temb_table;      // [n_train][C0] fp32 sinusoid table
  float* temb_h;          // [4*C0]
  float* temb_emb;        // [4*C0]
  float* bias_eff;        // [temb_total] of the forward in flight (points into bias_tab once cached)
  float* bias_scratch;    // [temb_total]
  float* bias_tab = nullptr;          // [n_train][temb_total]: conv1 bias + time embedding per TIMESTEP, filled on first use -- it depends
  std::vector<char> bias_valid;       // on t and the weights only, so every later forward at that timestep launches no GEMV
  TextKV tkv;
  // level-1 fallback for controllers without a descriptor: materialise-and-call-back (pnpi_set_attention_callback)
  pnpi_attn_callback attn_cb = nullptr;
  float* gn_bwd_ws = nullptr; size_t gn_bwd_ws_floats = 0;   // partial sums of the three-phase GroupNorm backward (grown on demand)
  void* attn_cb_user = nullptr;
  float* attn_buf = nullptr; size_t attn_buf_bytes = 0;   // caller-owned device buffer the probabilities are materialised in
  std::unordered_map<std::string, Slot> slots;
  UNetW unet;
  VaeW vae;
  ClipW clip;
  int* rows_ident = nullptr;   // [max(max_rows, 8)][4] identity attention-row table (text encoder)
  int rows_ident_n = 0;
  std::vector<float> ac;
  float final_alpha;
  bool sched_set;
  pnpi_counters ctr;
  CtrlDev cd;
  std::vector<char> host_stage;
  bool prof_on = false;
  std::vector<ProfRec> prof;
  Tape* tape = nullptr;          // non-null while a forward is being recorded for a backward pass
  // A split-K launch of the UNet forward whose combine has not run yet (api_graph.inc: op_conv defers it, the next op either is the
  // GroupNorm that reads the tensor -- it then sums the slabs itself -- or runs the combine first).  pend_keep: somebody besides that
  // GroupNorm reads the fp16 tensor (residual / skip connection), so the fused kernel must also store it.
  GemmP pend; bool pend_on = false, pend_keep = true, defer_ok = false;
};
