// Host-side launch wrappers for the gfx950 kernels (internal C++ interface; the public boundary is include/pnpi.h).
#pragma once
#include "common.h"

// ---------------------------------------------------------------------------------------------------------------
// Implicit-GEMM convolution / linear:  out[m][n] = alpha * sum_k A(m,k) * W[n][k] + bias[n] (+ res[m][n])
//   A(m,k): NHWC gather from up to two concatenated sources (x1: channels [0,C1), x2: [C1,C1+C2)); m=(b,yo,xo),
//   k=(tap,c) with tap=(r,s) for 3x3; optional nearest-2x upsample folded into the address map; zero padding.
//   W: [N][K] fp16, K = ksize*ksize*(C1+C2), tap-major (KRSC).
// Columns n >= vt_col0 are written transposed per batch item into outT (fp16 or fp32):
//   outT[(b*(N-vt_col0) + (n-vt_col0))*vt_ld + (m % rows_per_batch)]   -- used for V^T (attention) and NCHW fp32 outputs.
// ---------------------------------------------------------------------------------------------------------------
struct GemmP {
  const half_t* x1; const half_t* x2;
  int C1, C2, ldx1, ldx2;
  int B, H, W, Ho, Wo;
  int ksize, stride, pad, ups;
  const half_t* w; int ldw;
  int M, N, K;
  const float* bias;
  const half_t* res; int ldres;
  float alpha;
  half_t* out; int ldo;
  void* outT; int vt_col0, vt_ld, vt_f32, rows_per_batch;
  int vt_perm16;   // transposed columns: token t of each aligned 16-token group is stored at position vt_perm16_pos(t) (see below)
  float* slab; int splitk, kchunks_per_split;
  float* stats;   // optional: per (m-tile, channel) sum / sum-of-squares of the fp16 output, [gridDim.x][N][2] (GroupNorm fusion)
  int geglu;      // N columns are [x(32) | gate(32)] interleaved groups; output has N/2 columns: x * gelu(gate)
  int epi_lds;    // set by launch_igemm: coalesced LDS-staged epilogue is applicable
  int bias_init;  // set by launch_igemm: the LDS-DMA kernel starts its accumulators at the bias (alpha == 1, no split-K) and its epilogue adds none
  int res_late;   // set by launch_igemm: the LDS epilogue adds the residual in its store loop instead of staging it
  int k_order;    // set by launch_igemm for the ping-pong kernel: 1 = 3 x 3 convolutions walk k channel-slab-major (nine taps per 64-channel slab in a row)
  int gx, gy, gz, tile_order;   // set by the launcher: logical tile grid and XCD-aware traversal order (see igemm_dma_kernel)
  // nbatch > 1: the launch holds nbatch independent problems of the same shape (grid z = problem index; no split-K): problem z reads
  // x1 + z * sx1, w + z * sw and writes out + z * sout (elements) / outT + z * soutT (bytes).  The attention backward's per-head GEMMs.
  int nbatch;
  long sx1, sw, sout, soutT;
};
void gemm_defaults(GemmP& p);
// ws: fp32 scratch for split-K slabs (ws_bytes available). force_cfg: -1 auto, 0 = 128x128, 1 = 64x64, 2 = 64x64 split-K.
// stats_tile_rows (out): rows per m-tile of the p.stats partials actually produced, 0 if this launch produced none
// deferred (optional): a launch that ends in the vectorised split-K combine (alpha == 1) does NOT run it; *deferred receives the launch
// parameters (splitk > 1; otherwise splitk = 1) and the caller either runs launch_splitk_reduce or gives the slabs to a consumer that
// sums them itself in the same order (launch_groupnorm_slab)
int launch_igemm(GemmP p, float* ws, size_t ws_bytes, hipStream_t st, int force_cfg = -1, int force_split = 0,
                 int* cfg_used = nullptr, int* stats_tile_rows = nullptr, GemmP* deferred = nullptr);
int launch_splitk_reduce(const GemmP& p, hipStream_t st);
int igemm_init();  // sets dynamic-LDS attributes once
// tile configuration id / split-K of the most recent launch_igemm and the <BM, BN, BKT, NST, WGM, ABL, WK> of its igemm_dma_kernel (profiling)
void igemm_last_launch(int* cfg, int* split, int* geom7);
// measured tile table: 1 = exact {M, N, K, ksize} entry, 2 = the same layer at the nearest row count (<= 4x away in M), 0 = none (cost model)
int igemm_table_lookup(int M, int N, int K, int ksize, int* cfg, int* split, int* entry_m);
int igemm_set_tuning(const char* key, int value);   // process-wide tuning knobs; 0 on success, -1 unknown key
void igemm_set_dma(int on);  // 1 (default): LDS-DMA kernel where applicable; 0: register-staged v1 kernel everywhere

// ---------------------------------------------------------------------------------------------------------------
// Normalisation / elementwise
// ---------------------------------------------------------------------------------------------------------------
// GroupNorm over NHWC (two-source concat allowed). partial: [B][nchunk][G][2] fp32 scratch.
int launch_groupnorm(const half_t* x1, const half_t* x2, int C1, int C2, int B, int HW, int G, float eps,
                     const float* gamma, const float* beta, int silu, half_t* out, float* partial, hipStream_t st);
// GroupNorm whose statistics were produced by the GEMM epilogues of the tensor's producer(s): st1/st2 are [tiles][C][2]
// per-channel partial sums with tpb1/tpb2 m-tiles per batch item.
int launch_groupnorm_fused(const half_t* x1, const half_t* x2, int C1, int C2, int B, int HW, int G, float eps,
                           const float* gamma, const float* beta, int silu, half_t* out, const float* st1, int tpb1,
                           const float* st2, int tpb2, float* scratch, hipStream_t st);
void norm_set_tuning_gn_inline_rows(int v);   // tuning "gn_inline_rows"
// GroupNorm (one launch, small maps) whose FIRST source is still the split-K slabs of its producer GEMM: channel c < C1 of pixel m is
// fp16(sum_z slab[z][m][c] + bias[c] + res[m][c]) -- the bits splitk_reduce_vec_kernel would have stored -- summed by the block that
// normalises it; that fp16 tensor is also written to sum_out when non-null (a later consumer needs it: residual / skip connection).
struct GnSlab { const float* slab; int splitk; size_t stride; const float* bias; const half_t* res; int ldres; half_t* sum_out; };
bool groupnorm_slab_ok(int C1, int C2, int HW, int G);      // shapes the one-launch kernel takes
int launch_groupnorm_slab(const GnSlab& sl, const half_t* x2, int C1, int C2, int B, int HW, int G, float eps, const float* gamma,
                          const float* beta, int silu, half_t* out, hipStream_t st);
int launch_layernorm(const half_t* x, int M, int C, float eps, const float* gamma, const float* beta, half_t* out,
                     hipStream_t st);
int launch_geglu(const half_t* x, int M, int I, half_t* out, hipStream_t st);       // x [M][2I] -> out [M][I]
int launch_softmax_rows(half_t* x, int M, int N, int ld, hipStream_t st);            // in place, fp32 internally
int launch_softmax_rows_f32(float* x, size_t M, int N, hipStream_t st);                  // in place, [M][N] contiguous
int launch_f32_rows_to_f16_padded(const float* in, size_t M, int N, int ld, half_t* out, hipStream_t st);
int launch_gemv(const float* x, int K, const half_t* W, int N, const float* bias, const float* bias2, int silu_in,
                float* out, hipStream_t st);
// weight repack (PyTorch layouts -> fp16 KRSC / head-padded rows, fp32 vectors)
// ilv_half > 0: rows [0, ilv_half) and [ilv_half, 2*ilv_half) are interleaved in groups of 32 (GEGLU x / gate pairing)
int launch_repack_matrix(const void* src, int src_f16, int rows, int cols, int taps, half_t* dst, int dst_ld, int cin_pad,
                         int row0, int dh, int Dp, hipStream_t st, int ilv_half = 0);
int launch_repack_vec(const void* src, int src_f16, int n, float* dst, hipStream_t st, int ilv_half = 0);
int launch_nchw_f32_to_nhwc_f16(const float* in, int B, int C, int HW, int Cp, half_t* out, hipStream_t st);
int launch_f32_to_f16(const float* in, size_t n, half_t* out, hipStream_t st);
int launch_img_u8_to_nhwc(const uint8_t* img, int n, int HW, int Cp, half_t* out, hipStream_t st);
int launch_dec_to_u8(const float* nchw, int n, int HW, uint8_t* out_hwc, hipStream_t st);
int launch_scale_f32(const float* in, size_t n, float s, float* out, hipStream_t st);
int launch_gather_rows_f32(const float* in, const int* rows, int nrows, size_t row_elems, float* out, hipStream_t st);
// CLIP text embeddings: out[m][:] = fp16(tok_emb[ids[m]] + pos_emb[m % T]); quick_gelu in place; fp16 -> fp32
int launch_embed_tokens(const int* ids, int M, int T, int H, int vocab, const half_t* tok_emb, const half_t* pos_emb, half_t* out,
                        hipStream_t st);
int launch_quick_gelu(half_t* x, size_t n, hipStream_t st);
int launch_f16_to_f32(const half_t* in, size_t n, float* out, hipStream_t st);

// ---------------------------------------------------------------------------------------------------------------
// Attention
// ---------------------------------------------------------------------------------------------------------------
// Key order of a "permuted" V^T (GemmP::vt_perm16 / AttnP::vt_perm): inside every aligned group of 16 tokens the two middle quads are
// swapped -- stored order [0-3, 8-11, 4-7, 12-15] -- so that the 8 keys one lane feeds to the P V MFMA (accumulator rows 4h + {0..3, 8..11}
// of a 16-key step) are ONE aligned 16-byte chunk: a single conflict-free ds_read_b128 instead of two 2-way-conflicted ds_read_b64.
__host__ __device__ inline int vt_perm16_pos(int tok) { return tok ^ ((((tok >> 2) ^ (tok >> 3)) & 1) ? 12 : 0); }

struct AttnP {
  const half_t* q; int ldq, q_off;
  const half_t* k; int ldk, k_off;
  const half_t* vt; int ldv;      // Vt[((row*heads + head)*Dp + d)*ldv + tok]
  half_t* o; int ldo;             // o[(row*Nq + tok)*ldo + head*dh + d]
  int heads, Nq, Nk, Dp, dh;
  float scale;
  const int* rows;                // device [nrows][4] = {out_row, q_row, k_row, v_row}
  int nrows;
  int causal = 0;                 // 1: key j is masked for query i when j > i (CLIP text encoder)
  int vt_perm = 0;                // 1: vt is stored in the permuted key order (only the 64-wide LDS-DMA self-attention kernel reads it)
  int aug = 0;                    // 1: column d = dh of every K row and every V row (V^T row dh) holds 1.0 (the producer's bias wrote it):
                                  //    the 64-wide LDS-DMA kernel then takes the max shift and the row sum through the MFMAs (attn.hip, AUG)
  float* lse = nullptr;           // optional [out_row][heads][Nq]: log2-domain log-sum-exp of every query (recording forward of the null-text path)
};
int launch_attn_flash(const AttnP& p, hipStream_t st);
int attn_set_tuning_pipe(int v);        // tuning "attn_pipe" (ablation builds only): the half-tile software-pipelined forms of the 64-wide LDS-DMA kernel
bool attn_flash_uses_dma64(int Dp, int Nk, int causal);

// Flash-style attention backward (null-text path): dQ / dK / dV of softmax(scale Q K^T) V without the [N][N] matrices in memory.
struct BwdMat { const half_t* p; long hs; int ld; int w; };   // (head, row, col) -> p[head * hs + row * ld + col]; columns >= w read as zero (w % 8 == 0)
struct AttnBwdP {
  BwdMat b1, b2;                  // the workgroup's own ("block") rows, held in registers: DQ: Q, dO;  DK: K, V;  DV: K
  BwdMat l1, l2;                  // the rows it walks ("loop" side, staged in LDS):        DQ: K, V;   DK: Q, dO; DV: Q, dO
                                  // (the transposed operand of the accumulation -- K^T, Q^T, dO^T -- is made from these tiles in LDS)
  int nb, nl, heads;              // block rows, loop rows
  float scale;
  float* lse;                     // [heads][queries] log2-domain log-sum-exp: written by DQ, read by DK / DV
  float* dsum;                    // [heads][queries] sum_k P dP                   (same)
  half_t* out; long out_hs; int out_ld, out_w;   // out[head * out_hs + block_row * out_ld + d], d < out_w
  const half_t* o = nullptr; long o_hs = 0; int o_ld = 0;   // DQ only, optional: the forward's output O (head * o_hs + query * o_ld + d) -- with it
                                  // `lse` is an INPUT (the forward kernel's) and D = rowsum(dO o O): the kernel's first pass over the keys is skipped
  int nsplit = 1;                 // DK / DV with few keys (cross-attention): the loop rows are split over nsplit workgroups per (key tile, head), each
  float* part = nullptr;          // writing fp32 partial sums part[((split * heads + head) * nb + block_row) * out_w + d]; summed in order by
};                                // launch_attn_bwd_reduce
int launch_attn_bwd_reduce(const AttnBwdP& p, hipStream_t st);
int launch_attn_bwd_flash(const AttnBwdP& p, int mode /*0 DQ, 1 DK, 2 DV*/, int Dp, hipStream_t st);   // -1: head width not instantiated

// Cross-attention with the Prompt-to-Prompt edit fused in (one (src,tgt) row pair per grid.z entry).
struct CrossEditP {
  const half_t* q; int ldq, q_off;
  const half_t* k; int ldk, k_off;
  const half_t* vt; int ldv;
  half_t* o; int ldo;
  int heads, Nq, Nk, Dp, dh;
  float scale;
  const int* pairs;               // device [npairs][2] = {src_row, tgt_row}
  int npairs;
  const half_t* mmatT;            // device [npairs][96][96] fp16: mmatT[j][w] = Mmat[w][j] (zero padded)
  const float* c1; const float* c2;   // device [npairs][96] blend coefficients for the current step
  const float* lb_alpha;          // device [npairs][lb_planes][96] LocalBlend token selectors (nullable)
  float* lb_acc;                  // device [npairs][nslots][lb_planes][Nq] accumulators (nullable)
  int lb_planes = 2;              // 2: {src, tgt} blend-word selectors; 4: + {src, tgt} substruct-word selectors (LocalBlend substruct_words)
  int lb_slot0, lb_nslots;        // this layer's first slot (slot = lb_slot0 + head)
  int write_src;                  // also store the source row's output (tests); the executor leaves that to the flash kernel
};
int launch_attn_cross_edit(const CrossEditP& p, hipStream_t st);

// ---------------------------------------------------------------------------------------------------------------
// Scheduler / latent step kernels (fp32, NCHW, bit-exact w.r.t. the reference's elementwise order)
// ---------------------------------------------------------------------------------------------------------------
// x_next = sqrt(a_next) * ((x - sqrt(1-a_t) * eps) / sqrt(a_t)) + sqrt(1-a_next) * eps
int launch_ddim_move(const float* x, const float* eps, float a_from, float a_to, size_t n, float* out, hipStream_t st);
// Classifier-free guidance + DDIM denoise step (+ direct-inversion offset). See include/pnpi.h pnpi_cfg_ddim_prev.
int launch_ddim_prev_recon(const float* x, const float* eps, float a_from, float a_to, const float* ref, float lr, const float* mask, size_t n,
                           float* out, float* x0_out, hipStream_t st);
int launch_cfg_ddim_prev(const float* eps, const float* x, int nimg, int rows_per_img, size_t row_elems, float gscale,
                         float a_t, float a_prev, const float* noise_loss, int offset_rows, const float* target,
                         float offset_scale, float* offset_out, float* x_out, hipStream_t st, const float* prox_thr = nullptr,
                         int prox_mode = 0, const float* recon_ref = nullptr, float recon_lr = 0.f, int dilate = 0, int lat_h = 0,
                         int lat_w = 0, const float* inv_ref = nullptr);   // recon_ref [nimg][row_elems]: reconstruction guidance; inv_ref
                                                                          // [nimg][row_elems]: inversion guidance (both need prox_mode)
// threshold of the proximal-guidance step: quantile q of |eps_c - eps_u| over the rows of each image (torch.quantile, linear)
int launch_quantile_abs_diff(const float* eps, int nimg, int rows_per_img, size_t row_elems, float q, float* thr_out, hipStream_t st);
int launch_fill_f32(float* p, int n, float v, hipStream_t st);
int launch_local_blend(const float* lb_acc, int nslots, int map_hw, int lat_hw, int C, float th, float* latents,
                       int nimg, hipStream_t st, int planes = 2, float th_sub = 0.3f);   // planes 4: planes 2, 3 are the substruct maps (no pooling, th_sub)

// ---------------------------------------------------------------------------------------------------------------
// Activation-gradient kernels of the null-text path (bwd.hip)
// ---------------------------------------------------------------------------------------------------------------
int launch_layernorm_bwd(const half_t* x, const half_t* dy, int M, int C, float eps, const float* gamma, half_t* dx, hipStream_t st);
int launch_groupnorm_bwd(const half_t* x1, const half_t* x2, int C1, int C2, int B, int HW, int G, float eps, const float* gamma,
                         const float* beta, int silu, const half_t* dy, half_t* dx, hipStream_t st);
// Three chip-wide phases instead of one block per group; dx goes straight into the gradient buffers of the two concat sources
// (p == nullptr: that source needs no gradient; acc: add to what is there).  scratch: groupnorm_bwd2_scratch_floats() floats.
struct GnbOut { half_t* p; int ld; int acc; };
size_t groupnorm_bwd2_scratch_floats(int B, int HW, int G);
int launch_groupnorm_bwd2(const half_t* x1, const half_t* x2, int C1, int C2, int B, int HW, int G, float eps, const float* gamma, const float* beta,
                          int silu, const half_t* dy, GnbOut o1, GnbOut o2, float* scratch, hipStream_t st);
int launch_geglu_bwd(const half_t* h, const half_t* dy, int M, int I, half_t* dh, hipStream_t st);
int launch_softmax_bwd_rows(const float* P, const float* dP, size_t R, int N, int ld, float scale, half_t* dS, hipStream_t st);
int launch_accumulate_f16(half_t* dst, const half_t* src, size_t n, hipStream_t st);
int launch_sumpool2x2(const half_t* dup, int B, int H, int W, int C, half_t* dx, hipStream_t st);
int launch_zero_stuff2(const half_t* dy, int B, int Ho, int Wo, int C, half_t* out, hipStream_t st);
int launch_repack_dgrad(const half_t* w, int N, int Npad, int taps, int Cin, half_t* wd, hipStream_t st);
int launch_strided_add_f16(half_t* dst, const half_t* src, int ld, int off, size_t R, int C, int accumulate, hipStream_t st);
int launch_add_f16_to_f32(float* dst, const half_t* src, size_t n, float scale, hipStream_t st);
int launch_null_text_loss(const float* eps_u, const float* eps_c, const float* x, const float* target, int n, float w, float c_x, float c_e,
                          float grad_scale, float* d_eps_u, float* loss, hipStream_t st);
int launch_adam_step(float* p, float* m, float* v, const float* g, int n, int k, float lr, float inv_scale, hipStream_t st);
int launch_pad_heads_f16(const half_t* src, size_t R, int heads, int dh, int Dp, half_t* dst, hipStream_t st);
int launch_transpose_f16(const half_t* src, int ld_src, int R, int Cc, half_t* dst, int ld_dst, hipStream_t st, int nbatch = 1, long s_src = 0,
                         long s_dst = 0);   // nbatch matrices, s_src / s_dst elements apart
