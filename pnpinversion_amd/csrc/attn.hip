// Attention on MFMA for the SD-1.x head sizes (40/80/160 padded to 64/96/160), with the Prompt-to-Prompt controller fused in.
//
// Reference: the hooked CrossAttention forward, models/p2p/attention_control.py:20-47
//   (sim = q k^T * scale -> softmax -> controller(attn) -> attn v), and the controllers
//   AttentionControl.__call__ :178-190, AttentionControlEdit.forward :269-282, AttentionReplace/Refine/Reweight :301-363,
//   LocalBlend map accumulation via AttentionStore :221-234.
//
// Dataflow (per wavefront = 32 query rows, swapped-operand form):
//   S^T = K Q^T   : MFMA A-operand = K tile rows (keys), B-operand = Q rows  -> lane l owns query (l & 31) and 16 of the
//                   32 keys of a tile in its accumulator registers (the other 16 live in lane l ^ 32).
//   softmax       : in registers, one __shfl_xor(.,32) per reduction; running max / sum (flash style) for self-attention.
//   O^T = V^T P^T : A-operand = V^T rows (head-dim), B-operand = P straight from the accumulator registers (the MFMA k-slot
//                   permutation is applied to the V^T fragment instead, two ds_read_b64 per fragment), so P never touches LDS.
// Score matrices are never materialised (the reference materialises [B*8, N, N] fp32: 2.1 GB at 64x64).
// Self-attention replacement ("tgt probabilities := src probabilities", attention_control.py:258-263) costs nothing here:
// the target row simply takes Q and K from the source row (rows[] indirection) and keeps its own V.
#include <stdlib.h>

#include "ops.h"

static constexpr int KV_TILE = 64;
static constexpr int SV_LD = KV_TILE + 4;  // halfs; 136-byte rows: conflict-free ds_read_b64 over 32 rows

// PF: prefetch the next K / V^T tile into registers under the current tile's work (self-attention: many key tiles, few workgroups);
// without it the tile is loaded and stored in one go (cross-attention: two tiles, thousands of workgroups -- the extra registers of the
// prefetch would only cost occupancy there).
template <int DP, bool PF>
__global__ void __launch_bounds__(256) attn_flash_kernel(AttnP p) {
  constexpr int KS = DP / 16;  // k-steps over the head dim for S
  constexpr int OT = DP / 32;  // 32-wide output tiles over the head dim
  constexpr int SK_LD = DP + 8;
  __shared__ __attribute__((aligned(16))) half_t sK[KV_TILE * SK_LD];
  __shared__ __attribute__((aligned(16))) half_t sV[DP * SV_LD];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, ql = lane & 31;
  // XCD-aware order: workgroup ids are dealt round-robin to the 8 XCDs; give each XCD a contiguous run of (row, head)
  // groups so that one group's K / V^T stay in one L2 instead of being fetched by all eight.
  const int nqt = (p.Nq + 127) >> 7, T = nqt * p.heads * p.nrows, per = (T + 7) >> 3;
  const int tix = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
  if (tix >= T) return;
  const int qt = tix % nqt, head = (tix / nqt) % p.heads;
  const int* rw = p.rows + (tix / (nqt * p.heads)) * 4;
  const int orow = rw[0], qrow = rw[1], krow = rw[2], vrow = rw[3];
  const int qtok = qt * 128 + wave * 32 + ql;
  const bool qok = qtok < p.Nq;

  half8 qf[KS];
  {
    const half_t* qp = p.q + ((size_t)qrow * p.Nq + (qok ? qtok : 0)) * p.ldq + p.q_off + head * DP + h * 8;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[ks] = qok ? ldg_half8(qp + ks * 16) : zero_half8();
  }
  floatx16 O[OT];
#pragma unroll
  for (int ot = 0; ot < OT; ++ot)
#pragma unroll
    for (int r = 0; r < 16; ++r) O[ot][r] = 0.f;
  float mrun = -INFINITY, lrun = 0.f;
  const float c = p.scale * 1.44269504088896340736f;

  const half_t* kbase = p.k + (size_t)krow * p.Nk * p.ldk + p.k_off + head * DP;
  const half_t* vbase = p.vt + ((size_t)vrow * p.heads + head) * DP * (size_t)p.ldv;

  // One K / V^T tile per thread in registers: the NEXT tile's global loads are issued right after this tile was published in LDS and
  // fly under its MFMAs and softmax (the loop used to load, store, synchronise and only then compute: every 64-key tile paid a full
  // global round trip -- 16 of them in a row at the 32 x 32 level, with at most a few blocks per CU to cover for each other).
  constexpr int NKR = KV_TILE * (DP / 8) / 256, NVR = DP * (KV_TILE / 8) / 256;
  static_assert(KV_TILE * (DP / 8) % 256 == 0 && DP * (KV_TILE / 8) % 256 == 0, "a tile is a whole number of 16-byte vectors per thread");
  half8 rk[NKR], rv[NVR];
  auto gload = [&](int kv0) {
#pragma unroll
    for (int i = 0; i < NKR; ++i) {
      const int idx = tid + i * 256, r = idx / (DP / 8), v = idx - r * (DP / 8);
      const int tok = kv0 + r;
      rk[i] = tok < p.Nk ? ldg_half8(kbase + (size_t)tok * p.ldk + v * 8) : zero_half8();
    }
#pragma unroll
    for (int i = 0; i < NVR; ++i) {
      const int idx = tid + i * 256, d = idx >> 3, v = idx & 7;
      const int tok0 = kv0 + v * 8;
      const half_t* src = vbase + (size_t)d * p.ldv + tok0;
      if (tok0 + 8 <= p.Nk) {
        rv[i] = ldg_half8(src);
      } else {
        half8 val;
#pragma unroll
        for (int j = 0; j < 8; ++j) val[j] = (tok0 + j < p.Nk) ? src[j] : (half_t)0.f;
        rv[i] = val;
      }
    }
  };
  if (PF) gload(0);
  for (int kv0 = 0; kv0 < p.Nk; kv0 += KV_TILE) {
    __syncthreads();                    // every wave is done reading the previous tile
    if (!PF) gload(kv0);
#pragma unroll
    for (int i = 0; i < NKR; ++i) {
      const int idx = tid + i * 256, r = idx / (DP / 8), v = idx - r * (DP / 8);
      *reinterpret_cast<half8*>(sK + r * SK_LD + v * 8) = rk[i];
    }
#pragma unroll
    for (int i = 0; i < NVR; ++i) {
      const int idx = tid + i * 256, d = idx >> 3, v = idx & 7;
      const half8 val = rv[i];
      half4 lo = {val[0], val[1], val[2], val[3]}, hi = {val[4], val[5], val[6], val[7]};
      *reinterpret_cast<half4*>(sV + d * SV_LD + v * 8) = lo;
      *reinterpret_cast<half4*>(sV + d * SV_LD + v * 8 + 4) = hi;
    }
    __syncthreads();
    if (PF && kv0 + KV_TILE < p.Nk) gload(kv0 + KV_TILE);

    floatx16 s[2];
#pragma unroll
    for (int st = 0; st < 2; ++st) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[st][r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        half8 kf = *reinterpret_cast<const half8*>(sK + (st * 32 + ql) * SK_LD + ks * 16 + h * 8);
        s[st] = mfma32(kf, qf[ks], s[st]);
      }
    }
    // Only the last key tile can hold out-of-range keys: the mask is a wave-uniform branch, not per-tile VALU work.
    if (kv0 + KV_TILE > p.Nk) {
#pragma unroll
      for (int st = 0; st < 2; ++st)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (kv0 + st * 32 + acc_row(r, lane) >= p.Nk) s[st][r] = -INFINITY;
    }
    if (p.causal) {   // wave-uniform flag: key j > query i is masked (every query keeps key 0, so the running max stays finite)
#pragma unroll
      for (int st = 0; st < 2; ++st)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (kv0 + st * 32 + acc_row(r, lane) > qtok) s[st][r] = -INFINITY;
    }
    float mloc = s[0][0];
#pragma unroll
    for (int st = 0; st < 2; ++st)
#pragma unroll
      for (int r = 0; r < 16; ++r) mloc = fmaxf(mloc, s[st][r]);
    mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
    // Deferred rescale (flash-attention style): the running reference max only moves when some row's tile max exceeds it by
    // more than 8 in the exp2 domain, so P <= 2^8 (fp16 has the range) and O / l are rescaled on few tiles only.  The
    // decision precedes this tile's exponentials and every earlier P.V is already in O, so all terms share one scale.
    const bool grow = (mloc - mrun) * c > 8.0f;
    if (__any(grow)) {
      const float mnew = fmaxf(mrun, mloc);
      const float alpha = __builtin_amdgcn_exp2f((mrun - mnew) * c);   // mrun = -inf on the first tile -> 0
      lrun *= alpha;
#pragma unroll
      for (int ot = 0; ot < OT; ++ot)
#pragma unroll
        for (int r = 0; r < 16; ++r) O[ot][r] *= alpha;
      mrun = mnew;
    }
    const float mc = mrun * c;
    float psum = 0.f;
    half8 pf[2][2];
#pragma unroll
    for (int st = 0; st < 2; ++st)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(s[st][r], c, -mc));
        psum += pv;
        pf[st][r >> 3][r & 7] = (half_t)pv;
      }
    lrun += psum;
#pragma unroll
    for (int ot = 0; ot < OT; ++ot) {
#pragma unroll
      for (int st = 0; st < 2; ++st)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const half_t* vb = sV + (ot * 32 + ql) * SV_LD + st * 32 + t * 16 + 4 * h;
          half4 lo = *reinterpret_cast<const half4*>(vb);
          half4 hi = *reinterpret_cast<const half4*>(vb + 8);
          half8 vf = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          O[ot] = mfma32(vf, pf[st][t], O[ot]);
        }
    }
  }
  const float ltot = lrun + __shfl_xor(lrun, 32, 64);
  const float inv = 1.f / ltot;
  // recording forward of the null-text path: log2 sum_k 2^(c S) per query, for the backward kernel (mrun is the reference exponent of lrun)
  if (p.lse && qok && h == 0) p.lse[((size_t)orow * p.heads + head) * p.Nq + qtok] = mrun * c + __log2f(ltot);
  if (qok) {
    half_t* op = p.o + ((size_t)orow * p.Nq + qtok) * p.ldo + head * p.dh;
#pragma unroll
    for (int ot = 0; ot < OT; ++ot)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        int d = ot * 32 + 8 * g + 4 * h;
        if (d + 3 < p.dh) {
          half4 o4 = {(half_t)(O[ot][4 * g] * inv), (half_t)(O[ot][4 * g + 1] * inv), (half_t)(O[ot][4 * g + 2] * inv),
                      (half_t)(O[ot][4 * g + 3] * inv)};
          *reinterpret_cast<half4*>(op + d) = o4;
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (d + j < p.dh) op[d + j] = (half_t)(O[ot][4 * g + j] * inv);
        }
      }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Self-attention fast path for the 64-wide (padded d = 40) heads with Nk % 64 == 0 -- the 4096-token layers that dominate
// attention time.  Same math as attn_flash_kernel<64>; K / V^T tiles are staged by asynchronous LDS-DMA into a two-stage
// ring with the XOR-swizzled 128-byte-row layout of gemm.hip (byte(R, s) = (R>>4)*2048 + (R&7)*256 + ((R>>3)&1)*128 +
// ((s ^ (R&7))*16)), so the next tile flies under the current tile's MFMAs and no staging VGPRs / ds_writes are spent.
// ------------------------------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

// KSQ: 16-wide k-steps of S = K Q^T actually run -- 3 when the head is at most 48 wide (d = 40: columns 48 .. 63 of q and k are zero
// padding, their products exact zeros): a quarter of the S MFMAs and K fragment reads, bit-identical results.
//
// AUG (round 4; d = 40 heads, AttnP::aug): the kernel is VALU-bound -- per 64-key tile a wave issues 14 MFMAs (448 cycles) but ~940 cycles of
// VALU work on its 32 scores per lane (quarter-rate v_exp_f32 528, the exponent's fma 124, the row-sum adds 132, max / convert 160).  Two of
// those passes move into the MFMAs through ONE padding dimension of the 40 -> 64 head (the producer projection's bias writes 1.0 into column
// d = 40 of every K row and every V row, api.hip b_qkv_aug):
//   * Q is pre-scaled by scale * log2(e) and carries -m_run in its column 40, so S' = K Q'^T comes out of the matrix pipe already in the
//     log2 domain and already shifted by the running maximum: P = exp2(S') with no per-element fma.  m_run is kept exactly representable
//     in fp16 (it is a reference point, not the true maximum: the deferred-rescale threshold of 8 covers the rounding);
//   * V^T row 40 is all ones, so O^T row 40 accumulates sum_k P -- with the same fp16-rounded P the numerator uses: no per-element add.
// Same results to rounding (Q is rounded once more after the fp32 scaling: independent half-ulp errors, no bias); ~30 % fewer VALU cycles per tile.
template <bool VPERM, int KSQ = 4, bool AUG = false>
__global__ void __launch_bounds__(256) attn_flash_dma64_kernel(AttnP p) {
  constexpr int DP = 64, KS = 4, OT = 2;
  constexpr int STAGE = 2 * 64 * 128;   // K tile + V^T tile, bytes
  __shared__ __attribute__((aligned(1024))) char smem[2 * STAGE];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, ql = lane & 31;
  // XCD-aware order: workgroup ids are dealt round-robin to the 8 XCDs; give each XCD a contiguous run of (row, head)
  // groups so that one group's K / V^T stay in one L2 instead of being fetched by all eight.
  const int nqt = (p.Nq + 127) >> 7, T = nqt * p.heads * p.nrows, per = (T + 7) >> 3;
  const int tix = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
  if (tix >= T) return;
  const int qt = tix % nqt, head = (tix / nqt) % p.heads;
  const int* rw = p.rows + (tix / (nqt * p.heads)) * 4;
  const int orow = rw[0], qrow = rw[1], krow = rw[2], vrow = rw[3];
  const int qtok = qt * 128 + wave * 32 + ql;
  const bool qok = qtok < p.Nq;

  half8 qf[KSQ];
  {
    const half_t* qp = p.q + ((size_t)qrow * p.Nq + (qok ? qtok : 0)) * p.ldq + p.q_off + head * DP + h * 8;
#pragma unroll
    for (int ks = 0; ks < KSQ; ++ks) qf[ks] = qok ? ldg_half8(qp + ks * 16) : zero_half8();
  }
  const float c = p.scale * 1.44269504088896340736f;
  if (AUG) {
    // S in the log2 domain straight from the MFMA.  The factor is applied in fp32 and the product rounded once: an fp16 factor
    // (0.228149 for 0.22811 at d = 40) would put a correlated +1.7e-4 relative error -- a temperature bias -- on every logit
#pragma unroll
    for (int ks = 0; ks < KSQ; ++ks)
#pragma unroll
      for (int j = 0; j < 8; ++j) qf[ks][j] = (half_t)((float)qf[ks][j] * c);
  }
  floatx16 O[OT];
#pragma unroll
  for (int ot = 0; ot < OT; ++ot)
#pragma unroll
    for (int r = 0; r < 16; ++r) O[ot][r] = 0.f;
  float mrun = AUG ? 0.f : -INFINITY, lrun = 0.f;     // AUG: the reference point carried in Q's column 40 (log2 domain, fp16-representable)

  // DMA lane roles: instruction j (1 KiB = 8 rows) of a 64-row tile; this wave issues j = 2*wave, 2*wave+1 for K and for V^T
  const half_t* kptr[2];
  const half_t* vptr[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int j = wave * 2 + i;
    const int line = 4 * (j & 1) + (lane >> 4);
    const int R = (j >> 1) * 16 + ((lane >> 3) & 1) * 8 + line;
    const int ch = (lane & 7) ^ line;
    kptr[i] = p.k + ((size_t)krow * p.Nk + R) * p.ldk + p.k_off + head * DP + ch * 8;            // + kv0 * ldk per tile
    vptr[i] = p.vt + (((size_t)vrow * p.heads + head) * DP + R) * (size_t)p.ldv + ch * 8;        // + kv0 per tile
  }
  auto issue = [&](int buf, int kv0) {
    char* sK = smem + buf * STAGE;
    char* sV = sK + 64 * 128;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(kptr[i] + (size_t)kv0 * p.ldk), (lds_ptr_t)(sK + (wave * 2 + i) * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(vptr[i] + kv0), (lds_ptr_t)(sV + (wave * 2 + i) * 1024), 16, 0, 0);
    }
  };
  const int x7 = ql & 7;
  const int lane_row_off = (ql >> 4) * 2048 + (ql & 7) * 256 + ((ql >> 3) & 1) * 128;

  const int ntiles = p.Nk / 64;
  issue(0, 0);
  for (int it = 0; it < ntiles; ++it) {
    const int buf = it & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (it + 1 < ntiles) issue(buf ^ 1, (it + 1) * 64);
    const char* sK = smem + buf * STAGE + lane_row_off;
    const char* sV = sK + 64 * 128;

    floatx16 s[2];
#pragma unroll
    for (int st = 0; st < 2; ++st) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[st][r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < KSQ; ++ks) {
        half8 kf = *reinterpret_cast<const half8*>(sK + st * 4096 + (((ks * 2 + h) ^ x7) * 16));
        s[st] = mfma32(kf, qf[ks], s[st]);
      }
    }
    float mloc = s[0][0];
#pragma unroll
    for (int st = 0; st < 2; ++st)
#pragma unroll
      for (int r = 0; r < 16; ++r) mloc = fmaxf(mloc, s[st][r]);
    mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
    half8 pf[2][2];
    float psum = 0.f;
    if (AUG) {
      // s is S' = c S - m_run already.  The first tile always moves the reference point (its scores were computed against 0), later tiles
      // when the tile's maximum is more than 8 above it; the new point is rounded to fp16 so that Q's column 40 holds it exactly.
      const bool grow = it == 0 || mloc > 8.0f;
      float delta = 0.f;
      if (__any(grow)) {
        const float target = mrun + ((it == 0 || mloc > 0.f) ? mloc : 0.f);
        const half_t mh = (half_t)target;
        const float mnew = (float)mh;
        delta = mnew - mrun;                                          // exact: both ends are fp16 values
        const float alpha = __builtin_amdgcn_exp2f(-delta);
#pragma unroll
        for (int ot = 0; ot < OT; ++ot)
#pragma unroll
          for (int r = 0; r < 16; ++r) O[ot][r] *= alpha;             // row 40 (the running sum) included
        mrun = mnew;
        if (h == 1) qf[2][0] = (half_t)(-mnew);                       // column 40 = k-step 2, upper half-wave, element 0
#pragma unroll
        for (int st = 0; st < 2; ++st)
#pragma unroll
          for (int r = 0; r < 16; ++r) s[st][r] -= delta;             // this tile was computed against the old point (rare path)
      }
#pragma unroll
      for (int st = 0; st < 2; ++st)
#pragma unroll
        for (int r = 0; r < 16; ++r) pf[st][r >> 3][r & 7] = (half_t)__builtin_amdgcn_exp2f(s[st][r]);
    } else {
    const bool grow = (mloc - mrun) * c > 8.0f;     // deferred rescale, see attn_flash_kernel
    if (__any(grow)) {
      const float mnew = fmaxf(mrun, mloc);
      const float alpha = __builtin_amdgcn_exp2f((mrun - mnew) * c);
      lrun *= alpha;
#pragma unroll
      for (int ot = 0; ot < OT; ++ot)
#pragma unroll
        for (int r = 0; r < 16; ++r) O[ot][r] *= alpha;
      mrun = mnew;
    }
    const float mc = mrun * c;
#pragma unroll
    for (int st = 0; st < 2; ++st)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(s[st][r], c, -mc));
        psum += pv;
        pf[st][r >> 3][r & 7] = (half_t)pv;
      }
    }
    lrun += psum;
#pragma unroll
    for (int ot = 0; ot < OT; ++ot) {
#pragma unroll
      for (int st = 0; st < 2; ++st)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          // keys st*32 + t*16 + 4h + [0,4) and + 8: 16-byte chunks 4*st + 2*t and + 1 of row d, sub-offset 8h bytes
          const int c0 = st * 4 + t * 2;
          half8 vf;
          if (VPERM) {
            // permuted key order (vt_perm16_pos): chunk c0 + h IS this lane's 8 keys, in accumulator-row order
            vf = *reinterpret_cast<const half8*>(sV + ot * 4096 + (((c0 + h) ^ x7) * 16));
          } else {
            half4 lo = *reinterpret_cast<const half4*>(sV + ot * 4096 + ((c0 ^ x7) * 16) + 8 * h);
            half4 hi = *reinterpret_cast<const half4*>(sV + ot * 4096 + (((c0 + 1) ^ x7) * 16) + 8 * h);
            vf = half8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          }
          O[ot] = mfma32(vf, pf[st][t], O[ot]);
        }
    }
  }
  float ltot;
  if (AUG) {                       // O^T row 40 = accumulator 4 of the second 32-row tile in the lower half-wave
    const float lrow = O[1][4], lother = __shfl_xor(lrow, 32, 64);
    ltot = h == 0 ? lrow : lother;
  } else {
    ltot = lrun + __shfl_xor(lrun, 32, 64);
  }
  const float inv = 1.f / ltot;
  // recording forward of the null-text path: log2 sum_k 2^(c S) per query, for the backward kernel (mrun is the reference exponent of lrun)
  if (p.lse && qok && h == 0) p.lse[((size_t)orow * p.heads + head) * p.Nq + qtok] = (AUG ? mrun : mrun * c) + __log2f(ltot);
  if (qok) {
    half_t* op = p.o + ((size_t)orow * p.Nq + qtok) * p.ldo + head * p.dh;
#pragma unroll
    for (int ot = 0; ot < OT; ++ot)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        int d = ot * 32 + 8 * g + 4 * h;
        if (d + 3 < p.dh) {
          half4 o4 = {(half_t)(O[ot][4 * g] * inv), (half_t)(O[ot][4 * g + 1] * inv), (half_t)(O[ot][4 * g + 2] * inv),
                      (half_t)(O[ot][4 * g + 3] * inv)};
          *reinterpret_cast<half4*>(op + d) = o4;
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (d + j < p.dh) op[d + j] = (half_t)(O[ot][4 * g + j] * inv);
        }
      }
  }
}

#ifdef PNPI_ABLATIONS      // `python -m pnpinversion_amd.build --ablations` only: a measured-and-not-kept arm, kept reproducible (profiles/README.md, round 6)
// EXPERIMENT (round 6, tuning "attn_pipe" = 1 / 2): attn_flash_dma64_kernel<true, 3, true> software-pipelined at HALF-tile
// granularity inside each wave.  A 64-key tile is two 32-key halves; in half g the wave issues the three S' = K Q'^T MFMAs of half g + 1
// (independent of everything else in the half), exponentiates the scores of half g and issues the four V^T P MFMAs of half g -- so that the
// quarter-rate v_exp_f32 stream sits between MFMAs of its OWN instruction stream (the only place the matrix pipe and the VALU of a SIMD
// overlap: tools/micro/mfma_valu_overlap) without holding a second full tile of scores (16 + 16 score registers = today's 32).  The
// reference-point test of the AUG form runs per half; a change is applied before the next half's scores are issued, so nothing computed
// ahead needs a correction.  K and V^T tiles ride separate two-slot rings: K(it) is read in halves 2it - 1 and 2it, V^T(it) in 2it and
// 2it + 1, so K(it + 2) is requested at the top of half 2it + 1 and V^T(it + 2) at the top of half 2it + 2, each two halves before its
// first read; one counted vmcnt + one barrier per half.
// the placement of one half's main block: every MFMA followed by its share of the half's VALU work (16 v_exp_f32, 8 v_cvt_pk, the
// maximum chain of the next half's scores), the fragment reads ahead of their MFMAs.  Masks: 0x008 MFMA, 0x002 VALU, 0x100 DS read.
#define SGB(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)
#define PIPE_SCHED()                                                     \
  do {                                                                   \
    SGB(0x100, 3);                                                       \
    SGB(0x008, 1); SGB(0x002, 4);                                        \
    SGB(0x008, 1); SGB(0x002, 4);                                        \
    SGB(0x008, 1); SGB(0x100, 2); SGB(0x002, 6);                         \
    SGB(0x008, 1); SGB(0x100, 2); SGB(0x002, 5);                         \
    SGB(0x008, 1); SGB(0x002, 5);                                        \
    SGB(0x008, 1); SGB(0x002, 5);                                        \
    SGB(0x008, 1); SGB(0x002, 8);                                        \
  } while (0)
template <int PLACE>      // 0: the compiler's own schedule of the half; 1: the sched_group_barrier placement above
__global__ void __launch_bounds__(256) attn_flash_pipe64_kernel(AttnP p) {
  constexpr int DP = 64;
  constexpr int KT = 64 * 128;          // one K tile or one V^T tile, bytes
  __shared__ __attribute__((aligned(1024))) char smem[4 * KT];      // [K slot 0 | V slot 0 | K slot 1 | V slot 1]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, ql = lane & 31;
  const int nqt = (p.Nq + 127) >> 7, T = nqt * p.heads * p.nrows, per = (T + 7) >> 3;
  const int tix = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
  if (tix >= T) return;
  const int qt = tix % nqt, head = (tix / nqt) % p.heads;
  const int* rw = p.rows + (tix / (nqt * p.heads)) * 4;
  const int orow = rw[0], qrow = rw[1], krow = rw[2], vrow = rw[3];
  const int qtok = qt * 128 + wave * 32 + ql;
  const bool qok = qtok < p.Nq;
  half8 qf[3];
  {
    const half_t* qp = p.q + ((size_t)qrow * p.Nq + (qok ? qtok : 0)) * p.ldq + p.q_off + head * DP + h * 8;
    const float c = p.scale * 1.44269504088896340736f;
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) {
      qf[ks] = qok ? ldg_half8(qp + ks * 16) : zero_half8();
#pragma unroll
      for (int j = 0; j < 8; ++j) qf[ks][j] = (half_t)((float)qf[ks][j] * c);
    }
  }
  floatx16 O[2];
#pragma unroll
  for (int ot = 0; ot < 2; ++ot)
#pragma unroll
    for (int r = 0; r < 16; ++r) O[ot][r] = 0.f;
  float mrun = 0.f;
  const half_t* kptr[2];
  const half_t* vptr[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int j = wave * 2 + i;
    const int line = 4 * (j & 1) + (lane >> 4);
    const int R = (j >> 1) * 16 + ((lane >> 3) & 1) * 8 + line;
    const int ch = (lane & 7) ^ line;
    kptr[i] = p.k + ((size_t)krow * p.Nk + R) * p.ldk + p.k_off + head * DP + ch * 8;
    vptr[i] = p.vt + (((size_t)vrow * p.heads + head) * DP + R) * (size_t)p.ldv + ch * 8;
  }
  auto issue_k = [&](int slot, int kv0) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(kptr[i] + (size_t)kv0 * p.ldk), (lds_ptr_t)(smem + slot * 2 * KT + (wave * 2 + i) * 1024), 16, 0, 0);
  };
  auto issue_v = [&](int slot, int kv0) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(vptr[i] + kv0), (lds_ptr_t)(smem + slot * 2 * KT + KT + (wave * 2 + i) * 1024), 16, 0, 0);
  };
  const int x7 = ql & 7;
  const int lane_row_off = (ql >> 4) * 2048 + (ql & 7) * 256 + ((ql >> 3) & 1) * 128;
  auto scores = [&](int slot, int st) {
    const char* sK = smem + slot * 2 * KT + lane_row_off + st * 4096;
    floatx16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) s = mfma32(*reinterpret_cast<const half8*>(sK + (((ks * 2 + h) ^ x7) * 16)), qf[ks], s);
    return s;
  };
  auto rowmax = [&](const floatx16& s) {
    float m = s[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) m = fmaxf(m, s[r]);
    return fmaxf(m, __shfl_xor(m, 32, 64));
  };
  auto move_reference = [&](floatx16& s, float mloc, bool first) {      // as in attn_flash_dma64_kernel's AUG branch, on a 32-key half
    const bool grow = first || mloc > 8.0f;
    if (__any(grow)) {
      const float target = mrun + ((first || mloc > 0.f) ? mloc : 0.f);
      const half_t mh = (half_t)target;
      const float mnew = (float)mh;
      const float delta = mnew - mrun;
      const float alpha = __builtin_amdgcn_exp2f(-delta);
#pragma unroll
      for (int ot = 0; ot < 2; ++ot)
#pragma unroll
        for (int r = 0; r < 16; ++r) O[ot][r] *= alpha;
      mrun = mnew;
      if (h == 1) qf[2][0] = (half_t)(-mnew);
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] -= delta;
    }
  };
  auto exp_pv = [&](const floatx16& s, int slot, int st) {
    half8 pf[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) pf[r >> 3][r & 7] = (half_t)__builtin_amdgcn_exp2f(s[r]);
    const char* sV = smem + slot * 2 * KT + KT + lane_row_off;
#pragma unroll
    for (int ot = 0; ot < 2; ++ot)
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int c0 = st * 4 + t * 2;
        O[ot] = mfma32(*reinterpret_cast<const half8*>(sV + ot * 4096 + (((c0 + h) ^ x7) * 16)), pf[t], O[ot]);
      }
  };
  const int nt = p.Nk / 64;
  issue_k(0, 0);
  issue_v(0, 0);
  if (nt > 1) { issue_k(1, 64); asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
  else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  floatx16 sa = scores(0, 0), sb;
  float mloc = rowmax(sa);
  for (int it = 0; it < nt; ++it) {
    const int slot = it & 1;
    const bool more = it + 1 < nt;
    // ---- half 2 it: V^T(it) lands; V^T(it + 1) is requested into the other slot (V^T(it - 1) was last read in half 2 it - 1)
    if (more) asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (more) issue_v(slot ^ 1, (it + 1) * 64);
    move_reference(sa, mloc, it == 0);
    sb = scores(slot, 1);
    exp_pv(sa, slot, 0);
    mloc = rowmax(sb);
    if (PLACE == 1) PIPE_SCHED();
    // ---- half 2 it + 1: K(it + 1) lands; K(it + 2) is requested into K(it)'s slot (last read just above)
    if (more) asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (it + 2 < nt) issue_k(slot, (it + 2) * 64);
    move_reference(sb, mloc, false);
    if (more) sa = scores(slot ^ 1, 0);
    exp_pv(sb, slot, 1);
    if (more) mloc = rowmax(sa);
    if (PLACE == 1) PIPE_SCHED();
  }
  const float lrow = O[1][4], lother = __shfl_xor(lrow, 32, 64);
  const float ltot = h == 0 ? lrow : lother;
  const float inv = 1.f / ltot;
  if (p.lse && qok && h == 0) p.lse[((size_t)orow * p.heads + head) * p.Nq + qtok] = mrun + __log2f(ltot);
  if (qok) {
    half_t* op = p.o + ((size_t)orow * p.Nq + qtok) * p.ldo + head * p.dh;
#pragma unroll
    for (int ot = 0; ot < 2; ++ot)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d = ot * 32 + 8 * g + 4 * h;
        if (d + 3 < p.dh) {
          half4 o4 = {(half_t)(O[ot][4 * g] * inv), (half_t)(O[ot][4 * g + 1] * inv), (half_t)(O[ot][4 * g + 2] * inv), (half_t)(O[ot][4 * g + 3] * inv)};
          *reinterpret_cast<half4*>(op + d) = o4;
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (d + j < p.dh) op[d + j] = (half_t)(O[ot][4 * g + j] * inv);
        }
      }
  }
}
static int g_attn_pipe = 0;
int attn_set_tuning_pipe(int v) { g_attn_pipe = v; return 0; }
#else
static const int g_attn_pipe = 0;
int attn_set_tuning_pipe(int v) { return v == 0 ? 0 : -2; }      // the instance exists only in the ablation library
#endif

// which launches the 64-wide LDS-DMA kernel takes (the producer of V^T may then write the permuted key order)
bool attn_flash_uses_dma64(int Dp, int Nk, int causal) {
  static const bool no_dma = getenv("PNPI_ATTN_NODMA") != nullptr;
  return Dp == 64 && Nk % 64 == 0 && Nk >= 128 && !no_dma && !causal;
}

int launch_attn_flash(const AttnP& p, hipStream_t st) {
  if (p.nrows <= 0) return 0;
  if ((p.ldq & 7) || (p.ldk & 7) || (p.ldv & 7) || (p.q_off & 7) || (p.k_off & 7) || (p.dh & 3) || (p.ldo & 3)) return -3;
  const int total = ((p.Nq + 127) / 128) * p.heads * p.nrows;
  dim3 grid((unsigned)(((total + 7) / 8) * 8), 1, 1);
  static const bool no_dma = getenv("PNPI_ATTN_NODMA") != nullptr;
  if (p.Dp == 64 && p.Nk % 64 == 0 && p.Nk >= 128 && !no_dma && !p.causal) {
    static const bool ksq3 = !(getenv("PNPI_ATTN_KSQ4") != nullptr);      // PNPI_ATTN_KSQ4: always four k-steps (A/B)
    static const bool no_aug = getenv("PNPI_ATTN_NOAUG") != nullptr;         // A/B: ignore AttnP::aug
    if (p.aug && p.dh == 40 && ksq3 && !no_aug) {
#ifdef PNPI_ABLATIONS
      if (p.vt_perm && g_attn_pipe == 1) attn_flash_pipe64_kernel<0><<<grid, 256, 0, st>>>(p);
      else if (p.vt_perm && g_attn_pipe == 2) attn_flash_pipe64_kernel<1><<<grid, 256, 0, st>>>(p);
      else
#endif
      if (p.vt_perm) attn_flash_dma64_kernel<true, 3, true><<<grid, 256, 0, st>>>(p);
      else attn_flash_dma64_kernel<false, 3, true><<<grid, 256, 0, st>>>(p);
    } else if (p.dh <= 48 && ksq3) {
      if (p.vt_perm) attn_flash_dma64_kernel<true, 3><<<grid, 256, 0, st>>>(p);
      else attn_flash_dma64_kernel<false, 3><<<grid, 256, 0, st>>>(p);
    } else {
      if (p.vt_perm) attn_flash_dma64_kernel<true><<<grid, 256, 0, st>>>(p);
      else attn_flash_dma64_kernel<false><<<grid, 256, 0, st>>>(p);
    }
    return (int)hipGetLastError();
  }
  if (p.vt_perm) return -6;        // only the kernel above reads the permuted layout
  const bool pf = p.Nk > 2 * KV_TILE;
#define LAUNCH_FLASH(D) (pf ? attn_flash_kernel<D, true><<<grid, 256, 0, st>>>(p) : attn_flash_kernel<D, false><<<grid, 256, 0, st>>>(p))
  switch (p.Dp) {
    case 32: LAUNCH_FLASH(32); break;
    case 64: LAUNCH_FLASH(64); break;
    case 96: LAUNCH_FLASH(96); break;
    case 128: LAUNCH_FLASH(128); break;
    case 160: LAUNCH_FLASH(160); break;
    default: return -5;
  }
#undef LAUNCH_FLASH
  return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------------
// Cross-attention (<= 96 text tokens) for one (source, target) conditional row pair with the P2P edit fused in:
//   P_src = softmax(Q_src K_src^T), P_tgt = softmax(Q_tgt K_tgt^T)
//   P_tgt' = c1[j] * (P_src . Mmat)[j] + c2[j] * P_tgt[j]          (no renormalisation, as in the reference)
//   O_src = P_src V_src ; O_tgt = P_tgt' V_tgt
//   LocalBlend accumulators += sum_j lb_alpha[i][j] * P_i[q][j]    (i = src: unedited map, i = tgt: post-edit map)
// c1/c2 encode AttentionReplace / AttentionRefine (+ AttentionReweight) and the cross_replace_alpha schedule:
//   c1 = a_t * eq * alphas ; c2 = a_t * eq * (1 - alphas) + (1 - a_t)       (attention_control.py:276-277,303-304,319-323,340-345)
// ------------------------------------------------------------------------------------------------------------------
static constexpr int CE_KEYS = 96;
static constexpr int CE_LD = CE_KEYS + 4;  // halfs per row of the V^T / Mmat^T tiles (200 B, 8-byte aligned)

template <int DP>
__global__ void __launch_bounds__(256) attn_cross_edit_kernel(CrossEditP p) {
  constexpr int KS = DP / 16, OT = DP / 32, SK_LD = DP + 8;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  half_t* sK = reinterpret_cast<half_t*>(smem_raw);   // [2][96][SK_LD]   (src, tgt)
  half_t* sV = sK + 2 * CE_KEYS * SK_LD;              // [2][DP][CE_LD]
  half_t* sM = sV + 2 * DP * CE_LD;                   // [96][CE_LD]      Mmat^T[j][w]
  float* sC = reinterpret_cast<float*>(sM + CE_KEYS * CE_LD);  // c1[96], c2[96], lb_alpha[4][96] (blend src / tgt, substruct src / tgt)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, ql = lane & 31;
  const int head = blockIdx.y, pair = blockIdx.z;
  const int rows2[2] = {p.pairs[pair * 2 + 0], p.pairs[pair * 2 + 1]};
  const int qtok = blockIdx.x * 128 + wave * 32 + ql;
  const bool qok = qtok < p.Nq;

  // ---- stage K, V^T (both rows), Mmat^T and the coefficient vectors.  A site at the 16 x 16 / 8 x 8 level is 16 / 8 workgroups, so a
  // workgroup's own latency is the launch's: every staging loop issues eight 16-byte loads per thread before the first store (the loops
  // used to pair each load with its store -- and V^T went element by element: ~170 dependent round trips per thread at DP = 160), and the
  // query fragments of both rows are requested before the barrier instead of after it.
  half8 qf2[2][KS];
#pragma unroll
  for (int which = 0; which < 2; ++which) {
    const half_t* qp = p.q + ((size_t)rows2[which] * p.Nq + (qok ? qtok : 0)) * p.ldq + p.q_off + head * DP + h * 8;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf2[which][ks] = qok ? ldg_half8(qp + ks * 16) : zero_half8();
  }
  auto staged8 = [&](int n, auto load, auto store) {
    for (int base = tid; base < n; base += 256 * 8) {
      half8 r[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { const int idx = base + u * 256; r[u] = idx < n ? load(idx) : zero_half8(); }
#pragma unroll
      for (int u = 0; u < 8; ++u) { const int idx = base + u * 256; if (idx < n) store(idx, r[u]); }
    }
  };
  {
    constexpr int KV8 = CE_KEYS * (DP / 8);                 // 16-byte vectors of one row's K block
    staged8(2 * KV8,
            [&](int idx) {
              const int which = idx / KV8, i = idx - which * KV8, r = i / (DP / 8), v = i - r * (DP / 8);
              const half_t* kbase = p.k + (size_t)rows2[which] * p.Nk * p.ldk + p.k_off + head * DP;
              return r < p.Nk ? ldg_half8(kbase + (size_t)r * p.ldk + v * 8) : zero_half8();
            },
            [&](int idx, half8 val) {
              const int which = idx / KV8, i = idx - which * KV8, r = i / (DP / 8), v = i - r * (DP / 8);
              *reinterpret_cast<half8*>(sK + (which * CE_KEYS + r) * SK_LD + v * 8) = val;
            });
    constexpr int VV8 = DP * (CE_KEYS / 8);                 // V^T rows are ldv (a multiple of 8) halfs long: whole 16-byte vectors
    staged8(2 * VV8,
            [&](int idx) {
              const int which = idx / VV8, i = idx - which * VV8, d = i / (CE_KEYS / 8), v = i - d * (CE_KEYS / 8);
              const half_t* vbase = p.vt + ((size_t)rows2[which] * p.heads + head) * DP * (size_t)p.ldv;
              half8 val = v * 8 < p.ldv ? ldg_half8(vbase + (size_t)d * p.ldv + v * 8) : zero_half8();
#pragma unroll
              for (int j = 0; j < 8; ++j)
                if (v * 8 + j >= p.Nk) val[j] = (half_t)0.f;      // keys past Nk (the row's pad columns) contribute nothing
              return val;
            },
            [&](int idx, half8 val) {
              const int which = idx / VV8, i = idx - which * VV8, d = i / (CE_KEYS / 8), v = i - d * (CE_KEYS / 8);
              half4 lo = {val[0], val[1], val[2], val[3]}, hi = {val[4], val[5], val[6], val[7]};
              half_t* dst = sV + (which * DP + d) * CE_LD + v * 8;        // rows are 200 bytes: 8-byte aligned
              *reinterpret_cast<half4*>(dst) = lo;
              *reinterpret_cast<half4*>(dst + 4) = hi;
            });
    const half_t* mm = p.mmatT + (size_t)pair * CE_KEYS * CE_KEYS;
    staged8(CE_KEYS * (CE_KEYS / 8),
            [&](int idx) { return ldg_half8(mm + (size_t)idx * 8); },
            [&](int idx, half8 val) {
              const int j = idx / (CE_KEYS / 8), v = idx - j * (CE_KEYS / 8);
              half4 lo = {val[0], val[1], val[2], val[3]}, hi = {val[4], val[5], val[6], val[7]};
              half_t* dst = sM + j * CE_LD + v * 8;
              *reinterpret_cast<half4*>(dst) = lo;
              *reinterpret_cast<half4*>(dst + 4) = hi;
            });
    for (int idx = tid; idx < CE_KEYS; idx += 256) {
      sC[idx] = p.c1[pair * CE_KEYS + idx];
      sC[CE_KEYS + idx] = p.c2[pair * CE_KEYS + idx];
#pragma unroll
      for (int pl = 0; pl < 4; ++pl)
        sC[(2 + pl) * CE_KEYS + idx] = (p.lb_alpha && pl < p.lb_planes) ? p.lb_alpha[(pair * p.lb_planes + pl) * CE_KEYS + idx] : 0.f;
    }
  }
  __syncthreads();

  const float c = p.scale * 1.44269504088896340736f;
  floatx16 P[2][3];  // normalised probabilities, [src/tgt][key tile]
#pragma unroll
  for (int which = 0; which < 2; ++which) {
    const half8* qf = qf2[which];
    float mx = -INFINITY;
#pragma unroll
    for (int st = 0; st < 3; ++st) {
      floatx16 s;
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        half8 kf = *reinterpret_cast<const half8*>(sK + (which * CE_KEYS + st * 32 + ql) * SK_LD + ks * 16 + h * 8);
        s = mfma32(kf, qf[ks], s);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        int key = st * 32 + acc_row(r, lane);
        float v = key < p.Nk ? s[r] : -INFINITY;
        s[r] = v;
        mx = fmaxf(mx, v);
      }
      P[which][st] = s;
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int st = 0; st < 3; ++st)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float e = exp2f((P[which][st][r] - mx) * c);
        P[which][st][r] = e;
        sum += e;
      }
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.f / sum;
#pragma unroll
    for (int st = 0; st < 3; ++st)
#pragma unroll
      for (int r = 0; r < 16; ++r) P[which][st][r] *= inv;
  }

  // fp16 MFMA B-operand fragments of P_src (k-slot order = accumulator register order)
  half8 pfs[3][2];
#pragma unroll
  for (int st = 0; st < 3; ++st)
#pragma unroll
    for (int r = 0; r < 16; ++r) pfs[st][r >> 3][r & 7] = (half_t)P[0][st][r];

  // mapped^T[j][q] = sum_w Mmat^T[j][w] * P_src^T[w][q]; then the blend, written over P_tgt
#pragma unroll
  for (int jt = 0; jt < 3; ++jt) {
    floatx16 mp;
#pragma unroll
    for (int r = 0; r < 16; ++r) mp[r] = 0.f;
#pragma unroll
    for (int st = 0; st < 3; ++st)
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const half_t* mb = sM + (jt * 32 + ql) * CE_LD + st * 32 + t * 16 + 4 * h;
        half4 lo = *reinterpret_cast<const half4*>(mb);
        half4 hi = *reinterpret_cast<const half4*>(mb + 8);
        half8 mf = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        mp = mfma32(mf, pfs[st][t], mp);
      }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      int j = jt * 32 + acc_row(r, lane);
      P[1][jt][r] = sC[j] * mp[r] + sC[CE_KEYS + j] * P[1][jt][r];
    }
  }

  // LocalBlend accumulators (deterministic: each (slot, branch, query) word has exactly one writer per launch)
  if (p.lb_acc) {
#pragma unroll
    for (int pl = 0; pl < 4; ++pl) {
      if (pl >= p.lb_planes) break;            // planes 2, 3: the substruct-word selectors on the same two probability maps
      const int which = pl & 1;
      float acc = 0.f;
#pragma unroll
      for (int st = 0; st < 3; ++st)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          int j = st * 32 + acc_row(r, lane);
          acc += sC[(2 + pl) * CE_KEYS + j] * P[which][st][r];
        }
      acc += __shfl_xor(acc, 32, 64);
      if (h == 0 && qok) {
        float* dst = p.lb_acc + (((size_t)pair * p.lb_nslots + p.lb_slot0 + head) * p.lb_planes + pl) * p.Nq + qtok;
        *dst += acc;
      }
    }
  }

  // O = P V.  By default only the TARGET row is written: the source row's output comes from the plain flash kernel, so that it
  // is bit-identical to the controller-free passes (the direct-inversion offset only cancels against an identical forward).
#pragma unroll
  for (int which = 0; which < 2; ++which) {
    if (which == 0 && !p.write_src) continue;
    half8 pf[3][2];
#pragma unroll
    for (int st = 0; st < 3; ++st)
#pragma unroll
      for (int r = 0; r < 16; ++r) pf[st][r >> 3][r & 7] = (half_t)P[which][st][r];
    half_t* op = p.o + ((size_t)rows2[which] * p.Nq + (qok ? qtok : 0)) * p.ldo + head * p.dh;
#pragma unroll
    for (int ot = 0; ot < OT; ++ot) {
      floatx16 o;
#pragma unroll
      for (int r = 0; r < 16; ++r) o[r] = 0.f;
#pragma unroll
      for (int st = 0; st < 3; ++st)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const half_t* vb = sV + (which * DP + ot * 32 + ql) * CE_LD + st * 32 + t * 16 + 4 * h;
          half4 lo = *reinterpret_cast<const half4*>(vb);
          half4 hi = *reinterpret_cast<const half4*>(vb + 8);
          half8 vf = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          o = mfma32(vf, pf[st][t], o);
        }
      if (qok) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          int d = ot * 32 + 8 * g + 4 * h;
          if (d + 3 < p.dh) {
            half4 o4 = {(half_t)o[4 * g], (half_t)o[4 * g + 1], (half_t)o[4 * g + 2], (half_t)o[4 * g + 3]};
            *reinterpret_cast<half4*>(op + d) = o4;
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
              if (d + j < p.dh) op[d + j] = (half_t)o[4 * g + j];
          }
        }
      }
    }
  }
}

template <int DP>
static size_t ce_lds_bytes() {
  return (size_t)(2 * CE_KEYS * (DP + 8) + 2 * DP * CE_LD + CE_KEYS * CE_LD) * sizeof(half_t) + 6 * CE_KEYS * sizeof(float);
}

template <int DP>
static int launch_ce(const CrossEditP& p, hipStream_t st) {
  static DeviceOnce attr_once;
  const size_t lds = ce_lds_bytes<DP>();
  if (int r = once_per_device(attr_once, [&]() { return (int)hipFuncSetAttribute((const void*)attn_cross_edit_kernel<DP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); })) return r;
  dim3 grid((p.Nq + 127) / 128, p.heads, p.npairs);
  attn_cross_edit_kernel<DP><<<grid, 256, lds, st>>>(p);
  return (int)hipGetLastError();
}

int launch_attn_cross_edit(const CrossEditP& p, hipStream_t st) {
  if (p.npairs <= 0) return 0;
  if (p.Nk > CE_KEYS) return -6;
  if ((p.ldq & 7) || (p.ldk & 7) || (p.q_off & 7) || (p.k_off & 7) || (p.dh & 3) || (p.ldo & 3)) return -3;
  switch (p.Dp) {
    case 32: return launch_ce<32>(p, st);
    case 64: return launch_ce<64>(p, st);
    case 96: return launch_ce<96>(p, st);
    case 128: return launch_ce<128>(p, st);
    case 160: return launch_ce<160>(p, st);
    default: return -5;
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Flash-style attention BACKWARD (null-text path: the gradient has to pass through every self-attention of the UNet).
//   O = softmax(scale Q K^T) V;  dV = P^T dO;  dP = dO V^T;  dS = P o (dP - D), D[q] = sum_k P dP;  dQ = scale dS K;  dK = scale dS^T Q
// without S, P, dP, dS ever in memory (the materialised form wrote and re-read 537 MB of fp32 scores per 64 x 64 site).
// One kernel, three modes.  A workgroup owns 128 "block" rows (4 waves x 32: the MFMA B operand, in registers) and walks the "loop"
// rows in tiles of LT (the A operand, staged in LDS):   tile[loop][block] = L1 . B1^T   (and L2 . B2^T)
//   DQ  block = queries (Q, dO), loop = keys (K, V).  Pass 1: lse2[q] = log2 sum_k 2^(c S) and D[q], online (running max), stored for
//       the other two modes.  Pass 2: dS^T = P^T o (dP^T - D)  ->  dQ^T += K^T . dS^T
//   DK  block = keys (K, V), loop = queries (Q, dO):  dS = P o (dP - D)  ->  dK^T += Q^T . dS
//   DV  block = keys (K),    loop = queries (Q):                              dV^T += dO^T . P
// As in the forward kernel the probability / dS tile goes from the accumulator registers straight into the next MFMA as its B operand
// (the k-slot permutation is applied to the transposed loop-side operand: two 8-byte reads per fragment).  Every output element has one
// writer and a fixed summation order: deterministic.  10 tile products per (query tile, key tile) pair against the 5 of the
// materialised form -- the price of no atomics and no second workspace.
// ------------------------------------------------------------------------------------------------------------------
template <int DP, int LT, int MODE, bool PF = false>
__global__ void __launch_bounds__(256) attn_bwd_flash_kernel(AttnBwdP p) {
  constexpr int KS = DP / 16, OT = DP / 32, NT = LT / 32;
  constexpr int SK_LD = DP + 8, LT_LD = LT + 4;
  __shared__ __attribute__((aligned(16))) half_t sL1[LT * SK_LD];
  __shared__ __attribute__((aligned(16))) half_t sL2[MODE == 2 ? 8 : LT * SK_LD];
  __shared__ __attribute__((aligned(16))) half_t sLT[DP * LT_LD];
  __shared__ float sLse[LT], sD[LT];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, ql = lane & 31;
  const int nbt = (p.nb + 127) >> 7, T = nbt * p.heads * p.nsplit, per = (T + 7) >> 3;
  const int tix = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
  if (tix >= T) return;
  const int bt = tix % nbt, head = (tix / nbt) % p.heads, split = tix / (nbt * p.heads);
  const int brow = bt * 128 + wave * 32 + ql;
  const bool bok = brow < p.nb;
  // this workgroup's share of the loop rows (everything unless the launch is split; whole LT tiles)
  const int lchunk = ((p.nl + p.nsplit - 1) / p.nsplit + LT - 1) / LT * LT;
  const int l_begin = split * lchunk, l_end = min(p.nl, l_begin + lchunk);

  half8 f1[KS], f2[KS];
  {
    const half_t* s1 = p.b1.p + (size_t)head * p.b1.hs + (size_t)(bok ? brow : 0) * p.b1.ld + h * 8;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) f1[ks] = (bok && ks * 16 + h * 8 < p.b1.w) ? ldg_half8(s1 + ks * 16) : zero_half8();
    if constexpr (MODE != 2) {
      const half_t* s2 = p.b2.p + (size_t)head * p.b2.hs + (size_t)(bok ? brow : 0) * p.b2.ld + h * 8;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) f2[ks] = (bok && ks * 16 + h * 8 < p.b2.w) ? ldg_half8(s2 + ks * 16) : zero_half8();
    }
  }
  // The transposed loop-side operand (K^T for dQ, Q^T for dK, dO^T for dV) is the transpose of a row-major tile this workgroup loads
  // anyway: it is written to LDS a second time, element-wise, as [d][row] (no transposed copies in memory, no transpose launches).
  auto put_t = [&](const half8& val, int r, int v) {
#pragma unroll
    for (int j = 0; j < 8; ++j) sLT[(v * 8 + j) * LT_LD + r] = val[j];
  };
  auto stage_rows = [&](const BwdMat& M, half_t* dst, int l0, bool rows, bool tr) {
    constexpr int NV = LT * (DP / 8);
    for (int idx = tid; idx < NV; idx += 256) {
      const int r = idx / (DP / 8), v = idx - r * (DP / 8);
      const int row = l0 + r;
      const half8 val = (row < p.nl && v * 8 < M.w) ? ldg_half8(M.p + (size_t)head * M.hs + (size_t)row * M.ld + v * 8) : zero_half8();
      if (rows) *reinterpret_cast<half8*>(dst + r * SK_LD + v * 8) = val;
      if (tr) put_t(val, r, v);
    }
  };
  floatx16 s[NT], dp[NT];
  auto tiles = [&]() {
#pragma unroll
    for (int st = 0; st < NT; ++st) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[st][r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
        s[st] = mfma32(*reinterpret_cast<const half8*>(sL1 + (st * 32 + ql) * SK_LD + ks * 16 + h * 8), f1[ks], s[st]);
      if constexpr (MODE != 2) {
#pragma unroll
        for (int r = 0; r < 16; ++r) dp[st][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
          dp[st] = mfma32(*reinterpret_cast<const half8*>(sL2 + (st * 32 + ql) * SK_LD + ks * 16 + h * 8), f2[ks], dp[st]);
      }
    }
  };
  const float c = p.scale * 1.44269504088896340736f;
  float lse2 = 0.f, dq = 0.f;
  if (MODE == 0 && p.o) {
    // the forward kernel left the log-sum-exp, and D[q] = sum_k P dP = sum_d dO[q][d] O[q][d]: no first pass over the keys
    float part = 0.f;
    if (bok) {
      const half_t* so = p.o + (size_t)head * p.o_hs + (size_t)brow * p.o_ld + h * 8;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
        if (ks * 16 + h * 8 < p.b2.w) {
          const half8 ov = ldg_half8(so + ks * 16);
#pragma unroll
          for (int j = 0; j < 8; ++j) part = __builtin_fmaf((float)ov[j], (float)f2[ks][j], part);
        }
    }
    dq = part + __shfl_xor(part, 32, 64);
    lse2 = bok ? p.lse[(size_t)head * p.nb + brow] : 0.f;
    if (h == 0 && bok) p.dsum[(size_t)head * p.nb + brow] = dq;
  } else if constexpr (MODE == 0) {
    float mrun = -INFINITY, lrun = 0.f, drun = 0.f;
    for (int l0 = l_begin; l0 < l_end; l0 += LT) {
      __syncthreads();
      stage_rows(p.l1, sL1, l0, true, false);
      stage_rows(p.l2, sL2, l0, true, false);
      __syncthreads();
      tiles();
      if (l0 + LT > p.nl) {
#pragma unroll
        for (int st = 0; st < NT; ++st)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            if (l0 + st * 32 + acc_row(r, lane) >= p.nl) s[st][r] = -INFINITY;
      }
      float mloc = s[0][0];
#pragma unroll
      for (int st = 0; st < NT; ++st)
#pragma unroll
        for (int r = 0; r < 16; ++r) mloc = fmaxf(mloc, s[st][r]);
      mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
      const float mnew = fmaxf(mrun, mloc);          // finite: every tile starts inside the key range
      const float alpha = __builtin_amdgcn_exp2f((mrun - mnew) * c);   // first tile: 2^-inf = 0
      const float mc = mnew * c;
      float ps = 0.f, pd = 0.f;
#pragma unroll
      for (int st = 0; st < NT; ++st)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(s[st][r], c, -mc));
          ps += pv;
          pd = __builtin_fmaf(pv, dp[st][r], pd);
        }
      lrun = __builtin_fmaf(lrun, alpha, ps);
      drun = __builtin_fmaf(drun, alpha, pd);
      mrun = mnew;
    }
    const float ltot = lrun + __shfl_xor(lrun, 32, 64), dtot = drun + __shfl_xor(drun, 32, 64);
    lse2 = mrun * c + __log2f(ltot);
    dq = dtot / ltot;
    if (h == 0 && bok) {
      p.lse[(size_t)head * p.nb + brow] = lse2;
      p.dsum[(size_t)head * p.nb + brow] = dq;
    }
  }

  floatx16 acc[OT];
#pragma unroll
  for (int ot = 0; ot < OT; ++ot)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[ot][r] = 0.f;
  // PF (the 64-wide heads of the 4096-token sites, one workgroup per CU): the next tile's global loads are issued into registers right after
  // this tile was published in LDS and fly under its MFMAs -- without it every tile pays a full global round trip between two barriers.
  constexpr int NVR = LT * (DP / 8) / 256;
  static_assert(!PF || LT * (DP / 8) % 256 == 0, "prefetch: whole vectors per thread");
  half8 r1[PF ? NVR : 1], r2[PF ? NVR : 1];       // r2: V (dQ) / dO (dK, dV) rows
  auto gload = [&](int l0) {
    auto rows = [&](const BwdMat& M, half8* dst) {
#pragma unroll
      for (int i = 0; i < NVR; ++i) {
        const int idx = tid + i * 256, r = idx / (DP / 8), v = idx - r * (DP / 8);
        const int row = l0 + r;
        dst[i] = (row < p.nl && v * 8 < M.w) ? ldg_half8(M.p + (size_t)head * M.hs + (size_t)row * M.ld + v * 8) : zero_half8();
      }
    };
    rows(p.l1, r1);
    rows(p.l2, r2);
  };
  auto sstore = [&]() {
#pragma unroll
    for (int i = 0; i < NVR; ++i) {
      const int idx = tid + i * 256, r = idx / (DP / 8), v = idx - r * (DP / 8);
      *reinterpret_cast<half8*>(sL1 + r * SK_LD + v * 8) = r1[i];
      if constexpr (MODE != 2) *reinterpret_cast<half8*>(sL2 + r * SK_LD + v * 8) = r2[i];
      put_t(MODE == 2 ? r2[i] : r1[i], r, v);
    }
  };
  if constexpr (PF) gload(l_begin);
  for (int l0 = l_begin; l0 < l_end; l0 += LT) {
    __syncthreads();
    if constexpr (PF) {
      sstore();
    } else {
      stage_rows(p.l1, sL1, l0, true, MODE != 2);
      stage_rows(p.l2, sL2, l0, MODE != 2, MODE == 2);
    }
    if constexpr (MODE != 0) {
      if (tid < LT) {
        const int row = l0 + tid;
        sLse[tid] = row < p.nl ? p.lse[(size_t)head * p.nl + row] : INFINITY;     // 2^(x - inf) = 0: rows past the end contribute nothing
        sD[tid] = row < p.nl ? p.dsum[(size_t)head * p.nl + row] : 0.f;
      }
    }
    __syncthreads();
    if constexpr (PF) { if (l0 + LT < l_end) gload(l0 + LT); }
    tiles();
    half8 pf[NT][2];
#pragma unroll
    for (int st = 0; st < NT; ++st)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int lr = st * 32 + acc_row(r, lane);
        float lz, dz;
        if constexpr (MODE == 0) { lz = (l0 + lr < p.nl) ? lse2 : INFINITY; dz = dq; }
        else { lz = sLse[lr]; dz = sD[lr]; }
        const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(s[st][r], c, -lz));
        float val = pv;
        if constexpr (MODE != 2) val = pv * (dp[st][r] - dz);
        pf[st][r >> 3][r & 7] = (half_t)val;
      }
#pragma unroll
    for (int ot = 0; ot < OT; ++ot)
#pragma unroll
      for (int st = 0; st < NT; ++st)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const half_t* vb = sLT + (ot * 32 + ql) * LT_LD + st * 32 + t * 16 + 4 * h;
          const half4 lo = *reinterpret_cast<const half4*>(vb);
          const half4 hi = *reinterpret_cast<const half4*>(vb + 8);
          const half8 vf = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          acc[ot] = mfma32(vf, pf[st][t], acc[ot]);
        }
  }
  if (bok && p.nsplit > 1) {            // fp32 partial sums of this share of the loop rows; launch_attn_bwd_reduce adds the shares in order
    const float osc = MODE == 2 ? 1.f : p.scale;
    float* pp = p.part + (((size_t)split * p.heads + head) * p.nb + brow) * p.out_w;
#pragma unroll
    for (int ot = 0; ot < OT; ++ot)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int d = ot * 32 + acc_row(r, lane);
        if (d < p.out_w) pp[d] = acc[ot][r] * osc;
      }
  } else if (bok) {
    const float osc = MODE == 2 ? 1.f : p.scale;
    half_t* op = p.out + (size_t)head * p.out_hs + (size_t)brow * p.out_ld;
#pragma unroll
    for (int ot = 0; ot < OT; ++ot)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d = ot * 32 + 8 * g + 4 * h;
        if (d + 3 < p.out_w) {
          half4 o4 = {(half_t)(acc[ot][4 * g] * osc), (half_t)(acc[ot][4 * g + 1] * osc), (half_t)(acc[ot][4 * g + 2] * osc),
                      (half_t)(acc[ot][4 * g + 3] * osc)};
          *reinterpret_cast<half4*>(op + d) = o4;
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (d + j < p.out_w) op[d + j] = (half_t)(acc[ot][4 * g + j] * osc);
        }
      }
  }
}

__global__ void __launch_bounds__(256) attn_bwd_reduce_kernel(AttnBwdP p) {
  const size_t per = (size_t)p.heads * p.nb * p.out_w;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < per; i += (size_t)gridDim.x * 256) {
    float s = 0.f;
    for (int sp = 0; sp < p.nsplit; ++sp) s += p.part[(size_t)sp * per + i];
    const int d = (int)(i % p.out_w);
    const size_t hr = i / p.out_w;
    const int row = (int)(hr % p.nb), head = (int)(hr / p.nb);
    p.out[(size_t)head * p.out_hs + (size_t)row * p.out_ld + d] = (half_t)s;
  }
}
int launch_attn_bwd_reduce(const AttnBwdP& p, hipStream_t st) {
  if (p.nsplit < 2 || !p.part) return -3;
  const size_t per = (size_t)p.heads * p.nb * p.out_w;
  int blocks = (int)((per + 255) / 256);
  if (blocks > 1024) blocks = 1024;
  attn_bwd_reduce_kernel<<<blocks, 256, 0, st>>>(p);
  return (int)hipGetLastError();
}

template <int DP, int LT>
static int launch_abf(const AttnBwdP& p, int mode, hipStream_t st) {
  if (p.nsplit < 1 || (p.nsplit > 1 && (mode == 0 || !p.part))) return -3;
  const int T = ((p.nb + 127) >> 7) * p.heads * p.nsplit;
  const dim3 grid((unsigned)(((T + 7) / 8) * 8));
  static const bool pf_on = !(getenv("PNPI_ABF_PREFETCH") && atoi(getenv("PNPI_ABF_PREFETCH")) == 0);   // PNPI_ABF_PREFETCH=0: never the register-prefetch variant (A/B)
  if constexpr (DP == 64 || DP == 96) {
    // many loop tiles per workgroup and at most ~one workgroup per CU: prefetch the next tile into registers
    if (pf_on && p.nl >= 8 * LT && T <= 512) {
      if (mode == 0) attn_bwd_flash_kernel<DP, LT, 0, true><<<grid, 256, 0, st>>>(p);
      else if (mode == 1) attn_bwd_flash_kernel<DP, LT, 1, true><<<grid, 256, 0, st>>>(p);
      else attn_bwd_flash_kernel<DP, LT, 2, true><<<grid, 256, 0, st>>>(p);
      return (int)hipGetLastError();
    }
  }
  if (mode == 0) attn_bwd_flash_kernel<DP, LT, 0><<<grid, 256, 0, st>>>(p);
  else if (mode == 1) attn_bwd_flash_kernel<DP, LT, 1><<<grid, 256, 0, st>>>(p);
  else attn_bwd_flash_kernel<DP, LT, 2><<<grid, 256, 0, st>>>(p);
  return (int)hipGetLastError();
}
int launch_attn_bwd_flash(const AttnBwdP& p, int mode, int Dp, hipStream_t st) {
  if (p.nb <= 0 || p.nl <= 0 || mode < 0 || mode > 2) return -3;
  switch (Dp) {
    case 32: return launch_abf<32, 64>(p, mode, st);
    case 64: return launch_abf<64, 64>(p, mode, st);
    case 96: return launch_abf<96, 64>(p, mode, st);
    case 160: return launch_abf<160, 32>(p, mode, st);
    default: return -1;
  }
}
