// decoder.hip — the reranker's LM forward on the MI355X: last-position [no, yes] logits of a Qwen3-style
// decoder (what Qwen3Reranker.compute_logits takes from `self.lm(**inputs).logits[:, -1, :]`,
// core/rerank/Reranker_Qwen3.py:41-49, for the left-padded batches process_inputs builds, :29-39).
//
//   x = embed[ids]
//   per layer:  h = RMSNorm(x)·w_in;  [q|k|v] = h·Wqkvᵀ  (GQA: n_q heads of q, n_kv heads of k and v)
//               q, k <- RoPE(RMSNorm_head(q)·w_qn), RoPE(RMSNorm_head(k)·w_kn)       (Qwen3: per-head q/k norm;
//                       applied inside the attention kernel, on its operand fragments)
//               ctx = causal softmax(q·kᵀ/sqrt(dh)) v over the sequence's real tokens;  x += ctx·Woᵀ
//               h = RMSNorm(x)·w_post;  [g|u] = h·Wguᵀ;  x += (silu(g)·u)·Wdownᵀ
//   logits[s] = (RMSNorm(x[s, L-1])·w_final) · lm_head[{no, yes}]ᵀ
//
// The GEMMs are the encoder's MFMA ping-pong kernels (rarc_enc_gemm, encoder.hip: fp16 operands, fp32
// accumulation); attention is the encoder's wave-per-32-queries MFMA kernel generalised to head_dim 128, grouped
// K/V heads, a causal mask and left padding.  Activations and the residual stream are fp16, as in the reference's
// `torch_dtype=torch.float16` model; norms, RoPE, softmax and the final dot products compute in fp32.
#include "rarc_common.h"
#include <string.h>
#include <stdlib.h>
#include <stdio.h>

extern "C" int rarc_enc_gemm(const uint16_t* d_a, const uint16_t* d_w, const uint16_t* d_bias, uint16_t* d_c, int m,
                             int n, int k, int act, void* stream);

bool rarc_gemm_swiglu_fused(int m, int n, int k);  // encoder.hip: act = 3 available for this shape
bool rarc_gemm_norm_fusable(int m, int n, int k);  // encoder.hip: the row-scale / residual epilogues are available for this shape
int rarc_gemm_fused_norm(const uint16_t* d_a, const uint16_t* d_w, const void* d_rowscale_or_zero, uint16_t* d_c, int m, int n, int k,
                         int act, void* stream, float* d_ssq = nullptr);   // act 16 / 19: C = act(rowscale ⊙ A·Wᵀ); 32: C += A·Wᵀ in
                                                                           // place; 96: the same + sums of squares per 32 columns


typedef _Float16 half4v __attribute__((ext_vector_type(4)));

// ---- embedding gather: x[t] = embed[ids[t]] (ids clamped into the table: memory safety, the host validates) ----
__global__ __launch_bounds__(256) void rarc_lm_embed_kernel(const int32_t* ids, const half_t* embed, int n_tokens, int H,
                                                            int vocab, half_t* x) {
  const int t = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (t >= n_tokens) return;
  int id = ids[t];
  id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
  const half8* src = (const half8*)(embed + (size_t)id * H);
  half8* dst = (half8*)(x + (size_t)t * H);
  for (int c = lane; c < H / 8; c += 64) dst[c] = src[c];
}

// ---- RMSNorm over the hidden dimension, optionally after a residual add --------------------------------------
// delta != null:  x[t] += delta[t] (fp16 add, written back), then y[t] = x[t] · rsqrt(mean(x²) + eps) · w
__global__ __launch_bounds__(256) void rarc_lm_rmsnorm_kernel(half_t* x, const half_t* delta, const half_t* w, float eps,
                                                              int n_tokens, int H, half_t* y) {
  const int t = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (t >= n_tokens) return;
  half_t* xr = x + (size_t)t * H;
  float ss = 0.f;
  for (int c = lane * 8; c < H; c += 512) {
    half8 v = *(const half8*)(xr + c);
    if (delta) {
      const half8 dv = *(const half8*)(delta + (size_t)t * H + c);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (half_t)((float)v[e] + (float)dv[e]);  // one rounding, like an fp16 tensor add
      *(half8*)(xr + c) = v;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) ss = __builtin_fmaf((float)v[e], (float)v[e], ss);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
  if (!y) return;
  const float inv = 1.0f / __builtin_sqrtf(ss / (float)H + eps);
  for (int c = lane * 8; c < H; c += 512) {
    const half8 v = *(const half8*)(xr + c), wv = *(const half8*)(w + c);
    half8 o8;
#pragma unroll
    for (int e = 0; e < 8; ++e) o8[e] = (half_t)((float)wv[e] * (float)(half_t)((float)v[e] * inv));  // weight * x.to(fp16)
    *(half8*)(y + (size_t)t * H + c) = o8;
  }
}

// ---- r[t] from the partial sums of squares the residual epilogue left (ssq [n_tokens][n_parts], one per 32 columns) -----
__global__ __launch_bounds__(256) void rarc_lm_rowscale_parts_kernel(const float* __restrict__ ssq, int n_parts, float eps, int n_tokens,
                                                                     int H, float* __restrict__ r) {
  // eight lanes per row, 16 bytes each per step (a row's 32 partials are one 128-byte line: read by one thread per row the
  // loads were strided by a line each, 25 us per call); fixed summation order: per lane ascending, then the 8-lane tree
  const int t = blockIdx.x * 32 + (threadIdx.x >> 3), sub = threadIdx.x & 7;
  float ss = 0.f;
  if (t < n_tokens) {
    const float* p = ssq + (size_t)t * n_parts;
    for (int i = 4 * sub; i < n_parts; i += 32) {
      const float4 v = *(const float4*)(p + i);
      ss += (v.x + v.y) + (v.z + v.w);
    }
  }
  ss += __shfl_xor(ss, 1, 64);
  ss += __shfl_xor(ss, 2, 64);
  ss += __shfl_xor(ss, 4, 64);
  if (t < n_tokens && sub == 0) r[t] = 1.0f / __builtin_sqrtf(ss / (float)H + eps);
}

// ---- rotary table: per position a row of DH halves, cos of the DH/2 pairs then their sin ---------------------------------
// inv_freq_i = theta^(-2i/DH); angle = pos * inv_freq_i in fp32, as the reference's rotary module computes them, then cast
// to the activations' dtype before use.  Positions run over the PADDED sequence (the reference passes no position_ids).
// Computed once per forward and read from cache by the attention kernel (computing sincosf where it is used cost
// 374 us per layer at 51 200 tokens: the angles reach thousands of radians and need the accurate routine).  Planar
// (cos[DH/2] | sin[DH/2], round 3; it was interleaved pairs) so that a 16-byte load is a packed-fp16 operand.
__global__ __launch_bounds__(256) void rarc_lm_rope_table_kernel(int L, int DH, float theta, half2_t* table) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= L * (DH / 2)) return;
  const int pos = idx / (DH / 2), i = idx % (DH / 2);
  const float inv_freq = __builtin_exp2f(-(2.0f * (float)i / (float)DH) * __builtin_log2f(theta));
  float sn, cs;
  sincosf((float)pos * inv_freq, &sn, &cs);
  half_t* row = (half_t*)table + (size_t)pos * DH;
  row[i] = (half_t)cs;
  row[DH / 2 + i] = (half_t)sn;
}

// ---- the arithmetic of RMSNorm + rotary embedding on 8 halves at a time (shared by every kernel below, so that they agree
// bit for bit) ------------------------------------------------------------------------------------------------------
// lm_scale8: RN_f16(x * inv) with x fp16, inv fp32 — the reference's `hidden.float() * rsqrt(...)` followed by `.to(fp16)`.
// Written as a C cast hipcc picks v_fma_mixlo_f16 in one kernel and v_mul_f32 + v_cvt_f16_f32 in another, which round
// differently in the rare double-rounding cases (two kernels that must agree then differ in ~1e-3 of their outputs by one fp16
// ulp — measured); the instruction is spelled out: one v_fma_mix per element, straight into the packed halves.
__device__ __forceinline__ half8 lm_scale8(const half8 x, const float inv) {
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  const u32x4 in = __builtin_bit_cast(u32x4, x);
  u32x4 out;
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    uint32_t o;
    asm("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %0, %1, %2, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
        : "=&v"(o) : "v"(in[w]), "v"(inv));
    out[w] = o;
  }
  return __builtin_bit_cast(half8, out);
}
// rotation of the pairs (a[e], b[e]) = elements (i, i + DH/2) by the angles whose cos / sin are cs / sn, in packed fp16
// arithmetic (two roundings: the product b*sn, then the fused a*cs -+ it; the reference's fp16 tensors round three times):
//   lo = a*cs - b*sn,  hi = b*cs + a*sn
__device__ __forceinline__ void lm_rotate8(const half8 a, const half8 b, const half8 cs, const half8 sn, half8& lo, half8& hi) {
  lo = __builtin_elementwise_fma(a, cs, -(b * sn));
  hi = __builtin_elementwise_fma(b, cs, a * sn);
}

// ---- per-head RMSNorm, then rotary embedding (HF rotate_half convention), on MFMA operand fragments ---------------
// f[ks][e] is element 16 ks + 8 hh + e of one head row (the lane pair (l, l ^ 32) holds the whole row), so the rotation
// partner i + DH/2 of an element sits in the same lane (fragment ks + KS/2) and the row's sum of squares needs one
// exchange.  Roundings: x·inv -> fp16 (lm_scale8), ·weight -> fp16, rotation in fp16 (lm_rotate8).  `cs_row`: the row of the
// rotary table (cos[DH/2] | sin[DH/2]).
template <int DH>
__device__ __forceinline__ void lm_norm_rope(half8 (&f)[DH / 16], const half_t* __restrict__ w, float eps,
                                             const half2_t* __restrict__ cs_row2, int hh) {
  constexpr int KS = DH / 16;
  const half_t* cs_row = (const half_t*)cs_row2;
  // every load of the weights and of the row's cos / sin is issued before the first use (round 3: inside the loop below
  // each k-step waited for its own four loads — four memory round trips, ~9 k cycles per (q head, query block) unit)
  half8 w0[KS / 2], w1[KS / 2], cs[KS / 2], sn[KS / 2];
#pragma unroll
  for (int ks = 0; ks < KS / 2; ++ks) {
    const int i0 = 16 * ks + 8 * hh;
    w0[ks] = *(const half8*)(w + i0);
    w1[ks] = *(const half8*)(w + i0 + DH / 2);
    cs[ks] = *(const half8*)(cs_row + i0);
    sn[ks] = *(const half8*)(cs_row + DH / 2 + i0);
  }
  float ss = 0.f;
#pragma unroll
  for (int ks = 0; ks < KS; ++ks)
#pragma unroll
    for (int e = 0; e < 8; ++e) ss = __builtin_fmaf((float)f[ks][e], (float)f[ks][e], ss);
  ss += __shfl_xor(ss, 32, 64);
  const float inv = 1.0f / __builtin_sqrtf(ss / (float)DH + eps);
#pragma unroll
  for (int ks = 0; ks < KS / 2; ++ks) {
    const half8 a = w0[ks] * lm_scale8(f[ks], inv), b = w1[ks] * lm_scale8(f[ks + KS / 2], inv);
    lm_rotate8(a, b, cs[ks], sn[ks], f[ks], f[ks + KS / 2]);
  }
}

// ---- causal attention with grouped K/V heads, left padding and an optional shared KEY/VALUE PREFIX -------------------
// One WORKGROUP (4 waves) per (sequence, K/V head, group of four (q head, 32-query block) units); one wave per unit.
// Units of a (sequence, K/V head) are ordered query-block major, so the four waves of a workgroup want (nearly) the same
// keys: every 32-key tile is fetched ONCE per workgroup with coalesced 256-byte row loads, k gets its per-head RMSNorm
// and rotary embedding there (a row lives in 16 lanes: the rotation partner i + DH/2 is eight lanes away, the sum of
// squares four exchanges), and lands in LDS as the MFMA operand image — K row-major with a 16-byte pad per row
// (conflict-free ds_read_b128 fragments), V transposed [d][key] with key pairs packed into 4-byte writes.
//   S^T = K · Q^T       v_mfma_f32_32x32x16_f16, A = K fragments from LDS, B = the wave's Q rows (registers; RMSNorm +
//                       rotary applied to the fragments as they are loaded, lm_norm_rope): lane l owns query l & 31, the
//                       softmax statistics of a query sit in the lane pair (l, l ^ 32)
//   O^T += V^T · P^T    A = V^T fragments from LDS, B = P^T assembled in registers
// Round 3 — software pipeline: the raw rows of tile t+1 are fetched into registers (K chunk, its rotary row, the V pair:
// 32 VGPRs) BEFORE tile t is multiplied, and the LDS image is double buffered, so a tile costs one barrier and the
// global-memory latency of its rows hides under the previous tile's MFMAs and softmax (it was fully exposed: two
// barriers and ~15 k cycles per 32-key tile, 1.4 ms per layer at 640 x 256 tokens).
// Round 3 — prefix: the (query, document) prompts a reranker scores for ONE query share everything up to the document
// (chat prefix, instruction, query: ~80 of ~220 tokens), and in a causal LM the keys and values of those tokens do not
// depend on what follows.  `pf` names a cache of raw k | v rows computed once per query (rarc_lm_prefix_kv); a
// sequence with a prefix attends to cache rows [pstart, P) and then to its own rows.  Keys are walked in a VIRTUAL index
// u: u < P is cache row u, u >= P is own row start + (u - P) — contiguous, no tile is spent on left padding.  A key's
// rotary position is u (+ start without a prefix, which makes it the index in the padded sequence: the reference's
// position ids), a query row i sits at u = P + (i - start).  Keys limited to [u_min, u of the query].
// ---- online softmax of one 32-key tile in the log2 domain: t = s * (scale * log2 e), p = 2^(t - m) -------------------------
// st[r] is the raw score of this lane's query against key k0 + 8 (r >> 2) + 4 hh + (r & 3).  An INTERIOR tile (every key of
// it visible to every query of the wave: all but the first tile of a left-padded or prefixed sequence and the tile on the
// diagonal) needs no mask and no -inf guards; the two forms are separate instantiations chosen by ONE wave-uniform branch
// per tile (written as `interior ? a : b` inside the unrolled loops hipcc emitted a branch per pair of scores).  The running
// maximum is only raised when some query's tile maximum exceeds it by more than 2^DEFER (wave-uniform decision): otherwise
// p = 2^(t - m_old) <= 2^DEFER fits fp16 with the same relative precision, the accumulators need no rescale (64 multiplies
// per tile), and sum and output stay consistent because both use the same m.  Returns the tile's probabilities as the packed
// fp16 pairs pk[r >> 1] = (p[r], p[r + 1]).
#ifndef LM_DEFER
#define LM_DEFER_ON 1
#else
#define LM_DEFER_ON LM_DEFER
#endif
template <bool INTERIOR, int MB>
__device__ __forceinline__ void lm_softmax_tile(const f32x16& st, float scale2, int k0, int hh, int u_min, int uq, float& m_run,
                                                float& l_run, f32x16 (&o)[MB], uint32_t (&pk)[8]) {
  constexpr float DEFER = 11.0f;
  float tv[16];
  float tmax = -INFINITY;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    if (INTERIOR) {
      tv[r] = st[r] * scale2;
    } else {
      const int key = k0 + 8 * (r >> 2) + 4 * hh + (r & 3);
      tv[r] = (key >= u_min && key <= uq) ? st[r] * scale2 : -INFINITY;
    }
    tmax = fmaxf(tmax, tv[r]);
  }
  {
    // (inline asm, not the builtin: with the SAME value in both operands hipcc 7.2 keeps only the first result of the
    //  builtin — the generated code took max(sw[0], sw[0]) — measured as wrong logits; the two v_nop are the wait states
    //  the VALU-write -> permlane-read hazard asks for)
    float a = tmax, bcopy = tmax;
    asm volatile("v_nop\n\tv_nop\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(bcopy));
    tmax = fmaxf(a, bcopy);   // max over the lane pair (l, l ^ 32)
  }
  const bool defer = LM_DEFER_ON && __builtin_amdgcn_ballot_w64(tmax <= m_run + DEFER) == ~0ull;
  const float m_new = defer ? m_run : fmaxf(m_run, tmax);
  if (!defer) {
    const float corr = (m_run == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(m_run - m_new);
    l_run *= corr;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[mb][r] *= corr;
    m_run = m_new;
  }
  float psum = 0.f;
#pragma unroll
  for (int r = 0; r < 16; r += 2) {
    float p0 = __builtin_amdgcn_exp2f(tv[r] - m_new), p1 = __builtin_amdgcn_exp2f(tv[r + 1] - m_new);
    if (!INTERIOR) {   // (m_new may still be -inf for a query with no visible key so far: -inf - -inf is not a number)
      p0 = (tv[r] == -INFINITY) ? 0.f : p0;
      p1 = (tv[r + 1] == -INFINITY) ? 0.f : p1;
    }
    psum += p0 + p1;
    const half2_t h2 = {(half_t)p0, (half_t)p1};
    pk[r >> 1] = __builtin_bit_cast(uint32_t, h2);
  }
  {
    float a = psum, bcopy = psum;
    asm volatile("v_nop\n\tv_nop\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(bcopy));
    psum = a + bcopy;
  }
  l_run += psum;
}

#ifdef LM_ATTN_TIMELINE   // measurement builds: s_memtime stamps of one workgroup's wave 0 (tools/lab/lm_attn_timeline.py)
__device__ unsigned long long g_lm_tl[512];
extern "C" int rarc_lm_debug_timeline(unsigned long long* host, int n) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_lm_tl), sizeof(unsigned long long) * (size_t)n);
}
#define LM_TL(i) do { const int tl_k_ = (i); if (tl_on && tl_k_ < 512) g_lm_tl[tl_k_] = __builtin_readcyclecounter(); } while (0)
#else
#define LM_TL(i) do { } while (0)
#endif
struct LmAttnPrefix {
  const half_t* kv;        // [n_prefix][P][prs]: raw k rows (n_kv*DH) | v rows (n_kv*DH) of this layer; null = none
  const int32_t* pidx;     // [n_seq]: cache sequence of each sequence (-1: none)
  const int32_t* pstart;   // [n_prefix]: first real row of each cache sequence (left padded)
  int P, prs;
};
template <int DH>
__global__ __launch_bounds__(256, 2) void rarc_lm_attention_kernel(const half_t* __restrict__ qkv,
                                                                   const int32_t* __restrict__ start, int L, int n_q, int n_kv,
                                                                   int q_blocks, int wg_per_kv, const half_t* __restrict__ qn_w,
                                                                   const half_t* __restrict__ kn_w, float eps,
                                                                   const half2_t* __restrict__ rope, int rope_rows,
                                                                   half_t* __restrict__ ctx, const LmAttnPrefix pf) {
  constexpr int KS = DH / 16;
  constexpr int MB = DH / 32;
  constexpr int VROW = 40;              // halves per V^T row: 32 keys + pad (16-byte aligned rows)
  constexpr int KROW = DH + 8;          // halves per K row: DH + 16 bytes
  constexpr int CH = DH / 8;            // 16-byte chunks per row
  constexpr int RPP = 64 / CH;          // K rows per wave pass (4 at DH = 128, 8 at DH = 64)
  constexpr int NP = 8 / RPP;           // passes per wave and tile
  constexpr int VI = (16 * (CH / 4) + 63) / 64;   // V items per lane and tile (1)
  __shared__ __attribute__((aligned(16))) half_t kimg[2][32 * KROW];
  __shared__ __attribute__((aligned(16))) half_t vt[2][DH * VROW];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int G = n_q / n_kv, units = G * q_blocks;           // units of one (sequence, K/V head)
  // Workgroup ids go round-robin over the 8 XCDs, each with its own L2: taken as they come, the workgroups of one
  // (sequence, K/V head) — which read the SAME k / v rows — would sit on different XCDs and fetch every tile from HBM up
  // to wg_per_kv times.  XCD x instead owns a contiguous range of the logical ids, so they share an L2 (and run together).
  const int xcd = blockIdx.x & 7, per = gridDim.x >> 3, rem = gridDim.x & 7;
  const int lid = xcd * per + (xcd < rem ? xcd : rem) + (blockIdx.x >> 3);
  const int wg = lid % wg_per_kv, bk = lid / wg_per_kv;
  const int b = bk / n_kv, kvh = bk % n_kv;
  const int u = wg * 4 + wave;                              // this wave's unit (may be past the end: it only helps staging)
  const bool live = u < units;
  const int qb = live ? u / G : 0, hd = kvh * G + (live ? u % G : 0);
  int s0 = start[b];
  s0 = s0 < 0 ? 0 : (s0 > L - 1 ? L - 1 : s0);
  const int pq = pf.kv ? pf.pidx[b] : -1;
  const int P = pq >= 0 ? pf.P : 0;
  int u_min = 0;
  if (pq >= 0) { u_min = pf.pstart[pq]; u_min = u_min < 0 ? 0 : (u_min > P - 1 ? P - 1 : u_min); }
  const int pos_base = pq >= 0 ? 0 : s0;                    // rotary position of virtual key u = u + pos_base
  const int col = lane & 31, hh = lane >> 5;
  const size_t rs = (size_t)(n_q + 2 * n_kv) * DH;          // row stride of the fused qkv in halves
  const half_t* qbase = qkv + (size_t)b * L * rs + (size_t)hd * DH;
  const half_t* kbase = qkv + (size_t)b * L * rs + (size_t)(n_q + kvh) * DH;
  const half_t* vbase = qkv + (size_t)b * L * rs + (size_t)(n_q + n_kv + kvh) * DH;
  const half_t* pkbase = pq >= 0 ? pf.kv + (size_t)pq * P * pf.prs + (size_t)kvh * DH : nullptr;
  const half_t* pvbase = pq >= 0 ? pkbase + (size_t)n_kv * DH : nullptr;
  const float scale2 = (DH == 128 ? 0.08838834764831845f : 0.125f) * 1.44269504088896340736f;  // log2(e) / sqrt(DH)
  const int q0 = qb * 32;
  const int qpos = (q0 + col < L) ? q0 + col : L - 1;
  const int uq = P + (qpos - s0);                           // this lane's query in the virtual index (< P: a padding row)
  const int uq_first = P + (q0 - s0);                       // the wave's first query: keys up to it are visible to all 32
  auto rope_row = [&](int uu) -> const half2_t* {
    int r = uu + pos_base;
    r = r < 0 ? 0 : (r > rope_rows - 1 ? rope_rows - 1 : r);
    return rope + (size_t)r * (DH / 2);
  };

#ifdef LM_ATTN_TIMELINE
  const bool tl_on = blockIdx.x == gridDim.x / 2 + 3 && threadIdx.x == 0;
  int tl_i = 1;
#endif
  f32x16 o[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) o[mb] = (f32x16){0};
  float m_run = -INFINITY, l_run = 0.f;
  // causal: no key beyond the block's last query (virtual index, exclusive)
  const int k_end = live ? P + (((q0 + 32 < L) ? q0 + 32 : L) - s0) : 0;
  // the workgroup's last unit has the largest query block: its key range is the workgroup's
  const int last_u = (wg * 4 + 3 < units ? wg * 4 + 3 : units - 1), last_q0 = (last_u / G) * 32;
  const int wg_k_end = P + (((last_q0 + 32 < L) ? last_q0 + 32 : L) - s0);

  // staging roles: K row = 8*wave + pass*RPP + lane / CH, chunk c = lane % CH; the k-norm weights and this lane's
  // rotary sign are fixed for the whole kernel
  const int kc = lane % CH, kr_in = lane / CH;
  const half8 kw = *(const half8*)(kn_w + 8 * kc);
  const bool upper = kc >= CH / 2;                           // this chunk holds elements i + DH/2 of the rotation pairs
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  // row pointers of virtual key uu (cache row or own row; clamped into what exists: out-of-range keys are masked)
  auto krow_ptr = [&](int uu) -> const half_t* {
    if (uu < P) return pkbase + (size_t)(uu < 0 ? 0 : uu) * pf.prs;
    int i = uu - P + s0;
    return kbase + (size_t)(i > L - 1 ? L - 1 : i) * rs;
  };
  auto vrow_ptr = [&](int uu) -> const half_t* {
    if (uu < P) return pvbase + (size_t)(uu < 0 ? 0 : uu) * pf.prs;
    int i = uu - P + s0;
    return vbase + (size_t)(i > L - 1 ? L - 1 : i) * rs;
  };
  // ---- prefetch registers: the raw rows of the NEXT tile ----
  half8 pk_x[NP];
  u32x4 pk_c0[NP], pk_c1[NP];
  half8 pv0[VI], pv1[VI];
  auto prefetch = [&](int k0) {
#pragma unroll
    for (int pass = 0; pass < NP; ++pass) {
      const int uu = k0 + 8 * wave + pass * RPP + kr_in;
      pk_x[pass] = *(const half8*)(krow_ptr(uu) + 8 * kc);
      const half_t* cs_row = (const half_t*)rope_row(uu) + 8 * (kc % (CH / 2));
      pk_c0[pass] = *(const u32x4*)cs_row;                // cos of the chunk's 8 pairs
      pk_c1[pass] = *(const u32x4*)(cs_row + DH / 2);     // their sin
    }
#pragma unroll
    for (int it = 0; it < VI; ++it) {
      const int i = lane + 64 * it;
      if (i < 16 * (CH / 4)) {
        const int p = i & 15, c8 = (CH / 4) * wave + (i >> 4);
        pv0[it] = *(const half8*)(vrow_ptr(k0 + 2 * p) + 8 * c8);
        pv1[it] = *(const half8*)(vrow_ptr(k0 + 2 * p + 1) + 8 * c8);
      }
    }
  };
  // ---- registers -> LDS image `buf`: K rows 8*wave .. +7 (RMSNorm, rotary), this wave's d-chunks of V^T ----
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int pass = 0; pass < NP; ++pass) {
      const int r = 8 * wave + pass * RPP + kr_in;
      const half8 x = pk_x[pass];
      float ss = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) ss = __builtin_fmaf((float)x[e], (float)x[e], ss);
      // (the two halves of the row first, then across the chunks: the order the resident kernel's lanes — which hold both
      //  halves — sum in)
      ss += __shfl_xor(ss, CH / 2, 64);
#pragma unroll
      for (int o2 = 1; o2 < CH / 2; o2 <<= 1) ss += __shfl_xor(ss, o2, 64);
      const float inv = 1.0f / __builtin_sqrtf(ss / (float)DH + eps);
      const half8 xn = kw * lm_scale8(x, inv);
      u32x4 mine = __builtin_bit_cast(u32x4, xn), other;
#pragma unroll
      for (int w4 = 0; w4 < 4; ++w4) other[w4] = (uint32_t)__shfl_xor((int)mine[w4], CH / 2, 64);
      const half8 pn = __builtin_bit_cast(half8, other);
      const half8 cs = __builtin_bit_cast(half8, pk_c0[pass]), sn = __builtin_bit_cast(half8, pk_c1[pass]);
      // lower half: a cos - b sin with a = own, b = partner; upper half: b cos + a sin with b = own, a = partner (lm_rotate8)
      const half8 t = pn * sn;
      const half8 out = __builtin_elementwise_fma(xn, cs, upper ? t : -t);
      *(half8*)(kimg[buf] + r * KROW + 8 * kc) = out;
    }
#pragma unroll
    for (int it = 0; it < VI; ++it) {
      const int i = lane + 64 * it;
      if (i < 16 * (CH / 4)) {
        const int p = i & 15, c8 = (CH / 4) * wave + (i >> 4);
#pragma unroll
        for (int e = 0; e < 8; ++e) *(half2_t*)(vt[buf] + (8 * c8 + e) * VROW + 2 * p) = (half2_t){pv0[it][e], pv1[it][e]};
      }
    }
  };

  LM_TL(0);
  const int k_first = (u_min / 32) * 32;
  if (k_first < wg_k_end) prefetch(k_first);               // the first tile's rows travel while q is loaded and normed
  half8 qf[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) qf[ks] = *(const half8*)(qbase + (size_t)qpos * rs + 16 * ks + 8 * hh);
  lm_norm_rope<DH>(qf, qn_w, eps, rope_row(uq), hh);
  LM_TL(tl_i++);
  int buf = 0;
  for (int k0 = k_first; k0 < wg_k_end; k0 += 32, buf ^= 1) {
    // image `buf` was last read while tile k0 - 64 was multiplied; every wave has passed the barrier of tile k0 - 32
    // since, i.e. finished with it
    store_tile(buf);
    LM_TL(tl_i++);
    __syncthreads();
    LM_TL(tl_i++);
    if (k0 + 32 < wg_k_end) prefetch(k0 + 32);               // in flight under this tile's MFMAs
    if (k0 >= k_end) continue;                               // beyond this wave's causal range: it only staged

    const half_t* kim = kimg[buf];
    const half_t* vim = vt[buf];
    f32x16 st = {0};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const half8 kf = *(const half8*)(kim + col * KROW + 16 * ks + 8 * hh);
      st = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[ks], st, 0, 0, 0);
    }
    RARC_MFMA_SETTLE(st);
    // ---- online softmax in the log2 domain: t = s * (scale * log2 e), p = 2^(t - m) ----
    // An INTERIOR tile (every key of it visible to every query of the wave: all but the first tile of a left-padded or
    // prefixed sequence and the tile on the diagonal) needs no mask, no -inf guards.  The running maximum is only raised
    // when some query's tile maximum exceeds it by more than 2^DEFER (wave-uniform decision): otherwise p = 2^(t - m_old)
    // <= 2^DEFER fits fp16 with the same relative precision, the accumulators need no rescale (64 multiplies per tile),
    // and sum and output stay consistent because both use the same m.
#ifndef LM_INTERIOR
#define LM_INTERIOR 1
#endif
#ifndef LM_DEFER
#define LM_DEFER 1
#endif
#ifndef LM_SWAP_STATS
#define LM_SWAP_STATS 1
#endif
#ifndef LM_SWAP_P
#define LM_SWAP_P 1
#endif
    constexpr float DEFER = 11.0f;
    const bool interior = LM_INTERIOR && k0 >= u_min && k0 + 31 <= uq_first;
    float tv[16];
    float tmax = -INFINITY;
    if (interior) {
#pragma unroll
      for (int r = 0; r < 16; ++r) { tv[r] = st[r] * scale2; tmax = fmaxf(tmax, tv[r]); }
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = k0 + 8 * (r >> 2) + 4 * hh + (r & 3);
        tv[r] = (key >= u_min && key <= uq) ? st[r] * scale2 : -INFINITY;
        tmax = fmaxf(tmax, tv[r]);
      }
    }
    if (LM_SWAP_STATS) {
      // (inline asm, not the builtin: with the SAME value in both operands hipcc 7.2 keeps only the first result of the
      //  builtin — the generated code took max(sw[0], sw[0]) — measured as wrong logits; the two v_nop are the wait states
      //  the VALU-write -> permlane-read hazard asks for)
      float a = tmax, bcopy = tmax;
      asm volatile("v_nop\n\tv_nop\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(bcopy));
      const float sw[2] = {a, bcopy};
      tmax = fmaxf(sw[0], sw[1]);   // max over the lane pair (l, l ^ 32)
    } else {
      tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    }
    const bool defer = LM_DEFER && __builtin_amdgcn_ballot_w64(tmax <= m_run + DEFER) == ~0ull;
    const float m_new = defer ? m_run : fmaxf(m_run, tmax);
    if (!defer) {
      const float corr = (m_run == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(m_run - m_new);
      l_run *= corr;
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[mb][r] *= corr;
      m_run = m_new;
    }
    float psum = 0.f;
    uint32_t pk[8];
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      float p0, p1;
      if (interior) {
        p0 = __builtin_amdgcn_exp2f(tv[r] - m_new);
        p1 = __builtin_amdgcn_exp2f(tv[r + 1] - m_new);
      } else {   // (m_new may still be -inf for a query with no visible key so far: -inf - -inf is not a number)
        p0 = (tv[r] == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(tv[r] - m_new);
        p1 = (tv[r + 1] == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(tv[r + 1] - m_new);
      }
      psum += p0 + p1;
      const half2_t h2 = {(half_t)p0, (half_t)p1};
      pk[r >> 1] = __builtin_bit_cast(uint32_t, h2);
    }
    if (LM_SWAP_STATS) {
      float a = psum, bcopy = psum;
      asm volatile("v_nop\n\tv_nop\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(bcopy));
      psum = a + bcopy;
    } else {
      psum += __shfl_xor(psum, 32, 64);
    }
    l_run += psum;
    // ---- O^T += V^T · P^T.  k-step ks contracts keys 16ks .. 16ks+15: half hh needs keys 16ks + 8hh .. + 7.  A lane holds
    // key groups g = 0..3 (keys 8g + 4hh .. + 3) as the word pairs pk[2g], pk[2g+1]; one v_permlane32_swap per word with
    // vdst = group 2ks, src = group 2ks + 1 leaves [own | partner's] of group 2ks in the low half and [partner's | own] of
    // group 2ks + 1 in the high half: exactly the two B fragments (the four ds_bpermute + selects this replaces) ----
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      u32x4 bw;
      if (LM_SWAP_P) {
        const auto w0 = __builtin_amdgcn_permlane32_swap(pk[4 * ks], pk[4 * ks + 2], false, false);
        const auto w1 = __builtin_amdgcn_permlane32_swap(pk[4 * ks + 1], pk[4 * ks + 3], false, false);
        bw = (u32x4){w0[0], w1[0], w0[1], w1[1]};
      } else {
        const uint32_t mine0 = hh ? pk[4 * ks + 2] : pk[4 * ks], mine1 = hh ? pk[4 * ks + 3] : pk[4 * ks + 1];
        const uint32_t send0 = hh ? pk[4 * ks] : pk[4 * ks + 2], send1 = hh ? pk[4 * ks + 1] : pk[4 * ks + 3];
        const uint32_t recv0 = (uint32_t)__shfl_xor((int)send0, 32, 64), recv1 = (uint32_t)__shfl_xor((int)send1, 32, 64);
        bw = hh ? (u32x4){recv0, recv1, mine0, mine1} : (u32x4){mine0, mine1, recv0, recv1};
      }
      const half8 pfr = __builtin_bit_cast(half8, bw);
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
        const half8 vf = *(const half8*)(vim + (32 * mb + col) * VROW + 16 * ks + 8 * hh);
        o[mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pfr, o[mb], 0, 0, 0);
      }
    }
    RARC_MFMA_SETTLE(o);
    LM_TL(tl_i++);
  }
  LM_TL(tl_i++);
  if (live && q0 + col < L) {
    const float inv = l_run > 0.f ? 1.f / l_run : 0.f;  // (a padding query attends to nothing: zeros)
    half_t* out = ctx + ((size_t)b * L + q0 + col) * (size_t)n_q * DH + (size_t)hd * DH;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const half4v w = {(half_t)(o[mb][4 * g] * inv), (half_t)(o[mb][4 * g + 1] * inv),
                          (half_t)(o[mb][4 * g + 2] * inv), (half_t)(o[mb][4 * g + 3] * inv)};
        *(half4v*)(out + 32 * mb + 8 * g + 4 * hh) = w;
      }
  }
}

// ---- the same attention for SHORT sequences: the whole key range of a (sequence, K/V head) resident in LDS -----------
// Round 3.  The streaming kernel above stages every 32-key tile once per workgroup of four (q head, query block) units —
// at 16 units per (sequence, K/V head) that is four times the per-head RMSNorm + rotary work on k and four times the
// V transposition, one barrier per tile, and ~6 k shader cycles per tile for 16 MFMAs (s_memtime timeline: 2.1-5.5 k cycles
// for the tile's norm / rotary / LDS image, 3.8 k for a masked tile's scores + softmax + P·V with two workgroups sharing the
// SIMDs).  When P + L keys fit in the CU's 160 KiB (288 keys at head_dim 128: every reranker prompt whose remainder after
// the shared prefix is <= ~200 tokens) ONE workgroup of 8 waves owns the (sequence, K/V head):
//   phase 1  all 512 threads build the operand images of ALL keys — K rows (norm + rotary) and V^T — with every global load
//            of a group of passes issued before its first use: one exposed memory latency, each key prepared once;
//   barrier
//   phase 2  the G x q_blocks units are dealt to the waves largest first, boustrophedon (wave w: units w, 15 - w, 16 + w, ...:
//            the causal triangle balances to ~87 %), and a wave runs its units with NO further barrier and no global load
//            in the key loop (the next unit's q rows are fetched under the current one): S^T = K·Q^T of tile t + 1 is issued
//            before the softmax of tile t, so one wave keeps the matrix pipe busy under its own VALU work.
// Same arithmetic, same roundings and the same summation order per query as the streaming kernel (tiles of 32 keys in
// ascending order, deferred rescale): the two kernels return identical bits (tests/test_gpu_reranker_lm.py).
__host__ __device__ inline int lm_attn_vrow(int keys) { return keys <= 40 ? 40 : 40 + 128 * ((keys - 40 + 127) / 128); }
template <int DH, int NP>   // NP: passes of K rows the staging registers hold (kcap <= NP * 64 keys at head_dim 128, NP * 128 at 64)
__global__ __launch_bounds__(512, 1) void rarc_lm_attention_resident_kernel(const half_t* __restrict__ qkv,
                                                                            const int32_t* __restrict__ start, int L, int n_q,
                                                                            int n_kv, int q_blocks, const half_t* __restrict__ qn_w,
                                                                            const half_t* __restrict__ kn_w, float eps,
                                                                            const half2_t* __restrict__ rope, int rope_rows,
                                                                            half_t* __restrict__ ctx, const LmAttnPrefix pf,
                                                                            int kcap, int VROW) {
  constexpr int KS = DH / 16;
  constexpr int MB = DH / 32;
  constexpr int KROW = DH + 8;          // halves per K row: DH + 16 bytes
  constexpr int CH = DH / 8;            // 16-byte chunks per row
  constexpr int LPR = CH / 2;           // lanes per K row: a lane owns chunks c and c + CH/2 — both halves of its rotation pairs
  constexpr int RPW = 64 / LPR;         // K rows per wave and pass
  constexpr int RPP = 8 * RPW;          // K rows per workgroup pass (64 at DH = 128, 128 at DH = 64)
  extern __shared__ __attribute__((aligned(16))) char lm_attn_smem[];
  half_t* kimg = (half_t*)lm_attn_smem;                 // [kcap][KROW]
  half_t* vt = kimg + (size_t)kcap * KROW;              // [DH][VROW]: VROW = 40 mod 128 halves, the bank pattern of the streaming kernel's 40
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int G = n_q / n_kv, units = G * q_blocks;
  const int xcd = blockIdx.x & 7, per = gridDim.x >> 3, rem = gridDim.x & 7;
  const int lid = xcd * per + (xcd < rem ? xcd : rem) + (blockIdx.x >> 3);
  const int b = lid / n_kv, kvh = lid % n_kv;
  int s0 = start[b];
  s0 = s0 < 0 ? 0 : (s0 > L - 1 ? L - 1 : s0);
  const int pq = pf.kv ? pf.pidx[b] : -1;
  const int P = pq >= 0 ? pf.P : 0;
  int u_min = 0;
  if (pq >= 0) { u_min = pf.pstart[pq]; u_min = u_min < 0 ? 0 : (u_min > P - 1 ? P - 1 : u_min); }
  const int pos_base = pq >= 0 ? 0 : s0;
  const int col = lane & 31, hh = lane >> 5;
  const size_t rs = (size_t)(n_q + 2 * n_kv) * DH;
  const half_t* seq_base = qkv + (size_t)b * L * rs;
  const half_t* kbase = seq_base + (size_t)(n_q + kvh) * DH;
  const half_t* vbase = seq_base + (size_t)(n_q + n_kv + kvh) * DH;
  const half_t* pkbase = pq >= 0 ? pf.kv + (size_t)pq * P * pf.prs + (size_t)kvh * DH : nullptr;
  const half_t* pvbase = pq >= 0 ? pkbase + (size_t)n_kv * DH : nullptr;
  const float scale2 = (DH == 128 ? 0.08838834764831845f : 0.125f) * 1.44269504088896340736f;
  auto rope_row = [&](int uu) -> const half2_t* {
    int r = uu + pos_base;
    r = r < 0 ? 0 : (r > rope_rows - 1 ? rope_rows - 1 : r);
    return rope + (size_t)r * (DH / 2);
  };
  // (bit blends, not branches — hipcc turns a select between two address computations back into a divergent branch, and a
  //  load under a divergent branch is waited for before the next one is issued)
  auto row_ptr = [&](int uu, const half_t* pre, const half_t* own) -> const half_t* {
    const int i = uu - P + s0;
    const uint32_t m = uu < P ? 0xffffffffu : 0u;
    const uint32_t off = (((uint32_t)(uu < 0 ? 0 : uu) * (uint32_t)pf.prs) & m) | (((uint32_t)(i > L - 1 ? L - 1 : i) * (uint32_t)rs) & ~m);
    const uint64_t m64 = ((uint64_t)m << 32) | m;
    const uint64_t base = ((uint64_t)pre & m64) | ((uint64_t)own & ~m64);
    return (const half_t*)base + off;
  };
  auto krow_ptr = [&](int uu) -> const half_t* { return row_ptr(uu, pkbase, kbase); };
  auto vrow_ptr = [&](int uu) -> const half_t* { return row_ptr(uu, pvbase, vbase); };
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#ifdef LM_ATTN_TIMELINE
  const bool tl_on = blockIdx.x == gridDim.x / 2 + 3 && threadIdx.x == 0;
  int tl_i = 1;
#endif
  LM_TL(0);
  const int k_first = (u_min / 32) * 32;
  const int k_total = P + (L - s0);                           // virtual keys [k_first, k_total) exist
  int n_tiles = (k_total - k_first + 31) / 32;
  if (n_tiles * 32 > kcap) n_tiles = kcap / 32;               // (the host only launches this kernel when everything fits)

  // ---- the wave's unit list: j-th largest unit = (query block q_blocks - 1 - j / G, q head j % G) ----
  auto unit_of = [&](int r) -> int { return (r & 1) ? 8 * r + 7 - wave : 8 * r + wave; };
  half8 qraw[KS];
  auto load_q = [&](int j) {
    const int qb = q_blocks - 1 - j / G, hd = kvh * G + j % G;
    const int q0 = qb * 32, qpos = (q0 + col < L) ? q0 + col : L - 1;
    const half_t* qp = seq_base + (size_t)qpos * rs + (size_t)hd * DH;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qraw[ks] = *(const half8*)(qp + 16 * ks + 8 * hh);
  };

  // ---- phase 1: every global load of the workgroup's keys is issued before the first one is used (one workgroup per CU:
  // nothing else would hide a second round trip), then K rows -> RMSNorm, rotary -> row-major image, V rows -> V^T.
  // The loads are unconditional (rows past the end are clamped onto the last one, only the LDS writes are predicated):
  // with a branch per pass hipcc interleaves loads, waits and unpacking ----
  {
    constexpr int MAXV = NP;                                    // V items per thread
    const int kc = lane % LPR, kr_in = lane / LPR;
    const half8 kw_lo = *(const half8*)(kn_w + 8 * kc), kw_hi = *(const half8*)(kn_w + DH / 2 + 8 * kc);
    const int rows = n_tiles * 32;                              // <= NP * RPP (host)
    const int items = n_tiles * 16 * CH;                        // (key pair, 8-d chunk) items of V
    half8 xl[NP], xh[NP];
    u32x4 c0[NP], c1[NP];
#pragma unroll
    for (int g = 0; g < NP; ++g) {
      const int uu = k_first + g * RPP + wave * RPW + kr_in;
      const half_t* kp = krow_ptr(uu) + 8 * kc;
      xl[g] = *(const half8*)kp;
      xh[g] = *(const half8*)(kp + DH / 2);
      const half_t* cs_row = (const half_t*)rope_row(uu) + 8 * kc;
      c0[g] = *(const u32x4*)cs_row;                      // cos of the chunk's 8 pairs
      c1[g] = *(const u32x4*)(cs_row + DH / 2);           // their sin
    }
    half8 v0[MAXV], v1[MAXV];
#pragma unroll
    for (int g = 0; g < MAXV; ++g) {
      const int i = threadIdx.x + 512 * g;
      const int p = i & 15, c8 = (i >> 4) % CH, tile = i / (16 * CH);
      const int uu = k_first + 32 * tile + 2 * p;
      v0[g] = *(const half8*)(vrow_ptr(uu) + 8 * c8);
      v1[g] = *(const half8*)(vrow_ptr(uu + 1) + 8 * c8);
    }
    LM_TL(tl_i++);   // loads issued
    // (every loaded register passes through an opaque statement: nothing of the processing below moves up between the loads)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int g = 0; g < NP; ++g) asm volatile("" : "+v"(xl[g]), "+v"(xh[g]), "+v"(c0[g]), "+v"(c1[g]));
#pragma unroll
    for (int g = 0; g < MAXV; ++g) asm volatile("" : "+v"(v0[g]), "+v"(v1[g]));
    __builtin_amdgcn_sched_barrier(0);
    LM_TL(tl_i++);   // loads arrived
#pragma unroll
    for (int g = 0; g < NP; ++g) {
      const int r = g * RPP + wave * RPW + kr_in;
      float ss = 0.f, ss_hi = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) ss = __builtin_fmaf((float)xl[g][e], (float)xl[g][e], ss);
#pragma unroll
      for (int e = 0; e < 8; ++e) ss_hi = __builtin_fmaf((float)xh[g][e], (float)xh[g][e], ss_hi);
      ss += ss_hi;
#pragma unroll
      for (int o2 = 1; o2 < LPR; o2 <<= 1) ss += __shfl_xor(ss, o2, 64);
      const float inv = 1.0f / __builtin_sqrtf(ss / (float)DH + eps);
      const half8 a = kw_lo * lm_scale8(xl[g], inv), b = kw_hi * lm_scale8(xh[g], inv);
      half8 lo, hi;
      lm_rotate8(a, b, __builtin_bit_cast(half8, c0[g]), __builtin_bit_cast(half8, c1[g]), lo, hi);
      if (r < rows) {
        *(half8*)(kimg + r * KROW + 8 * kc) = lo;
        *(half8*)(kimg + r * KROW + DH / 2 + 8 * kc) = hi;
      }
    }
    LM_TL(tl_i++);   // K image written
#pragma unroll
    for (int g = 0; g < MAXV; ++g) {
      const int i = threadIdx.x + 512 * g;
      if (i < items) {
        const int p = i & 15, c8 = (i >> 4) % CH, tile = i / (16 * CH);
#pragma unroll
        for (int e = 0; e < 8; ++e) *(half2_t*)(vt + (8 * c8 + e) * VROW + 32 * tile + 2 * p) = (half2_t){v0[g][e], v1[g][e]};
      }
    }
  }
  LM_TL(tl_i++);     // V^T written
  __syncthreads();
  LM_TL(tl_i++);     // barrier passed

  // ---- phase 2: this wave's units, no barrier from here on ----
  for (int r = 0;; ++r) {
    const int j = unit_of(r);
    if (j >= units) break;
    const int qb = q_blocks - 1 - j / G, hd = kvh * G + j % G;
    const int q0 = qb * 32;
    const int qpos = (q0 + col < L) ? q0 + col : L - 1;
    const int uq = P + (qpos - s0);
    const int uq_first = P + (q0 - s0);
    half8 qf[KS];
    load_q(j);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[ks] = qraw[ks];
    // (the q-norm weights are the same for every unit: hoisted out of this loop they would sit in 32 registers through the
    //  key loops — spilled, reloaded from scratch per unit; the pointer goes through an opaque copy instead)
    const half_t* qn_w_unit = qn_w;
    asm volatile("" : "+s"(qn_w_unit));
    lm_norm_rope<DH>(qf, qn_w_unit, eps, rope_row(uq), hh);
    LM_TL(tl_i++);   // unit: q ready
    f32x16 o[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) o[mb] = (f32x16){0};
    float m_run = -INFINITY, l_run = 0.f;
    int k_end = P + (((q0 + 32 < L) ? q0 + 32 : L) - s0);     // causal: no key beyond the block's last query (exclusive)
    if (k_end > k_first + n_tiles * 32) k_end = k_first + n_tiles * 32;
    auto qk_tile = [&](int k0) -> f32x16 {
      const half_t* kim = kimg + (size_t)(k0 - k_first) * KROW;
      f32x16 st = {0};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const half8 kf = *(const half8*)(kim + col * KROW + 16 * ks + 8 * hh);
        st = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[ks], st, 0, 0, 0);
      }
      RARC_MFMA_SETTLE(st);
      return st;
    };
    f32x16 st = {0};
    if (k_first < k_end) st = qk_tile(k_first);
    for (int k0 = k_first; k0 < k_end; k0 += 32) {
      f32x16 st_next = {0};
      if (k0 + 32 < k_end) st_next = qk_tile(k0 + 32);          // the matrix pipe works on the next tile under this softmax
      uint32_t pk[8];
      if (k0 >= u_min && k0 + 31 <= uq_first) lm_softmax_tile<true, MB>(st, scale2, k0, hh, u_min, uq, m_run, l_run, o, pk);
      else lm_softmax_tile<false, MB>(st, scale2, k0, hh, u_min, uq, m_run, l_run, o, pk);
      const half_t* vim = vt + (k0 - k_first);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const auto w0 = __builtin_amdgcn_permlane32_swap(pk[4 * ks], pk[4 * ks + 2], false, false);
        const auto w1 = __builtin_amdgcn_permlane32_swap(pk[4 * ks + 1], pk[4 * ks + 3], false, false);
        const u32x4 bw = (u32x4){w0[0], w1[0], w0[1], w1[1]};
        const half8 pfr = __builtin_bit_cast(half8, bw);
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
          const half8 vf = *(const half8*)(vim + (32 * mb + col) * VROW + 16 * ks + 8 * hh);
          o[mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pfr, o[mb], 0, 0, 0);
        }
      }
      RARC_MFMA_SETTLE(o);   // (the loop's register moves read the accumulators at once: rarc_common.h)
      st = st_next;
    }
    LM_TL(tl_i++);   // unit: key loop done
    if (q0 + col < L) {
      const float inv = l_run > 0.f ? 1.f / l_run : 0.f;
      half_t* out = ctx + ((size_t)b * L + q0 + col) * (size_t)n_q * DH + (size_t)hd * DH;
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const half4v w = {(half_t)(o[mb][4 * g] * inv), (half_t)(o[mb][4 * g + 1] * inv),
                            (half_t)(o[mb][4 * g + 2] * inv), (half_t)(o[mb][4 * g + 3] * inv)};
          *(half4v*)(out + 32 * mb + 8 * g + 4 * hh) = w;
        }
    }
  }
  LM_TL(tl_i++);     // all units stored
}

// ---- SwiGLU: h[t][j] = silu(g[t][j]) · u[t][j] from the fused gate/up GEMM output, whose columns come in groups of 16:
// 8 gates, then the 8 ups of the same features (RarcLmLayer.gate_up_w, rarc.h).  Only for shapes the GEMM's own
// SwiGLU epilogue (encoder.hip, act = 3) does not take: small batches.
__global__ __launch_bounds__(256) void rarc_lm_swiglu_kernel(const half_t* gu, int n_tokens, int I, half_t* h) {
  const size_t n8 = (size_t)n_tokens * I / 8;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
    const size_t t = i / (I / 8), c = (i % (I / 8)) * 8;
    const half8 g = *(const half8*)(gu + t * 2 * I + 2 * c), u = *(const half8*)(gu + t * 2 * I + 2 * c + 8);
    half8 o8;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float gf = (float)g[e];
      const half_t act = (half_t)(gf / (1.0f + __expf(-gf)));  // silu, rounded to fp16 like the activation tensor
      o8[e] = (half_t)((float)act * (float)u[e]);
    }
    *(half8*)(h + t * I + c) = o8;
  }
}

// ---- last position: final RMSNorm, then the two logits (no, yes) = <normed, lm_head[id]> --------------------------
__global__ __launch_bounds__(64) void rarc_lm_last_logits_kernel(const half_t* x, const half_t* w_final, const half_t* lm_head,
                                                                 float eps, int row_stride, int row_off, int H, int no_id,
                                                                 int yes_id, half_t* out) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const half_t* xr = x + ((size_t)b * row_stride + row_off) * H;  // [T][H] rows: stride L, offset L-1; compact rows: 1, 0
  float ss = 0.f;
  for (int c = lane; c < H; c += 64) ss = __builtin_fmaf((float)xr[c], (float)xr[c], ss);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
  const float inv = 1.0f / __builtin_sqrtf(ss / (float)H + eps);
  float a_no = 0.f, a_yes = 0.f;
  for (int c = lane; c < H; c += 64) {
    const float nv = (float)(half_t)((float)w_final[c] * (float)(half_t)((float)xr[c] * inv));
    a_no = __builtin_fmaf(nv, (float)lm_head[(size_t)no_id * H + c], a_no);
    a_yes = __builtin_fmaf(nv, (float)lm_head[(size_t)yes_id * H + c], a_yes);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    a_no += __shfl_xor(a_no, o, 64);
    a_yes += __shfl_xor(a_yes, o, 64);
  }
  if (lane == 0) {
    out[2 * b] = (half_t)a_no;      // logits are an fp16 tensor in the reference's fp16 model
    out[2 * b + 1] = (half_t)a_yes;
  }
}

// ---- last positions: dst[s] = src[s*L + L-1] for s < n_seq, zero rows up to n_rows (a multiple of 128 for the GEMMs) ----
__global__ __launch_bounds__(256) void rarc_lm_gather_last_kernel(const half_t* src, int L, int W, int n_seq, int n_rows, half_t* dst) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (r >= n_rows) return;
  const half8* from = (const half8*)(src + ((size_t)r * L + (L - 1)) * W);
  half8* to = (half8*)(dst + (size_t)r * W);
  for (int c = lane; c < W / 8; c += 64) to[c] = r < n_seq ? from[c] : (half8){0};
}

// ---- host ------------------------------------------------------------------------------------------------------
static inline size_t lm_align(size_t x) { return (x + 255) & ~(size_t)255; }

extern "C" size_t rarc_lm_workspace_bytes(const RarcLmModel* m, int n_tokens) {
  if (!m || n_tokens <= 0) return 0;
  // (buffers hold the token count rounded up to 256 rows: a 128-row remainder is padded inside, see lm_forward)
  const size_t T = ((size_t)n_tokens + 255) / 256 * 256, qkv = (size_t)(m->n_q_heads + 2 * m->n_kv_heads) * m->head_dim;
  return 3 * lm_align(T * m->hidden * 2)                        // x, h (normed), delta (projection outputs)
         + lm_align(T * qkv * 2)                                // fused q | k | v
         + lm_align(T * (size_t)m->n_q_heads * m->head_dim * 2)  // attention context
         + lm_align(T * 2 * (size_t)m->inter * 2)               // fused gate | up
         + lm_align(T * (size_t)m->inter * 2)                   // silu(gate) * up
         + lm_align((T + 4096) * (size_t)(m->head_dim / 2) * 4)   // rotary (cos, sin) table: prefix + sequence positions
         + lm_align(T * 4)                                       // row scales of the folded RMSNorms
         + lm_align(T * (size_t)(m->hidden / 32) * 4);           // their partial sums of squares (one per 32 columns)
}

// ---- prefix K/V cache (see LmAttnPrefix): per layer [n_prefix * P][2 * n_kv * head_dim] raw k | v rows -----------------
extern "C" size_t rarc_lm_prefix_cache_bytes(const RarcLmModel* m, int n_prefix_tokens) {
  if (!m || n_prefix_tokens <= 0) return 0;
  return (size_t)m->n_layers * lm_align((size_t)n_prefix_tokens * 2 * m->n_kv_heads * m->head_dim * 2);
}

// k | v columns of the fused qkv rows -> cache rows (16-byte pieces, one wave per row)
__global__ __launch_bounds__(256) void rarc_lm_copy_kv_kernel(const half_t* __restrict__ qkv, int n_rows, int rs, int q_cols, int kv_cols,
                                                              half_t* __restrict__ cache) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (r >= n_rows) return;
  const half8* from = (const half8*)(qkv + (size_t)r * rs + q_cols);
  half8* to = (half8*)(cache + (size_t)r * kv_cols);
  for (int c = lane; c < kv_cols / 8; c += 64) to[c] = from[c];
}

// mode 0: logits of the last positions (d_out_f16); `use` = optional prefix cache the sequences attend to first
// mode 1: fill `fill` (a prefix cache) with the k | v rows of every layer for these sequences; no logits, and the last
//         layer stops after its q|k|v projection (nothing else of it is ever read)
struct LmPrefixUse { const half_t* cache; const int32_t* pidx; const int32_t* pstart; int n_prefix, P; };
static int lm_forward(const RarcLmModel* m, const int32_t* d_ids, const int32_t* d_start, int n_seq, int seq_len, int no_id,
                      int yes_id, void* d_ws, size_t ws_bytes, uint16_t* d_out_f16, void* stream, const LmPrefixUse* use,
                      half_t* fill) {
  RARC_REQUIRE(m && m->layers && d_ids && d_start && d_ws && (d_out_f16 || fill), RARC_E_INVALID, "rarc_lm_yes_no_logits: null pointer");
  RARC_REQUIRE(m->embed && m->lm_head && m->final_norm && m->zero_bias, RARC_E_INVALID, "rarc_lm_yes_no_logits: incomplete model");
  const int H = m->hidden, I = m->inter, DH = m->head_dim, NQ = m->n_q_heads, NKV = m->n_kv_heads;
  RARC_REQUIRE(n_seq > 0 && seq_len > 0 && m->n_layers > 0, RARC_E_INVALID, "rarc_lm_yes_no_logits: empty batch or model");
  RARC_REQUIRE((DH == 64 || DH == 128) && NQ > 0 && NKV > 0 && NQ % NKV == 0, RARC_E_UNSUPPORTED,
               "rarc_lm_yes_no_logits: head_dim must be 64 or 128 and the q heads a multiple of the kv heads");
  const int QKV = (NQ + 2 * NKV) * DH, QD = NQ * DH;
  RARC_REQUIRE(H % 128 == 0 && I % 128 == 0 && QKV % 128 == 0 && QD % 64 == 0 && H <= 8192, RARC_E_UNSUPPORTED,
               "rarc_lm_yes_no_logits: hidden, inter and the fused qkv width must be multiples of 128");
  RARC_REQUIRE(m->vocab > 0 && no_id >= 0 && yes_id >= 0 && no_id < m->vocab && yes_id < m->vocab, RARC_E_INVALID,
               "rarc_lm_yes_no_logits: token ids outside the vocabulary");
  const long long t_ll = (long long)n_seq * seq_len;
  RARC_REQUIRE(t_ll % 128 == 0 && t_ll < (1ll << 31), RARC_E_UNSUPPORTED,
               "rarc_lm_yes_no_logits: n_seq*seq_len must be a multiple of 128 (got %lld)", t_ll);
  const int T = (int)t_ll;
  RARC_REQUIRE(ws_bytes >= rarc_lm_workspace_bytes(m, T), RARC_E_WORKSPACE, "rarc_lm_yes_no_logits: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  char* w = (char*)d_ws;
  // The 256-row tile kernels (and the SwiGLU epilogue that only they have) take M % 256 == 0: a batch of T = 256 j + 128
  // tokens runs its GEMMs on Tg = T + 128 rows, the extra rows zero (128 rows of waste against the 128-row kernels for
  // the whole GEMM, which is what an odd multiple of 128 used to get).  Everything that is not a GEMM works on the T real rows.
  const int Tg = (T % 256 && T >= 2048) ? T + 128 : T;
  const size_t Tw = ((size_t)T + 255) / 256 * 256;
  const size_t th = lm_align(Tw * H * 2);
  half_t* x = (half_t*)w;
  half_t* h = (half_t*)(w + th);
  half_t* delta = (half_t*)(w + 2 * th);
  half_t* qkv = (half_t*)(w + 3 * th);
  half_t* ctx = (half_t*)((char*)qkv + lm_align(Tw * QKV * 2));
  half_t* gu = (half_t*)((char*)ctx + lm_align(Tw * QD * 2));
  half_t* act = (half_t*)((char*)gu + lm_align(Tw * 2 * I * 2));
  half2_t* rope = (half2_t*)((char*)act + lm_align(Tw * I * 2));
  float* rowscale = (float*)((char*)rope + lm_align((Tw + 4096) * (size_t)(DH / 2) * 4));
  float* ssq = (float*)((char*)rowscale + lm_align(Tw * 4));
  if (Tg > T) {   // the GEMMs' A operands of the padding rows: h / x (q|k|v, gate|up), ctx (output projection); act follows
    RARC_HIP_CHECK(hipMemsetAsync(h + (size_t)T * H, 0, (size_t)(Tg - T) * H * 2, s));
    RARC_HIP_CHECK(hipMemsetAsync(x + (size_t)T * H, 0, (size_t)(Tg - T) * H * 2, s));
    RARC_HIP_CHECK(hipMemsetAsync(ctx + (size_t)T * QD, 0, (size_t)(Tg - T) * QD * 2, s));
  }
  // RMSNorm folded into the projections (RarcLmLayer.qkv_w_folded / gate_up_w_folded): when every projection of a layer runs
  // a tile kernel with the row-scale / residual epilogues, the norm passes disappear — the output and down projections add
  // into x in place and leave per-row partial sums of squares (one per 32 columns), rarc_lm_rowscale_parts_kernel turns those
  // into r = rsqrt(mean(x²) + eps), q|k|v and gate|up multiply x by the folded weights and scale their rows by r.  (It was: read x, read delta, write x, write the normed copy — 0.41 ms of a
  // 5.9 ms layer at 163 840 tokens.)  RARC_LM_FUSE_NORM=0: the separate passes (A/B runs, tests).
  const char* fuse_env = getenv("RARC_LM_FUSE_NORM");
  bool fuse = !(fuse_env && atoi(fuse_env) == 0) && rarc_gemm_norm_fusable(Tg, QKV, H) && rarc_gemm_norm_fusable(Tg, H, QD) &&
              rarc_gemm_norm_fusable(Tg, 2 * I, H) && rarc_gemm_swiglu_fused(Tg, 2 * I, H) && rarc_gemm_norm_fusable(Tg, H, I);
  for (int l = 0; l < m->n_layers && fuse; ++l) fuse = m->layers[l].qkv_w_folded && m->layers[l].gate_up_w_folded;
  auto row_scales = [&]() -> int {   // from the partial sums the residual epilogue (act 96) left: 4 H / 32 bytes per row instead of 2 H
    hipLaunchKernelGGL(rarc_lm_rowscale_parts_kernel, dim3((Tg + 31) / 32), dim3(256), 0, s, (const float*)ssq, H / 32, m->rms_eps, Tg, H,
                       rowscale);
    RARC_HIP_CHECK(hipGetLastError());
    return RARC_OK;
  };
  const int tb = (T + 3) / 4;

  const int P = use ? use->P : 0;
  const int rope_rows = P + seq_len;
  RARC_REQUIRE(rope_rows <= T + 4096, RARC_E_UNSUPPORTED, "rarc_lm_yes_no_logits: prefix of %d rows is too long for this batch", P);
  const size_t kv_cols = (size_t)2 * NKV * DH;
  const size_t cache_layer = use ? lm_align((size_t)use->n_prefix * P * kv_cols * 2) : (fill ? lm_align((size_t)T * kv_cols * 2) : 0);
  hipLaunchKernelGGL(rarc_lm_rope_table_kernel, dim3((rope_rows * (DH / 2) + 255) / 256), dim3(256), 0, s, rope_rows, DH,
                     m->rope_theta, rope);
  RARC_HIP_CHECK(hipGetLastError());

  hipLaunchKernelGGL(rarc_lm_embed_kernel, dim3(tb), dim3(256), 0, s, d_ids, (const half_t*)m->embed, T, H, m->vocab, x);
  RARC_HIP_CHECK(hipGetLastError());
  // last layer on the last positions only, when its compact buffers fit into the (then idle) gate|up buffer
  const int Mp = (n_seq + 127) / 128 * 128;
  const bool last_only = 6 * 256 + (size_t)Mp * ((size_t)QD + 3 * (size_t)H + 3 * (size_t)I) * 2 <= (size_t)T * 2 * I * 2;
  const int q_blocks = (seq_len + 31) / 32, wg_per_kv = ((NQ / NKV) * q_blocks + 3) / 4;  // attention workgroups per (sequence, K/V head)
  // short sequences: all P + seq_len keys of a (sequence, K/V head) resident in LDS (rarc_lm_attention_resident_kernel);
  // RARC_LM_ATTN=stream forces the streaming kernel (A/B runs, tests)
  const int attn_kcap = (P + seq_len + 31) / 32 * 32, attn_vrow = lm_attn_vrow(attn_kcap);
  const size_t attn_lds = (size_t)attn_kcap * (DH + 8) * 2 + (size_t)DH * attn_vrow * 2;
  const char* attn_env = getenv("RARC_LM_ATTN");   // (read per call: the tests switch it inside one process)
  const bool attn_stream_only = attn_env && !strcmp(attn_env, "stream");
  const int attn_rpp = DH == 128 ? 64 : 128;                  // K rows one pass of its staging covers
  const int attn_np = (attn_kcap + attn_rpp - 1) / attn_rpp;   // passes its staging holds in registers (<= 5)
  const bool resident = !attn_stream_only && attn_lds <= 160 * 1024 && attn_np <= 5;
  if (resident) {
    static RarcPerDevice attr_dev;
    size_t& attr = attr_dev.cur();
    if (!attr) {
#define LM_RES_ATTR(DHV, NPV) RARC_HIP_CHECK(hipFuncSetAttribute((const void*)rarc_lm_attention_resident_kernel<DHV, NPV>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024))
      LM_RES_ATTR(128, 2); LM_RES_ATTR(128, 3); LM_RES_ATTR(128, 4); LM_RES_ATTR(128, 5);
      LM_RES_ATTR(64, 2); LM_RES_ATTR(64, 3); LM_RES_ATTR(64, 4); LM_RES_ATTR(64, 5);
#undef LM_RES_ATTR
      attr = 1;
    }
  }
  for (int l = 0; l < m->n_layers; ++l) {
    const RarcLmLayer& Ly = m->layers[l];
    if (fuse && l > 0) {   // x already holds the previous layer's MLP output; its row scales were taken after the down projection
      if (int rc = rarc_gemm_fused_norm((const uint16_t*)x, Ly.qkv_w_folded, rowscale, (uint16_t*)qkv, Tg, QKV, H, 16, stream)) return rc;
    } else {
      // (layer 0: plain norm; later layers: the previous layer's MLP output is added here, then normed)
      hipLaunchKernelGGL(rarc_lm_rmsnorm_kernel, dim3(tb), dim3(256), 0, s, x, (l && !fuse) ? (const half_t*)delta : (const half_t*)nullptr,
                         (const half_t*)Ly.in_norm, m->rms_eps, T, H, h);
      RARC_HIP_CHECK(hipGetLastError());
      if (int rc = rarc_enc_gemm_zero_bias((const uint16_t*)h, Ly.qkv_w, m->zero_bias, (uint16_t*)qkv, Tg, QKV, H, 0, stream)) return rc;
    }
    if (fill) {   // mode 1: this layer's raw k | v rows go to the cache
      hipLaunchKernelGGL(rarc_lm_copy_kv_kernel, dim3(tb), dim3(256), 0, s, (const half_t*)qkv, T, QKV, QD, (int)kv_cols,
                         (half_t*)((char*)fill + (size_t)l * cache_layer));
      RARC_HIP_CHECK(hipGetLastError());
      if (l == m->n_layers - 1) return RARC_OK;
    }
    LmAttnPrefix pf{nullptr, nullptr, nullptr, 0, (int)kv_cols};
    if (use) pf = LmAttnPrefix{(const half_t*)((const char*)use->cache + (size_t)l * cache_layer), use->pidx, use->pstart, P, (int)kv_cols};
    if (resident) {
#define LM_RES_LAUNCH(DHV, NPV)                                                                                              \
      hipLaunchKernelGGL((rarc_lm_attention_resident_kernel<DHV, NPV>), dim3(n_seq * NKV), dim3(512), attn_lds, s, (const half_t*)qkv, \
                         d_start, seq_len, NQ, NKV, q_blocks, (const half_t*)Ly.q_norm, (const half_t*)Ly.k_norm, m->rms_eps,    \
                         (const half2_t*)rope, rope_rows, ctx, pf, attn_kcap, attn_vrow)
      if (DH == 128) {
        if (attn_np <= 2) LM_RES_LAUNCH(128, 2); else if (attn_np == 3) LM_RES_LAUNCH(128, 3);
        else if (attn_np == 4) LM_RES_LAUNCH(128, 4); else LM_RES_LAUNCH(128, 5);
      } else {
        if (attn_np <= 2) LM_RES_LAUNCH(64, 2); else if (attn_np == 3) LM_RES_LAUNCH(64, 3);
        else if (attn_np == 4) LM_RES_LAUNCH(64, 4); else LM_RES_LAUNCH(64, 5);
      }
#undef LM_RES_LAUNCH
    } else if (DH == 128)
      hipLaunchKernelGGL(rarc_lm_attention_kernel<128>, dim3(n_seq * NKV * wg_per_kv), dim3(256), 0, s, (const half_t*)qkv, d_start,
                         seq_len, NQ, NKV, q_blocks, wg_per_kv, (const half_t*)Ly.q_norm, (const half_t*)Ly.k_norm, m->rms_eps,
                         (const half2_t*)rope, rope_rows, ctx, pf);
    else
      hipLaunchKernelGGL(rarc_lm_attention_kernel<64>, dim3(n_seq * NKV * wg_per_kv), dim3(256), 0, s, (const half_t*)qkv, d_start,
                         seq_len, NQ, NKV, q_blocks, wg_per_kv, (const half_t*)Ly.q_norm, (const half_t*)Ly.k_norm, m->rms_eps,
                         (const half2_t*)rope, rope_rows, ctx, pf);
    RARC_HIP_CHECK(hipGetLastError());
#ifdef LM_DUMP_CTX
    if (l == 0 && getenv("RARC_LM_DUMP_CTX")) {
      hipStreamSynchronize(s);
      size_t nb = (size_t)T * QD * 2;
      void* hb = malloc(nb);
      hipMemcpy(hb, ctx, nb, hipMemcpyDeviceToHost);
      FILE* f = fopen(getenv("RARC_LM_DUMP_CTX"), "wb"); fwrite(hb, 1, nb, f); fclose(f); free(hb);
    }
#endif
    if (l == m->n_layers - 1 && last_only && !fill) {
      // Only the last position's logits are wanted, and after the last layer's attention nothing mixes positions any
      // more: its output projection, MLP and residual adds run on the n_seq last rows alone (gathered, padded to a
      // multiple of 128 rows) instead of on all T tokens.
      char* cw = (char*)gu;  // the full-size gate|up buffer is idle in this layer: the compact buffers live there
      half_t* ctx_l = (half_t*)cw;                 cw += lm_align((size_t)Mp * QD * 2);
      half_t* x_l = (half_t*)cw;                   cw += lm_align((size_t)Mp * H * 2);
      half_t* h_l = (half_t*)cw;                   cw += lm_align((size_t)Mp * H * 2);
      half_t* d_l = (half_t*)cw;                   cw += lm_align((size_t)Mp * H * 2);
      half_t* act_l = (half_t*)cw;                 cw += lm_align((size_t)Mp * I * 2);
      half_t* gu_l = (half_t*)cw;
      const int gb = (Mp + 3) / 4;
      hipLaunchKernelGGL(rarc_lm_gather_last_kernel, dim3(gb), dim3(256), 0, s, (const half_t*)ctx, seq_len, QD, n_seq, Mp, ctx_l);
      hipLaunchKernelGGL(rarc_lm_gather_last_kernel, dim3(gb), dim3(256), 0, s, (const half_t*)x, seq_len, H, n_seq, Mp, x_l);
      RARC_HIP_CHECK(hipGetLastError());
      if (int rc = rarc_enc_gemm_zero_bias((const uint16_t*)ctx_l, Ly.o_w, m->zero_bias, (uint16_t*)d_l, Mp, H, QD, 0, stream)) return rc;
      hipLaunchKernelGGL(rarc_lm_rmsnorm_kernel, dim3(gb), dim3(256), 0, s, x_l, (const half_t*)d_l, (const half_t*)Ly.post_norm,
                         m->rms_eps, Mp, H, h_l);
      RARC_HIP_CHECK(hipGetLastError());
      if (rarc_gemm_swiglu_fused(Mp, 2 * I, H)) {
        if (int rc = rarc_enc_gemm_zero_bias((const uint16_t*)h_l, Ly.gate_up_w, m->zero_bias, (uint16_t*)act_l, Mp, 2 * I, H, 3, stream)) return rc;
      } else {
        if (int rc = rarc_enc_gemm_zero_bias((const uint16_t*)h_l, Ly.gate_up_w, m->zero_bias, (uint16_t*)gu_l, Mp, 2 * I, H, 0, stream)) return rc;
        hipLaunchKernelGGL(rarc_lm_swiglu_kernel, dim3(512), dim3(256), 0, s, (const half_t*)gu_l, Mp, I, act_l);
        RARC_HIP_CHECK(hipGetLastError());
      }
      if (int rc = rarc_enc_gemm_zero_bias((const uint16_t*)act_l, Ly.down_w, m->zero_bias, (uint16_t*)d_l, Mp, H, I, 0, stream)) return rc;
      hipLaunchKernelGGL(rarc_lm_rmsnorm_kernel, dim3(gb), dim3(256), 0, s, x_l, (const half_t*)d_l, (const half_t*)nullptr,
                         m->rms_eps, Mp, H, (half_t*)nullptr);
      RARC_HIP_CHECK(hipGetLastError());
      hipLaunchKernelGGL(rarc_lm_last_logits_kernel, dim3(n_seq), dim3(64), 0, s, (const half_t*)x_l, (const half_t*)m->final_norm,
                         (const half_t*)m->lm_head, m->rms_eps, 1, 0, H, no_id, yes_id, (half_t*)d_out_f16);
      RARC_HIP_CHECK(hipGetLastError());
      return RARC_OK;
    }
    if (fuse) {
      if (int rc = rarc_gemm_fused_norm((const uint16_t*)ctx, Ly.o_w, m->zero_bias, (uint16_t*)x, Tg, H, QD, 96, stream, ssq)) return rc;
      if (int rc = row_scales()) return rc;
      if (int rc = rarc_gemm_fused_norm((const uint16_t*)x, Ly.gate_up_w_folded, rowscale, (uint16_t*)act, Tg, 2 * I, H, 19, stream)) return rc;
      if (int rc = rarc_gemm_fused_norm((const uint16_t*)act, Ly.down_w, m->zero_bias, (uint16_t*)x, Tg, H, I, 96, stream, ssq)) return rc;
      if (l + 1 < m->n_layers)
        if (int rc = row_scales()) return rc;
      continue;
    }
    if (int rc = rarc_enc_gemm_zero_bias((const uint16_t*)ctx, Ly.o_w, m->zero_bias, (uint16_t*)delta, Tg, H, QD, 0, stream)) return rc;
    hipLaunchKernelGGL(rarc_lm_rmsnorm_kernel, dim3(tb), dim3(256), 0, s, x, (const half_t*)delta, (const half_t*)Ly.post_norm,
                       m->rms_eps, T, H, h);
    RARC_HIP_CHECK(hipGetLastError());
    if (rarc_gemm_swiglu_fused(Tg, 2 * I, H)) {  // silu(gate)·up in the GEMM's epilogue: the [T][2I] tensor never exists
      if (int rc = rarc_enc_gemm_zero_bias((const uint16_t*)h, Ly.gate_up_w, m->zero_bias, (uint16_t*)act, Tg, 2 * I, H, 3, stream)) return rc;
    } else {
      if (int rc = rarc_enc_gemm_zero_bias((const uint16_t*)h, Ly.gate_up_w, m->zero_bias, (uint16_t*)gu, Tg, 2 * I, H, 0, stream)) return rc;
      hipLaunchKernelGGL(rarc_lm_swiglu_kernel, dim3(2048), dim3(256), 0, s, (const half_t*)gu, Tg, I, act);
      RARC_HIP_CHECK(hipGetLastError());
    }
    if (int rc = rarc_enc_gemm_zero_bias((const uint16_t*)act, Ly.down_w, m->zero_bias, (uint16_t*)delta, Tg, H, I, 0, stream)) return rc;
  }
  if (!fuse) {   // the last layer's MLP output joins the residual stream (no norm output wanted: y = null)
    hipLaunchKernelGGL(rarc_lm_rmsnorm_kernel, dim3(tb), dim3(256), 0, s, x, (const half_t*)delta, (const half_t*)nullptr, m->rms_eps,
                       T, H, (half_t*)nullptr);
    RARC_HIP_CHECK(hipGetLastError());
  }
  hipLaunchKernelGGL(rarc_lm_last_logits_kernel, dim3(n_seq), dim3(64), 0, s, (const half_t*)x, (const half_t*)m->final_norm,
                     (const half_t*)m->lm_head, m->rms_eps, seq_len, seq_len - 1, H, no_id, yes_id, (half_t*)d_out_f16);
  RARC_HIP_CHECK(hipGetLastError());
  return RARC_OK;
}

extern "C" int rarc_lm_yes_no_logits(const RarcLmModel* m, const int32_t* d_ids, const int32_t* d_start, int n_seq,
                                     int seq_len, int no_id, int yes_id, void* d_ws, size_t ws_bytes,
                                     uint16_t* d_out_f16, void* stream) {
  RARC_RANGE();
  RARC_REQUIRE(d_out_f16, RARC_E_INVALID, "rarc_lm_yes_no_logits: null pointer");
  return lm_forward(m, d_ids, d_start, n_seq, seq_len, no_id, yes_id, d_ws, ws_bytes, d_out_f16, stream, nullptr, nullptr);
}

extern "C" int rarc_lm_prefix_kv(const RarcLmModel* m, const int32_t* d_ids, const int32_t* d_start, int n_prefix, int prefix_len,
                                 void* d_ws, size_t ws_bytes, void* d_cache, size_t cache_bytes, void* stream) {
  RARC_RANGE();
  RARC_REQUIRE(m && d_cache, RARC_E_INVALID, "rarc_lm_prefix_kv: null pointer");
  RARC_REQUIRE(n_prefix > 0 && prefix_len > 0 && cache_bytes >= rarc_lm_prefix_cache_bytes(m, n_prefix * prefix_len), RARC_E_INVALID,
               "rarc_lm_prefix_kv: cache too small");
  return lm_forward(m, d_ids, d_start, n_prefix, prefix_len, 0, 0, d_ws, ws_bytes, nullptr, stream, nullptr, (half_t*)d_cache);
}

extern "C" int rarc_lm_yes_no_logits_prefixed(const RarcLmModel* m, const int32_t* d_ids, const int32_t* d_start, int n_seq,
                                              int seq_len, const int32_t* d_prefix_of, const void* d_cache, int n_prefix,
                                              int prefix_len, const int32_t* d_prefix_start, int no_id, int yes_id, void* d_ws,
                                              size_t ws_bytes, uint16_t* d_out_f16, void* stream) {
  RARC_RANGE();
  RARC_REQUIRE(m && d_prefix_of && d_cache && d_prefix_start && d_out_f16, RARC_E_INVALID, "rarc_lm_yes_no_logits_prefixed: null pointer");
  RARC_REQUIRE(n_prefix > 0 && prefix_len > 0 && prefix_len <= 4096, RARC_E_INVALID, "rarc_lm_yes_no_logits_prefixed: bad prefix shape");
  const LmPrefixUse use{(const half_t*)d_cache, d_prefix_of, d_prefix_start, n_prefix, prefix_len};
  return lm_forward(m, d_ids, d_start, n_seq, seq_len, no_id, yes_id, d_ws, ws_bytes, d_out_f16, stream, &use, nullptr);
}
