// encoder_f32.hip — the encoder forward at the REFERENCE's precision (fp32-class), precision = "fp32" of the provider.
//
// HuggingFaceEmbeddings builds `SentenceTransformer(model_name, **model_kwargs)` with no dtype
// (core/file_management/embeddings/huggingface.py:96-98) and calls `.encode` (:122-126): an fp32 forward.  The fp16
// forward of encoder.hip is a 1e-3-class approximation of it; this file is the mode whose embeddings agree with an
// fp32 forward to fp32 rounding noise, so that ids and scores downstream are the reference's.
//
// The dense contractions still run on the fp16 MFMA (v_mfma_f32_32x32x16_f16, fp32 accumulate) — as SPLIT operands:
//   x·s = hi + lo,  hi = fp16(x·s),  lo = fp16(x·s − hi)        (s a power of two per row: both roundings lose < 2^-22 |x|)
//   C = A·Wᵀ = [A_lo | A_hi | A_hi] · [W_hi | W_lo | W_hi]ᵀ / (s_a[m]·s_w[n])          (the lo·lo term, 2^-22 relative, is dropped)
// i.e. ONE fp16 GEMM over K' = 3K with the small cross terms accumulated first; fp16 products are exact in fp32, the
// accumulator is fp32, the row scales are powers of two (exact).  Per-row scales put the row maximum in [2^13, 2^14):
// lo is a normal fp16 number for every element within 2^-17 of the row maximum, and an element below that is
// represented to 2^-39 of the maximum even if subnormal halves were flushed.
// Everything between the GEMMs is fp32: residual stream, LayerNorm (two-pass statistics), exact erf GELU (libm erff),
// softmax (libm expf), pooling and the canonical L2 normalisation.  Attention (QKᵀ, PV): head_dim 64 on the fp16 MFMA over split
// operands like the GEMMs (rarc_e32_attention_split_kernel, round 4), head_dim 32 on the fp32 MFMA (exact fp32 products).
#include "rarc_common.h"
#include <cstring>
#include <cstdlib>

int rarc_gemm_f16_f32out(const uint16_t* a, const uint16_t* w, float* c, int m, int n, int k, hipStream_t s);  // encoder.hip
int rarc_gemm_f16_f32out_parts(const uint16_t* a, const uint16_t* w, float* c, int m, int n, int k, int max_parts, int* parts,
                               hipStream_t s);   // encoder.hip: up to max_parts fp32 partial slabs c[part][m][n] (small batches)
bool rarc_gemm_f16_gelu_split_takes(int m, int n, int k3);   // encoder.hip: would the fused FFN1 (below) take this shape?
int rarc_gemm_f16_gelu_split(const uint16_t* a3, const uint16_t* w3, const float* ra, const float* rw, const float* bias,
                             const float* sg, uint16_t* out3, int m, int n, int k3, hipStream_t s);   // 1 = not taken

typedef _Float16 half4_t __attribute__((ext_vector_type(4)));

// power-of-two scale that puts `mx` into [2^13, 2^14) (1 for a zero / non-finite maximum); inv = 1/s, both exact
__device__ __forceinline__ void e32_scale_of(float mx, float& s, float& inv) {
  s = 1.f; inv = 1.f;
  if (mx > 0.f && mx < __builtin_inff()) {
    int e;
    (void)frexpf(mx, &e);            // mx = m·2^e, m in [0.5, 1)
    int ex = 14 - e;
    ex = ex > 100 ? 100 : (ex < -100 ? -100 : ex);
    s = ldexpf(1.f, ex);
    inv = ldexpf(1.f, -ex);
  }
}

__device__ __forceinline__ void e32_split4(const float4 v, float s, half4_t& hi, half4_t& lo) {
  const float x[4] = {v.x * s, v.y * s, v.z * s, v.w * s};
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    hi[e] = (half_t)x[e];
    lo[e] = (half_t)(x[e] - (float)hi[e]);   // the difference is exact in fp32
  }
}

__device__ __forceinline__ float e32_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float e32_wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
// 256-thread block reductions (fixed order: wave tree, then waves 0..3); `slot` = 4 floats of LDS per call site
__device__ __forceinline__ float e32_block_sum(float v, float* slot) {
  v = e32_wave_sum(v);
  if ((threadIdx.x & 63) == 0) slot[threadIdx.x >> 6] = v;
  __syncthreads();
  return ((slot[0] + slot[1]) + (slot[2] + slot[3]));
}
__device__ __forceinline__ float e32_block_max(float v, float* slot) {
  v = e32_wave_max(v);
  if ((threadIdx.x & 63) == 0) slot[threadIdx.x >> 6] = v;
  __syncthreads();
  return fmaxf(fmaxf(slot[0], slot[1]), fmaxf(slot[2], slot[3]));
}

// One workgroup (256 threads) per row of n <= 1024·NV elements; thread t owns the float4 chunks t + 256·i.
template <int NV>
struct E32Row {
  float4 v[NV];
};

// Fragment-major split image of the query path (round 6; rarc_e32_skinny_gemm_kernel reads it): element (m, c) of a
// [M][n] matrix lives where the lane of a v_mfma_f32_32x32x16_f16 B operand wants it — per (32-token block, 16-wide k step):
// the hi fragment then the lo fragment, each 64 lanes x 8 halves = 1 KiB contiguous (lane = token + 32·(k half), natural k
// order inside a fragment).  Offset in halves of the hi value of (m, c); the lo value sits 512 halves further.
__device__ __forceinline__ size_t e32_frag_offset(size_t m, int c, int n) {
  return ((((m >> 5) * (size_t)(n >> 4) + (size_t)(c >> 4)) * 2) * 64 + (m & 31) + 32 * ((c >> 3) & 1)) * 8 + (c & 7);
}

// row -> split image [lo | hi | hi] (3n halves) + the row's inverse scale; frag: the fragment-major image instead (out3 is
// then the base of the whole image and `row` the row's index in it)
template <int NV>
__device__ __forceinline__ void e32_store_split(const E32Row<NV>& r, int n, half_t* out3, float* ra_out, float* slot,
                                                bool frag = false, size_t row = 0) {
  float mx = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i)
    if ((threadIdx.x + 256 * i) * 4 < n)
      mx = fmaxf(fmaxf(mx, fmaxf(fabsf(r.v[i].x), fabsf(r.v[i].y))), fmaxf(fabsf(r.v[i].z), fabsf(r.v[i].w)));
  mx = e32_block_max(mx, slot);
  float s, inv;
  e32_scale_of(mx, s, inv);
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (threadIdx.x + 256 * i) * 4;
    if (c < n) {
      half4_t hi, lo;
      e32_split4(r.v[i], s, hi, lo);
      if (frag) {
        half_t* o = out3 + e32_frag_offset(row, c, n);
        *(half4_t*)o = hi;
        *(half4_t*)(o + 512) = lo;
      } else {
        *(half4_t*)(out3 + c) = lo;
        *(half4_t*)(out3 + n + c) = hi;
        *(half4_t*)(out3 + 2 * n + c) = hi;
      }
    }
  }
  if (threadIdx.x == 0) *ra_out = inv;
}

// LayerNorm of the row in place (two-pass fp32 statistics, population variance, 1/sqrt exact-rounded)
// gamma / beta (and whatever else a row pass multiplies the normalised row with) are fetched by the CALLER in front of its
// first dependent load: behind the two block reductions' barriers the loads would be one more round trip to L2 on the critical
// path of a pass that is nothing but latency at 32 rows (the query path)
template <int NV>
__device__ __forceinline__ void e32_load_row_params(const float* p, int n, float4 (&out)[NV]) {
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (threadIdx.x + 256 * i) * 4;
    out[i] = c < n ? *(const float4*)(p + c) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
}
template <int NV>
__device__ __forceinline__ void e32_layernorm(E32Row<NV>& r, int n, const float4 (&gam)[NV], const float4 (&bet)[NV], float eps,
                                              float* slot) {
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i)
    if ((threadIdx.x + 256 * i) * 4 < n) s += (r.v[i].x + r.v[i].y) + (r.v[i].z + r.v[i].w);
  const float mean = e32_block_sum(s, slot) / (float)n;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i)
    if ((threadIdx.x + 256 * i) * 4 < n) {
      const float a = r.v[i].x - mean, b = r.v[i].y - mean, c = r.v[i].z - mean, d = r.v[i].w - mean;
      q += (a * a + b * b) + (c * c + d * d);
    }
  const float var = e32_block_sum(q, slot + 4) / (float)n;
  const float rstd = 1.0f / sqrtf(var + eps);
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (threadIdx.x + 256 * i) * 4;
    if (c < n) {
      const float4 g = gam[i], b = bet[i];
      r.v[i].x = (r.v[i].x - mean) * rstd * g.x + b.x;
      r.v[i].y = (r.v[i].y - mean) * rstd * g.y + b.y;
      r.v[i].z = (r.v[i].z - mean) * rstd * g.z + b.z;
      r.v[i].w = (r.v[i].w - mean) * rstd * g.w + b.w;
    }
  }
}

__device__ __forceinline__ float e32_gelu(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f)); }

// Row epilogue of a split GEMM.  MODE:
//   0  v = P·ra·rw + bias                      -> out32
//   1  v = gelu(P·ra·rw + bias)                -> split image
//   2  v = LayerNorm(P·ra·rw + bias + resid)   -> out32 (may alias resid) and split image
//   3  v = P (plain fp32 rows, no scales)      -> split image
// Round 4 — FFN1 with its GELU fused into the GEMM's epilogue (rarc_gemm256_f16_kernel<5>, encoder.hip).  The epilogue of a
// 256 x 256 tile cannot know the maximum of a 4096-wide row, so the power-of-two scale of the GELU output's split image is
// fixed BEFORE the GEMM from a bound on the row:
//     |gelu(v_j)| <= |v_j| = |sum_k x_k W1[j][k] + b1[j]| <= sum_k |x_k| c[k] + c[K],   c[k] = max_j |W1[j][k]|, c[K] = max_j |b1[j]|
// (RarcEnc32Layer.f1_colmax, computed from the fp32 weights when the model is loaded).  The bound only has to be SAFE (it puts
// the row maximum below 2^14, fp16 ends at 2^16) and not absurdly loose: an element below 2^-17 of the bound keeps an absolute
// error of 2^-39 of the bound (header of this file).  This one follows a hot input channel and a heavy weight row exactly and
// overshoots by the random-sign slack of the sum, ~2^5..2^7 (measured: the fused forward is as close to float64 as the
// unfused one, also on rows over eight decades with a hot channel — tests/test_gpu_encoder_f32.py); the first form tried,
// ||x||_2 · max_j ||W1_j||_2, overshot 2^11 on such rows and cost a factor 2.4 in the embedding's error there.
// The row pass in front of FFN1 (MODE 2, the LayerNorm) computes it: sg_out[m] = scale, sg_out[rows + m] = 1/scale.
struct E32Fuse {
  const float* colmax = nullptr; int k = 0; float* sg_out = nullptr; size_t sg_rows = 0;
  bool frag = false;   // the query path: the split image is written fragment-major (e32_frag_offset)
};

// slab 0 + slab 1 + ... of a split-K product, IN ORDER (the sum is what an unsplit k loop would have rounded differently, but it
// is the same for every consumer); the loads of three slabs go out together — a runtime loop of load-then-add is a chain of
// round trips to L2, and a single query's FFN2 product comes in sixteen slabs.
__device__ __forceinline__ float4 e32_sum_parts(const float* __restrict__ src, int n_parts, size_t part_stride) {
  float4 p = *(const float4*)src;
  int sp = 1;
  for (; sp + 2 < n_parts; sp += 3) {
    const float4 a = *(const float4*)(src + (size_t)sp * part_stride);
    const float4 b = *(const float4*)(src + (size_t)(sp + 1) * part_stride);
    const float4 c = *(const float4*)(src + (size_t)(sp + 2) * part_stride);
    p.x += a.x; p.y += a.y; p.z += a.z; p.w += a.w;
    p.x += b.x; p.y += b.y; p.z += b.z; p.w += b.w;
    p.x += c.x; p.y += c.y; p.z += c.z; p.w += c.w;
  }
  for (; sp < n_parts; ++sp) {
    const float4 a = *(const float4*)(src + (size_t)sp * part_stride);
    p.x += a.x; p.y += a.y; p.z += a.z; p.w += a.w;
  }
  return p;
}

template <int NV, int MODE>
__global__ __launch_bounds__(256) void rarc_e32_epi_kernel(const float* P, const float* __restrict__ ra,
                                                           const float* __restrict__ rw, const float* __restrict__ bias,
                                                           const float* resid, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float eps, int n,
                                                           float* out32, half_t* __restrict__ out3,
                                                           float* __restrict__ ra_out, int n_parts, size_t part_stride,
                                                           const E32Fuse fz = E32Fuse()) {
  __shared__ float slot[12];
  const size_t m = blockIdx.x;
  E32Row<NV> r;
  const float ram = MODE == 3 ? 1.f : ra[m];
  float4 gam[NV], bet[NV], cmx[NV];
  float cm_last = 0.f;
  if (MODE == 2) {
    e32_load_row_params<NV>(gamma, n, gam);
    e32_load_row_params<NV>(beta, n, bet);
    if (fz.colmax) { e32_load_row_params<NV>(fz.colmax, n, cmx); cm_last = fz.colmax[fz.k]; }
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (threadIdx.x + 256 * i) * 4;
    if (c < n) {
      float4 p = e32_sum_parts(P + m * n + c, n_parts, part_stride);   // split-K partial slabs of a small batch's GEMM, summed in order
      if (MODE != 3) {
        const float4 w = *(const float4*)(rw + c), b = *(const float4*)(bias + c);
        // ra, rw are powers of two: the two scalings are exact, the bias add rounds once (as in x·Wᵀ + b)
        p.x = p.x * ram * w.x + b.x; p.y = p.y * ram * w.y + b.y; p.z = p.z * ram * w.z + b.z; p.w = p.w * ram * w.w + b.w;
      }
      if (MODE == 1) { p.x = e32_gelu(p.x); p.y = e32_gelu(p.y); p.z = e32_gelu(p.z); p.w = e32_gelu(p.w); }
      if (MODE == 2) {
        const float4 x = *(const float4*)(resid + m * n + c);
        p.x += x.x; p.y += x.y; p.z += x.z; p.w += x.w;
      }
      r.v[i] = p;
    } else {
      r.v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  if (MODE == 2) e32_layernorm<NV>(r, n, gam, bet, eps, slot);
  if (MODE == 2 && fz.colmax) {   // the scale FFN1's fused epilogue will split gelu(.) of this row with
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = (threadIdx.x + 256 * i) * 4;
      if (c < n) {
        const float4 cm = cmx[i];
        q += (fabsf(r.v[i].x) * cm.x + fabsf(r.v[i].y) * cm.y) + (fabsf(r.v[i].z) * cm.z + fabsf(r.v[i].w) * cm.w);
      }
    }
    const float l1 = e32_block_sum(q, slot);   // (slot[0..3]: every thread is past the LayerNorm's second barrier)
    const float bound = l1 * 1.0001f + cm_last;   // (1e-4: the rounding of the 1024-term sum, with room)
    float sgs, sgi;
    e32_scale_of(bound, sgs, sgi);
    if (threadIdx.x == 0) {
      fz.sg_out[m] = sgs;
      fz.sg_out[fz.sg_rows + m] = sgi;
    }
  }
  if (MODE == 0 || MODE == 2) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = (threadIdx.x + 256 * i) * 4;
      if (c < n) *(float4*)(out32 + m * n + c) = r.v[i];
    }
  }
  if (MODE != 0) e32_store_split<NV>(r, n, fz.frag ? out3 : out3 + m * 3 * n, ra_out + m, slot + 8, fz.frag, m);
}

// x = LayerNorm(word[id] + pos[t % L] + type0) -> fp32 rows + split image
__global__ __launch_bounds__(256) void rarc_e32_embed_kernel(const int32_t* __restrict__ ids, const float* __restrict__ word,
                                                             const float* __restrict__ pos, const float* __restrict__ type0,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             float eps, int L, int H, int vocab, float* __restrict__ out32,
                                                             half_t* __restrict__ out3, float* __restrict__ ra_out,
                                                             bool frag = false) {
  __shared__ float slot[12];
  const size_t t = blockIdx.x;
  int id = ids[t];
  id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
  E32Row<1> r;
  const int c = threadIdx.x * 4;
  if (c < H) {
    const float4 a = *(const float4*)(word + (size_t)id * H + c), b = *(const float4*)(pos + (size_t)(t % L) * H + c),
                 ty = *(const float4*)(type0 + c);
    // (word + type) + pos: the order BertEmbeddings adds them in (inputs_embeds + token_type_embeddings, then + position)
    r.v[0] = make_float4((a.x + ty.x) + b.x, (a.y + ty.y) + b.y, (a.z + ty.z) + b.z, (a.w + ty.w) + b.w);
  } else {
    r.v[0] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  float4 gam[1], bet[1];
  e32_load_row_params<1>(gamma, H, gam);
  e32_load_row_params<1>(beta, H, bet);
  e32_layernorm<1>(r, H, gam, bet, eps, slot);
  if (c < H) *(float4*)(out32 + t * H + c) = r.v[0];
  e32_store_split<1>(r, H, frag ? out3 : out3 + t * 3 * H, ra_out + t, slot + 8, frag, t);
}

// weight rows at load time: W fp32 [N][K] -> [W_hi | W_lo | W_hi] (3K halves per row) + inverse row scale
__global__ __launch_bounds__(256) void rarc_e32_split_weight_kernel(const float* __restrict__ W, int K, half_t* __restrict__ w3,
                                                                    float* __restrict__ rw) {
  __shared__ float slot[4];
  const size_t nrow = blockIdx.x;
  const float* w = W + nrow * K;
  float mx = 0.f;
  for (int c = threadIdx.x; c < K; c += 256) mx = fmaxf(mx, fabsf(w[c]));
  mx = e32_block_max(mx, slot);
  float s, inv;
  e32_scale_of(mx, s, inv);
  half_t* o = w3 + nrow * 3 * K;
  for (int c = threadIdx.x; c < K; c += 256) {
    const float x = w[c] * s;
    const half_t hi = (half_t)x, lo = (half_t)(x - (float)hi);
    o[c] = hi;
    o[K + c] = lo;
    o[2 * K + c] = hi;
  }
  if (threadIdx.x == 0) rw[nrow] = inv;
}

// ------------------------------------------------------------------------------------------
// Attention in fp32 on the fp32 MFMA (v_mfma_f32_32x32x2_f32: exact fp32 products, fp32 accumulate — an fmaf chain).
// P [M][3H] is the raw split product of the fused q|k|v projection; its scales and bias are applied as the values are
// loaded (no separate pass): value = P·ra[token]·rw[column] + bias[column].
// One wave per (sequence, head, block of 32 query rows); keys / values in tiles of 32; four waves per workgroup.
//   S^T = K · Q^T    A = K rows, B = Q rows.  The instruction contracts two k indices per step, one per lane half:
//                    half hh takes d = hh·DH/2 + s at step s, so a lane's operand stream is a CONTIGUOUS half row
//                    (DH/2 floats: float4 loads).  Lane (l&31 = query, hh) ends with 16 of the tile's 32 keys:
//                    key(r, hh) = 8(r>>2) + 4hh + (r&3); a query's softmax statistics live in the lane pair (l, l^32).
//   O^T += V^T · P^T step s contracts keys key(s, 0) and key(s, 1): the B operand is the lane's own probability
//                    register s (no cross-lane traffic), the A operand V[key(s, hh)][32mb + (l&31)] comes from the
//                    wave's LDS copy of the V tile (row stride DH + 8 floats: the two halves hit disjoint banks).
// Online softmax over key tiles with libm expf; keys >= lens[seq] masked; rows >= seq_len neither loaded past the
// end nor stored.
// ------------------------------------------------------------------------------------------
template <int DH, bool REL>   // REL: scores get the relative-position bias rel[head][key - query + rel_span - 1] (MPNet)
__global__ __launch_bounds__(256) void rarc_e32_attention_kernel(const float* __restrict__ P, const float* __restrict__ ra,
                                                                 const float* __restrict__ rw, const float* __restrict__ bias,
                                                                 const int32_t* __restrict__ lens, int L, int H, int n_heads,
                                                                 int q_blocks, int n_units, float* __restrict__ ctx,
                                                                 const float* __restrict__ rel, int rel_span, int n_parts,
                                                                 size_t part_stride) {
  constexpr int HD = DH / 2;     // floats of a q / k row one lane half contracts
  constexpr int MB = DH / 32;    // 32-row blocks of O^T
  constexpr int VS = DH + 8;     // row stride of the V tile in LDS (floats)
  __shared__ __attribute__((aligned(16))) float vs_all[4][32 * VS];
  __shared__ __attribute__((aligned(16))) float sb_all[4][6 * DH];   // rw | bias of this head's q, k, v columns
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int unit = blockIdx.x * 4 + wave;
  if (unit >= n_units) return;  // (no block-level barrier below: waves are independent)
  float* vs = vs_all[wave];
  float* sb = sb_all[wave];
  const int qb = unit % q_blocks, bh = unit / q_blocks;
  const int b = bh / n_heads, hd = bh % n_heads;
  const int len = lens[b] < 1 ? 1 : (lens[b] > L ? L : lens[b]);
  const int col = lane & 31, hh = lane >> 5;
  const size_t rs = (size_t)3 * H;
  const size_t tok0 = (size_t)b * L;
  const float scale = DH == 64 ? 0.125f : 0.17677669529663687f;  // 1/sqrt(DH)
  for (int i = lane; i < 3 * DH; i += 64) {
    const int part = i / DH, c = i % DH;   // 0 q, 1 k, 2 v
    sb[part * 2 * DH + c] = rw[part * H + hd * DH + c];
    sb[part * 2 * DH + DH + c] = bias[part * H + hd * DH + c];
  }
  __builtin_amdgcn_wave_barrier();
  const int q0 = qb * 32;
  const int qrow = (q0 + col < L) ? q0 + col : L - 1;
  // a lane's half row of q or k: P·ra·rw + bias over d = hh*HD .. hh*HD + HD - 1
  auto load_half = [&](size_t tok, int part, float (&dst)[HD]) {
    const float r = ra[tok];
    const float* src = P + tok * rs + part * H + hd * DH + hh * HD;
    const float* w = sb + part * 2 * DH + hh * HD;
#pragma unroll
    for (int c = 0; c < HD; c += 4) {
      const float4 p = e32_sum_parts(src + c, n_parts, part_stride);   // split-K partial slabs of a small batch's q|k|v product
      const float4 s4 = *(const float4*)(w + c), b4 = *(const float4*)(w + DH + c);
      dst[c] = p.x * r * s4.x + b4.x; dst[c + 1] = p.y * r * s4.y + b4.y;
      dst[c + 2] = p.z * r * s4.z + b4.z; dst[c + 3] = p.w * r * s4.w + b4.w;
    }
  };
  float qf[HD];
  load_half(tok0 + qrow, 0, qf);

  f32x16 o[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) o[mb] = (f32x16){0};
  float m_run = -INFINITY, l_run = 0.f;

  for (int k0 = 0; k0 < len; k0 += 32) {
    // ---- S^T tile: keys k0..k0+31 (rows) x this wave's 32 queries (columns) ----
    const int krow = (k0 + col < L) ? k0 + col : L - 1;
    float kf[HD];
    load_half(tok0 + krow, 1, kf);
    f32x16 st = {0};
#pragma unroll
    for (int s = 0; s < HD; ++s) st = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[s], qf[s], st, 0, 0, 0);
    // ---- V tile -> LDS [key][d] (scaled + biased), 16-byte pieces, coalesced rows ----
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = lane; i < 32 * (DH / 4); i += 64) {
      const int kr = i / (DH / 4), c4 = i % (DH / 4);
      const int vrow = (k0 + kr < L) ? k0 + kr : L - 1;
      const float r = ra[tok0 + vrow];
      const float4 p = e32_sum_parts(P + (tok0 + vrow) * rs + 2 * H + hd * DH + 4 * c4, n_parts, part_stride);
      const float4 s4 = *(const float4*)(sb + 4 * DH + 4 * c4), b4 = *(const float4*)(sb + 5 * DH + 4 * c4);
      *(float4*)(vs + kr * VS + 4 * c4) =
          make_float4(p.x * r * s4.x + b4.x, p.y * r * s4.y + b4.y, p.z * r * s4.z + b4.z, p.w * r * s4.w + b4.w);
    }
    // ---- softmax statistics of this lane's query over its 16 keys, then with the partner lane ----
    float pr[16];
    float tmax = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = k0 + 8 * (r >> 2) + 4 * hh + (r & 3);
      float v = st[r] * scale;
      if (REL) {   // (index clamped into the table: keys past the sequence are masked below anyway)
        int ri = key - (q0 + col) + rel_span - 1;
        ri = ri < 0 ? 0 : (ri > 2 * rel_span - 2 ? 2 * rel_span - 2 : ri);
        v += rel[(size_t)hd * (2 * rel_span - 1) + ri];
      }
      pr[r] = key < len ? v : -INFINITY;
      tmax = fmaxf(tmax, pr[r]);
    }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    const float m_new = fmaxf(m_run, tmax);
    const float corr = (m_run == -INFINITY) ? 0.f : expf(m_run - m_new);
    float psum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      pr[r] = (pr[r] == -INFINITY) ? 0.f : expf(pr[r] - m_new);
      psum += pr[r];
    }
    psum += __shfl_xor(psum, 32, 64);
    l_run = l_run * corr + psum;
    m_run = m_new;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[mb][r] *= corr;
    __builtin_amdgcn_wave_barrier();
    // ---- O^T += V^T · P^T: step s contracts keys key(s, 0), key(s, 1) ----
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const int key = 8 * (s >> 2) + 4 * hh + (s & 3);
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
        o[mb] = __builtin_amdgcn_mfma_f32_32x32x2f32(vs[key * VS + 32 * mb + col], pr[s], o[mb], 0, 0, 0);
    }
    // (the last O^T MFMA of the key loop is read — accumulators copied out of their AGPRs — right behind the loop's exit
    //  branch: a window that holds free instructions; the pad sits INSIDE the loop because the copies are placed after
    //  scheduling and would slip in front of a pad behind it.  tests/codeobj.py follows branches since round 5)
    RARC_MFMA_SETTLE(o);
  }
  // ---- store: lane (query, hh) holds d = 32mb + 8(r>>2) + 4hh + (r&3) ----
  if (q0 + col < L) {
    const float inv = l_run > 0.f ? 1.f / l_run : 0.f;
    float* out = ctx + (tok0 + q0 + col) * H + hd * DH;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *(float4*)(out + 32 * mb + 8 * g + 4 * hh) =
            make_float4(o[mb][4 * g] * inv, o[mb][4 * g + 1] * inv, o[mb][4 * g + 2] * inv, o[mb][4 * g + 3] * inv);
  }
}

// ------------------------------------------------------------------------------------------
// Round 4 — the same attention on the fp16 MFMA over SPLIT operands (head_dim 64: bge-base / bge-large / MPNet).
// The fp32 MFMA runs at 1/16 of the fp16 rate; at 512 tokens the kernel above cost a third of an ingest forward.  Here every
// operand is a (hi, lo) pair of fp16 images of the fp32 value times a power of two — exactly the GEMMs' form:
//   S^T = [K_lo | K_hi | K_hi] · [Q_hi | Q_lo | Q_hi]^T / (s_k[key] · s_q[query])          per-row scales
//   O^T += ( [V_lo | V_hi | V_hi]^T · [P_hi | P_lo | P_hi]^T ) / (s_v[block] · 2^11)       per (16 keys x 32 d) block scale
// fp16 products are exact in the fp32 accumulator; dropped: the lo·lo terms (2^-22 relative) and the tail of lo (< 2^-22 of
// the value, or 2^-38 of the row / block maximum where lo is subnormal).  p in [0, 1] is split as p·2^11.  Softmax itself
// (maximum, libm expf, sum) is fp32 as before.
// One WORKGROUP (4 waves) per (sequence, head, four consecutive 32-query blocks), one wave per block.  Every 32-key tile is
// fetched, scaled and split ONCE per workgroup: all four waves share the K image (row-major, 16-byte row pad), wave w builds
// the V^T image of block (keys 16(w&1).., d 32(w>>1)..) — its block maximum is a wave reduction, no exchange — with key pairs
// packed into 4-byte writes.  Raw rows of tile t+1 travel in registers while tile t is multiplied; the images are double
// buffered: one barrier per tile.  P^T fragments are assembled with v_permlane32_swap as in decoder.hip.
// Keys >= lens[seq] are masked and never read (clamped to the last real token): a result does not depend on the padding.
// Two waves per SIMD (launch bounds): the kernel is bound by VALU issue (softmax, splitting), not by the MFMAs.
// (Tried and dropped: skipping the accumulator rescale when no query's maximum rose + a mask-free instantiation for
//  interior tiles — together 2 % at best, and that build returned wrong rows at random on the GPU while either one alone
//  was exact; the cause was not found in the ISA, so neither is kept.)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void e32a_scale_of(float mx, float& s, float& inv) {   // like e32_scale_of, exponent within +-60:
  s = 1.f; inv = 1.f;                                                              // products of two inverses stay finite
  if (mx > 0.f && mx < __builtin_inff()) {
    int e;
    (void)frexpf(mx, &e);
    int ex = 14 - e;
    ex = ex > 60 ? 60 : (ex < -60 ? -60 : ex);
    s = ldexpf(1.f, ex);
    inv = ldexpf(1.f, -ex);
  }
}
__device__ __forceinline__ void e32a_split(float x, half_t& hi, half_t& lo) {
  hi = (half_t)x;
  lo = (half_t)(x - (float)hi);
}

template <bool REL>
__global__ __launch_bounds__(256, 2) void rarc_e32_attention_split_kernel(const float* __restrict__ P, const float* __restrict__ ra,
                                                                       const float* __restrict__ rw, const float* __restrict__ bias,
                                                                       const int32_t* __restrict__ lens, int L, int H, int n_heads,
                                                                       int q_blocks, int q_groups, float* __restrict__ ctx,
                                                                       const float* __restrict__ rel, int rel_span, int n_parts,
                                                                       size_t part_stride) {
  constexpr int DH = 64, MB = 2, KS = 4;
  constexpr int KROW = DH + 8;   // halves per K row (+16 bytes: conflict-free ds_read_b128 fragments)
  constexpr int VROW = 40;       // halves per V^T row: 32 keys + pad
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  __shared__ __attribute__((aligned(16))) half_t kh[2][32 * KROW], kl[2][32 * KROW];
  __shared__ __attribute__((aligned(16))) half_t vh[2][DH * VROW], vl[2][DH * VROW];
  __shared__ __attribute__((aligned(16))) float skinv[2][32];
  __shared__ float svinv[2][4];
  __shared__ __attribute__((aligned(16))) float sb[6 * DH];   // rw | bias of this head's q, k, v columns
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int qg = blockIdx.x % q_groups, bh = blockIdx.x / q_groups;
  const int b = bh / n_heads, hd = bh % n_heads;
  const int len = lens[b] < 1 ? 1 : (lens[b] > L ? L : lens[b]);
  const int col = lane & 31, hh = lane >> 5;
  const size_t rs = (size_t)3 * H;
  const size_t tok0 = (size_t)b * L;
  const int qb = 4 * qg + wave;
  const bool live = qb < q_blocks;
  for (int i = tid; i < 3 * DH; i += 256) {
    const int part = i / DH, c = i % DH;   // 0 q, 1 k, 2 v
    sb[part * 2 * DH + c] = rw[part * H + hd * DH + c];
    sb[part * 2 * DH + DH + c] = bias[part * H + hd * DH + c];
  }
  __syncthreads();

  auto load4 = [&](const float* src) {   // one 16-byte piece, split-K partial slabs summed in order
    return e32_sum_parts(src, n_parts, part_stride);
  };
  auto affine4 = [&](const float4 p, float r, const float* w) {   // value = P·ra·rw + bias
    const float4 s4 = *(const float4*)w, b4 = *(const float4*)(w + DH);
    return make_float4(p.x * r * s4.x + b4.x, p.y * r * s4.y + b4.y, p.z * r * s4.z + b4.z, p.w * r * s4.w + b4.w);
  };

  // ---- staging roles: K row r = tid >> 3, floats 8c..8c+7 (c = tid & 7); V block (t, mb) = (wave & 1, wave >> 1):
  //      key pair pp = lane & 7 (keys 16t + 2pp, + 1), floats 32mb + 4cc .. + 3 (cc = lane >> 3) ----
  const int kr = tid >> 3, kc = tid & 7;
  const int vt_ = wave & 1, vmb = wave >> 1, vpp = lane & 7, vcc = lane >> 3;
  float4 kraw0, kraw1, vraw0, vraw1;
  float kra, vra0, vra1;
  auto prefetch = [&](int k0) {
    const int krow = (k0 + kr < len) ? k0 + kr : len - 1;
    const float* ksrc = P + (tok0 + krow) * rs + H + hd * DH + 8 * kc;
    kraw0 = load4(ksrc);
    kraw1 = load4(ksrc + 4);
    kra = ra[tok0 + krow];
    const int key0 = k0 + 16 * vt_ + 2 * vpp;
    const int v0 = key0 < len ? key0 : len - 1, v1 = key0 + 1 < len ? key0 + 1 : len - 1;
    vraw0 = load4(P + (tok0 + v0) * rs + 2 * H + hd * DH + 32 * vmb + 4 * vcc);
    vraw1 = load4(P + (tok0 + v1) * rs + 2 * H + hd * DH + 32 * vmb + 4 * vcc);
    vra0 = ra[tok0 + v0];
    vra1 = ra[tok0 + v1];
  };
  auto store_tile = [&](int buf) {
    {   // K row piece: scale of the row = max over its 8 lanes
      const float4 a = affine4(kraw0, kra, sb + 2 * DH + 8 * kc), c4 = affine4(kraw1, kra, sb + 2 * DH + 8 * kc + 4);
      const float x[8] = {a.x, a.y, a.z, a.w, c4.x, c4.y, c4.z, c4.w};
      float mx = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) mx = fmaxf(mx, fabsf(x[e]));
      mx = fmaxf(mx, __shfl_xor(mx, 1, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 2, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 4, 64));
      float s, inv;
      e32a_scale_of(mx, s, inv);
      half8 hi, lo;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        half_t h, l;
        e32a_split(x[e] * s, h, l);
        hi[e] = h; lo[e] = l;
      }
      *(half8*)(kh[buf] + kr * KROW + 8 * kc) = hi;
      *(half8*)(kl[buf] + kr * KROW + 8 * kc) = lo;
      if (kc == 0) skinv[buf][kr] = inv;
    }
    {   // V block piece: scale of the (16 keys x 32 d) block = max over the wave
      const float4 a = affine4(vraw0, vra0, sb + 4 * DH + 32 * vmb + 4 * vcc), c4 = affine4(vraw1, vra1, sb + 4 * DH + 32 * vmb + 4 * vcc);
      const float x0[4] = {a.x, a.y, a.z, a.w}, x1[4] = {c4.x, c4.y, c4.z, c4.w};
      float mx = 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) mx = fmaxf(mx, fmaxf(fabsf(x0[e]), fabsf(x1[e])));
      mx = e32_wave_max(mx);
      float s, inv;
      e32a_scale_of(mx, s, inv);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        half_t h0, l0, h1, l1;
        e32a_split(x0[e] * s, h0, l0);
        e32a_split(x1[e] * s, h1, l1);
        const int d = 32 * vmb + 4 * vcc + e;
        *(half2_t*)(vh[buf] + d * VROW + 16 * vt_ + 2 * vpp) = (half2_t){h0, h1};
        *(half2_t*)(vl[buf] + d * VROW + 16 * vt_ + 2 * vpp) = (half2_t){l0, l1};
      }
      if (lane == 0) svinv[buf][2 * vt_ + vmb] = inv * 0.00048828125f;   // · 2^-11: p was split as p · 2^11
    }
  };

  prefetch(0);   // the first tile's rows travel while q is loaded and split
  const int q0 = qb * 32;
  half8 qh[KS], ql[KS];
  float sqinv = 1.f;
  if (live) {
    const int qrow = (q0 + col < L) ? q0 + col : L - 1;
    const float r = ra[tok0 + qrow];
    float x[KS][8];
    float mx = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const float* src = P + (tok0 + qrow) * rs + hd * DH + 16 * ks + 8 * hh;
      const float4 a = affine4(load4(src), r, sb + 16 * ks + 8 * hh), c4 = affine4(load4(src + 4), r, sb + 16 * ks + 8 * hh + 4);
      x[ks][0] = a.x; x[ks][1] = a.y; x[ks][2] = a.z; x[ks][3] = a.w;
      x[ks][4] = c4.x; x[ks][5] = c4.y; x[ks][6] = c4.z; x[ks][7] = c4.w;
#pragma unroll
      for (int e = 0; e < 8; ++e) mx = fmaxf(mx, fabsf(x[ks][e]));
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));   // the other half of the row lives in lane l ^ 32
    float s;
    e32a_scale_of(mx, s, sqinv);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        half_t h, l;
        e32a_split(x[ks][e] * s, h, l);
        qh[ks][e] = h; ql[ks][e] = l;
      }
  }
  const float scale = 0.125f;   // 1/sqrt(64): a power of two, folded into the scale product below
  f32x16 o[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) o[mb] = (f32x16){0};
  float m_run = -INFINITY, l_run = 0.f;

  int buf = 0;
  for (int k0 = 0; k0 < len; k0 += 32, buf ^= 1) {
    // image `buf` was last read while tile k0 - 64 was multiplied; every wave has passed the barrier of tile k0 - 32 since
    store_tile(buf);
    __syncthreads();
    if (k0 + 32 < len) prefetch(k0 + 32);   // in flight under this tile's MFMAs and softmax
    if (!live) continue;                    // (a wave without a query block only stages)

    // ---- S^T tile: keys k0..k0+31 (rows) x this wave's 32 queries (columns), small terms first ----
    f32x16 st = {0};
    const half_t* khp = kh[buf] + col * KROW + 8 * hh;
    const half_t* klp = kl[buf] + col * KROW + 8 * hh;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const half8 ah = *(const half8*)(khp + 16 * ks), al = *(const half8*)(klp + 16 * ks);
      st = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, qh[ks], st, 0, 0, 0);
      st = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, ql[ks], st, 0, 0, 0);
      st = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, qh[ks], st, 0, 0, 0);
    }
    RARC_MFMA_SETTLE(st);
    // ---- softmax statistics of this lane's query over its 16 keys, then with the partner lane ----
    float pr[16];
    float tmax = -INFINITY;
    const float fq = sqinv * scale;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 ski = *(const float4*)(skinv[buf] + 8 * g + 4 * hh);
      const float f[4] = {ski.x * fq, ski.y * fq, ski.z * fq, ski.w * fq};   // powers of two: exact
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int r = 4 * g + j;
        const int key = k0 + 8 * g + 4 * hh + j;
        float v = st[r] * f[j];
        if (REL) {   // (index clamped into the table: keys past the sequence are masked below anyway)
          int ri = key - (q0 + col) + rel_span - 1;
          ri = ri < 0 ? 0 : (ri > 2 * rel_span - 2 ? 2 * rel_span - 2 : ri);
          v += rel[(size_t)hd * (2 * rel_span - 1) + ri];
        }
        pr[r] = key < len ? v : -INFINITY;
        tmax = fmaxf(tmax, pr[r]);
      }
    }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    const float m_new = fmaxf(m_run, tmax);
    const float corr = (m_run == -INFINITY) ? 0.f : expf(m_run - m_new);
    float psum = 0.f;
    uint32_t pkh[8], pkl[8];
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      const float p0 = (pr[r] == -INFINITY) ? 0.f : expf(pr[r] - m_new);
      const float p1 = (pr[r + 1] == -INFINITY) ? 0.f : expf(pr[r + 1] - m_new);
      psum += p0;
      psum += p1;
      half_t h0, l0, h1, l1;
      e32a_split(p0 * 2048.f, h0, l0);
      e32a_split(p1 * 2048.f, h1, l1);
      pkh[r >> 1] = __builtin_bit_cast(uint32_t, (half2_t){h0, h1});
      pkl[r >> 1] = __builtin_bit_cast(uint32_t, (half2_t){l0, l1});
    }
    psum += __shfl_xor(psum, 32, 64);
    l_run = l_run * corr + psum;
    m_run = m_new;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[mb][r] *= corr;
    // ---- O^T += V^T · P^T.  k-step ks contracts keys 16ks .. 16ks+15: half hh needs keys 16ks + 8hh .. + 7.  A lane holds
    // key groups g = 0..3 (keys 8g + 4hh .. + 3) as the word pairs pk[2g], pk[2g+1]; one v_permlane32_swap per word with
    // vdst = group 2ks, src = group 2ks + 1 leaves the two B fragments in place (decoder.hip) ----
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const auto h0 = __builtin_amdgcn_permlane32_swap(pkh[4 * ks], pkh[4 * ks + 2], false, false);
      const auto h1 = __builtin_amdgcn_permlane32_swap(pkh[4 * ks + 1], pkh[4 * ks + 3], false, false);
      const auto l0 = __builtin_amdgcn_permlane32_swap(pkl[4 * ks], pkl[4 * ks + 2], false, false);
      const auto l1 = __builtin_amdgcn_permlane32_swap(pkl[4 * ks + 1], pkl[4 * ks + 3], false, false);
      const half8 bhi = __builtin_bit_cast(half8, ((u32x4){h0[0], h1[0], h0[1], h1[1]}));
      const half8 blo = __builtin_bit_cast(half8, ((u32x4){l0[0], l1[0], l0[1], l1[1]}));
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
        const half8 vfh = *(const half8*)(vh[buf] + (32 * mb + col) * VROW + 16 * ks + 8 * hh);
        const half8 vfl = *(const half8*)(vl[buf] + (32 * mb + col) * VROW + 16 * ks + 8 * hh);
        f32x16 t = {0};
        t = __builtin_amdgcn_mfma_f32_32x32x16_f16(vfl, bhi, t, 0, 0, 0);
        t = __builtin_amdgcn_mfma_f32_32x32x16_f16(vfh, blo, t, 0, 0, 0);
        t = __builtin_amdgcn_mfma_f32_32x32x16_f16(vfh, bhi, t, 0, 0, 0);
        RARC_MFMA_SETTLE(t);
        const float svi = svinv[buf][2 * ks + mb];
#pragma unroll
        for (int r = 0; r < 16; ++r) o[mb][r] = __builtin_fmaf(t[r], svi, o[mb][r]);
      }
    }
  }
  // ---- store: lane (query, hh) holds d = 32mb + 8(r>>2) + 4hh + (r&3) ----
  if (live && q0 + col < L) {
    const float inv = l_run > 0.f ? 1.f / l_run : 0.f;
    float* out = ctx + (tok0 + q0 + col) * H + hd * DH;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *(float4*)(out + 32 * mb + 8 * g + 4 * hh) =
            make_float4(o[mb][4 * g] * inv, o[mb][4 * g + 1] * inv, o[mb][4 * g + 2] * inv, o[mb][4 * g + 3] * inv);
  }
}

// ------------------------------------------------------------------------------------------
// Round 6 — attention of the QUERY PATH: sequences of at most 32 padded tokens (one query block, one key tile), head_dim 64 or 32.
// rarc_e32_attention_split_kernel is built for 512-token documents: shared K / V images of 32-key tiles in LDS, double
// buffered, ONE wave per 32-query block doing the softmax of 1024 scores — at one tile and one block that is a single wave's
// instruction stream, 11 µs per layer for a single query, more than any of its projections (a lone wave issues one VALU
// instruction per four cycles; the first version of THIS kernel, one wave per head with everything in registers, took 14 µs).
// Here a workgroup of FOUR waves owns a (sequence, head) and every phase is cut four ways on the 16 x 16 x 32 MFMA:
//   A  all 256 threads fetch q, k, v (8 dims of one token each; split-K slabs summed on the way in), scale rows of q and k,
//      columns of v (one power of two per COLUMN d, finer than the document kernel's per-block one), write (hi, lo) images;
//   B  wave (kb, qb) multiplies the 16-key x 16-query block S^T[kb][qb] (6 MFMAs), does the softmax of ITS 256 scores — four
//      expf per lane — and trades row maxima / sums with the wave of the other key half through LDS; p·2^11 -> (hi, lo) image;
//   C  wave (qb, d half) multiplies two 16-d x 16-query blocks of O^T (6 MFMAs) and writes its part of the epilogue.
// Same arithmetic class as the document kernel: (hi, lo) fp16 pairs of value · 2^e, products lo·hi, hi·lo, hi·hi in that order
// into fp32 accumulators, libm expf, fp32 statistics.  Keys >= lens[seq] are masked and their V rows read as zero.
// And the row pass that used to follow the attention — ctx -> split image for the output projection, a launch of its own on
// 32 workgroups — is this kernel's epilogue: the power-of-two scale is taken per (token, HEAD) from the head's own 64 outputs
// (no reduction across heads exists to wait for), rah[token][head] keeps its inverse, and the output projection's weight
// stream multiplies each wave's partial product — a wave's k slice lies inside ONE head — by it before the slices are added
// (E32SkinnyEpi.wave_scale).  ra_one[token] = 1: the projection's row pass has no row scale left to apply.
// ------------------------------------------------------------------------------------------
template <int DH, bool REL>   // DH: head_dim 64 (bge-base / -large, MPNet) or 32 (bge-small, MiniLM)
__global__ __launch_bounds__(256) void rarc_e32q_attention_kernel(const float* __restrict__ P, const float* __restrict__ ra,
                                                                  const float* __restrict__ rw, const float* __restrict__ bias,
                                                                  const int32_t* __restrict__ lens, int H, int n_heads,
                                                                  half_t* __restrict__ xq, float* __restrict__ rah,
                                                                  float* __restrict__ ra_one, const float* __restrict__ rel,
                                                                  int rel_span, int n_parts, size_t part_stride) {
  constexpr int L = 32, C8 = DH / 8;   // C8: threads per token row in phase A (8 dims each)
  constexpr int QROW = DH + 8;   // halves per q / k row (+16 bytes: conflict-free ds_read_b128 fragments)
  constexpr int TROW = L + 8;    // halves per V^T row (32 keys) / P^T row (32 keys)
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  __shared__ __attribute__((aligned(16))) half_t qh[L * QROW], ql[L * QROW], kh[L * QROW], kl[L * QROW];
  __shared__ __attribute__((aligned(16))) half_t vh[DH * TROW], vl[DH * TROW], ph[L * TROW], pl[L * TROW];
  __shared__ __attribute__((aligned(16))) float sqinv[L], skinv[L], svinv[DH];
  __shared__ uint32_t vmax[DH];
  __shared__ float smax[2][L], ssum[2][L], somax[2][L];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x / n_heads, hd = blockIdx.x % n_heads;
  const int len = lens[b] < 1 ? 1 : (lens[b] > L ? L : lens[b]);
  if (tid < DH) vmax[tid] = 0u;
  // ---- A: thread (token row = tid >> 3, c8 = tid & 7) owns dims 8 c8 .. + 7 of the row's q, k and v ----
  const bool loader = tid < L * C8;            // (head_dim 32: the first two waves fetch, all four multiply)
  const int row = loader ? tid / C8 : 0, c8 = tid % C8;
  const size_t mrow = (size_t)b * L + row;
  float x[3][8] = {};
  if (loader) {
    const float r = ra[mrow];
    const float* src = P + mrow * 3 * (size_t)H + hd * DH + 8 * c8;
#pragma unroll
    for (int part = 0; part < 3; ++part) {
      const float* w = rw + part * H + hd * DH + 8 * c8;
      const float* bb = bias + part * H + hd * DH + 8 * c8;
#pragma unroll
      for (int h4 = 0; h4 < 2; ++h4) {
        const float4 pv = e32_sum_parts(src + part * H + 4 * h4, n_parts, part_stride);
        const float4 s4 = *(const float4*)(w + 4 * h4), b4 = *(const float4*)(bb + 4 * h4);
        x[part][4 * h4] = pv.x * r * s4.x + b4.x; x[part][4 * h4 + 1] = pv.y * r * s4.y + b4.y;
        x[part][4 * h4 + 2] = pv.z * r * s4.z + b4.z; x[part][4 * h4 + 3] = pv.w * r * s4.w + b4.w;
      }
    }
  }
  if (row >= len || !loader) {   // a masked key's V row reads as zero: nothing of the padding reaches a result
#pragma unroll
    for (int e = 0; e < 8; ++e) x[2][e] = 0.f;
  }
  __syncthreads();    // vmax is zero
  if (loader) {
#pragma unroll
    for (int e = 0; e < 8; ++e) atomicMax(&vmax[8 * c8 + e], __float_as_uint(fabsf(x[2][e])));   // (non-negative floats order as integers)
  }
#pragma unroll
  for (int part = 0; part < 2; ++part) {   // q, k: one scale per row = the maximum over the row's eight lanes
    float mx = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) mx = fmaxf(mx, fabsf(x[part][e]));
    mx = fmaxf(mx, __shfl_xor(mx, 1, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 2, 64));
    if (C8 == 8) mx = fmaxf(mx, __shfl_xor(mx, 4, 64));
    float sc, inv;
    e32a_scale_of(mx, sc, inv);
    half8 hi, lo;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      half_t h_, l_;
      e32a_split(x[part][e] * sc, h_, l_);
      hi[e] = h_; lo[e] = l_;
    }
    if (loader) {
      *(half8*)((part ? kh : qh) + row * QROW + 8 * c8) = hi;
      *(half8*)((part ? kl : ql) + row * QROW + 8 * c8) = lo;
      if (c8 == 0) (part ? skinv : sqinv)[row] = inv;
    }
  }
  __syncthreads();    // every column's maximum is in
#pragma unroll
  for (int e = 0; e < 8; ++e) {   // v: one scale per column d; the images are V^T [d][key]
    if (!loader) break;
    const int d = 8 * c8 + e;
    float sc, inv;
    e32a_scale_of(__uint_as_float(vmax[d]), sc, inv);
    half_t h_, l_;
    e32a_split(x[2][e] * sc, h_, l_);
    vh[d * TROW + row] = h_;
    vl[d * TROW + row] = l_;
    if (row == 0) svinv[d] = inv;
  }
  __syncthreads();
  // ---- B: wave (kb, qb): S^T block of keys 16kb.. x queries 16qb..; lane (query col c, rq) ends with keys 16kb + 4rq + e ----
  const int c = lane & 15, rq = lane >> 4;
  const int kb = wave & 1, qb = wave >> 1;
  const int query = 16 * qb + c;
  f32x4 st = {0, 0, 0, 0};
  {
    constexpr int S2 = DH / 32;      // k steps of 32 over the head's dims
    half8 akh[S2], akl[S2], bqh[S2], bql[S2];
#pragma unroll
    for (int s2 = 0; s2 < S2; ++s2) {
      akh[s2] = *(const half8*)(kh + (16 * kb + c) * QROW + 32 * s2 + 8 * rq);
      akl[s2] = *(const half8*)(kl + (16 * kb + c) * QROW + 32 * s2 + 8 * rq);
      bqh[s2] = *(const half8*)(qh + query * QROW + 32 * s2 + 8 * rq);
      bql[s2] = *(const half8*)(ql + query * QROW + 32 * s2 + 8 * rq);
    }
#pragma unroll
    for (int s2 = 0; s2 < S2; ++s2) st = __builtin_amdgcn_mfma_f32_16x16x32_f16(akl[s2], bqh[s2], st, 0, 0, 0);
#pragma unroll
    for (int s2 = 0; s2 < S2; ++s2) st = __builtin_amdgcn_mfma_f32_16x16x32_f16(akh[s2], bql[s2], st, 0, 0, 0);
#pragma unroll
    for (int s2 = 0; s2 < S2; ++s2) st = __builtin_amdgcn_mfma_f32_16x16x32_f16(akh[s2], bqh[s2], st, 0, 0, 0);
    RARC_MFMA_SETTLE(st);
  }
  float pr[4];
  float tmax = -INFINITY;
  {
    const float4 ski = *(const float4*)(skinv + 16 * kb + 4 * rq);
    const float fq = sqinv[query] * (DH == 64 ? 0.125f : 0.17677669529663687f);   // 1/sqrt(head_dim); a power of two at 64
    const float f[4] = {ski.x * fq, ski.y * fq, ski.z * fq, ski.w * fq};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int key = 16 * kb + 4 * rq + e;
      float v = st[e] * f[e];
      if (REL) {
        int ri = key - query + rel_span - 1;
        ri = ri < 0 ? 0 : (ri > 2 * rel_span - 2 ? 2 * rel_span - 2 : ri);
        v += rel[(size_t)hd * (2 * rel_span - 1) + ri];
      }
      pr[e] = key < len ? v : -INFINITY;
      tmax = fmaxf(tmax, pr[e]);
    }
  }
  tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
  tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
  if (rq == 0) smax[kb][query] = tmax;
  __syncthreads();
  const float m_all = fmaxf(smax[0][query], smax[1][query]);   // (key 0 is never masked: finite)
  float psum = 0.f;
  {
    half4_t hi, lo;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float pe = (pr[e] == -INFINITY) ? 0.f : expf(pr[e] - m_all);
      psum += pe;
      half_t h_, l_;
      e32a_split(pe * 2048.f, h_, l_);
      hi[e] = h_; lo[e] = l_;
    }
    *(half4_t*)(ph + query * TROW + 16 * kb + 4 * rq) = hi;
    *(half4_t*)(pl + query * TROW + 16 * kb + 4 * rq) = lo;
  }
  psum += __shfl_xor(psum, 16, 64);
  psum += __shfl_xor(psum, 32, 64);
  if (rq == 0) ssum[kb][query] = psum;
  __syncthreads();
  // ---- C: wave (qb, dpair = wave & 1): O^T blocks d = 16 db.. for db = 2 dpair, 2 dpair + 1, x queries 16qb.. ----
  const int dpair = wave & 1;
  const float inv_l = 0.00048828125f / (ssum[0][query] + ssum[1][query]);   // · 2^-11: p was split as p · 2^11
  constexpr int NBW = DH / 32;     // 16-d blocks of O^T per wave
  float cval[NBW][4];
  float mo = 0.f;
  {
    const half8 bh = *(const half8*)(ph + query * TROW + 8 * rq), bl = *(const half8*)(pl + query * TROW + 8 * rq);
#pragma unroll
    for (int i = 0; i < NBW; ++i) {
      const int db = NBW * dpair + i;
      const half8 avh = *(const half8*)(vh + (16 * db + c) * TROW + 8 * rq), avl = *(const half8*)(vl + (16 * db + c) * TROW + 8 * rq);
      f32x4 o = {0, 0, 0, 0};
      o = __builtin_amdgcn_mfma_f32_16x16x32_f16(avl, bh, o, 0, 0, 0);
      o = __builtin_amdgcn_mfma_f32_16x16x32_f16(avh, bl, o, 0, 0, 0);
      o = __builtin_amdgcn_mfma_f32_16x16x32_f16(avh, bh, o, 0, 0, 0);
      RARC_MFMA_SETTLE(o);
      const float4 svi = *(const float4*)(svinv + 16 * db + 4 * rq);
      cval[i][0] = o[0] * svi.x * inv_l; cval[i][1] = o[1] * svi.y * inv_l;
      cval[i][2] = o[2] * svi.z * inv_l; cval[i][3] = o[3] * svi.w * inv_l;
#pragma unroll
      for (int e = 0; e < 4; ++e) mo = fmaxf(mo, fabsf(cval[i][e]));
    }
  }
  mo = fmaxf(mo, __shfl_xor(mo, 16, 64));
  mo = fmaxf(mo, __shfl_xor(mo, 32, 64));
  if (rq == 0) somax[dpair][query] = mo;
  __syncthreads();
  // ---- epilogue: the (token, head) block of the output projection's operand, split under its own scale ----
  float so, soinv;
  e32_scale_of(fmaxf(somax[0][query], somax[1][query]), so, soinv);
  const size_t mq = (size_t)b * L + query;
#pragma unroll
  for (int i = 0; i < NBW; ++i) {
    half4_t hi, lo;
    e32_split4(make_float4(cval[i][0], cval[i][1], cval[i][2], cval[i][3]), so, hi, lo);
    half_t* dst = xq + e32_frag_offset(mq, hd * DH + 16 * (NBW * dpair + i) + 4 * rq, H);
    *(half4_t*)dst = hi;
    *(half4_t*)(dst + 512) = lo;
  }
  if (dpair == 0 && rq == 0) {
    rah[mq * n_heads + hd] = soinv;
    if (hd == 0) ra_one[mq] = 1.f;
  }
}

// pooling over fp32 hidden states: CLS row or the mean of the real tokens; optional canonical L2 normalisation
__global__ __launch_bounds__(256) void rarc_e32_pool_kernel(const float* __restrict__ hidden, const int32_t* __restrict__ lens,
                                                            int L, int H, int mean, int normalize, float* __restrict__ out) {
  __shared__ float s_vec[1024];
  __shared__ float s_nr;
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* x = hidden + (size_t)b * L * H;
  if (mean) {
    const int len = lens[b] < 1 ? 1 : (lens[b] > L ? L : lens[b]);
    for (int c = tid; c < H; c += 256) {
      float acc = 0.f;
      for (int t = 0; t < len; ++t) acc += x[(size_t)t * H + c];
      s_vec[c] = acc / (float)len;
    }
  } else {
    for (int c = tid; c < H; c += 256) s_vec[c] = x[c];
  }
  __syncthreads();
  if (tid < 64) {
    float acc = 0.f;
    if (tid < 8)
      for (int m2 = tid; m2 < H; m2 += 8) acc = __builtin_fmaf(s_vec[m2], s_vec[m2], acc);
    float a[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = __shfl(acc, i, 64);
    if (tid == 0) s_nr = rarc_canon_tree(a);
  }
  __syncthreads();
  const float nr = s_nr;
  const float inv = (normalize && nr > 0.f) ? (float)(1.0 / (double)(float)sqrt((double)nr)) : 1.f;
  for (int c = tid; c < H; c += 256) out[(size_t)b * H + c] = s_vec[c] * inv;
}


// ------------------------------------------------------------------------------------------
// The QUERY PATH (round 6): forwards of at most 128 tokens — the call the reference issues per search, one query's
// `embed_query` (huggingface.py:136-145 -> VectorStore_Faiss.py:240).  At 32..128 tokens a projection is a weight STREAM: the
// FFN's 8M parameters against 32 rows of activations.  The tile GEMMs of encoder.hip run it as 128 x 128 tiles with split-K,
// one or two hundred workgroups each walking a k loop of dependent LDS-DMA stages: 15 µs per projection, 1.2 TB/s of weight
// bytes, on a kernel built for 256-row tiles.  rarc_e32_skinny_gemm_kernel is the other shape:
//   * the weights are the MFMA's A operand straight from HBM: a load-time image in FRAGMENT order (rarc_enc32_pack_query_weight:
//     per 32 output features and 16-wide k step the W_hi fragment then the W_lo fragment, 1 KiB each — every wave-wide load
//     instruction is one contiguous KiB; W_hi is read ONCE, not twice as in the [W_hi | W_lo | W_hi] rows the tile kernels
//     stream: 4 bytes per parameter instead of 6);
//   * the activations are the B operand from a fragment-major split image the row passes write (e32_frag_offset): L2 hits;
//   * a wave owns 32 features x KR k steps: ALL of its loads (2·KR weight KiB + 2·KR activation KiB per 32 tokens) are issued
//     before the first MFMA — the kernel is one memory round trip deep, not a pipeline;
//   * the four waves of a workgroup hold consecutive k slices of the same features and add their accumulators through LDS
//     in wave order (deterministic); workgroup y writes partial slab y — the consumers (attention loads, row passes) sum the
//     slabs in order, as they do for the tile kernels' split-K.
// Arithmetic per output element: the same three split products (lo·hi and hi·lo first, then hi·hi), fp16 products exact in
// the fp32 accumulator; only the GROUPING of the k range differs from the tile kernels' (slices of KR·16 here), i.e. the
// two forwards differ by fp32 summation order, not by method (tests/test_gpu_encoder_query.py bounds it).
// ------------------------------------------------------------------------------------------
// EPI 1 (FFN1): the whole k range inside ONE workgroup of WAVES waves (no partial slabs), and the row pass that used to follow —
// bias + GELU + split, a launch of its own on 32 workgroups — is this kernel's epilogue: gelu(acc·ra·rw + b)·sg -> (hi, lo) ->
// the fragment-major image FFN2 streams.  sg[m] is the power-of-two scale from the BOUND the LayerNorm pass in front of FFN1
// computes (E32Fuse.colmax: sum_k |x_k| max_j |W1[j][k]| + max_j |b1[j]|  >=  every |gelu(.)| of the row), as for the tile
// kernels' fused FFN1 (rarc_gemm256_f16_kernel<5>): no pass over the finished row is needed to know it.
struct E32SkinnyEpi {
  const float *ra = nullptr, *rw = nullptr, *bias = nullptr, *sg = nullptr;
  half_t* outq = nullptr;
  // EPI 0 only — the activations were split under one scale per (token, group of `ks_per_group` k steps) instead of one per
  // token (the query attention's per-head scales): wave_scale[token · n_groups + group] multiplies a wave's partial product
  // before the slices are added.  A wave's k slice must lie inside one group (KR <= ks_per_group).
  const float* wave_scale = nullptr;
  int ks_per_group = 0, n_groups = 0;
};

template <int MT, int KR, int WAVES, int EPI>   // MT: 32-token blocks (M = 32·MT), KR: k steps (of 16) per wave
__global__ __launch_bounds__(WAVES * 64) void rarc_e32_skinny_gemm_kernel(const half_t* __restrict__ wq,
                                                                          const half_t* __restrict__ xq, float* __restrict__ P,
                                                                          int N, int KS, int M, const E32SkinnyEpi ep) {
  constexpr int TS = 36;   // floats per token row of a wave's tile in LDS (32 + 4: the rows start on different banks)
  __shared__ __attribute__((aligned(16))) float red[WAVES][32 * TS];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nb = blockIdx.x, ks0 = (blockIdx.y * WAVES + wave) * KR;
  f32x16 acc[MT];
#pragma unroll
  for (int mb = 0; mb < MT; ++mb) acc[mb] = (f32x16){0};
  if (ks0 < KS) {   // (a k range that is not a multiple of four slices leaves the last workgroup's upper waves idle: zeros)
    const half8* wp = (const half8*)(wq + ((size_t)nb * KS + ks0) * 1024) + lane;
    half8 wh[KR], wl[KR];
#pragma unroll
    for (int i = 0; i < KR; ++i) {
      wh[i] = __builtin_nontemporal_load(wp + i * 128);        // streamed once per forward: do not displace the activations in L2
      wl[i] = __builtin_nontemporal_load(wp + i * 128 + 64);
    }
    half8 xh[MT][KR], xl[MT][KR];
#pragma unroll
    for (int mb = 0; mb < MT; ++mb) {
      const half8* xp = (const half8*)(xq + ((size_t)mb * KS + ks0) * 1024) + lane;
#pragma unroll
      for (int i = 0; i < KR; ++i) {
        xh[mb][i] = xp[i * 128];
        xl[mb][i] = xp[i * 128 + 64];
      }
    }
    // every load of the wave is in flight before the first MFMA waits for one: ONE memory round trip.  (Left to itself the
    // scheduler starts the chain behind the first four loads and issues the rest between its waits: three to four round trips.)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int mb = 0; mb < MT; ++mb) {
#pragma unroll
      for (int i = 0; i < KR; ++i) acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[i], xl[mb][i], acc[mb], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < KR; ++i) acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[i], xh[mb][i], acc[mb], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < KR; ++i) acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[i], xh[mb][i], acc[mb], 0, 0, 0);
    }
    RARC_MFMA_SETTLE(acc);
    if (EPI == 0 && ep.wave_scale) {   // powers of two: exact
#pragma unroll
      for (int mb = 0; mb < MT; ++mb) {
        const float ws = ep.wave_scale[(size_t)(mb * 32 + (lane & 31)) * ep.n_groups + ks0 / ep.ks_per_group];
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[mb][i] *= ws;
      }
    }
  }
  // lane (token = lane & 31, h = lane >> 5) holds features 8g + 4h + e in acc[4g + e]
  const int tok = lane & 31, h = lane >> 5;
  const int tr = threadIdx.x >> 3, c4 = threadIdx.x & 7;   // 8 threads write one token's 32 features: 128 contiguous bytes
#pragma unroll
  for (int mb = 0; mb < MT; ++mb) {
    if (mb) __syncthreads();
#pragma unroll
    for (int g = 0; g < 4; ++g)
      *(float4*)&red[wave][tok * TS + 8 * g + 4 * h] =
          make_float4(acc[mb][4 * g], acc[mb][4 * g + 1], acc[mb][4 * g + 2], acc[mb][4 * g + 3]);
    __syncthreads();
    if (threadIdx.x < 256) {
      float4 o = *(const float4*)&red[0][tr * TS + 4 * c4];
#pragma unroll
      for (int w = 1; w < WAVES; ++w) {     // wave order: the sum does not depend on which wave finished first
        const float4 a = *(const float4*)&red[w][tr * TS + 4 * c4];
        o.x += a.x; o.y += a.y; o.z += a.z; o.w += a.w;
      }
      if (EPI == 0) {
        *(float4*)(P + ((size_t)blockIdx.y * M + mb * 32 + tr) * N + nb * 32 + 4 * c4) = o;
      } else {
        const size_t m = (size_t)mb * 32 + tr;
        const int n = nb * 32 + 4 * c4;
        const float ram = ep.ra[m], sgm = ep.sg[m];
        const float4 w4 = *(const float4*)(ep.rw + n), b4 = *(const float4*)(ep.bias + n);
        // (ra, rw, sg are powers of two: exact; the bias add rounds once, as in x·Wᵀ + b — rarc_e32_epi_kernel<., 1>'s arithmetic)
        const float4 g4 = make_float4(e32_gelu(o.x * ram * w4.x + b4.x) * sgm, e32_gelu(o.y * ram * w4.y + b4.y) * sgm,
                                      e32_gelu(o.z * ram * w4.z + b4.z) * sgm, e32_gelu(o.w * ram * w4.w + b4.w) * sgm);
        half4_t hi, lo;
        e32_split4(g4, 1.f, hi, lo);
        half_t* dst = ep.outq + e32_frag_offset(m, n, N);
        *(half4_t*)dst = hi;
        *(half4_t*)(dst + 512) = lo;
      }
    }
  }
}

// [W_hi | W_lo | W_hi] rows (rarc_enc32_split_weight) -> the fragment-major image the skinny GEMM streams: one workgroup per
// (32 features, 16-wide k step): 2 fragments x 64 lanes x 8 halves
__global__ __launch_bounds__(128) void rarc_e32_pack_weight_kernel(const half_t* __restrict__ w3, int K, half_t* __restrict__ wq) {
  const int KS = K >> 4;
  const size_t nb = blockIdx.x / KS;
  const int ks = blockIdx.x % KS;
  const int frag = threadIdx.x >> 6, lane = threadIdx.x & 63;        // 0: hi, 1: lo
  const half_t* src = w3 + (nb * 32 + (lane & 31)) * 3 * (size_t)K + (size_t)frag * K + ks * 16 + 8 * (lane >> 5);
  *(uint4*)(wq + (((nb * KS + ks) * 2 + frag) * 64 + lane) * 8) = *(const uint4*)src;
}

extern "C" int rarc_enc32_pack_query_weight(const uint16_t* d_w3, int n, int k, uint16_t* d_wq, void* stream) {
  RARC_REQUIRE(d_w3 && d_wq, RARC_E_INVALID, "rarc_enc32_pack_query_weight: null pointer");
  RARC_REQUIRE(n > 0 && k > 0 && n % 32 == 0 && k % 128 == 0, RARC_E_UNSUPPORTED,
               "rarc_enc32_pack_query_weight: need n a multiple of 32 and k a multiple of 128 (got %d, %d)", n, k);
  hipLaunchKernelGGL(rarc_e32_pack_weight_kernel, dim3((unsigned)((size_t)(n / 32) * (k / 16))), dim3(128), 0, (hipStream_t)stream,
                     (const half_t*)d_w3, k, (half_t*)d_wq);
  RARC_HIP_CHECK(hipGetLastError());
  return RARC_OK;
}

#define E32Q_MAX_TOKENS 128   // the query path's largest forward
#define E32Q_MAX_PARTS 16     // partial slabs a skinny product may come in
#define E32Q_DEFAULT_WAVES 4  // waves per workgroup of the slab-writing weight stream (RARC_E32Q_WAVES overrides: A/B)

// one skinny product: P[part][m][n] (part < *parts) = partial sums of A·Wᵀ over the part's k slices.  KR (k steps per wave) is
// the largest of 8 / 4 / 2 that still gives the chip ~200 workgroups, bounded below by E32Q_MAX_PARTS slabs.
static int e32_skinny_gemm(const uint16_t* xq, const uint16_t* wq, float* P, int m, int n, int k, int* parts, hipStream_t s,
                           const float* wave_scale = nullptr, int ks_per_group = 0) {
  const int KS = k / 16, nb = n / 32, MT = m / 32;
  static const int force_kr = getenv("RARC_E32Q_KR") ? atoi(getenv("RARC_E32Q_KR")) : 0;   // (A/B: 8 / 4 / 2)
  static const int waves_env = getenv("RARC_E32Q_WAVES") ? atoi(getenv("RARC_E32Q_WAVES")) : 0;   // (A/B: 4 / 8 waves per workgroup)
  // waves per workgroup: 8 (two per SIMD) halves the number of partial slabs the consumer has to sum — the row passes and the
  // attention read every slab — at the same number of loads in flight per wave
  const int W = waves_env == 4 ? 4 : (waves_env == 8 ? 8 : E32Q_DEFAULT_WAVES);
  int kr = 8;
  // (measured flat, tools/r06/encq_ab.sh: 4 / 8 waves and KR 8 / 4 / 2 all land within 2 % of each other on a whole forward —
  //  0.97 .. 1.00 ms for bge-large's single query; fewer, larger slices were the best of them, hence the low bar of ~100 workgroups)
  while (kr > 2 && nb * ((KS + W * kr - 1) / (W * kr)) < 96) kr >>= 1;
  if (MT == 4 && kr > 4) kr = 4;                       // (registers: MT·KR fragments of activations are in flight)
  if (force_kr == 8 || force_kr == 4 || force_kr == 2) kr = (MT == 4 && force_kr > 4) ? 4 : force_kr;
  while (kr < 8 && (KS + W * kr - 1) / (W * kr) > E32Q_MAX_PARTS) kr <<= 1;
  if (wave_scale) while (kr > ks_per_group) kr >>= 1;      // a wave's slice inside one scale group
  const int S = (KS + W * kr - 1) / (W * kr);
  RARC_REQUIRE(S <= E32Q_MAX_PARTS && (MT == 1 || MT == 2 || MT == 4) && (MT < 4 || kr <= 4), RARC_E_UNSUPPORTED,
               "fp32-class query path: product %d x %d x %d not supported", m, n, k);
  *parts = S;
  const dim3 grid(nb, S);
  E32SkinnyEpi none;
  none.wave_scale = wave_scale; none.ks_per_group = ks_per_group; none.n_groups = ks_per_group ? KS / ks_per_group : 0;
#define E32Q_LAUNCH_W(MTV, KRV, WV)                                                                                    \
  hipLaunchKernelGGL((rarc_e32_skinny_gemm_kernel<MTV, KRV, WV, 0>), grid, dim3(64 * WV), 0, s, (const half_t*)wq, (const half_t*)xq, P, n, KS, m, none)
#define E32Q_LAUNCH(MTV, KRV) do { if (W == 8) E32Q_LAUNCH_W(MTV, KRV, 8); else E32Q_LAUNCH_W(MTV, KRV, 4); } while (0)
  if (MT == 1) { if (kr == 8) E32Q_LAUNCH(1, 8); else if (kr == 4) E32Q_LAUNCH(1, 4); else E32Q_LAUNCH(1, 2); }
  else if (MT == 2) { if (kr == 8) E32Q_LAUNCH(2, 8); else if (kr == 4) E32Q_LAUNCH(2, 4); else E32Q_LAUNCH(2, 2); }
  else { if (kr == 4) E32Q_LAUNCH(4, 4); else E32Q_LAUNCH(4, 2); }
#undef E32Q_LAUNCH
#undef E32Q_LAUNCH_W
  RARC_HIP_CHECK(hipGetLastError());
  return RARC_OK;
}

// FFN1 of the query path with bias + GELU + split as the product's epilogue (EPI 1): needs the whole k range in one
// workgroup of eight waves with all loads in flight — k <= 1024 at 32 / 64 tokens (KR 8), k <= 512 at 128 tokens (KR 4).
static bool e32_skinny_gelu_takes(int m, int k) {
  const char* e = getenv("RARC_E32Q_FUSE_GELU");   // (read per call: tests switch it between forwards)
  if (e && atoi(e) == 0) return false;
  const int KS = k / 16, MT = m / 32;
  return MT == 4 ? KS <= 32 : KS <= 64;
}
static int e32_skinny_gemm_gelu(const uint16_t* xq, const uint16_t* wq, const float* ra, const float* rw, const float* bias,
                                const float* sg, uint16_t* outq, int m, int n, int k, hipStream_t s) {
  const int KS = k / 16, nb = n / 32, MT = m / 32;
  E32SkinnyEpi ep;
  ep.ra = ra; ep.rw = rw; ep.bias = bias; ep.sg = sg; ep.outq = (half_t*)outq;
#define E32Q_LAUNCH(MTV, KRV)                                                                                          \
  hipLaunchKernelGGL((rarc_e32_skinny_gemm_kernel<MTV, KRV, 8, 1>), dim3(nb, 1), dim3(512), 0, s, (const half_t*)wq, (const half_t*)xq, \
                     (float*)nullptr, n, KS, m, ep)
  if (MT == 1) E32Q_LAUNCH(1, 8); else if (MT == 2) E32Q_LAUNCH(2, 8); else E32Q_LAUNCH(4, 4);
#undef E32Q_LAUNCH
  RARC_HIP_CHECK(hipGetLastError());
  return RARC_OK;
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
template <int MODE>
static int e32_epi(const float* P, const float* ra, const float* rw, const float* bias, const float* resid, const float* gamma,
                   const float* beta, float eps, int m, int n, float* out32, uint16_t* out3, float* ra_out, hipStream_t s,
                   int n_parts = 1, const E32Fuse fz = E32Fuse()) {
  const size_t part_stride = (size_t)m * n;
  RARC_REQUIRE(n % 4 == 0 && n <= 4096, RARC_E_UNSUPPORTED, "fp32-class encoder: row length %d (need a multiple of 4, <= 4096)", n);
  if (n <= 1024)
    hipLaunchKernelGGL((rarc_e32_epi_kernel<1, MODE>), dim3(m), dim3(256), 0, s, P, ra, rw, bias, resid, gamma, beta, eps, n,
                       out32, (half_t*)out3, ra_out, n_parts, part_stride, fz);
  else
    hipLaunchKernelGGL((rarc_e32_epi_kernel<4, MODE>), dim3(m), dim3(256), 0, s, P, ra, rw, bias, resid, gamma, beta, eps, n,
                       out32, (half_t*)out3, ra_out, n_parts, part_stride, fz);
  RARC_HIP_CHECK(hipGetLastError());
  return RARC_OK;
}

extern "C" int rarc_enc32_split_weight(const float* d_w, int n, int k, uint16_t* d_w3, float* d_rw, void* stream) {
  RARC_REQUIRE(d_w && d_w3 && d_rw && n > 0 && k > 0, RARC_E_INVALID, "rarc_enc32_split_weight: bad arguments");
  hipLaunchKernelGGL(rarc_e32_split_weight_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, d_w, k, (half_t*)d_w3, d_rw);
  RARC_HIP_CHECK(hipGetLastError());
  return RARC_OK;
}

extern "C" int rarc_enc32_split_rows(const float* d_x, int m, int k, uint16_t* d_a3, float* d_ra, void* stream) {
  RARC_REQUIRE(d_x && d_a3 && d_ra && m > 0 && k > 0, RARC_E_INVALID, "rarc_enc32_split_rows: bad arguments");
  return e32_epi<3>(d_x, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0.f, m, k, nullptr, d_a3, d_ra, (hipStream_t)stream);
}

extern "C" int rarc_enc32_gemm(const uint16_t* d_a3, const float* d_ra, const uint16_t* d_w3, const float* d_rw,
                               const float* d_bias, float* d_c, int m, int n, int k, void* stream) {
  RARC_REQUIRE(d_a3 && d_ra && d_w3 && d_rw && d_bias && d_c, RARC_E_INVALID, "rarc_enc32_gemm: null pointer");
  RARC_REQUIRE(m > 0 && n > 0 && k > 0 && m % 128 == 0 && n % 128 == 0 && k % 64 == 0, RARC_E_UNSUPPORTED,
               "rarc_enc32_gemm: need M, N multiples of 128 and K a multiple of 64 (got %d, %d, %d)", m, n, k);
  if (int rc = rarc_gemm_f16_f32out(d_a3, d_w3, d_c, m, n, 3 * k, (hipStream_t)stream)) return rc;
  return e32_epi<0>(d_c, d_ra, d_rw, d_bias, nullptr, nullptr, nullptr, 0.f, m, n, d_c, nullptr, nullptr, (hipStream_t)stream);
}

static inline size_t e32_align(size_t v) { return (v + 255) & ~(size_t)255; }
static inline int e32_p_slabs(size_t m) { return m <= E32Q_MAX_TOKENS ? E32Q_MAX_PARTS : (m <= 2048 ? 4 : 1); }   // [M][wide] slabs of the product buffer

// does the query path take this forward?  At most 128 tokens in whole 32-token blocks (32 / 64 / 128), every layer with its
// fragment-major weight images; RARC_E32_QUERY=0 keeps the tile kernels (A/B, and the comparison in tests).
static bool e32_query_takes(const RarcEnc32Model* model, long long m) {
  const char* e = getenv("RARC_E32_QUERY");   // (read per call: tests switch it between forwards)
  if (e && atoi(e) == 0) return false;
  if (!(m == 32 || m == 64 || m == 128)) return false;
  for (int l = 0; l < model->n_layers; ++l) {
    const RarcEnc32Layer& Ly = model->layers[l];
    if (!Ly.qkv_wq || !Ly.o_wq || !Ly.f1_wq || !Ly.f2_wq) return false;
  }
  return true;
}

extern "C" size_t rarc_enc32_workspace_bytes(int hidden, int inter, int n_tokens) {
  if (hidden <= 0 || inter <= 0 || n_tokens <= 0) return 0;
  const size_t M = (size_t)n_tokens, H = (size_t)hidden, I = (size_t)inter;
  const size_t wide = 3 * H > I ? 3 * H : I;
  return 2 * e32_align(M * H * 4)      // x (residual stream), ctx
         + e32_align(M * 3 * H * 2)    // split image of x / ctx
         + e32_align(M * 3 * I * 2)    // split image of the GELU output
         + e32_align(M * wide * 4 * e32_p_slabs(M))   // raw GEMM products (small batches: split-K partial slabs; the query path's come in up to sixteen)
         + 2 * e32_align(M * 4)        // row scales
         + e32_align(2 * M * 4)        // fused FFN1: output scales and their inverses
         + e32_align(M * 16 * 4);      // query path: the attention output's inverse scale per (token, head)
}

extern "C" int rarc_enc32_forward(const RarcEnc32Model* model, const int32_t* d_ids, const int32_t* d_lens, int n_seq,
                                  int seq_len, int normalize, void* d_ws, size_t ws_bytes, float* d_out, void* stream) {
  RARC_RANGE();
  RARC_REQUIRE(model && model->layers && d_ids && d_lens && d_ws && d_out, RARC_E_INVALID, "rarc_enc32_forward: null pointer");
  const int H = model->hidden, I = model->inter;
  RARC_REQUIRE(n_seq > 0 && seq_len > 0 && model->n_layers > 0, RARC_E_INVALID, "rarc_enc32_forward: empty batch or model");
  RARC_REQUIRE(model->vocab > 0 && model->max_pos >= seq_len, RARC_E_INVALID,
               "rarc_enc32_forward: model->vocab (%d) must be set and model->max_pos (%d) must cover seq_len (%d)",
               model->vocab, model->max_pos, seq_len);
  RARC_REQUIRE(H % 128 == 0 && H <= 1024 && I % 128 == 0 && I <= 4096 && model->heads > 0 &&
                   (H == model->heads * 64 || H == model->heads * 32) && seq_len <= 512,
               RARC_E_UNSUPPORTED, "rarc_enc32_forward: hidden %d / inter %d / heads %d / seq_len %d not supported", H, I,
               model->heads, seq_len);
  RARC_REQUIRE(!model->rel_bias || model->rel_span >= seq_len, RARC_E_INVALID,
               "rarc_enc32_forward: the relative-position bias spans %d positions, the batch is %d long", model->rel_span, seq_len);
  const long long m_ll = (long long)n_seq * seq_len;
  const bool query = e32_query_takes(model, m_ll);
  RARC_REQUIRE((m_ll % 128 == 0 || query) && m_ll < (1ll << 31), RARC_E_UNSUPPORTED,
               "rarc_enc32_forward: n_seq*seq_len must be a multiple of 128 — or 32 / 64 when every layer carries its query-path "
               "weight images (got %lld)", m_ll);
  const int M = (int)m_ll;
  RARC_REQUIRE(ws_bytes >= rarc_enc32_workspace_bytes(H, I, M), RARC_E_INVALID, "rarc_enc32_forward: workspace too small");
  hipStream_t hs = (hipStream_t)stream;
  char* w = (char*)d_ws;
  const size_t wide = (size_t)(3 * H > I ? 3 * H : I);
  float* x = (float*)w;                 w += e32_align((size_t)M * H * 4);
  float* ctx = (float*)w;               w += e32_align((size_t)M * H * 4);
  uint16_t* xs = (uint16_t*)w;          w += e32_align((size_t)M * 3 * H * 2);
  uint16_t* mids = (uint16_t*)w;        w += e32_align((size_t)M * 3 * I * 2);
  const int p_slabs = e32_p_slabs((size_t)M);                // [M][wide] slabs the product buffer holds
  float* P = (float*)w;                 w += e32_align((size_t)M * wide * 4 * p_slabs);
  float* ra_a = (float*)w;              w += e32_align((size_t)M * 4);
  float* ra_b = (float*)w;              w += e32_align((size_t)M * 4);
  float* sg = (float*)w;                w += e32_align((size_t)2 * M * 4);     // fused FFN1: [M] scales | [M] inverses
  float* rah = (float*)w;                                                      // query path: [M][heads] inverse scales of the attention output
  const float eps = model->ln_eps;

  hipLaunchKernelGGL(rarc_e32_embed_kernel, dim3(M), dim3(256), 0, hs, d_ids, model->word, model->pos, model->type0,
                     model->emb_g, model->emb_b, eps, seq_len, H, model->vocab, x, (half_t*)xs, ra_a, query);
  RARC_HIP_CHECK(hipGetLastError());
  const int q_blocks = (seq_len + 31) / 32;
  const int n_units = n_seq * model->heads * q_blocks;  // one wave each, four per workgroup
  const int max_parts = (int)(p_slabs * wide / (size_t)H);   // partial slabs of an [M][H] product that fit the buffer P
  const int max_parts_i = (int)(p_slabs * wide / (size_t)I);  // ... of an [M][I] product
  const int max_parts_qkv = (int)(p_slabs * wide / (size_t)(3 * H));   // ... of the [M][3H] q|k|v product
  int rc = RARC_OK;
  // head_dim 64: attention on the fp16 MFMA over split operands (round 4); RARC_E32_ATTN=mfma32 keeps the fp32-MFMA kernel (A/B)
  const char* attn_env = getenv("RARC_E32_ATTN");   // (read per call: tests switch it between forwards)
  const bool split_attention = !(attn_env && !strcmp(attn_env, "mfma32"));
  // FFN1's GELU in the GEMM epilogue: big batches (tile kernels), and the query path's weight stream
  const bool fuse_shape = query ? e32_skinny_gelu_takes(M, H) : rarc_gemm_f16_gelu_split_takes(M, I, 3 * H);
  E32Fuse fq;            // what every row pass is told: where the split image goes (query path: fragment-major)
  fq.frag = query;
  // one projection: the tile kernels over the [hi | lo | hi] rows, or the query path's weight stream over the fragment images
#define E32_PROJ(XS, W3, WQ, NN, KK, MAXP)                                                                       \
  (query ? e32_skinny_gemm(XS, WQ, P, M, NN, KK, &parts, hs) : rarc_gemm_f16_f32out_parts(XS, W3, P, M, NN, 3 * (KK), MAXP, &parts, hs))
  for (int l = 0; l < model->n_layers; ++l) {
    const RarcEnc32Layer& Ly = model->layers[l];
    // fused q|k|v projection; its scales and bias are applied by the attention kernel's loads
    int parts = 1;   // (small batches: split-K into partial slabs that the consumer — attention loads, epilogue kernels — sums)
    if ((rc = E32_PROJ(xs, Ly.qkv_w3, Ly.qkv_wq, 3 * H, H, max_parts_qkv)) != RARC_OK) return rc;
    const int qkv_parts = parts;
#define E32_ATTN_LAUNCH(DHV, RELV)                                                                                     \
    hipLaunchKernelGGL((rarc_e32_attention_kernel<DHV, RELV>), dim3((n_units + 3) / 4), dim3(256), 0, hs, P, ra_a, Ly.qkv_rw, \
                       Ly.qkv_b, d_lens, seq_len, H, model->heads, q_blocks, n_units, ctx, model->rel_bias, model->rel_span,     \
                       qkv_parts, (size_t)M * 3 * H)
    // the query path's own attention (sequences of 32 padded tokens, head_dim 64): one wave per (sequence, head), the split
    // image of its output written in place of the row pass below; RARC_E32Q_ATTN=0 keeps the document kernel + the row pass
    const char* qa_env = getenv("RARC_E32Q_ATTN");
    const int head_dim = H / model->heads;
    const bool query_attn = query && seq_len == 32 && model->heads <= 16 && !(qa_env && atoi(qa_env) == 0);
    if (query_attn) {
#define E32Q_ATTN_LAUNCH(DHV, RELV)                                                                                          \
      hipLaunchKernelGGL((rarc_e32q_attention_kernel<DHV, RELV>), dim3(n_seq * model->heads), dim3(256), 0, hs, P, ra_a,       \
                         Ly.qkv_rw, Ly.qkv_b, d_lens, H, model->heads, (half_t*)xs, rah, ra_b, model->rel_bias, model->rel_span, \
                         qkv_parts, (size_t)M * 3 * H)
      if (head_dim == 64) { if (model->rel_bias) E32Q_ATTN_LAUNCH(64, true); else E32Q_ATTN_LAUNCH(64, false); }
      else { if (model->rel_bias) E32Q_ATTN_LAUNCH(32, true); else E32Q_ATTN_LAUNCH(32, false); }
#undef E32Q_ATTN_LAUNCH
    } else if (H == model->heads * 64 && split_attention) {
      const int q_groups = (q_blocks + 3) / 4;
#define E32_SPLIT_LAUNCH(RELV)                                                                                               \
      hipLaunchKernelGGL((rarc_e32_attention_split_kernel<RELV>), dim3(n_seq * model->heads * q_groups), dim3(256), 0, hs, P, ra_a, \
                         Ly.qkv_rw, Ly.qkv_b, d_lens, seq_len, H, model->heads, q_blocks, q_groups, ctx, model->rel_bias,          \
                         model->rel_span, qkv_parts, (size_t)M * 3 * H)
      if (model->rel_bias) E32_SPLIT_LAUNCH(true); else E32_SPLIT_LAUNCH(false);
#undef E32_SPLIT_LAUNCH
    } else if (H == model->heads * 64) {
      if (model->rel_bias) E32_ATTN_LAUNCH(64, true); else E32_ATTN_LAUNCH(64, false);
    } else {
      if (model->rel_bias) E32_ATTN_LAUNCH(32, true); else E32_ATTN_LAUNCH(32, false);
    }
#undef E32_ATTN_LAUNCH
    RARC_HIP_CHECK(hipGetLastError());
    const bool fuse_gelu = fuse_shape && Ly.f1_colmax != nullptr;
    E32Fuse fz2 = fq;
    if (fuse_gelu) { fz2.colmax = Ly.f1_colmax; fz2.k = H; fz2.sg_out = sg; fz2.sg_rows = (size_t)M; }
    if (!query_attn && (rc = e32_epi<3>(ctx, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0.f, M, H, nullptr, xs, ra_b, hs, 1, fq)))
      return rc;
    // attention output projection -> x = LayerNorm(proj + x)
    if (query_attn) rc = e32_skinny_gemm(xs, Ly.o_wq, P, M, H, H, &parts, hs, rah, head_dim / 16);   // (per-head scales: a head = 4 or 2 k steps)
    else rc = E32_PROJ(xs, Ly.o_w3, Ly.o_wq, H, H, max_parts);
    if (rc != RARC_OK) return rc;
    if ((rc = e32_epi<2>(P, ra_b, Ly.o_rw, Ly.o_b, x, Ly.ln1_g, Ly.ln1_b, eps, M, H, x, xs, ra_a, hs, parts, fz2))) return rc;
    // FFN: the first projection with bias + GELU + split fused into its epilogue where the shape allows (big batches),
    // else the fp32 product and a row pass
    const float* ra_f2 = ra_b;
    if (fuse_gelu && query) {
      if ((rc = e32_skinny_gemm_gelu(xs, Ly.f1_wq, ra_a, Ly.f1_rw, Ly.f1_b, sg, mids, M, I, H, hs)) != RARC_OK) return rc;
      ra_f2 = sg + M;
    } else if (fuse_gelu) {
      if ((rc = rarc_gemm_f16_gelu_split(xs, Ly.f1_w3, ra_a, Ly.f1_rw, Ly.f1_b, sg, mids, M, I, 3 * H, hs)) != RARC_OK)
        return rc == 1 ? RARC_E_INVALID : rc;   // (the shape was asked about above)
      ra_f2 = sg + M;
    } else {
      if ((rc = E32_PROJ(xs, Ly.f1_w3, Ly.f1_wq, I, H, max_parts_i)) != RARC_OK) return rc;
      if ((rc = e32_epi<1>(P, ra_a, Ly.f1_rw, Ly.f1_b, nullptr, nullptr, nullptr, 0.f, M, I, nullptr, mids, ra_b, hs, parts, fq))) return rc;
    }
    if ((rc = E32_PROJ(mids, Ly.f2_w3, Ly.f2_wq, H, I, max_parts)) != RARC_OK) return rc;
    if ((rc = e32_epi<2>(P, ra_f2, Ly.f2_rw, Ly.f2_b, x, Ly.ln2_g, Ly.ln2_b, eps, M, H, x, xs, ra_a, hs, parts, fq))) return rc;
  }
#undef E32_PROJ
  hipLaunchKernelGGL(rarc_e32_pool_kernel, dim3(n_seq), dim3(256), 0, hs, x, d_lens, seq_len, H, (normalize & 2) ? 1 : 0,
                     normalize & 1, d_out);
  RARC_HIP_CHECK(hipGetLastError());
  return RARC_OK;
}
