// encoder.hip — BERT-family encoder forward (the embedding provider's hot loop).
//
// Replaces what HuggingFaceEmbeddings reaches through sentence-transformers:
//   core/file_management/embeddings/huggingface.py:122-126 (SentenceTransformer.encode)
//   [external] tokenizer -> BERT forward -> CLS pooling -> optional L2 normalise -> fp32
// Token ids in, embeddings out (tokenisation stays on the host; no vocab ships offline).
//
// Kernels (fp16 storage, fp32 accumulate / statistics):
//   rarc_enc_embed_ln   word + position + type embeddings -> LayerNorm            (HBM-bound)
//   rarc_enc_gemm       C = A·Wᵀ + bias [, GELU]   A [M][K], W [N][K] (torch Linear layout); by shape:
//                       256x256x64 / 256x128x64 ping-pong kernels (two wave groups alternate load and MFMA
//                       slots, LDS-DMA staged 5-6 phases ahead with counted waits), or the 128/256x128 kernel
//                       (2-4 stage pipeline, split-K option) for small batches              (MFMA-bound)
//   rarc_enc_attention  softmax(Q·Kᵀ/sqrt(dh) + mask)·V per (sequence, head), keys streamed in
//                       LDS tiles with an online softmax (sequence lengths <= 512)
//   rarc_enc_add_ln     LayerNorm(x + residual)                                   (HBM-bound)
//   rarc_enc_pool       CLS row -> fp32 [, L2 normalise with the canonical sum order of prep.hip]
#include "rarc_common.h"
#include <cstring>
#include <cstdlib>

// exact-form GELU 0.5·v·(1 + erf(v/√2)) with erf from Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, far below the
// fp16 rounding of the result): 1 - erf(x) = (a1 t + ... + a5 t^5)·exp(-x²), t = 1/(1 + p·x), x >= 0.
// About 16 instructions per element (two of them transcendental) against ~37 for the library erff — the GELU
// epilogue was 28 % of the FFN1 GEMM.
__device__ __forceinline__ float rarc_gelu_erf(float v) {
  const float ax = __builtin_fabsf(v);
  const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f * 0.70710678118654752440f, ax, 1.0f));
  float poly = 1.061405429f;
  poly = __builtin_fmaf(poly, t, -1.453152027f);
  poly = __builtin_fmaf(poly, t, 1.421413741f);
  poly = __builtin_fmaf(poly, t, -0.284496736f);
  poly = __builtin_fmaf(poly, t, 0.254829592f);
  const float e = __builtin_amdgcn_exp2f(v * v * (-0.5f * 1.44269504088896340736f));  // exp(-v²/2)
  const float hr = 0.5f * v * (poly * t * e);  // 0.5·v·(1 - erf(|v|/√2))
  return v >= 0.f ? v - hr : hr;
}

// SwiGLU epilogue (ACT 3; the reranker LM's gate|up GEMM, decoder.hip): the W rows are interleaved in groups of 8 —
// 8 gate rows, then the 8 up rows of the same features — so in every 32 x 32 MFMA block a lane's column groups g = 0, 2
// are gates and g = 1, 3 the matching ups, and the epilogue writes silu(gate)·up straight away: C is [M][N/2].
// Roundings as in the unfused path (GEMM output fp16, silu fp16, product fp16), so both give the same bits.
__device__ __forceinline__ half_t rarc_swiglu_f16(float gate_acc, float up_acc) {
  const float gf = (float)(half_t)gate_acc, uf = (float)(half_t)up_acc;
  const half_t act = (half_t)(gf / (1.0f + __expf(-gf)));
  return (half_t)((float)act * uf);
}

// ------------------------------------------------------------------------------------------
// GEMM  C[M][N] = A[M][K] · W[N][K]ᵀ + bias[N]   (M, N multiples of 128; K multiple of 64)
// ------------------------------------------------------------------------------------------
constexpr int GM = 128, GN = 128, GK = 64;  // smallest tile (M is padded to a multiple of GM by the host)

// blockIdx -> output tile.  Workgroups go to the 8 XCDs round-robin by id, every XCD has its own 4 MB L2, and the
// 32 workgroups resident on an XCD run their k loops roughly in step.  order 0 / 1 (the plain row- or column-major
// walks) hand an XCD 32 tiles that share ONE operand tile between them: the other operand is fetched 32 times over
// from the fabric, and the GEMMs sat at the ~6.4 TB/s the CUs can pull past their L2s (11 bytes per cycle per CU,
// MFMA pipe busy 38 %).  order 2: XCD x owns a contiguous range of a "grouped" tile order — 8 row-tiles deep, then
// the next column-tile — so its 32 resident tiles form an 8 x 4 block: 12 operand tiles serve 32 output tiles.
__device__ __forceinline__ void gemm_tile_of(int bid, int grid, int tiles_m, int tiles_n, int order, int& tm, int& tn) {
  if (order != 2) {
    tm = order ? bid / tiles_n : bid % tiles_m;
    tn = order ? bid % tiles_n : bid / tiles_m;
    return;
  }
  const int x = bid & 7, q = grid >> 3, rem = grid & 7;
  const int id = x * q + (x < rem ? x : rem) + (bid >> 3);  // XCD x: ids [x*q + min(x, rem), ... + q + (x < rem))
  constexpr int GROUP_M = 8;
  const int per_group = GROUP_M * tiles_n, g = id / per_group, first_m = g * GROUP_M;
  const int rows = tiles_m - first_m < GROUP_M ? tiles_m - first_m : GROUP_M;
  const int in_group = id - g * per_group;
  tm = first_m + in_group % rows;
  tn = in_group / rows;
}

// BM x 128 x 64 tiles, BM/32 x 2... waves laid out (BM/64) x 2, each computing a 64 x 64 block of C as 2 x 2
// v_mfma_f32_32x32x16_f16.  Operand tiles stream HBM -> LDS by LDS-DMA (8 rows x 128 B per instruction,
// XOR-swizzled on the source side: conflict-free ds_read_b128 fragments), STAGES deep:
//   BM = 128: 4 waves, 2 stages of 32 KB, two workgroups per CU;
//   BM = 256: 8 waves, 3 stages of 48 KB (two tiles in flight behind the one being multiplied), one
//             workgroup per CU — the W tile feeds twice as many rows, and the prefetch is one tile deeper.
//
// ACT 2 = split-K: blockIdx.y owns the k range [y*K, (y+1)*K) of rows of length ldk and writes its fp32 partial
// products (no bias) to ((float*)C)[y][M][N]; the LayerNorm kernel that follows sums the partials.
template <int ACT, int BM, int STAGES>  // ACT 0 = bias only, 1 = bias + exact (erf) GELU
__global__ __launch_bounds__(BM * 2, BM == 128 ? 2 : 1) void rarc_gemm_f16_kernel(const half_t* __restrict__ A,
                                                                                  const half_t* __restrict__ W,
                                                                                  const half_t* __restrict__ bias,
                                                                                  half_t* __restrict__ C, int M,
                                                                                  int N, int K, int ldk, int order) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [STAGES][A BM*128 B | W 16 KiB]
  constexpr int NW = BM / 32;                   // waves
  constexpr int A_BYTES = BM * GK * 2, W_BYTES = GN * GK * 2, ST_BYTES = A_BYTES + W_BYTES;
  constexpr int A_DPW = (BM / 8) / NW, W_DPW = (GN / 8) / NW;  // 1-KiB DMA instructions per wave per tile
  constexpr int DPW = A_DPW + W_DPW;
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int row = lane & 31, h = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  // XCD-aware tile order.  Workgroup ids go round-robin over the 8 XCDs (each with its own L2), so the operand
  // indexed by (id % tiles) is read once per XCD that needs it and the OTHER operand by all eight:
  //   order 0: tm = id % tiles_m  -> an XCD keeps a few row-blocks of A, every W tile crosses the fabric 8 times
  //   order 1: tn = id % tiles_n  -> an XCD keeps a few W tiles, A crosses 8 times   (M < N: small batches)
  const int tiles_m = M / BM, tiles_n = N / GN;
  const int bid = blockIdx.x;
  int tm, tn;
  gemm_tile_of(bid, (int)gridDim.x, tiles_m, tiles_n, order, tm, tn);
  const half_t* Ab = A + (size_t)tm * BM * ldk + (ACT == 2 ? (size_t)blockIdx.y * K : 0);
  const half_t* Wb = W + (size_t)tn * GN * ldk + (ACT == 2 ? (size_t)blockIdx.y * K : 0);

  const int drow = lane >> 3, dslot = lane & 7;
  auto issue = [&](int stage, int kt) {
#pragma unroll
    for (int j = 0; j < A_DPW; ++j) {
      const int i = wave * A_DPW + j;  // row-block of 8 rows: rows 8i .. 8i+7
      const int r = 8 * i + drow;
      const int c = dslot ^ ((r >> 1) & 7);
      __builtin_amdgcn_global_load_lds(RARC_GPTR(Ab + (size_t)r * ldk + kt * GK + c * 8),
                                       RARC_LPTR(smem + stage * ST_BYTES + i * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < W_DPW; ++j) {
      const int i = wave * W_DPW + j;
      const int r = 8 * i + drow;
      const int c = dslot ^ ((r >> 1) & 7);
      __builtin_amdgcn_global_load_lds(RARC_GPTR(Wb + (size_t)r * ldk + kt * GK + c * 8),
                                       RARC_LPTR(smem + stage * ST_BYTES + A_BYTES + i * 1024), 16, 0, 0);
    }
  };
  // fragment offsets inside a 32-row group (4 KiB): row*128 + ((chunk ^ sw) << 4)
  const int sw = (row >> 1) & 7;
  int xk[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) xk[kk] = row * 128 + (((2 * kk + h) ^ sw) << 4);

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (f32x16){0};

  const int KT = K / GK;
#pragma unroll
  for (int s0 = 0; s0 < STAGES - 1; ++s0)
    if (s0 < KT) issue(s0, s0);
  // fragments double-buffered in registers: the four ds_read_b128 of k-step kk+1 are in flight while the
  // four MFMAs of k-step kk run (in-order LDS returns: "at most 4 outstanding" == step kk has landed)
  half8 fa0[2], fa1[2], fb0[2], fb1[2];
#define GEMM_LDFRAG(BUF, KK)                                                                          \
  asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:4096\n\t"                          \
               "ds_read_b128 %2, %5\n\tds_read_b128 %3, %5 offset:4096"                               \
               : "=&v"(fa0[BUF]), "=&v"(fa1[BUF]), "=&v"(fb0[BUF]), "=&v"(fb1[BUF])                   \
               : "v"(abase + xk[KK]), "v"(wbase + xk[KK])                                             \
               : "memory")
#define GEMM_WAIT(BUF, N)                                                                             \
  asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(fa0[BUF]), "+v"(fa1[BUF]), "+v"(fb0[BUF]), "+v"(fb1[BUF]))
  int st = 0;
  for (int kt = 0; kt < KT; ++kt) {
    // tile kt has landed when at most the DMA of the (STAGES-2) newer tiles is outstanding
    if (STAGES > 2 && kt + STAGES - 2 < KT) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((STAGES - 2) * DPW) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    {  // refill the stage consumed in the previous iteration
      int ns = st + STAGES - 1;
      if (ns >= STAGES) ns -= STAGES;
      if (kt + STAGES - 1 < KT) issue(ns, kt + STAGES - 1);
    }
    const int abase = st * ST_BYTES + (2 * wm) * 4096;
    const int wbase = st * ST_BYTES + A_BYTES + (2 * wn) * 4096;
    GEMM_LDFRAG(0, 0);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int cur = kk & 1;
      if (kk == 0) { GEMM_LDFRAG(1, 1); GEMM_WAIT(0, 4); }
      else if (kk == 1) { GEMM_LDFRAG(0, 2); GEMM_WAIT(1, 4); }
      else if (kk == 2) { GEMM_LDFRAG(1, 3); GEMM_WAIT(0, 4); }
      else { GEMM_WAIT(1, 0); }
      // swapped roles: the W fragment is the A operand, so a lane ends up with ONE row m of C and 16 of
      // its columns n — four consecutive n per register quad: 8-byte stores instead of 2-byte ones
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb0[cur], fa0[cur], acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb1[cur], fa0[cur], acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb0[cur], fa1[cur], acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb1[cur], fa1[cur], acc[1][1], 0, 0, 0);
    }
    if (++st == STAGES) st = 0;
  }
#undef GEMM_LDFRAG
#undef GEMM_WAIT
  // epilogue: lane holds row m = lane & 31 of each 32x32 block and columns n = 8*(r>>2) + 4*h + (r&3) — its
  // pieces lie in 32 different rows of C, so a direct store touches 32 cache lines per instruction with 16-32
  // bytes each.  Each wave transposes its 64 x 64 block through LDS (the pipeline buffers are dead once every
  // wave is past the barrier below) and writes whole rows: 128 B (fp16) or 256 B (fp32 split-K partials).
  typedef _Float16 half4 __attribute__((ext_vector_type(4)));
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  if (ACT == 2) {
    constexpr int ST = 64 * 4 + 16;
    char* ep = smem + wave * 64 * ST;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int jj = 0; jj < 2; ++jj)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *(float4*)(ep + (i * 32 + row) * ST + (jj * 32 + 8 * g + 4 * h) * 4) =
              make_float4(acc[i][jj][4 * g], acc[i][jj][4 * g + 1], acc[i][jj][4 * g + 2], acc[i][jj][4 * g + 3]);
    __builtin_amdgcn_wave_barrier();
    const int r4 = lane >> 4, c = lane & 15;
    float* P = (float*)C + ((size_t)blockIdx.y * M + tm * BM + wm * 64) * N + tn * GN + wn * 64 + c * 4;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const int r = t * 4 + r4;
      *(float4*)(P + (size_t)r * N) = *(const float4*)(ep + r * ST + c * 16);
    }
  } else {
    constexpr int ST = 64 * 2 + 16;
    char* ep = smem + wave * 64 * ST;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int nl = jj * 32 + 8 * g + 4 * h;
          const half4 b4 = *(const half4*)(bias + tn * GN + wn * 64 + nl);
          half4 out;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float v = acc[i][jj][4 * g + e] + (float)b4[e];
            if (ACT == 1) v = rarc_gelu_erf(v);
            out[e] = (half_t)v;
          }
          *(half4*)(ep + (i * 32 + row) * ST + nl * 2) = out;
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
    const int r8 = lane >> 3, c = lane & 7;
    half_t* Cw = C + (size_t)(tm * BM + wm * 64) * N + tn * GN + wn * 64 + c * 8;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int r = t * 8 + r8;
      *(uint4*)(Cw + (size_t)r * N) = *(const uint4*)(ep + r * ST + c * 16);
    }
  }
}

// ---- epilogue variants of the reranker LM (round 3: RMSNorm folded into its neighbours) ----------------------
// ACT = base (0 none, 1 GELU, 3 SwiGLU, 4 fp32 out) | 16 (RS) | 32 (RES):
//   RES  the output joins the residual stream in place: C = fp16(C + fp16(A·Wᵀ)) — the fp16 tensor add the RMSNorm kernel
//        used to do on its way in;
//   RS   per-row scale: the kernel's `bias` argument is `const float* rowscale` [M] and C = act(rowscale[m] · A·Wᵀ) — with
//        W's columns pre-multiplied by the norm weight this is the projection of RMSNorm(x) computed from x itself.
// (RES: the old values are fetched at the START of the epilogue, `old`, so that their latency hides under the conversion and
//  the LDS transposition — read inside the store loop every piece waited for its own round trip: +7 % on these GEMMs)
__device__ __forceinline__ void gemm_store8(half_t* p, const uint4 v, const bool res, const half8 old) {
  if (res) *(half8*)p = old + __builtin_bit_cast(half8, v);
  else *(uint4*)p = v;
}
// RES with statistics: additionally the sum of squares of the 32-column segment this lane's 8 values belong to (4 lanes
// x 8 columns, fp32, fixed order) goes to ssq[row * (N / 32) + segment] — the RMSNorm that follows needs mean(x²) per row
// and would otherwise read the whole stream again for it (rarc_lm_rowscale_kernel sums the N / 32 partials instead).
__device__ __forceinline__ void gemm_store8_ss(half_t* p, const uint4 v, const half8 old, float* ssq, size_t slot, int lane) {
  const half8 nv = old + __builtin_bit_cast(half8, v);
  *(half8*)p = nv;
  float ss = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) ss = __builtin_fmaf((float)nv[e], (float)nv[e], ss);
  ss += __shfl_xor(ss, 1, 64);
  ss += __shfl_xor(ss, 2, 64);
  if ((lane & 3) == 0) ssq[slot] = ss;
}

// ------------------------------------------------------------------------------------------
// 256 x 256 x 64 tiles for the large GEMMs (QKV, FFN1 at M >= 8192): half the L2 -> LDS bytes per flop of the
// 256 x 128 kernel above, and an explicit ping-pong between the two waves of every SIMD.
//   8 waves = 2 (M) x 4 (N); wave (wr, wc) owns rows wr*128.. x cols wc*64.. of the tile: acc[8][4] blocks of 16 x 16
//   (v_mfma_f32_16x16x32_f16; rounds 1-2: acc[4][2] blocks of 32 x 32).
//   LDS 128 KiB = 2 parities x {A0, A1, B0, B1} x 16 KiB.  A "half" h of an operand holds, for every wave, half of
//   ITS rows: A_h = rows {wr*128 + h*64 + r}, B_h = cols {wc*64 + h*32 + r}.  A K tile is four slots — load A0, B0, B1 |
//   32 MFMAs (A0 x B0, A0 x B1) | load A1 | 32 MFMAs (A1 x B1, A1 x B0) — with the next tiles' halves restaged in the tails
//   of the two matrix slots; waves 4-7 run one slot behind waves 0-3: while one of a SIMD's two waves multiplies, the
//   other loads.  The slot structure, the staging schedule and the RAW / WAR argument are written out at G256_TILE2 below.
// ------------------------------------------------------------------------------------------
constexpr int G256_EP_STRIDE = 144, G256_EP_BYTES = 128 * G256_EP_STRIDE;  // epilogue staging, per wave
constexpr int G256_LDS = 8 * G256_EP_BYTES > 131072 ? 8 * G256_EP_BYTES : 131072;
// ACT 5 (round 4) — the fp32-class encoder's FFN1 with its activation fused (encoder_f32.hip): A and W are split images
// ([lo|hi|hi] x [hi|lo|hi], K' = 3K), the fp32 accumulator becomes v = acc·ra[row]·rw[col] + bias[col] (the exact expression of
// rarc_e32_epi_kernel), g = 0.5 v (1 + erff(v/sqrt 2)) with the library erff, and g·sg[row] is written straight back as the
// split image [lo | hi | hi] of the NEXT GEMM's A operand (row stride 3N halves) — no fp32 product matrix, no second pass.
// sg[row] is a power of two chosen BEFORE the GEMM from a bound on the row (|g| <= |v| <= ||x|| ||W_j|| + |b_j|), see
// encoder_f32.hip; ra, rw are powers of two as well, so the only roundings are the bias add, the GELU and the two halves.
struct GemmSplitEpi {
  const float* ra = nullptr;     // [M] inverse scales of the A rows
  const float* rw = nullptr;     // [N] inverse scales of the W rows
  const float* bias = nullptr;   // [N] fp32
  const float* sg = nullptr;     // [M] output scales
  // ACT 6 (wide.hip, round 5): no C at all — the fp32 product of row r and column q is a SCORE, compared with thr[q]; what
  // reaches it goes to query q's candidate list as (ordered score key << 32 | ~row).  N = 256 (one column tile).
  const float* thr = nullptr;            // [N]
  unsigned long long* cand = nullptr;    // [N][cap]
  uint32_t* count = nullptr;             // [N]
  uint32_t* status = nullptr;            // [N]
  uint32_t cap = 0, row0 = 0, n_valid = 0;
  // a query's list is `shards` (1 or 8) sub-lists of cap / shards entries with a counter each (count is [N][shards]); a workgroup
  // nominates into sub-list blockIdx.x % shards — 1000 workgroups adding to ONE counter per query serialise on it (wide.hip)
  uint32_t shards = 1;
};
__device__ __forceinline__ float gemm_gelu_libm(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f)); }

template <int ACT>
__global__ __launch_bounds__(512, 1) void rarc_gemm256_f16_kernel(const half_t* __restrict__ A,
                                                                  const half_t* __restrict__ W,
                                                                  const half_t* __restrict__ bias,
                                                                  half_t* __restrict__ C, int M, int N, int K,
                                                                  int order, float* __restrict__ ssq = nullptr,
                                                                  const GemmSplitEpi fx = GemmSplitEpi()) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int row = lane & 31, hh = lane >> 5;
  const int wr = wave >> 2, wc = wave & 3;
  const int tiles_m = M / 256, tiles_n = N / 256, n_tiles = tiles_m * tiles_n;
  // persistent: a workgroup walks tiles blockIdx.x, + gridDim.x, ... (gridDim.x a multiple of 8, so all of them sit
  // on its own XCD's share of the tile order).  The stores of one tile's epilogue drain while the next tile's
  // first operand tiles are fetched, instead of holding the CU until they are acknowledged.
  const int drow = lane >> 3, dslot = lane & 7;
  uint32_t soff[4][2];
#pragma unroll
  for (int which = 0; which < 4; ++which)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int lr = 8 * (wave * 2 + j) + drow;  // row inside the 128-row half-tile
      const int c = dslot ^ ((lr >> 1) & 7);     // XOR swizzle on the source side
      const int h = which & 1;
      const int trow = which < 2 ? (lr >> 6) * 128 + h * 64 + (lr & 63) : (lr >> 5) * 64 + h * 32 + (lr & 31);
      soff[which][j] = ((uint32_t)trow * (uint32_t)K + (uint32_t)c * 8u) * 2u;
    }
  for (int bid = blockIdx.x; bid < n_tiles; bid += gridDim.x) {
  int tm, tn;
  gemm_tile_of(bid, n_tiles, tiles_m, tiles_n, order, tm, tn);
  const half_t* Ab = A + (size_t)tm * 256 * K;
  const half_t* Wb = W + (size_t)tn * 256 * K;
  const int KT = K / GK;
  // (the eight offsets stay eight 32-bit registers: made opaque once per tile, or hipcc folds them into 64-bit per-lane
  //  pointers `A + soff` outside the tile loop — sixteen registers, of which the ACT 96 instantiation spilled fourteen and
  //  reloaded them, with vector-memory scratch loads, in front of every tile)
#pragma unroll
  for (int which = 0; which < 4; ++which) asm volatile("" : "+v"(soff[which][0]), "+v"(soff[which][1]));

  // staging: half-tile `which` (0 A0, 1 A1, 2 B0, 3 B1) of k tile kt into parity par; this wave's two KiB of it.
  // The lane's source offsets inside a tile fit 32 bits (256 rows x K halves) and do not depend on the tile: eight
  // registers for the whole kernel, added to the uniform tile base (64-bit pairs here spilled once the tile loop came).
  auto stage = [&](int par, int which, int kt) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int i = wave * 2 + j;      // 8-row block of the 128-row half-tile
      const char* base = (const char*)(which < 2 ? Ab : Wb) + (size_t)kt * (GK * 2);
      __builtin_amdgcn_global_load_lds(RARC_GPTR(base + soff[which][j]),
                                       RARC_LPTR(smem + par * 65536 + which * 16384 + i * 1024), 16, 0, 0);
    }
  };
  // Fragments for v_mfma_f32_16x16x32_f16 (round 3; rounds 1-2 ran 32x32x16): lane reads row (lane & 15) of a 16-row block and the
  // 16-byte chunk 4 s + (lane >> 4) of the row's eight, s = the k step of 32 inside the k tile.  Same LDS image and swizzle (a 16-lane
  // group covers 16 distinct slots: 8 (r & 1) + (c ^ (r >> 1))); (4 s + g) ^ sw = (g ^ sw) ^ 4 s, so step 1 is step 0's address ^ 64;
  // the 16-row blocks of a half are 2 KiB apart (offset field).  Why the smaller shape: per MAC it moves half the accumulator
  // bytes (4 B read + written per 32 MACs instead of per 16) for twice the operand bytes — 0.5 B instead of 0.625 B of register
  // traffic per MAC — and the LM / encoder GEMMs run at the board's power limit: the same schedule with the MFMAs swapped for
  // pairs of 16x16x32 (wrong results) ran 6-9 % faster on real operands and no faster on zeros (profiles/r03_gemm_seamless.txt);
  // the vendor's kernel for these shapes is MI16x16 as well.
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  int xs[2];
  {
    const int r16 = lane & 15, sw16 = (r16 >> 1) & 7;
    xs[0] = r16 * 128 + (((lane >> 4) ^ sw16) << 4);
    xs[1] = xs[0] ^ 64;
  }

  f32x4 acc[8][4];  // [16-row block of the wave's 128 rows][16-column block of its 64 columns]
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0, 0, 0, 0};
  half8 fa[2][4], fb0[2][2], fb1[2][2];  // [k step of 32][16-row block]: the A half in use (64 rows); both B halves (32 columns each)

#define G256_LOAD_A(PAR, H)                                                                             \
  _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                                    \
    const int ad = xs[ks] + ((PAR) * 65536 + (H) * 16384 + wr * 8192);                                  \
    asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:2048\n\t"                          \
                 "ds_read_b128 %2, %4 offset:4096\n\tds_read_b128 %3, %4 offset:6144"                   \
                 : "=&v"(fa[ks][0]), "=&v"(fa[ks][1]), "=&v"(fa[ks][2]), "=&v"(fa[ks][3]) : "v"(ad) : "memory"); \
  }
#define G256_LOAD_B(PAR, H, FB)                                                                         \
  _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                                    \
    const int ad = xs[ks] + ((PAR) * 65536 + 32768 + (H) * 16384 + wc * 4096);                          \
    asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:2048"                              \
                 : "=&v"(FB[ks][0]), "=&v"(FB[ks][1]) : "v"(ad) : "memory");                            \
  }
  // end of a load slot: counted wait for the staged data the NEXT slot reads, then this slot's own ds_reads
#define G256_WAIT(VM)                                                                                   \
  asm volatile("s_waitcnt vmcnt(" #VM ")\n\ts_waitcnt lgkmcnt(0)"                                       \
               : "+v"(fa[0][0]), "+v"(fa[0][1]), "+v"(fa[0][2]), "+v"(fa[0][3]), "+v"(fa[1][0]),        \
                 "+v"(fa[1][1]), "+v"(fa[1][2]), "+v"(fa[1][3]), "+v"(fb0[0][0]), "+v"(fb0[0][1]),      \
                 "+v"(fb0[1][0]), "+v"(fb0[1][1]), "+v"(fb1[0][0]), "+v"(fb1[0][1]), "+v"(fb1[1][0]), "+v"(fb1[1][1]) \
               :: "memory")
  // one matrix slot: the A half QM (four 16-row blocks) against both B halves — 2 k steps x 4 x 4 = 32 MFMAs, every accumulator
  // block once per k step (16 MFMAs apart).  Register-only MFMAs drift across s_barrier (they are pure to the optimiser): their
  // inputs are tied to an asm after the slot's first barrier and their results to one before its second.
#define G256_MMA2(QM)                                                                                   \
  asm volatile("" : "+v"(fb0[0][0]), "+v"(fb0[0][1]), "+v"(fb0[1][0]), "+v"(fb0[1][1]), "+v"(fb1[0][0]), "+v"(fb1[0][1]), \
                    "+v"(fb1[1][0]), "+v"(fb1[1][1]));                                                  \
  __builtin_amdgcn_sched_barrier(0);                                                                    \
  __builtin_amdgcn_s_setprio(1);                                                                        \
  _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                      \
    _Pragma("unroll") for (int b = 0; b < 4; ++b) {                                                     \
      acc[4 * (QM) + b][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb0[ks][0], fa[ks][b], acc[4 * (QM) + b][0], 0, 0, 0); \
      acc[4 * (QM) + b][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb0[ks][1], fa[ks][b], acc[4 * (QM) + b][1], 0, 0, 0); \
      acc[4 * (QM) + b][2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb1[ks][0], fa[ks][b], acc[4 * (QM) + b][2], 0, 0, 0); \
      acc[4 * (QM) + b][3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb1[ks][1], fa[ks][b], acc[4 * (QM) + b][3], 0, 0, 0); \
    }                                                                                                   \
  __builtin_amdgcn_s_setprio(0);                                                                        \
  _Pragma("unroll") for (int b = 0; b < 4; ++b)                                                         \
    asm volatile("" : "+v"(acc[4 * (QM) + b][0]), "+v"(acc[4 * (QM) + b][1]), "+v"(acc[4 * (QM) + b][2]), "+v"(acc[4 * (QM) + b][3])); \
  __builtin_amdgcn_sched_barrier(0)
#define G256_BAR() do { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } while (0)
  // One k tile t (parity par) = four slots, two per wave group and A half; waves 4-7 run one slot behind waves 0-3, so that
  // while one of a SIMD's two waves multiplies, the other loads:
  //   slot L_A: fragments of A0, B0, B1   M_A: 32 MFMAs, then stage A1 of tile t+1 (other parity; last read in L_B(t-1))
  //   slot L_B: fragments of A1           M_B: 32 MFMAs, then stage A0, B0, B1 of tile t+2 (this parity; last read in L_A(t))
  // (rounds 1-2 and most of round 3 ran eight slots of 8 MFMAs of 32x32x16 — one output quadrant each; with 4 slots and 4
  // barriers K = 256 / 512 gained 7-10 %, larger K nothing.)  The DMA of a slot is issued in the TAIL of its matrix slot, not
  // next to the fragment reads (2864 -> 2738 cycles per k tile when that was introduced).
  // Ordering (every wave passes every barrier):
  //   WAR  every half is staged two slots or more after its last ds_read, and readers wait for lgkmcnt(0) inside the slot
  //        that issued the read;
  //   RAW  a half is staged five slots before its first read; a wave's queue at the end of L_A holds M_B(t-1)'s six
  //        instructions (vmcnt(6): A1 of this tile, staged before them, has landed), at the end of L_B M_A(t)'s two (vmcnt(2):
  //        the next tile's A0, B0, B1 have landed); every wave waits for its own share, a barrier follows before anyone reads.
#define G256_TILE2(ST_A, ST_B, VMA, VMB)                                                                \
  {                                                                                                     \
    G256_LOAD_A(par, 0) G256_LOAD_B(par, 0, fb0) G256_LOAD_B(par, 1, fb1)                               \
    G256_WAIT(VMA);                                                                                     \
    G256_BAR();                                                                                         \
    G256_MMA2(0);                                                                                       \
    if (ST_A) stage(np, 1, kt + 1);                                                                     \
    G256_BAR();                                                                                         \
    G256_LOAD_A(par, 1)                                                                                 \
    G256_WAIT(VMB);                                                                                     \
    G256_BAR();                                                                                         \
    G256_MMA2(1);                                                                                       \
    if (ST_B) { stage(par, 0, kt + 2); stage(par, 2, kt + 2); stage(par, 3, kt + 2); }                  \
    G256_BAR();                                                                                         \
  }

  // prologue: all of k tile 0, then A0, B0, B1 of tile 1 (its A1 follows in phase 1 of tile 0, as in steady state)
  stage(0, 0, 0); stage(0, 2, 0); stage(0, 3, 0); stage(0, 1, 0);
  if (KT >= 2) {
    stage(1, 0, 1); stage(1, 2, 1); stage(1, 3, 1);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  G256_BAR();
  if (wr == 1) G256_BAR();  // waves 4-7 run one slot behind
  int kt = 0;
  for (; kt + 2 < KT; ++kt) {
    const int par = kt & 1, np = par ^ 1;
    G256_TILE2(true, true, 6, 2)
  }
  if (kt + 1 < KT) {  // second-to-last tile: only the last tile's A1 is still to stage
    const int par = kt & 1, np = par ^ 1;
    G256_TILE2(true, false, 6, 2)
    ++kt;
  }
  // RS (ACT | 16): the eight row scales of this lane's rows are fetched before the last k tile, so that their round trip
  // runs under its MFMAs instead of in front of the epilogue; they are the newest vector-memory operations from here on,
  // and the last tile's waits let exactly them stay outstanding
  float rs_pre[8] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f};
  if constexpr ((ACT & 16) != 0) {
#pragma unroll
    for (int i = 0; i < 8; ++i) rs_pre[i] = ((const float*)bias)[(size_t)tm * 256 + wr * 128 + i * 16 + (lane & 15)];
  }
  {  // last tile
    const int par = kt & 1, np = par ^ 1;
    (void)np;
    if constexpr ((ACT & 16) != 0) { G256_TILE2(false, false, 8, 8) }
    else { G256_TILE2(false, false, 0, 0) }
  }
  if (wr == 0) G256_BAR();
#undef G256_TILE2
#undef G256_MMA2
#undef G256_LOAD_A
#undef G256_LOAD_B
#undef G256_WAIT
#undef G256_BAR
  // epilogue: acc[i][j][e] is row wr*128 + 16 i + (lane & 15), column wc*64 + 16 j + 4 (lane >> 4) + e of the tile.  A lane's
  // 8-byte pieces lie in 16 different rows: stored directly, every instruction would touch 16 cache lines with 32
  // bytes each.  The pipeline buffers are dead now (every wave is past the last barrier), so each wave
  // transposes its 128 x 64 block through its own 18 KiB of LDS (row stride 144 B) and writes whole 128-byte rows.
  typedef _Float16 half4 __attribute__((ext_vector_type(4)));
  // (the lane index goes through an opaque copy: everything the epilogue derives from it would otherwise be hoisted
  //  out of the tile loop and kept in registers through the main loop, which has none to spare)
  int lane_e = lane;
  asm volatile("" : "+v"(lane_e));
  const int row_e = lane_e & 15, q_e = lane_e >> 4;
  char* ep = smem + wave * G256_EP_BYTES;
  constexpr int BASE = ACT & 15;
  constexpr bool RS = (ACT & 16) != 0, RES = (ACT & 32) != 0;
  float rs[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) rs[i] = rs_pre[i];
  // RES: the 16 old 16-byte pieces of the residual stream this lane adds into.  All 16 fetched here are 64 VGPRs next to the
  // 128 accumulator registers still live — 16 of them spilled to scratch (round 3's shipped build).  The first eight
  // go out now (their round trip hides under the conversion + transposition), the other eight once half of the accumulator
  // blocks have been converted and their registers are free (their round trip hides under the rest and the first stores).
  half8 oldv[16];
  const half_t* Co = C + (size_t)(tm * 256 + wr * 128 + (lane_e >> 3)) * N + tn * 256 + wc * 64 + (lane_e & 7) * 8;
  if constexpr (RES) {
#pragma unroll
    for (int t = 0; t < 8; ++t) oldv[t] = *(const half8*)(Co + (size_t)(t * 8) * N);
  }
  if constexpr (BASE == 4) {  // raw fp32 products, no bias (the split-operand GEMMs of encoder_f32.hip): C is float [M][N]
    // the wave's 128 x 64 block goes through its 18 KiB of staging in two 32-column halves of 128-byte rows
#pragma unroll
    for (int j = 0; j < 2; ++j) {
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int g = 0; g < 2; ++g)
          *(float4*)(ep + (i * 16 + row_e) * G256_EP_STRIDE + (16 * g + 4 * q_e) * 4) =
              make_float4(acc[i][2 * j + g][0], acc[i][2 * j + g][1], acc[i][2 * j + g][2], acc[i][2 * j + g][3]);
      __builtin_amdgcn_wave_barrier();
      const int r8 = lane_e >> 3, c = lane_e & 7;
      float* Cw = (float*)C + (size_t)(tm * 256 + wr * 128) * N + tn * 256 + wc * 64 + j * 32 + c * 4;
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const int r = t * 8 + r8;
        *(float4*)(Cw + (size_t)r * N) = *(const float4*)(ep + r * G256_EP_STRIDE + c * 16);
      }
      __builtin_amdgcn_wave_barrier();
    }
  } else if constexpr (BASE == 5) {  // gelu(acc·ra·rw + bias)·sg -> (hi, lo) halves -> split image [lo | hi | hi], row stride 3N
    float ra_r[8], sg_r[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const size_t rg = (size_t)tm * 256 + wr * 128 + i * 16 + row_e;
      ra_r[i] = fx.ra[rg];
      sg_r[i] = fx.sg[rg];
    }
    // the wave's 128 x 64 block in two 32-column halves: a staging row = 64 bytes of hi | 64 bytes of lo
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const int j = 2 * h + jj;
        const int col0 = tn * 256 + wc * 64 + j * 16 + 4 * q_e;
        const float4 w4 = *(const float4*)(fx.rw + col0), b4 = *(const float4*)(fx.bias + col0);
        const float wv[4] = {w4.x, w4.y, w4.z, w4.w}, bv[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          half4 hi4, lo4;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float v = acc[i][j][e] * ra_r[i] * wv[e] + bv[e];
            const float x = gemm_gelu_libm(v) * sg_r[i];
            const half_t hi = (half_t)x;
            hi4[e] = hi;
            lo4[e] = (half_t)(x - (float)hi);
          }
          char* dst = ep + (i * 16 + row_e) * G256_EP_STRIDE + (jj * 16 + 4 * q_e) * 2;
          *(half4*)dst = hi4;
          *(half4*)(dst + 64) = lo4;
        }
      }
      __builtin_amdgcn_wave_barrier();
      const int r8 = lane_e >> 3, c = lane_e & 7;
      const size_t ld3 = (size_t)3 * N;
      half_t* Cw = C + (size_t)(tm * 256 + wr * 128) * ld3 + tn * 256 + wc * 64 + h * 32;
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const int r = t * 8 + r8;
        const uint4 v = *(const uint4*)(ep + r * G256_EP_STRIDE + c * 16);
        half_t* rowp = Cw + (size_t)r * ld3;
        if (c < 4) {            // hi: stored twice
          *(uint4*)(rowp + N + c * 8) = v;
          *(uint4*)(rowp + 2 * (size_t)N + c * 8) = v;
        } else {                // lo
          *(uint4*)(rowp + (c - 4) * 8) = v;
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
  } else if constexpr (BASE == 6) {  // scores against per-column thresholds -> candidate lists; nothing is stored (wide.hip)
    float t16[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float4 t4 = *(const float4*)(fx.thr + tn * 256 + wc * 64 + j * 16 + 4 * q_e);
      t16[j][0] = t4.x; t16[j][1] = t4.y; t16[j][2] = t4.z; t16[j][3] = t4.w;
    }
    // A register (i, j, e) holds one query per group of 16 lanes (q_e) and 16 rows across the group.  Phase A counts, per column
    // (j, e), the wave's rows that reach the threshold and reserves their places in the query's list with ONE atomic per
    // column and group — all sixteen issued before any result is needed; phase B hands the places out.  (An atomic per
    // nomination made the GEMM 1.7x as long at k = 2000; an atomic per REGISTER, awaited before the next register was looked
    // at, still cost the growing chunks of every search 3x their time: under their loose thresholds a third of the registers
    // have a hit somewhere in the wave — 40 round trips to L2 per tile, one after the other.)
    const unsigned long long grp_mask = 0xFFFFull << (16 * q_e);
    const int leader = 16 * q_e;
    const uint32_t shard = blockIdx.x & (fx.shards - 1u), cap_s = fx.cap / fx.shards;
    uint32_t tot16[4][4], base16[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        uint32_t tot = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const bool hit = (uint32_t)(tm * 256 + wr * 128 + i * 16 + row_e) < fx.n_valid && acc[i][j][e] >= t16[j][e];
          tot += (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(hit) & grp_mask);
        }
        tot16[j][e] = tot;
        base16[j][e] = 0;
        if (tot && lane_e == leader)
          base16[j][e] = atomicAdd(&fx.count[(uint32_t)(tn * 256 + wc * 64 + j * 16 + 4 * q_e + e) * fx.shards + shard], tot);
      }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (tot16[j][e] == 0) continue;                      // (uniform over the group: its sixteen lanes go on together)
        const uint32_t q = (uint32_t)(tn * 256 + wc * 64 + j * 16 + 4 * q_e + e);
        uint32_t at = (uint32_t)__builtin_amdgcn_ds_bpermute(leader << 2, (int)base16[j][e]);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const uint32_t r = (uint32_t)(tm * 256 + wr * 128 + i * 16 + row_e);
          const float sc = acc[i][j][e];
          const bool hit = r < fx.n_valid && sc >= t16[j][e];
          const unsigned long long grp = __builtin_amdgcn_ballot_w64(hit) & grp_mask;
          if (hit) {
            const uint32_t pos = at + (uint32_t)__builtin_popcountll(grp & ((1ull << lane_e) - 1ull));
            if (pos < cap_s) fx.cand[(size_t)q * fx.cap + (size_t)shard * cap_s + pos] = rarc_candkey(sc, fx.row0 + r);
            else atomicOr(&fx.status[q], RARC_Q_OVERFLOW | RARC_Q_WHY_SEGMENT);
          }
          at += (uint32_t)__builtin_popcountll(grp);
        }
      }
  } else if constexpr (BASE == 3) {  // silu(gate)·up: the wave's 64 columns are 32 features -> 64-byte output rows
    // gate / up columns alternate in groups of eight: in a 16-column block the lanes with (lane >> 4) < 2 hold gate values of
    // features 4 q + e, the lanes 32 above them the up values of the same features.  One v_permlane32_swap per pair of registers
    // (e, e + 2) leaves the lower lane with (gate, up) of features 4 q + {0, 1} and the upper with those of 4 q + {2, 3}.
    typedef _Float16 half2v __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int i = 0; i < 8; ++i) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int nl = j * 16 + 4 * q_e;   // this lane's four columns inside the wave's block (gate if q_e < 2, else up)
        half4 b4 = {0, 0, 0, 0};
        if constexpr (!RS) b4 = *(const half4*)(bias + tn * 256 + wc * 64 + nl);
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[i][j][e] * rs[i] + (float)b4[e];
        // lower half: v0 v1 stay gate(f0) gate(f1), v2 v3 become up(f0) up(f1); upper half: v0 v1 become gate(f2) gate(f3)
        asm volatile("v_nop\n\tv_nop\n\tv_permlane32_swap_b32 %0, %1" : "+v"(v[0]), "+v"(v[2]));
        asm volatile("v_nop\n\tv_nop\n\tv_permlane32_swap_b32 %0, %1" : "+v"(v[1]), "+v"(v[3]));
        half2v out;
        out[0] = rarc_swiglu_f16(v[0], v[2]);
        out[1] = rarc_swiglu_f16(v[1], v[3]);
        *(half2v*)(ep + (i * 16 + row_e) * G256_EP_STRIDE + (j * 8 + 4 * (q_e & 1) + 2 * (q_e >> 1)) * 2) = out;
      }
    }
    __builtin_amdgcn_wave_barrier();
    const int r16 = lane_e >> 2, c = lane_e & 3;
    const int NO = N / 2;
    half_t* Cw = C + (size_t)(tm * 256 + wr * 128) * NO + tn * 128 + wc * 32 + c * 8;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int r = t * 16 + r16;
      *(uint4*)(Cw + (size_t)r * NO) = *(const uint4*)(ep + r * G256_EP_STRIDE + c * 16);
    }
  } else {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    if constexpr (RES) {
      if (i == 4) {   // acc[0..3] are dead: room for the second half of the old pieces (kept below the first half's conversion)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 8; t < 16; ++t) oldv[t] = *(const half8*)(Co + (size_t)(t * 8) * N);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int nl = j * 16 + 4 * q_e;  // column inside the wave's block
      half4 b4 = {0, 0, 0, 0};
      if constexpr (!RS) b4 = *(const half4*)(bias + tn * 256 + wc * 64 + nl);
      half4 out;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float v = acc[i][j][e] * rs[i] + (float)b4[e];
        if (BASE == 1) v = rarc_gelu_erf(v);
        out[e] = (half_t)v;
      }
      *(half4*)(ep + (i * 16 + row_e) * G256_EP_STRIDE + nl * 2) = out;
    }
  }
  __builtin_amdgcn_wave_barrier();
  {
    const int r8 = lane_e >> 3, c = lane_e & 7;
    half_t* Cw = C + (size_t)(tm * 256 + wr * 128) * N + tn * 256 + wc * 64 + c * 8;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const int r = t * 8 + r8;
      if constexpr ((ACT & 64) != 0)
        gemm_store8_ss(Cw + (size_t)r * N, *(const uint4*)(ep + r * G256_EP_STRIDE + c * 16), oldv[t], ssq,
                       (size_t)(tm * 256 + wr * 128 + r) * (N / 32) + tn * 8 + wc * 2 + (c >> 2), lane_e);
      else
        gemm_store8(Cw + (size_t)r * N, *(const uint4*)(ep + r * G256_EP_STRIDE + c * 16), RES, oldv[t]);
    }
  }
  }
  // every wave is done reading its staging block before the next tile's operand DMA lands in the same LDS
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  }
}

// ------------------------------------------------------------------------------------------
// Round 3 — the same 256 x 256 x 64 ping-pong schedule with NO SEAM between output tiles ("s" = seamless).
// The kernel above pays ~7 us per tile round next to 1.62 us per k tile (26 us of matrix work at K = 1024): the operand
// pipeline drains, the epilogue converts / transposes / stores with the matrix pipe idle (3 us of it is the per-CU issue
// rate of the stores), the next tile's first operands are fetched cold.  Here a workgroup's tiles form ONE stream of
// k tiles: the staging of the last two k tiles of an output tile already fetches the first two of the next (its A / W
// base pointers are computed a tile ahead), and the epilogue is taken apart by QUADRANT — the four 128 x 64 quadrants of
// a wave's block become final one per phase of the last k tile (A0B0, A0B1, A1B1, A1B0) and each is needed again only
// four phases later, when the next tile's first k tile (whose first MFMA takes C = 0) reaches the same quadrant.  So
// quadrant q is drained in the load slot of the phase AFTER it became final (q = 4 in the first slot of the next tile):
// fp16 conversion, a 32-row transposition through a private 2.5 KiB of LDS per wave (20 KiB next to the 128 KiB of
// pipeline), four 1-KiB stores of 16 rows x 64 bytes — under the other wave group's MFMAs, no barrier of its own, the
// pipeline's counted waits widened by exactly the stores in flight (vmcnt counts them, in order).
//   ACT 0: C = A·Wᵀ (no bias: the reranker LM's projections);  ACT 3: silu(gate)·up over interleaved gate / up columns.
// ------------------------------------------------------------------------------------------
#ifdef G256S_TIMELINE   // measurement builds: s_memtime at the start of every k tile of one workgroup's stream (tools/lab/gemm_seam_timeline.py)
__device__ unsigned long long g_g256s_tl[1024];
extern "C" int rarc_gemm_debug_timeline(unsigned long long* host, int n) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_g256s_tl), sizeof(unsigned long long) * (size_t)n);
}
#define G256S_TL() do { if (tl_on && tl_i < 1024) g_g256s_tl[tl_i++] = __builtin_readcyclecounter(); } while (0)
#else
#define G256S_TL() do { } while (0)
#endif
constexpr int G256S_EP_STRIDE = 80, G256S_EP_BYTES = 32 * G256S_EP_STRIDE;   // per-wave drain staging: 32 rows x (64 + 16) B
constexpr int G256S_LDS = 131072 + 8 * G256S_EP_BYTES;
template <int ACT>
__global__ __launch_bounds__(512, 1) void rarc_gemm256s_f16_kernel(const half_t* __restrict__ A,
                                                                   const half_t* __restrict__ W,
                                                                   half_t* __restrict__ C, int M, int N, int K,
                                                                   int order) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int row = lane & 31, hh = lane >> 5;
  const int wr = wave >> 2, wc = wave & 3;
  const int tiles_m = M / 256, tiles_n = N / 256, n_tiles = tiles_m * tiles_n;
  const int KT = K / GK;   // >= 4 (host)
  const int drow = lane >> 3, dslot = lane & 7;
  uint32_t soff[4][2];
#pragma unroll
  for (int which = 0; which < 4; ++which)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int lr = 8 * (wave * 2 + j) + drow;
      const int c = dslot ^ ((lr >> 1) & 7);
      const int h = which & 1;
      const int trow = which < 2 ? (lr >> 6) * 128 + h * 64 + (lr & 63) : (lr >> 5) * 64 + h * 32 + (lr & 31);
      soff[which][j] = ((uint32_t)trow * (uint32_t)K + (uint32_t)c * 8u) * 2u;
    }
  // staging: half-tile `which` (0 A0, 1 A1, 2 B0, 3 B1) of k tile kt of the output tile whose operand rows start at Ab / Wb
  auto stage = [&](int par, int which, const half_t* Ab, const half_t* Wb, int kt) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int i = wave * 2 + j;
      const char* base = (const char*)(which < 2 ? Ab : Wb) + (size_t)kt * (GK * 2);
      __builtin_amdgcn_global_load_lds(RARC_GPTR(base + soff[which][j]),
                                       RARC_LPTR(smem + par * 65536 + which * 16384 + i * 1024), 16, 0, 0);
    }
  };
  const int sw = (row >> 1) & 7;
  int xk[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) xk[kk] = row * 128 + (((2 * kk + hh) ^ sw) << 4);

  f32x16 acc[4][2];
  half8 fa[4][2], fb0[4], fb1[4];
  typedef _Float16 half4 __attribute__((ext_vector_type(4)));
  typedef uint32_t uint2v __attribute__((ext_vector_type(2)));
  typedef uint32_t uint4v __attribute__((ext_vector_type(4)));

#define G256_LOAD_A(PAR, H)                                                                             \
  _Pragma("unroll") for (int kk = 0; kk < 4; ++kk) {                                                    \
    const int ad = xk[kk] + ((PAR) * 65536 + (H) * 16384 + wr * 8192);                                  \
    asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:4096"                               \
                 : "=&v"(fa[kk][0]), "=&v"(fa[kk][1]) : "v"(ad) : "memory");                            \
  }
#define G256_LOAD_B(PAR, H, FB)                                                                         \
  _Pragma("unroll") for (int kk = 0; kk < 4; ++kk) {                                                    \
    const int ad = xk[kk] + ((PAR) * 65536 + 32768 + (H) * 16384 + wc * 4096);                          \
    asm volatile("ds_read_b128 %0, %1" : "=&v"(FB[kk]) : "v"(ad) : "memory");                           \
  }
#define G256_WAIT()                                                                                     \
  asm volatile("s_waitcnt lgkmcnt(0)"                                                                   \
               : "+v"(fa[0][0]), "+v"(fa[0][1]), "+v"(fa[1][0]), "+v"(fa[1][1]), "+v"(fa[2][0]),        \
                 "+v"(fa[2][1]), "+v"(fa[3][0]), "+v"(fa[3][1]), "+v"(fb0[0]), "+v"(fb0[1]),            \
                 "+v"(fb0[2]), "+v"(fb0[3]), "+v"(fb1[0]), "+v"(fb1[1]), "+v"(fb1[2]), "+v"(fb1[3])     \
               :: "memory")
#define G256_MMA(QM, QN, FB)                                                                            \
  asm volatile("" : "+v"(FB[0]), "+v"(FB[1]), "+v"(FB[2]), "+v"(FB[3]));                                \
  __builtin_amdgcn_sched_barrier(0);                                                                    \
  __builtin_amdgcn_s_setprio(1);                                                                        \
  _Pragma("unroll") for (int kk = 0; kk < 4; ++kk) {                                                    \
    acc[2 * (QM)][QN] = __builtin_amdgcn_mfma_f32_32x32x16_f16(FB[kk], fa[kk][0], acc[2 * (QM)][QN], 0, 0, 0); \
    acc[2 * (QM) + 1][QN] = __builtin_amdgcn_mfma_f32_32x32x16_f16(FB[kk], fa[kk][1], acc[2 * (QM) + 1][QN], 0, 0, 0); \
  }                                                                                                     \
  __builtin_amdgcn_s_setprio(0);                                                                        \
  asm volatile("" : "+v"(acc[2 * (QM)][QN]), "+v"(acc[2 * (QM) + 1][QN]));                              \
  __builtin_amdgcn_sched_barrier(0)
#define G256_BAR() do { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } while (0)
  // drain of quadrant (QM, QN) of the output tile (TM, TN): acc[2QM + i][QN] is rows wr*128 + (2QM + i)*32 + row, cols
  // wc*64 + QN*32 + (8g + 4hh .. +3) of the tile.  Per 32-row block: transposition through the wave's own staging (LDS
  // operations of one wave execute in order: no wait between its writes and reads, and the next block's writes come after
  // this block's reads), then 16-byte pieces of whole 64-byte (ACT 3: 32-byte) row segments to C; the registers are
  // cleared for the next output tile.
#define G256S_DRAIN(QM, QN, TM, TN)                                                                     \
  {                                                                                                     \
    int lane_e = lane;                                                                                  \
    asm volatile("" : "+v"(lane_e));   /* (nothing derived from the lane index is hoisted out of the loop) */ \
    const int row_e = lane_e & 31, hh_e = lane_e >> 5;                                                  \
    const int ep_off = 131072 + wave * G256S_EP_BYTES;   /* LDS byte address of the wave's staging (dynamic LDS starts at 0) */ \
    _Pragma("unroll") for (int i2 = 0; i2 < 2; ++i2) {                                                  \
      f32x16& blk = acc[2 * (QM) + i2][QN];                                                             \
      /* the staging accesses are inline asm: to the compiler an LDS read after an LDS-DMA may alias the DMA's */ \
      /* destination, and it drains the whole operand pipeline (vmcnt(0)) in front of it */             \
      if constexpr (ACT == 3) {                                                                         \
        _Pragma("unroll") for (int g = 0; g < 4; g += 2) {                                              \
          half4 out;                                                                                    \
          _Pragma("unroll") for (int e = 0; e < 4; ++e) out[e] = rarc_swiglu_f16(blk[4 * g + e], blk[4 * g + 4 + e]); \
          asm volatile("ds_write_b64 %0, %1" :: "v"(ep_off + row_e * G256S_EP_STRIDE + (4 * g + 4 * hh_e) * 2), \
                       "v"(__builtin_bit_cast(uint2v, out)) : "memory");                              \
        }                                                                                               \
        const int r32 = lane_e >> 1, c = lane_e & 1;                                                    \
        uint4v v;                                                                                       \
        asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(ep_off + r32 * G256S_EP_STRIDE + c * 16) : "memory"); \
        half_t* Cw = C + (size_t)((TM) * 256 + wr * 128 + (2 * (QM) + i2) * 32 + r32) * (N / 2) + (TN) * 128 + wc * 32 + (QN) * 16 + c * 8; \
        *(uint4v*)Cw = v;                                                                               \
      } else {                                                                                          \
        _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                                 \
          const half4 out = {(half_t)blk[4 * g], (half_t)blk[4 * g + 1], (half_t)blk[4 * g + 2], (half_t)blk[4 * g + 3]}; \
          asm volatile("ds_write_b64 %0, %1" :: "v"(ep_off + row_e * G256S_EP_STRIDE + (8 * g + 4 * hh_e) * 2), \
                       "v"(__builtin_bit_cast(uint2v, out)) : "memory");                              \
        }                                                                                               \
        const int r16 = lane_e >> 2, c = lane_e & 3;                                                    \
        uint4v v0, v1;                                                                                  \
        asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:1280\n\ts_waitcnt lgkmcnt(0)"   \
                     : "=&v"(v0), "=&v"(v1) : "v"(ep_off + r16 * G256S_EP_STRIDE + c * 16) : "memory"); \
        half_t* Cw = C + (size_t)((TM) * 256 + wr * 128 + (2 * (QM) + i2) * 32 + r16) * N + (TN) * 256 + wc * 64 + (QN) * 32 + c * 8; \
        *(uint4v*)Cw = v0;                                                                              \
        *(uint4v*)(Cw + (size_t)16 * N) = v1;                                                           \
      }                                                                                                 \
      blk = (f32x16){0};                                                                                \
    }                                                                                                   \
    asm volatile("" ::: "memory");   /* the stores are issued before this slot's operand staging */      \
    __builtin_amdgcn_sched_barrier(0);                                                                  \
  }
  constexpr int SD = ACT == 3 ? 2 : 4;   // stores per quadrant drain and wave
  // the vector-memory wait at the end of a load slot: all but the 10 DMA instructions of the five newest slots, plus the
  // stores issued in those slots (vmcnt counts stores too, in order; in the stream's order a drain's stores precede
  // their slot's DMA).  sel 0: none in the window, 1: one drain slot (2 quadrants = 2 SD stores), 2: two drain slots,
  // 3: everything (the last two k tiles of the stream)
  auto vm_wait = [&](int sel) __attribute__((always_inline)) {
    if (sel == 0) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    else if (sel == 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(10 + 2 * SD) : "memory");
    else if (sel == 2) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(10 + 4 * SD) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };

  int bid = blockIdx.x;
  if (bid >= n_tiles) return;
  int tm, tn;
  gemm_tile_of(bid, n_tiles, tiles_m, tiles_n, order, tm, tn);
  const half_t* Ab = A + (size_t)tm * 256 * K;
  const half_t* Wb = W + (size_t)tn * 256 * K;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (f32x16){0};
  // prologue of the stream: all of k tile 0, then A0, B0, B1 of k tile 1 (its A1 follows in phase 1 of k tile 0)
  stage(0, 0, Ab, Wb, 0); stage(0, 2, Ab, Wb, 0); stage(0, 3, Ab, Wb, 0); stage(0, 1, Ab, Wb, 0);
  stage(1, 0, Ab, Wb, 1); stage(1, 2, Ab, Wb, 1); stage(1, 3, Ab, Wb, 1);
  asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  G256_BAR();
  if (wr == 1) G256_BAR();  // waves 4-7 run one slot behind
  int par = 0, np = 1;
  int ptm = 0, ptn = 0;     // the previous output tile (its second drain runs in this tile's first slot)
  bool first = true;
  int nbid = bid + gridDim.x;
  bool has_next = nbid < n_tiles;
  int ntm = 0, ntn = 0;
  if (has_next) gemm_tile_of(nbid, n_tiles, tiles_m, tiles_n, order, ntm, ntn);
  const half_t* nAb = A + (size_t)ntm * 256 * K;
  const half_t* nWb = W + (size_t)ntn * 256 * K;
  // ONE loop over the workgroup's stream of k tiles; what varies at the seams (where the look-ahead's operands come from,
  // whether a drain runs, how many stores sit in the wait's window) is wave-uniform run-time state, not code variants
  // (seven instantiations of the four phases spilled the accumulators: the allocator lost track of 128 live registers).
#ifdef G256S_TIMELINE
  const bool tl_on = blockIdx.x == 100 && threadIdx.x == 0;
  int tl_i = 0;
#endif
  for (int kt = 0;;) {
    // the interior k tiles of an output tile (2 .. KT-3): operands of this tile, no drain, no stores in any window — the
    // loop of the kernel with a seam, with nothing decided at run time (with the seam logic in every phase a k tile took
    // 1.95 us instead of 1.51: ~160 cycles of scalar selects and branches per load slot, and the load slots are the
    // critical path of the ping-pong)
    for (; kt >= 2 && kt + 2 < KT; ++kt) {
      G256S_TL();
#ifdef G256S_STAGE_IN_M   // experiment: the operand DMA issued in the tail of the matrix slot instead of next to the fragment reads
      G256_LOAD_A(par, 0) G256_LOAD_B(par, 0, fb0)
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      G256_WAIT();
      G256_BAR();
      G256_MMA(0, 0, fb0);
      stage(np, 1, Ab, Wb, kt + 1);
      G256_BAR();
      G256_LOAD_B(par, 1, fb1)
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      G256_WAIT();
      G256_BAR();
      G256_MMA(0, 1, fb1);
      stage(par, 0, Ab, Wb, kt + 2);
      G256_BAR();
      G256_LOAD_A(par, 1)
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      G256_WAIT();
      G256_BAR();
      G256_MMA(1, 1, fb1);
      stage(par, 2, Ab, Wb, kt + 2);
      G256_BAR();
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      G256_WAIT();
      G256_BAR();
      G256_MMA(1, 0, fb0);
      stage(par, 3, Ab, Wb, kt + 2);
      G256_BAR();
#else
      G256_LOAD_A(par, 0) G256_LOAD_B(par, 0, fb0)
      stage(np, 1, Ab, Wb, kt + 1);
      asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
      G256_WAIT();
      G256_BAR();
      G256_MMA(0, 0, fb0);
      G256_BAR();
      G256_LOAD_B(par, 1, fb1)
      stage(par, 0, Ab, Wb, kt + 2);
      asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
      G256_WAIT();
      G256_BAR();
      G256_MMA(0, 1, fb1);
      G256_BAR();
      G256_LOAD_A(par, 1)
      stage(par, 2, Ab, Wb, kt + 2);
      asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
      G256_WAIT();
      G256_BAR();
      G256_MMA(1, 1, fb1);
      G256_BAR();
      stage(par, 3, Ab, Wb, kt + 2);
      asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
      G256_WAIT();
      G256_BAR();
      G256_MMA(1, 0, fb0);
      G256_BAR();
#endif
      par ^= 1; np ^= 1;
    }
    // ---- a k tile at a seam: 0, 1, KT-2 or KT-1 ----
    G256S_TL();
    const int k1 = kt + 1, k2 = kt + 2;
    const bool in1 = k1 < KT, in2 = k2 < KT;
    const half_t* s1A = in1 ? Ab : nAb;
    const half_t* s1W = in1 ? Wb : nWb;
    const half_t* s2A = in2 ? Ab : nAb;
    const half_t* s2W = in2 ? Wb : nWb;
    const int s1k = in1 ? k1 : k1 - KT, s2k = in2 ? k2 : k2 - KT;
    const bool st1 = in1 || has_next, st2 = in2 || has_next;
    const bool ending = !has_next && kt + 2 >= KT;                    // the last two k tiles of the stream
    const bool drain_b = kt == 0 && !first, drain_a = kt == KT - 1;
    // stores in the five-slot window at the end of each phase's load slot (see vm_wait)
    // (a drain in the tail of phase p's matrix slot precedes the DMA of phase p + 1's load slot: it belongs to that slot.
    //  Drain slots of a tile: phase 4 of its last k tile and phase 2 of the next tile's first k tile.)
    const bool after1 = kt == 1 && !first;
    const int w1 = ending ? 3 : (drain_b ? 1 : (after1 ? 1 : 0));
    const int w2 = ending ? 3 : (drain_b ? 2 : (after1 ? 1 : 0));
    const int w3 = ending ? 3 : (drain_b ? 2 : 0);
    const int w4 = ending ? 3 : (drain_b ? 2 : (drain_a ? 1 : 0));
    // ---- phase 1: quadrant A0 x B0 ----
    G256_LOAD_A(par, 0) G256_LOAD_B(par, 0, fb0)
    if (st1) stage(np, 1, s1A, s1W, s1k);
    vm_wait(w1);
    G256_WAIT();
    G256_BAR();
    G256_MMA(0, 0, fb0);
    // the second drain of the previous output tile, in the TAIL of this matrix slot: the wave has issued its eight MFMAs and
    // would wait ~200 cycles at the barrier for the other group's load slot anyway (s_memtime per k tile: a load slot is
    // ~460 cycles, a matrix slot ~256; in a load slot the same work extended the phase by its full length)
    if (drain_b) { G256S_DRAIN(1, 1, ptm, ptn) G256S_DRAIN(1, 0, ptm, ptn) }
    G256_BAR();
    // ---- phase 2: A0 x B1 ----
    G256_LOAD_B(par, 1, fb1)
    if (st2) stage(par, 0, s2A, s2W, s2k);
    vm_wait(w2);
    G256_WAIT();
    G256_BAR();
    G256_MMA(0, 1, fb1);
    G256_BAR();
    // ---- phase 3: A1 x B1; in its matrix slot's tail the first drain (quadrants A0B0, A0B1 are final) ----
    G256_LOAD_A(par, 1)
    if (st2) stage(par, 2, s2A, s2W, s2k);
    vm_wait(w3);
    G256_WAIT();
    G256_BAR();
    G256_MMA(1, 1, fb1);
    if (drain_a) { G256S_DRAIN(0, 0, tm, tn) G256S_DRAIN(0, 1, tm, tn) }
    G256_BAR();
    // ---- phase 4: A1 x B0 (no LDS reads) ----
    if (st2) stage(par, 3, s2A, s2W, s2k);
    vm_wait(w4);
    G256_WAIT();
    G256_BAR();
    G256_MMA(1, 0, fb0);
    G256_BAR();
    par ^= 1; np ^= 1;
    if (++kt < KT) continue;
    if (!has_next) break;
    // ---- the next output tile of the stream ----
    ptm = tm; ptn = tn;
    bid = nbid; tm = ntm; tn = ntn; Ab = nAb; Wb = nWb;
    first = false;
    kt = 0;
    nbid = bid + gridDim.x;
    has_next = nbid < n_tiles;
    if (has_next) gemm_tile_of(nbid, n_tiles, tiles_m, tiles_n, order, ntm, ntn);
    nAb = A + (size_t)ntm * 256 * K;
    nWb = W + (size_t)ntn * 256 * K;
  }
  if (wr == 0) G256_BAR();
  G256S_DRAIN(1, 1, tm, tn) G256S_DRAIN(1, 0, tm, tn)
#undef G256S_DRAIN
#undef G256_LOAD_A
#undef G256_LOAD_B
#undef G256_WAIT
#undef G256_MMA
#undef G256_BAR
}

// ------------------------------------------------------------------------------------------
// The same ping-pong schedule on 256 x 128 tiles, for N = 1024 (attention output, FFN2: 256 tiles at M = 8192)
// and for shapes whose 256 x 256 tile count would leave a ragged last round.
//   8 waves = 2 (M) x 4 (N); wave (wr, wc) owns rows wr*128.. x cols wc*32..: acc[8][2] 16x16 blocks (v_mfma_f32_16x16x32_f16).
//   LDS 144 KiB = a ring of 3 k tiles x {A0, A1, B} x 16 KiB (A_h as above; B = the 128 rows of W).
//   Two phases per k tile (16 MFMAs each): A0 x B (reads A0, B), A1 x B (reads A1; B stays in registers).
//   Restaging: phase 2 of tile t refills A0, B of its own slot with tile t+3; phase 1 of tile t with A1 of
//   tile t+2 (slot of tile t-1) — 5 phases of lead, up to 80 KiB in flight; the wait at the end of L(p) lets
//   the stages of the last four phases (12 DMA instructions per wave) stay outstanding.
// ------------------------------------------------------------------------------------------
constexpr int G128_EP_STRIDE = 80, G128_EP_BYTES = 128 * G128_EP_STRIDE;
constexpr int G128_LDS = 3 * 49152;
static_assert(8 * 128 * 144 <= G128_LDS, "fp32 epilogue staging of the 256 x 128 kernel");
template <int ACT>
__global__ __launch_bounds__(512, 1) void rarc_gemm256x128_f16_kernel(const half_t* __restrict__ A,
                                                                      const half_t* __restrict__ W,
                                                                      const half_t* __restrict__ bias,
                                                                      half_t* __restrict__ C, int M, int N, int K,
                                                                      int order, float* __restrict__ ssq = nullptr) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int row = lane & 31, hh = lane >> 5;
  const int wr = wave >> 2, wc = wave & 3;
  const int tiles_m = M / 256, tiles_n = N / 128;
  const int bid = blockIdx.x;
  int tm, tn;
  gemm_tile_of(bid, (int)gridDim.x, tiles_m, tiles_n, order, tm, tn);
  const half_t* Ab = A + (size_t)tm * 256 * K;
  const half_t* Wb = W + (size_t)tn * 128 * K;
  const int KT = K / GK;  // >= 3 (launcher)

  const int drow = lane >> 3, dslot = lane & 7;
  auto stage = [&](int slot, int which, int kt) __attribute__((always_inline)) {  // which: 0 A0, 1 A1, 2 B
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int i = wave * 2 + j;
      const int lr = 8 * i + drow;
      const int c = dslot ^ ((lr >> 1) & 7);
      const half_t* src = which < 2 ? Ab + (size_t)((lr >> 6) * 128 + which * 64 + (lr & 63)) * K : Wb + (size_t)lr * K;
      __builtin_amdgcn_global_load_lds(RARC_GPTR(src + kt * GK + c * 8),
                                       RARC_LPTR(smem + slot * 49152 + which * 16384 + i * 1024), 16, 0, 0);
    }
  };
  // 16x16x32 fragments, as in the 256 x 256 kernel: row (lane & 15) of a 16-row block, chunk 4 s + (lane >> 4); step 1 = step 0 ^ 64
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  int xs[2];
  {
    const int r16 = lane & 15, sw16 = (r16 >> 1) & 7;
    xs[0] = r16 * 128 + (((lane >> 4) ^ sw16) << 4);
    xs[1] = xs[0] ^ 64;
  }

  f32x4 acc[8][2];  // [16-row block of the wave's 128 rows][16-column block of its 32 columns]
#pragma unroll
  for (int i = 0; i < 8; ++i) { acc[i][0] = (f32x4){0, 0, 0, 0}; acc[i][1] = (f32x4){0, 0, 0, 0}; }
  half8 fa[2][4], fb[2][2];

#define G128_LOAD_A(SLOT, H)                                                                            \
  _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                                    \
    const int ad = xs[ks] + ((SLOT) * 49152 + (H) * 16384 + wr * 8192);                                 \
    asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:2048\n\t"                          \
                 "ds_read_b128 %2, %4 offset:4096\n\tds_read_b128 %3, %4 offset:6144"                   \
                 : "=&v"(fa[ks][0]), "=&v"(fa[ks][1]), "=&v"(fa[ks][2]), "=&v"(fa[ks][3]) : "v"(ad) : "memory"); \
  }
#define G128_LOAD_B(SLOT)                                                                               \
  _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                                    \
    const int ad = xs[ks] + ((SLOT) * 49152 + 32768 + wc * 4096);                                       \
    asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:2048"                              \
                 : "=&v"(fb[ks][0]), "=&v"(fb[ks][1]) : "v"(ad) : "memory");                            \
  }
#define G128_WAIT(VM)                                                                                   \
  asm volatile("s_waitcnt vmcnt(" #VM ")\n\ts_waitcnt lgkmcnt(0)"                                       \
               : "+v"(fa[0][0]), "+v"(fa[0][1]), "+v"(fa[0][2]), "+v"(fa[0][3]), "+v"(fa[1][0]),        \
                 "+v"(fa[1][1]), "+v"(fa[1][2]), "+v"(fa[1][3]), "+v"(fb[0][0]), "+v"(fb[0][1]), "+v"(fb[1][0]), \
                 "+v"(fb[1][1])                                                                         \
               :: "memory")
#define G128_MMA(H)                                                                                     \
  asm volatile("" : "+v"(fb[0][0]), "+v"(fb[0][1]), "+v"(fb[1][0]), "+v"(fb[1][1]));                    \
  __builtin_amdgcn_sched_barrier(0);                                                                    \
  __builtin_amdgcn_s_setprio(1);                                                                        \
  _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                      \
    _Pragma("unroll") for (int b = 0; b < 4; ++b) {                                                     \
      acc[4 * (H) + b][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[ks][0], fa[ks][b], acc[4 * (H) + b][0], 0, 0, 0); \
      acc[4 * (H) + b][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[ks][1], fa[ks][b], acc[4 * (H) + b][1], 0, 0, 0); \
    }                                                                                                   \
  __builtin_amdgcn_s_setprio(0);                                                                        \
  _Pragma("unroll") for (int b = 0; b < 4; ++b) asm volatile("" : "+v"(acc[4 * (H) + b][0]), "+v"(acc[4 * (H) + b][1])); \
  __builtin_amdgcn_sched_barrier(0)
#define G128_BAR() do { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } while (0)
  // (round 3: a phase's DMA is issued in the tail of its matrix slot, as in the 256 x 256 kernel; the wait at the end of a
  //  load slot then sees the stages of the last THREE phases outstanding instead of four: 10 / 8 instructions)
#define G128_TILE(ST_A1, ST_AB, VM1, VM2)                                                               \
  {                                                                                                     \
    G128_LOAD_A(sl, 0) G128_LOAD_B(sl)                                                                  \
    G128_WAIT(VM1);                                                                                     \
    G128_BAR();                                                                                         \
    G128_MMA(0);                                                                                        \
    if (ST_A1) stage(slp, 1, kt + 2);                                                                   \
    G128_BAR();                                                                                         \
    G128_LOAD_A(sl, 1)                                                                                  \
    G128_WAIT(VM2);                                                                                     \
    G128_BAR();                                                                                         \
    G128_MMA(1);                                                                                        \
    if (ST_AB) { stage(sl, 0, kt + 3); stage(sl, 2, kt + 3); }                                          \
    G128_BAR();                                                                                         \
    slp = sl; sl = sl == 2 ? 0 : sl + 1;                                                                \
  }
  // prologue in steady-state order: tiles 0, 1 whole, A0 and B of tile 2 (its A1 follows in phase 1 of tile 0)
  stage(0, 0, 0); stage(0, 2, 0); stage(0, 1, 0);
  stage(1, 0, 1); stage(1, 2, 1); stage(1, 1, 1);
  stage(2, 0, 2); stage(2, 2, 2);
  asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
  G128_BAR();
  if (wr == 1) G128_BAR();
  int kt = 0, sl = 0, slp = 2;  // sl = kt % 3, slp = (kt + 2) % 3
  for (; kt + 3 < KT; ++kt) G128_TILE(true, true, 10, 8)
  G128_TILE(true, false, 10, 8)  // tile KT-3
  ++kt;
  G128_TILE(false, false, 6, 2)  // tile KT-2
  ++kt;
  G128_TILE(false, false, 0, 0)  // tile KT-1
  if (wr == 0) G128_BAR();
#undef G128_TILE
#undef G128_LOAD_A
#undef G128_LOAD_B
#undef G128_WAIT
#undef G128_MMA
#undef G128_BAR
  // epilogue: acc[i][j][e] is row wr*128 + 16 i + (lane & 15), column wc*32 + 16 j + 4 (lane >> 4) + e; transposed through LDS like above
  const int row16 = lane & 15, q4 = lane >> 4;
  typedef _Float16 half4 __attribute__((ext_vector_type(4)));
  char* ep = smem + wave * G128_EP_BYTES;
  constexpr int BASE = ACT & 15;
  constexpr bool RS = (ACT & 16) != 0, RES = (ACT & 32) != 0;
  float rs[8] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f};
  if constexpr (RS) {
#pragma unroll
    for (int i = 0; i < 8; ++i) rs[i] = ((const float*)bias)[(size_t)tm * 256 + wr * 128 + i * 16 + row16];
  }
  half8 oldv[8];
  if constexpr (RES) {
    const half_t* Co = C + (size_t)(tm * 256 + wr * 128 + (lane >> 2)) * N + tn * 128 + wc * 32 + (lane & 3) * 8;
#pragma unroll
    for (int t = 0; t < 8; ++t) oldv[t] = *(const half8*)(Co + (size_t)(t * 16) * N);
  }
  if constexpr (BASE == 4) {  // raw fp32 products, no bias: the wave's 128 x 32 block as 128-byte rows (144-byte staging rows)
    char* ep4 = smem + wave * (128 * 144);
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        *(float4*)(ep4 + (i * 16 + row16) * 144 + (16 * j + 4 * q4) * 4) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
    __builtin_amdgcn_wave_barrier();
    const int r8 = lane >> 3, c = lane & 7;
    float* Cw = (float*)C + (size_t)(tm * 256 + wr * 128) * N + tn * 128 + wc * 32 + c * 4;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const int r = t * 8 + r8;
      *(float4*)(Cw + (size_t)r * N) = *(const float4*)(ep4 + r * 144 + c * 16);
    }
    return;
  }
  if constexpr (BASE == 3) {  // silu(gate)·up (see rarc_swiglu_f16): the wave's 32 columns are 16 features -> 32-byte rows
    // (gate | up of a feature sit 32 lanes apart: one v_permlane32_swap per register pair, as in the 256 x 256 kernel)
    typedef _Float16 half2v __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int i = 0; i < 8; ++i) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int nl = j * 16 + 4 * q4;
        half4 b4 = {0, 0, 0, 0};
        if constexpr (!RS) b4 = *(const half4*)(bias + tn * 128 + wc * 32 + nl);
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[i][j][e] * rs[i] + (float)b4[e];
        asm volatile("v_nop\n\tv_nop\n\tv_permlane32_swap_b32 %0, %1" : "+v"(v[0]), "+v"(v[2]));
        asm volatile("v_nop\n\tv_nop\n\tv_permlane32_swap_b32 %0, %1" : "+v"(v[1]), "+v"(v[3]));
        half2v out;
        out[0] = rarc_swiglu_f16(v[0], v[2]);
        out[1] = rarc_swiglu_f16(v[1], v[3]);
        *(half2v*)(ep + (i * 16 + row16) * G128_EP_STRIDE + (j * 8 + 4 * (q4 & 1) + 2 * (q4 >> 1)) * 2) = out;
      }
    }
    __builtin_amdgcn_wave_barrier();
    const int r32 = lane >> 1, c = lane & 1;
    const int NO = N / 2;
    half_t* Cw = C + (size_t)(tm * 256 + wr * 128) * NO + tn * 64 + wc * 16 + c * 8;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int r = t * 32 + r32;
      *(uint4*)(Cw + (size_t)r * NO) = *(const uint4*)(ep + r * G128_EP_STRIDE + c * 16);
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int nl = 16 * j + 4 * q4;
      half4 b4 = {0, 0, 0, 0};
      if constexpr (!RS) b4 = *(const half4*)(bias + tn * 128 + wc * 32 + nl);
      half4 out;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float v = acc[i][j][e] * rs[i] + (float)b4[e];
        if (BASE == 1) v = rarc_gelu_erf(v);
        out[e] = (half_t)v;
      }
      *(half4*)(ep + (i * 16 + row16) * G128_EP_STRIDE + nl * 2) = out;
    }
  }
  __builtin_amdgcn_wave_barrier();
  {
    const int r16 = lane >> 2, c = lane & 3;
    half_t* Cw = C + (size_t)(tm * 256 + wr * 128) * N + tn * 128 + wc * 32 + c * 8;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int r = t * 16 + r16;
      if constexpr ((ACT & 64) != 0)
        gemm_store8_ss(Cw + (size_t)r * N, *(const uint4*)(ep + r * G128_EP_STRIDE + c * 16), oldv[t], ssq,
                       (size_t)(tm * 256 + wr * 128 + r) * (N / 32) + tn * 4 + wc, lane);
      else
        gemm_store8(Cw + (size_t)r * N, *(const uint4*)(ep + r * G128_EP_STRIDE + c * 16), RES, oldv[t]);
    }
  }
}

// ------------------------------------------------------------------------------------------
// The ping-pong schedule for SMALL batches: 128 x 128 tiles, at most one workgroup per CU (<= 256 tiles), where the
// one-barrier kernel leaves a CU's single wave per SIMD waiting on LDS and DMA latencies in turn.
//   8 waves = 2 (M) x 4 (N); wave (wr, wc) owns rows wr*64.. x cols wc*32..: acc[2] 32x32 blocks.
//   LDS 128 KiB = a ring of 4 k tiles x {A, B} x 16 KiB.  One phase per k tile (8 MFMAs): L reads A (8) + B (4)
//   fragments and restages the slot read in the PREVIOUS phase with the tile three ahead (4 DMA instructions per
//   wave, 96 KiB in flight); the wait at the end of L(p) leaves the stages of L(p-1), L(p) outstanding (vmcnt 8).
//   ACT 2 = split-K slice blockIdx.y -> fp32 partials (summed by the LayerNorm kernel), as in the first kernel.
// ------------------------------------------------------------------------------------------
constexpr int G128S_LDS = 4 * 32768;
template <int ACT>
__global__ __launch_bounds__(512, 1) void rarc_gemm128pp_f16_kernel(const half_t* __restrict__ A,
                                                                    const half_t* __restrict__ W,
                                                                    const half_t* __restrict__ bias,
                                                                    half_t* __restrict__ C, int M, int N, int K, int ldk,
                                                                    int order, float* __restrict__ ssq = nullptr,
                                                                    const GemmSplitEpi fx = GemmSplitEpi()) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int row = lane & 31, hh = lane >> 5;
  const int wr = wave >> 2, wc = wave & 3;
  const int tiles_m = M / 128, tiles_n = N / 128;
  const int bid = blockIdx.x;
  int tm, tn;
  gemm_tile_of(bid, (int)gridDim.x, tiles_m, tiles_n, order, tm, tn);
  const half_t* Ab = A + (size_t)tm * 128 * ldk + (ACT == 2 ? (size_t)blockIdx.y * K : 0);
  const half_t* Wb = W + (size_t)tn * 128 * ldk + (ACT == 2 ? (size_t)blockIdx.y * K : 0);
  const int KT = K / GK;  // >= 4 (launcher)

  const int drow = lane >> 3, dslot = lane & 7;
  auto stage = [&](int slot, int kt) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int i = wave * 2 + j;
      const int lr = 8 * i + drow;
      const int c = dslot ^ ((lr >> 1) & 7);
      __builtin_amdgcn_global_load_lds(RARC_GPTR(Ab + (size_t)lr * ldk + kt * GK + c * 8),
                                       RARC_LPTR(smem + slot * 32768 + i * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds(RARC_GPTR(Wb + (size_t)lr * ldk + kt * GK + c * 8),
                                       RARC_LPTR(smem + slot * 32768 + 16384 + i * 1024), 16, 0, 0);
    }
  };
  const int sw = (row >> 1) & 7;
  int xk[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) xk[kk] = row * 128 + (((2 * kk + hh) ^ sw) << 4);

  f32x16 acc[2];
  acc[0] = (f32x16){0};
  acc[1] = (f32x16){0};
  half8 fa[4][2], fb[4];
#define G128S_BAR() do { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } while (0)
#define G128S_TILE(ST, VM)                                                                              \
  {                                                                                                     \
    _Pragma("unroll") for (int kk = 0; kk < 4; ++kk) {                                                  \
      const int ad = xk[kk] + (sl * 32768 + wr * 8192);                                                 \
      asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:4096"                             \
                   : "=&v"(fa[kk][0]), "=&v"(fa[kk][1]) : "v"(ad) : "memory");                          \
    }                                                                                                   \
    _Pragma("unroll") for (int kk = 0; kk < 4; ++kk) {                                                  \
      const int ad = xk[kk] + (sl * 32768 + 16384 + wc * 4096);                                         \
      asm volatile("ds_read_b128 %0, %1" : "=&v"(fb[kk]) : "v"(ad) : "memory");                         \
    }                                                                                                   \
    if (ST) stage(slp, kt + 3);                                                                         \
    asm volatile("s_waitcnt vmcnt(" #VM ")\n\ts_waitcnt lgkmcnt(0)"                                     \
                 : "+v"(fa[0][0]), "+v"(fa[0][1]), "+v"(fa[1][0]), "+v"(fa[1][1]), "+v"(fa[2][0]),      \
                   "+v"(fa[2][1]), "+v"(fa[3][0]), "+v"(fa[3][1]), "+v"(fb[0]), "+v"(fb[1]),            \
                   "+v"(fb[2]), "+v"(fb[3])                                                             \
                 :: "memory");                                                                          \
    G128S_BAR();                                                                                        \
    asm volatile("" : "+v"(fb[0]), "+v"(fb[1]), "+v"(fb[2]), "+v"(fb[3]));                              \
    __builtin_amdgcn_sched_barrier(0);                                                                  \
    __builtin_amdgcn_s_setprio(1);                                                                      \
    _Pragma("unroll") for (int kk = 0; kk < 4; ++kk) {                                                  \
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[kk], fa[kk][0], acc[0], 0, 0, 0);              \
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[kk], fa[kk][1], acc[1], 0, 0, 0);              \
    }                                                                                                   \
    __builtin_amdgcn_s_setprio(0);                                                                      \
    asm volatile("" : "+v"(acc[0]), "+v"(acc[1]));                                                      \
    __builtin_amdgcn_sched_barrier(0);                                                                  \
    G128S_BAR();                                                                                        \
    slp = sl; sl = (sl + 1) & 3;                                                                        \
  }
  // prologue: tiles 0, 1, 2 (tile 3 follows in the first phase, into the slot nobody has read yet)
  stage(0, 0); stage(1, 1); stage(2, 2);
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  G128S_BAR();
  if (wr == 1) G128S_BAR();  // waves 4-7 run one slot behind
  int kt = 0, sl = 0, slp = 3;  // sl = kt & 3; slp = slot of tile kt-1 = slot of tile kt+3
  for (; kt + 3 < KT; ++kt) G128S_TILE(true, 8)
  G128S_TILE(false, 4)  // tile KT-3
  ++kt;
  G128S_TILE(false, 0)  // tile KT-2
  ++kt;
  G128S_TILE(false, 0)  // tile KT-1
  if (wr == 0) G128S_BAR();
#undef G128S_TILE
#undef G128S_BAR
  // epilogue: acc[i] is rows wr*64 + i*32 + row, cols wc*32 + (8g + 4hh .. +3); transposed through LDS into whole
  // 64-byte (fp16) / 128-byte (fp32 partial) row segments
  typedef _Float16 half4 __attribute__((ext_vector_type(4)));
  constexpr int BASE = ACT & 15;
  constexpr bool RS = (ACT & 16) != 0, RES = (ACT & 32) != 0;
  if (BASE == 2) {
    constexpr int ST = 32 * 4 + 16;
    char* ep = smem + wave * 64 * ST;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *(float4*)(ep + (i * 32 + row) * ST + (8 * g + 4 * hh) * 4) =
            make_float4(acc[i][4 * g], acc[i][4 * g + 1], acc[i][4 * g + 2], acc[i][4 * g + 3]);
    __builtin_amdgcn_wave_barrier();
    const int r8 = lane >> 3, c = lane & 7;
    float* P = (float*)C + ((size_t)blockIdx.y * M + tm * 128 + wr * 64) * N + tn * 128 + wc * 32 + c * 4;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int r = t * 8 + r8;
      *(float4*)(P + (size_t)r * N) = *(const float4*)(ep + r * ST + c * 16);
    }
  } else if (BASE == 5) {  // gelu(acc·ra·rw + bias)·sg -> (hi, lo) halves -> split image [lo | hi | hi], row stride 3N (as the
                           // 256 x 256 kernel's ACT 5: FFN1 of the fp32-class encoder at 768..1024 tokens, one tile per CU)
    constexpr int ST = 64 + 64 + 16;     // a staging row: 64 bytes of hi | 64 bytes of lo
    char* ep = smem + wave * 64 * ST;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const size_t rg = (size_t)tm * 128 + wr * 64 + i * 32 + row;
      const float ra_r = fx.ra[rg], sg_r = fx.sg[rg];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int nl = 8 * g + 4 * hh;
        const int col0 = tn * 128 + wc * 32 + nl;
        const float4 w4 = *(const float4*)(fx.rw + col0), b4 = *(const float4*)(fx.bias + col0);
        const float wv[4] = {w4.x, w4.y, w4.z, w4.w}, bv[4] = {b4.x, b4.y, b4.z, b4.w};
        half4 hi4, lo4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float v = acc[i][4 * g + e] * ra_r * wv[e] + bv[e];
          const float x = gemm_gelu_libm(v) * sg_r;
          const half_t hi = (half_t)x;
          hi4[e] = hi;
          lo4[e] = (half_t)(x - (float)hi);
        }
        char* dst = ep + (i * 32 + row) * ST + nl * 2;
        *(half4*)dst = hi4;
        *(half4*)(dst + 64) = lo4;
      }
    }
    __builtin_amdgcn_wave_barrier();
    const int r8 = lane >> 3, c = lane & 7;
    const size_t ld3 = (size_t)3 * N;
    half_t* Cw = C + (size_t)(tm * 128 + wr * 64) * ld3 + tn * 128 + wc * 32;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int r = t * 8 + r8;
      const uint4 v = *(const uint4*)(ep + r * ST + c * 16);
      half_t* rowp = Cw + (size_t)r * ld3;
      if (c < 4) {            // hi: stored twice
        *(uint4*)(rowp + N + c * 8) = v;
        *(uint4*)(rowp + 2 * (size_t)N + c * 8) = v;
      } else {                // lo
        *(uint4*)(rowp + (c - 4) * 8) = v;
      }
    }
  } else if (BASE == 6) {  // scores against per-column thresholds -> candidate lists, nothing stored (as the 256 x 256 kernel's ACT 6;
                           // wide.hip's chunks of at most one 256 x 256 tile per CU: four of these tiles per CU pipeline, one does not)
    // register (i, 4g + e): rows wr*64 + i*32 + row, column wc*32 + 8g + 4hh + e — one query per half-wave (hh), 32 rows across it;
    // counted first, one atomic per column and half-wave, then written (see the 256 x 256 kernel's ACT 6)
    const unsigned long long grp_mask = hh ? 0xFFFFFFFF00000000ull : 0x00000000FFFFFFFFull;
    const int leader = 32 * hh;
    const uint32_t shard = blockIdx.x & (fx.shards - 1u), cap_s = fx.cap / fx.shards;
    float t16[4][4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 t4 = *(const float4*)(fx.thr + tn * 128 + wc * 32 + 8 * g + 4 * hh);
      t16[g][0] = t4.x; t16[g][1] = t4.y; t16[g][2] = t4.z; t16[g][3] = t4.w;
    }
    uint32_t tot16[4][4], base16[4][4];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        uint32_t tot = 0;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const bool hit = (uint32_t)(tm * 128 + wr * 64 + i * 32 + row) < fx.n_valid && acc[i][4 * g + e] >= t16[g][e];
          tot += (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(hit) & grp_mask);
        }
        tot16[g][e] = tot;
        base16[g][e] = 0;
        if (tot && lane == leader)
          base16[g][e] = atomicAdd(&fx.count[(uint32_t)(tn * 128 + wc * 32 + 8 * g + 4 * hh + e) * fx.shards + shard], tot);
      }
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (tot16[g][e] == 0) continue;
        const uint32_t q = (uint32_t)(tn * 128 + wc * 32 + 8 * g + 4 * hh + e);
        uint32_t at = (uint32_t)__builtin_amdgcn_ds_bpermute(leader << 2, (int)base16[g][e]);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const uint32_t r = (uint32_t)(tm * 128 + wr * 64 + i * 32 + row);
          const float sc = acc[i][4 * g + e];
          const bool hit = r < fx.n_valid && sc >= t16[g][e];
          const unsigned long long grp = __builtin_amdgcn_ballot_w64(hit) & grp_mask;
          if (hit) {
            const uint32_t pos = at + (uint32_t)__builtin_popcountll(grp & ((1ull << lane) - 1ull));
            if (pos < cap_s) fx.cand[(size_t)q * fx.cap + (size_t)shard * cap_s + pos] = rarc_candkey(sc, fx.row0 + r);
            else atomicOr(&fx.status[q], RARC_Q_OVERFLOW | RARC_Q_WHY_SEGMENT);
          }
          at += (uint32_t)__builtin_popcountll(grp);
        }
      }
  } else {
    constexpr int ST = 32 * 2 + 16;
    char* ep = smem + wave * 64 * ST;
    half8 oldv[4];
    if constexpr (RES) {
      const half_t* Co = C + (size_t)(tm * 128 + wr * 64 + (lane >> 2)) * N + tn * 128 + wc * 32 + (lane & 3) * 8;
#pragma unroll
      for (int t = 0; t < 4; ++t) oldv[t] = *(const half8*)(Co + (size_t)(t * 16) * N);
    }
    float rs[2] = {1.f, 1.f};
    if constexpr (RS) {
#pragma unroll
      for (int i = 0; i < 2; ++i) rs[i] = ((const float*)bias)[(size_t)tm * 128 + wr * 64 + i * 32 + row];
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int nl = 8 * g + 4 * hh;
        half4 b4 = {0, 0, 0, 0};
        if constexpr (!RS) b4 = *(const half4*)(bias + tn * 128 + wc * 32 + nl);
        half4 out;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float v = acc[i][4 * g + e] * rs[i] + (float)b4[e];
          if (BASE == 1) v = rarc_gelu_erf(v);
          out[e] = (half_t)v;
        }
        *(half4*)(ep + (i * 32 + row) * ST + nl * 2) = out;
      }
    }
    __builtin_amdgcn_wave_barrier();
    const int r16 = lane >> 2, c = lane & 3;
    half_t* Cw = C + (size_t)(tm * 128 + wr * 64) * N + tn * 128 + wc * 32 + c * 8;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int r = t * 16 + r16;
      if constexpr ((ACT & 64) != 0)
        gemm_store8_ss(Cw + (size_t)r * N, *(const uint4*)(ep + r * ST + c * 16), oldv[t], ssq,
                       (size_t)(tm * 128 + wr * 64 + r) * (N / 32) + tn * 4 + wc, lane);
      else
        gemm_store8(Cw + (size_t)r * N, *(const uint4*)(ep + r * ST + c * 16), RES, oldv[t]);
    }
  }
}

// ------------------------------------------------------------------------------------------
// LayerNorm helpers: one wave per row, fp32 statistics, H <= 1024 (H multiple of 64)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// A lane owns the 8-column chunks lane and lane + 64 of its row (H <= 1024): 16-byte accesses, all in registers.
typedef _Float16 half8v __attribute__((ext_vector_type(8)));
struct LnRow { float x[2][8]; };

__device__ __forceinline__ void ln_row(LnRow& r, int H, const half_t* gamma, const half_t* beta, float eps, half_t* out,
                                       int lane) {
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int e = 0; e < 8; ++e) s += r.x[i][e];  // chunks past H hold zeros
  const float mean = wave_sum(s) / (float)H;
  float v = 0.f;
#pragma unroll
  for (int i = 0; i < 2; ++i)
    if ((lane + 64 * i) * 8 < H) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float d = r.x[i][e] - mean; v += d * d; }
    }
  const float rstd = rsqrtf(wave_sum(v) / (float)H + eps);
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = (lane + 64 * i) * 8;
    if (c < H) {
      const half8v g = *(const half8v*)(gamma + c), b = *(const half8v*)(beta + c);
      half8v o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (half_t)((r.x[i][e] - mean) * rstd * (float)g[e] + (float)b[e]);
      *(half8v*)(out + c) = o;
    }
  }
}

__global__ __launch_bounds__(256) void rarc_embed_ln_kernel(const int32_t* ids, const half_t* word, const half_t* pos,
                                                            const half_t* type0, const half_t* gamma,
                                                            const half_t* beta, float eps, int n_tokens, int L, int H,
                                                            int vocab, half_t* out) {
  const int lane = threadIdx.x & 63;
  const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (t >= n_tokens) return;
  int id = ids[t];  // ids outside [0, vocab) never index outside the table (the host binding rejects them)
  id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
  const half_t* w = word + (size_t)id * H;
  const half_t* p = pos + (size_t)(t % L) * H;
  LnRow r;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = (lane + 64 * i) * 8;
    if (c < H) {
      const half8v a = *(const half8v*)(w + c), b = *(const half8v*)(p + c), ty = *(const half8v*)(type0 + c);
#pragma unroll
      for (int e = 0; e < 8; ++e) r.x[i][e] = (float)a[e] + (float)b[e] + (float)ty[e];
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) r.x[i][e] = 0.f;
    }
  }
  ln_row(r, H, gamma, beta, eps, out + (size_t)t * H, lane);
}

// out = LayerNorm(x + resid); with n_parts > 0, x is instead fp16(bias + the sum of n_parts fp32 split-K partial
// products parts[p][n_rows][H]), summed in the fixed order p = 0, 1, ...
__global__ __launch_bounds__(256) void rarc_add_ln_kernel(const half_t* x_in, const float* parts, int n_parts,
                                                          const half_t* bias, const half_t* resid, const half_t* gamma,
                                                          const half_t* beta, float eps, int n_rows, int H,
                                                          half_t* out) {
  const int lane = threadIdx.x & 63;
  const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (t >= n_rows) return;
  LnRow r;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = (lane + 64 * i) * 8;
    if (c < H) {
      const half8v b = *(const half8v*)(resid + (size_t)t * H + c);
      if (n_parts > 0) {
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = 0.f;
        for (int p = 0; p < n_parts; ++p) {
          const float4* src = (const float4*)(parts + ((size_t)p * n_rows + t) * H + c);
          const float4 lo = src[0], hi = src[1];
          acc[0] += lo.x; acc[1] += lo.y; acc[2] += lo.z; acc[3] += lo.w;
          acc[4] += hi.x; acc[5] += hi.y; acc[6] += hi.z; acc[7] += hi.w;
        }
        const half8v bs = *(const half8v*)(bias + c);
#pragma unroll
        for (int e = 0; e < 8; ++e) r.x[i][e] = (float)(half_t)(acc[e] + (float)bs[e]) + (float)b[e];
      } else {
        const half8v a = *(const half8v*)(x_in + (size_t)t * H + c);
#pragma unroll
        for (int e = 0; e < 8; ++e) r.x[i][e] = (float)a[e] + (float)b[e];
      }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) r.x[i][e] = 0.f;
    }
  }
  ln_row(r, H, gamma, beta, eps, out + (size_t)t * H, lane);
}

// ------------------------------------------------------------------------------------------
// Attention: qkv [n_seq*L][3H] (q | k | v, heads contiguous inside each), ctx [n_seq*L][H].
// One workgroup per (sequence, head, block of 64 query rows); wave w owns rows w, w+4, ... (16 of
// them) and keeps their online-softmax state in registers (running max, running sum, and the output
// accumulator with lane <-> output dimension).  Keys/values stream through LDS in 64-key tiles, each
// loaded ONCE per workgroup; for the scores lane <-> key of the tile.
// ------------------------------------------------------------------------------------------
template <int DH>
__global__ __launch_bounds__(256) void rarc_attention_kernel(const half_t* qkv, const int32_t* lens, int L, int H,
                                                             int n_heads, int q_blocks, half_t* ctx) {
  constexpr int KT = 64, RPW = 16;  // keys per tile, query rows per wave
  __shared__ float ks[KT][DH + 1];
  __shared__ float vs[KT][DH + 1];
  __shared__ float qs[64][DH + 1];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int qb = blockIdx.x % q_blocks, bh = blockIdx.x / q_blocks;
  const int b = bh / n_heads, hd = bh % n_heads;
  const int len = lens[b] < 1 ? 1 : (lens[b] > L ? L : lens[b]);
  const size_t base = (size_t)b * L * 3 * H;
  const float scale = DH == 64 ? 0.125f : 0.17677669529663687f;  // 1/sqrt(DH)
  const int q0 = qb * 64;
  for (int i = threadIdx.x; i < 64 * DH; i += 256) {
    const int r = i / DH, c = i % DH;
    qs[r][c] = (q0 + r < L) ? (float)qkv[base + (size_t)(q0 + r) * 3 * H + hd * DH + c] : 0.f;
  }
  float m[RPW], l[RPW], o[RPW];
#pragma unroll
  for (int r = 0; r < RPW; ++r) { m[r] = -INFINITY; l[r] = 0.f; o[r] = 0.f; }
  for (int k0 = 0; k0 < len; k0 += KT) {
    __syncthreads();
    for (int i = threadIdx.x; i < KT * DH; i += 256) {
      const int kr = i / DH, kc = i % DH;
      const int kj = k0 + kr;
      float kv = 0.f, vv = 0.f;
      if (kj < len) {
        kv = (float)qkv[base + (size_t)kj * 3 * H + H + hd * DH + kc];
        vv = (float)qkv[base + (size_t)kj * 3 * H + 2 * H + hd * DH + kc];
      }
      ks[kr][kc] = kv;
      vs[kr][kc] = vv;
    }
    __syncthreads();
    const bool klive = k0 + lane < len;
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
      const int qr = 4 * r + wave;  // row inside the 64-row block
      if (q0 + qr >= L) continue;   // wave-uniform
      float s = -INFINITY;
      if (klive) {
        float a = 0.f;
#pragma unroll 16
        for (int d = 0; d < DH; ++d) a = __builtin_fmaf(qs[qr][d], ks[lane][d], a);
        s = a * scale;
      }
      float tmax = s;
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) tmax = fmaxf(tmax, __shfl_xor(tmax, off, 64));
      const float mnew = fmaxf(m[r], tmax);
      const float p = (s == -INFINITY) ? 0.f : __expf(s - mnew);
      const float corr = (m[r] == -INFINITY) ? 0.f : __expf(m[r] - mnew);
      l[r] = l[r] * corr + wave_sum(p);
      float acc = o[r] * corr;
      for (int j = 0; j < KT; ++j) acc = __builtin_fmaf(__shfl(p, j, 64), vs[j][lane & (DH - 1)], acc);
      o[r] = acc;
      m[r] = mnew;
    }
  }
#pragma unroll
  for (int r = 0; r < RPW; ++r) {
    const int qi = q0 + 4 * r + wave;
    if (qi < L && lane < DH) ctx[((size_t)b * L + qi) * H + hd * DH + lane] = (half_t)(l[r] > 0.f ? o[r] / l[r] : 0.f);
  }
}

// ------------------------------------------------------------------------------------------
// MFMA attention: one wave per (sequence, head, block of 32 query rows); keys/values in tiles of 32.
//   S^T = K · Q^T      v_mfma_f32_32x32x16_f16, A = K rows, B = Q rows (both 16-byte row loads from
//                      the fused qkv matrix).  "Swapped": lane l then owns QUERY column l&31 and 16
//                      of the tile's 32 keys, so the softmax statistics of a query live in one lane
//                      pair (l, l^32) — max / sum are 15 in-lane ops and one cross-lane exchange.
//   O^T += V^T · P^T   A = V^T fragments read from an LDS image the loader writes transposed
//                      ([d][key], 80-byte rows), B = P^T assembled in registers: a lane keeps the
//                      fp16 pairs of its own keys and swaps the other half with lane l^32.
// Online softmax across key tiles (running max / sum per query, accumulator rescaled per lane).
// Keys >= lens[seq] are masked; rows >= seq_len are neither loaded past the end nor stored.
// ------------------------------------------------------------------------------------------
template <int DH, bool REL>   // REL: scores get the relative-position bias rel[head][key - query + rel_span - 1] (MPNet)
__global__ __launch_bounds__(256) void rarc_attention_mfma_kernel(const half_t* __restrict__ qkv,
                                                                  const int32_t* __restrict__ lens, int L, int H,
                                                                  int n_heads, int q_blocks, int n_units,
                                                                  half_t* __restrict__ ctx, const float* __restrict__ rel,
                                                                  int rel_span) {
  constexpr int KS = DH / 16;  // k-steps of the QK^T product
  constexpr int MB = DH / 32;  // 32-row blocks of O^T
  constexpr int VROW = 40;     // halves per row of the transposed V image (32 keys + 8 padding)
  __shared__ __attribute__((aligned(16))) half_t vt_all[4][DH * VROW];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int unit = blockIdx.x * 4 + wave;
  if (unit >= n_units) return;  // (no block-level barrier below: waves are independent)
  half_t* vt = vt_all[wave];
  const int qb = unit % q_blocks, bh = unit / q_blocks;
  const int b = bh / n_heads, hd = bh % n_heads;
  const int len = lens[b] < 1 ? 1 : (lens[b] > L ? L : lens[b]);  // never an empty softmax, never past the sequence
  const int col = lane & 31, hh = lane >> 5;
  const size_t rs = (size_t)3 * H;  // row stride of qkv in halves
  const half_t* base = qkv + (size_t)b * L * rs + hd * DH;
  const float scale = DH == 64 ? 0.125f : 0.17677669529663687f;  // 1/sqrt(DH)
  const int q0 = qb * 32;
  const int qrow = (q0 + col < L) ? q0 + col : L - 1;

  half8 qf[KS];  // B operand of S^T: this lane's query row, k = 16*ks + 8*hh ..
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) qf[ks] = *(const half8*)(base + (size_t)qrow * rs + 16 * ks + 8 * hh);

  f32x16 o[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) o[mb] = (f32x16){0};
  float m_run = -INFINITY, l_run = 0.f;

  for (int k0 = 0; k0 < len; k0 += 32) {
    // ---- S^T tile: keys k0..k0+31 (rows) x this wave's 32 queries (columns) ----
    const int krow = (k0 + col < L) ? k0 + col : L - 1;
    f32x16 st = {0};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const half8 kf = *(const half8*)(base + H + (size_t)krow * rs + 16 * ks + 8 * hh);
      st = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[ks], st, 0, 0, 0);
    }
    RARC_MFMA_SETTLE(st);
    // ---- V tile -> transposed LDS image vt[d][key] (each lane: 16-byte row pieces, written as halves) ----
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = lane; i < 32 * (DH / 8); i += 64) {
      const int kr = i / (DH / 8), c8 = i % (DH / 8);
      const int vrow = (k0 + kr < L) ? k0 + kr : L - 1;
      const half8 v = *(const half8*)(base + 2 * H + (size_t)vrow * rs + 8 * c8);
#pragma unroll
      for (int e = 0; e < 8; ++e) vt[(8 * c8 + e) * VROW + kr] = v[e];
    }
    // ---- softmax statistics of this lane's query over its 16 keys, then with the partner lane ----
    float s[16];
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
      s[r] = key < len ? v : -INFINITY;
      tmax = fmaxf(tmax, s[r]);
    }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    const float m_new = fmaxf(m_run, tmax);
    const float corr = (m_run == -INFINITY) ? 0.f : __expf(m_run - m_new);
    float psum = 0.f;
    uint32_t pk[8];  // fp16 pairs: pk[2g], pk[2g+1] = keys 8g + 4hh + {0,1}, {2,3}
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      const float p0 = (s[r] == -INFINITY) ? 0.f : __expf(s[r] - m_new);
      const float p1 = (s[r + 1] == -INFINITY) ? 0.f : __expf(s[r + 1] - m_new);
      psum += p0 + p1;
      const half2_t h2 = {(half_t)p0, (half_t)p1};
      pk[r >> 1] = __builtin_bit_cast(uint32_t, h2);
    }
    psum += __shfl_xor(psum, 32, 64);
    l_run = l_run * corr + psum;
    m_run = m_new;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[mb][r] *= corr;
    __builtin_amdgcn_wave_barrier();
    // ---- O^T += V^T · P^T over the tile's 32 keys (two k-steps of 16) ----
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      // B fragment of lane (query, hh): keys 16ks + 8hh + 0..7 = group g = 2ks + hh: keys 8g+{0..3} sit in
      // lane (query, 0) and 8g+{4..7} in lane (query, 1), both in that lane's pk[2g], pk[2g+1]
      const int gm = 2 * ks + hh, go = 2 * ks + (1 - hh);  // my group; the group my partner needs from me
      const uint32_t mine0 = hh ? (ks ? pk[6] : pk[2]) : (ks ? pk[4] : pk[0]);
      const uint32_t mine1 = hh ? (ks ? pk[7] : pk[3]) : (ks ? pk[5] : pk[1]);
      const uint32_t send0 = hh ? (ks ? pk[4] : pk[0]) : (ks ? pk[6] : pk[2]);
      const uint32_t send1 = hh ? (ks ? pk[5] : pk[1]) : (ks ? pk[7] : pk[3]);
      (void)gm; (void)go;
      const uint32_t recv0 = (uint32_t)__shfl_xor((int)send0, 32, 64), recv1 = (uint32_t)__shfl_xor((int)send1, 32, 64);
      typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
      const u32x4 bw = hh ? (u32x4){recv0, recv1, mine0, mine1} : (u32x4){mine0, mine1, recv0, recv1};
      const half8 pf = __builtin_bit_cast(half8, bw);
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
        const half8 vf = *(const half8*)(vt + (32 * mb + col) * VROW + 16 * ks + 8 * hh);
        o[mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf, o[mb], 0, 0, 0);
      }
    }
    // (the last O^T MFMA of the key loop is read — accumulators copied out of their AGPRs — right behind the loop's exit
    //  branch: a window that holds free instructions; the pad sits INSIDE the loop because the copies are placed after
    //  scheduling and would slip in front of a pad behind it.  tests/codeobj.py follows branches since round 5)
    RARC_MFMA_SETTLE(o);
  }
  // ---- store: lane (query, hh) holds d = 32mb + 8(r>>2) + 4hh + (r&3) ----
  if (q0 + col < L) {
    const float inv = l_run > 0.f ? 1.f / l_run : 0.f;
    half_t* out = ctx + ((size_t)b * L + q0 + col) * H + hd * DH;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        typedef _Float16 half4 __attribute__((ext_vector_type(4)));
        const half4 w = {(half_t)(o[mb][4 * g] * inv), (half_t)(o[mb][4 * g + 1] * inv),
                         (half_t)(o[mb][4 * g + 2] * inv), (half_t)(o[mb][4 * g + 3] * inv)};
        *(half4*)(out + 32 * mb + 8 * g + 4 * hh) = w;
      }
  }
}

// ------------------------------------------------------------------------------------------
// Round 4 — the same attention with the key / value tiles SHARED by a workgroup (sequences longer than one query block).
// Above, every wave fetches the K rows of its head straight from the q|k|v matrix (16-byte pieces of rows 3H halves apart) and
// transposes its own copy of the V tile with 2-byte LDS writes: at 512 tokens sixteen waves do that work for one (sequence,
// head).  Here one WORKGROUP (4 waves) takes four consecutive 32-query blocks of one (sequence, head); each 32-key tile is
// fetched once per workgroup — K by all four waves into a row-major image (16-byte row pad: conflict-free ds_read_b128
// fragments), V by waves 2 and 3 into the transposed image with key PAIRS packed into 4-byte writes — the raw rows of tile t+1
// travel in registers while tile t is multiplied, the images are double buffered (one barrier per tile), and the P^T
// fragments are assembled with v_permlane32_swap (decoder.hip).  Same MFMAs on the same operands in the same order, same
// softmax arithmetic: the output is bit-identical to rarc_attention_mfma_kernel's (tests/test_gpu_encoder.py).
// ------------------------------------------------------------------------------------------
template <int DH, bool REL>
__global__ __launch_bounds__(256) void rarc_attention_mfma_shared_kernel(const half_t* __restrict__ qkv,
                                                                         const int32_t* __restrict__ lens, int L, int H,
                                                                         int n_heads, int q_blocks, int q_groups,
                                                                         half_t* __restrict__ ctx, const float* __restrict__ rel,
                                                                         int rel_span) {
  constexpr int KS = DH / 16;   // k-steps of the QK^T product
  constexpr int MB = DH / 32;   // 32-row blocks of O^T
  constexpr int CH = DH / 8;    // 16-byte chunks per row
  constexpr int KROW = DH + 8;  // halves per K row (+16 bytes)
  constexpr int VROW = 40;      // halves per row of the transposed V image (32 keys + 8 padding)
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  __shared__ __attribute__((aligned(16))) half_t kimg[2][32 * KROW];
  __shared__ __attribute__((aligned(16))) half_t vt[2][DH * VROW];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int qg = blockIdx.x % q_groups, bh = blockIdx.x / q_groups;
  const int b = bh / n_heads, hd = bh % n_heads;
  const int len = lens[b] < 1 ? 1 : (lens[b] > L ? L : lens[b]);  // never an empty softmax, never past the sequence
  const int col = lane & 31, hh = lane >> 5;
  const size_t rs = (size_t)3 * H;  // row stride of qkv in halves
  const half_t* base = qkv + (size_t)b * L * rs + hd * DH;
  const float scale = DH == 64 ? 0.125f : 0.17677669529663687f;  // 1/sqrt(DH)
  const int qb = 4 * qg + wave;
  const bool live = qb < q_blocks;
  const int q0 = qb * 32;

  // staging roles: K item (row tid / CH, chunk tid % CH) for tid < 32 CH; V item (key pair p, chunk c8) for the
  // 16 CH threads from 128 on (waves 2 and 3 at head_dim 64)
  const bool k_role = tid < 32 * CH;
  const int kr = tid / CH, kc = tid % CH;
  const int vi = tid - 128;
  const bool v_role = vi >= 0 && vi < 16 * CH;
  const int vp = vi & 15, vc8 = vi >> 4;
  half8 kraw, vraw0, vraw1;
  auto prefetch = [&](int k0) {
    if (k_role) {
      const int krow = (k0 + kr < L) ? k0 + kr : L - 1;
      kraw = *(const half8*)(base + H + (size_t)krow * rs + 8 * kc);
    }
    if (v_role) {
      const int r0 = (k0 + 2 * vp < L) ? k0 + 2 * vp : L - 1, r1 = (k0 + 2 * vp + 1 < L) ? k0 + 2 * vp + 1 : L - 1;
      vraw0 = *(const half8*)(base + 2 * H + (size_t)r0 * rs + 8 * vc8);
      vraw1 = *(const half8*)(base + 2 * H + (size_t)r1 * rs + 8 * vc8);
    }
  };
  auto store_tile = [&](int buf) {
    if (k_role) *(half8*)(kimg[buf] + kr * KROW + 8 * kc) = kraw;
    if (v_role) {
#pragma unroll
      for (int e = 0; e < 8; ++e) *(half2_t*)(vt[buf] + (8 * vc8 + e) * VROW + 2 * vp) = (half2_t){vraw0[e], vraw1[e]};
    }
  };

  prefetch(0);
  half8 qf[KS];  // B operand of S^T: this lane's query row, k = 16*ks + 8*hh ..
  if (live) {
    const int qrow = (q0 + col < L) ? q0 + col : L - 1;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[ks] = *(const half8*)(base + (size_t)qrow * rs + 16 * ks + 8 * hh);
  }
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
    const half_t* kim = kimg[buf];
    const half_t* vim = vt[buf];
    f32x16 st = {0};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const half8 kf = *(const half8*)(kim + col * KROW + 16 * ks + 8 * hh);
      st = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[ks], st, 0, 0, 0);
    }
    RARC_MFMA_SETTLE(st);
    // ---- softmax statistics of this lane's query over its 16 keys, then with the partner lane ----
    float sc[16];
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
      sc[r] = key < len ? v : -INFINITY;
      tmax = fmaxf(tmax, sc[r]);
    }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    const float m_new = fmaxf(m_run, tmax);
    const float corr = (m_run == -INFINITY) ? 0.f : __expf(m_run - m_new);
    float psum = 0.f;
    uint32_t pk[8];  // fp16 pairs: pk[2g], pk[2g+1] = keys 8g + 4hh + {0,1}, {2,3}
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      const float p0 = (sc[r] == -INFINITY) ? 0.f : __expf(sc[r] - m_new);
      const float p1 = (sc[r + 1] == -INFINITY) ? 0.f : __expf(sc[r + 1] - m_new);
      psum += p0 + p1;
      const half2_t h2 = {(half_t)p0, (half_t)p1};
      pk[r >> 1] = __builtin_bit_cast(uint32_t, h2);
    }
    psum += __shfl_xor(psum, 32, 64);
    l_run = l_run * corr + psum;
    m_run = m_new;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[mb][r] *= corr;
    // ---- O^T += V^T · P^T over the tile's 32 keys (two k-steps of 16): B fragments by v_permlane32_swap ----
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const auto w0 = __builtin_amdgcn_permlane32_swap(pk[4 * ks], pk[4 * ks + 2], false, false);
      const auto w1 = __builtin_amdgcn_permlane32_swap(pk[4 * ks + 1], pk[4 * ks + 3], false, false);
      const half8 pf = __builtin_bit_cast(half8, ((u32x4){w0[0], w1[0], w0[1], w1[1]}));
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
        const half8 vf = *(const half8*)(vim + (32 * mb + col) * VROW + 16 * ks + 8 * hh);
        o[mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf, o[mb], 0, 0, 0);
      }
    }
    // (the last O^T MFMA of the key loop is read — accumulators copied out of their AGPRs — right behind the loop's exit
    //  branch: a window that holds free instructions; the pad sits INSIDE the loop because the copies are placed after
    //  scheduling and would slip in front of a pad behind it.  tests/codeobj.py follows branches since round 5)
    RARC_MFMA_SETTLE(o);
  }
  // ---- store: lane (query, hh) holds d = 32mb + 8(r>>2) + 4hh + (r&3) ----
  if (live && q0 + col < L) {
    const float inv = l_run > 0.f ? 1.f / l_run : 0.f;
    half_t* out = ctx + ((size_t)b * L + q0 + col) * H + hd * DH;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        typedef _Float16 half4 __attribute__((ext_vector_type(4)));
        const half4 w = {(half_t)(o[mb][4 * g] * inv), (half_t)(o[mb][4 * g + 1] * inv),
                         (half_t)(o[mb][4 * g + 2] * inv), (half_t)(o[mb][4 * g + 3] * inv)};
        *(half4*)(out + 32 * mb + 8 * g + 4 * hh) = w;
      }
  }
}

// CLS pooling (+ optional L2 normalisation with the canonical order of prep.hip)
__global__ __launch_bounds__(64) void rarc_pool_kernel(const half_t* hidden, int L, int H, int normalize, float* out) {
  const int lane = threadIdx.x, j = lane & 7;
  const int b = blockIdx.x;
  const half_t* x = hidden + (size_t)b * L * H;  // row of the [CLS] token
  float acc = 0.f;
  if (lane < 8)
    for (int m2 = j; m2 < H; m2 += 8) { const float f = (float)x[m2]; acc = __builtin_fmaf(f, f, acc); }
  float a[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = __shfl(acc, i, 64);
  const float nr = rarc_canon_tree(a);
  const float inv = (normalize && nr > 0.f) ? (float)(1.0 / (double)(float)sqrt((double)nr)) : 1.f;
  for (int c = lane; c < H; c += 64) out[(size_t)b * H + c] = (float)x[c] * inv;
}

// Mean pooling over the sequence's real tokens (sentence-transformers' default for all-MiniLM / gte style models):
// out[b] = sum_{t < len} hidden[b, t] / len in fp32, then the optional L2 normalisation (canonical order: the sum of
// squares of the pooled fp32 vector runs through the same 8 chains).  One workgroup per sequence.
__global__ __launch_bounds__(256) void rarc_pool_mean_kernel(const half_t* hidden, const int32_t* lens, int L, int H,
                                                             int normalize, float* out) {
  __shared__ float s_vec[1024];
  __shared__ float s_nr;
  const int b = blockIdx.x, tid = threadIdx.x;
  const int len = lens[b] < 1 ? 1 : (lens[b] > L ? L : lens[b]);
  const half_t* x = hidden + (size_t)b * L * H;
  for (int c = tid; c < H; c += 256) {
    float acc = 0.f;
    for (int t = 0; t < len; ++t) acc += (float)x[(size_t)t * H + c];   // token order: fixed, sequential
    s_vec[c] = acc / (float)len;
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
static int gemm_attrs() {
  constexpr int lds_small = 2 * (128 * GK * 2 + GN * GK * 2), lds_big = 3 * (256 * GK * 2 + GN * GK * 2);
  constexpr int lds_deep = 4 * (128 * GK * 2 + GN * GK * 2);
  static RarcPerDevice attr_dev;
  size_t& attr = attr_dev.cur();
  if (attr) return RARC_OK;
  RARC_HIP_CHECK(hipFuncSetAttribute((const void*)rarc_gemm_f16_kernel<0, 128, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_small));
  RARC_HIP_CHECK(hipFuncSetAttribute((const void*)rarc_gemm_f16_kernel<1, 128, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_small));
  RARC_HIP_CHECK(hipFuncSetAttribute((const void*)rarc_gemm_f16_kernel<0, 256, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_big));
  RARC_HIP_CHECK(hipFuncSetAttribute((const void*)rarc_gemm_f16_kernel<1, 256, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_big));
  RARC_HIP_CHECK(hipFuncSetAttribute((const void*)rarc_gemm_f16_kernel<0, 128, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_deep));
  RARC_HIP_CHECK(hipFuncSetAttribute((const void*)rarc_gemm_f16_kernel<1, 128, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_deep));
  RARC_HIP_CHECK(hipFuncSetAttribute((const void*)rarc_gemm_f16_kernel<2, 128, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_deep));
  RARC_HIP_CHECK(hipFuncSetAttribute((const void*)rarc_gemm256_f16_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, G256_LDS));
  RARC_HIP_CHECK(hipFuncSetAttribute((const void*)rarc_gemm256_f16_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, G256_LDS));
  RARC_HIP_CHECK(hipFuncSetAttribute((const void*)rarc_gemm256_f16_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, G256_LDS));
  RARC_HIP_CHECK(hipFuncSetAttribute((const void*)rarc_gemm256_f16_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, G256_LDS));
  RARC_HIP_CHECK(hipFuncSetAttribute((const void*)rarc_gemm256_f16_kernel<5>, hipFuncAttributeMaxDynamicSharedMemorySize, G256_LDS));
  RARC_HIP_CHECK(hipFuncSetAttribute((const void*)rarc_gemm256_f16_kernel<6>, hipFuncAttributeMaxDynamicSharedMemorySize, G256_LDS));
  RARC_HIP_CHECK(hipFuncSetAttribute((const void*)rarc_gemm256_f16_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, G256_LDS));
  RARC_HIP_CHECK(hipFuncSetAttribute((const void*)rarc_gemm256_f16_kernel<19>, hipFuncAttributeMaxDynamicSharedMemorySize, G256_LDS));
  RARC_HIP_CHECK(hipFuncSetAttribute((const void*)rarc_gemm256_f16_kernel<32>, hipFuncAttributeMaxDynamicSharedMemorySize, G256_LDS));
  RARC_HIP_CHECK(hipFuncSetAttribute((const void*)rarc_gemm256x128_f16_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, G128_LDS));
  RARC_HIP_CHECK(hipFuncSetAttribute((const void*)rarc_gemm256x128_f16_kernel<19>, hipFuncAttributeMaxDynamicSharedMemorySize, G128_LDS));
  RARC_HIP_CHECK(hipFuncSetAttribute((const void*)rarc_gemm256x128_f16_kernel<32>, hipFuncAttributeMaxDynamicSharedMemorySize, G128_LDS));
  RARC_HIP_CHECK(hipFuncSetAttribute((const void*)rarc_gemm128pp_f16_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, G128S_LDS));
  RARC_HIP_CHECK(hipFuncSetAttribute((const void*)rarc_gemm128pp_f16_kernel<32>, hipFuncAttributeMaxDynamicSharedMemorySize, G128S_LDS));
  RARC_HIP_CHECK(hipFuncSetAttribute((const void*)rarc_gemm256_f16_kernel<96>, hipFuncAttributeMaxDynamicSharedMemorySize, G256_LDS));
  RARC_HIP_CHECK(hipFuncSetAttribute((const void*)rarc_gemm256x128_f16_kernel<96>, hipFuncAttributeMaxDynamicSharedMemorySize, G128_LDS));
  RARC_HIP_CHECK(hipFuncSetAttribute((const void*)rarc_gemm128pp_f16_kernel<96>, hipFuncAttributeMaxDynamicSharedMemorySize, G128S_LDS));
  RARC_HIP_CHECK(hipFuncSetAttribute((const void*)rarc_gemm256s_f16_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, G256S_LDS));
  RARC_HIP_CHECK(hipFuncSetAttribute((const void*)rarc_gemm256s_f16_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, G256S_LDS));
  RARC_HIP_CHECK(hipFuncSetAttribute((const void*)rarc_gemm256x128_f16_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, G128_LDS));
  RARC_HIP_CHECK(hipFuncSetAttribute((const void*)rarc_gemm256x128_f16_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, G128_LDS));
  RARC_HIP_CHECK(hipFuncSetAttribute((const void*)rarc_gemm256x128_f16_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, G128_LDS));
  RARC_HIP_CHECK(hipFuncSetAttribute((const void*)rarc_gemm256x128_f16_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, G128_LDS));
  RARC_HIP_CHECK(hipFuncSetAttribute((const void*)rarc_gemm128pp_f16_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, G128S_LDS));
  RARC_HIP_CHECK(hipFuncSetAttribute((const void*)rarc_gemm128pp_f16_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, G128S_LDS));
  RARC_HIP_CHECK(hipFuncSetAttribute((const void*)rarc_gemm128pp_f16_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, G128S_LDS));
  RARC_HIP_CHECK(hipFuncSetAttribute((const void*)rarc_gemm128pp_f16_kernel<5>, hipFuncAttributeMaxDynamicSharedMemorySize, G128S_LDS));
  RARC_HIP_CHECK(hipFuncSetAttribute((const void*)rarc_gemm128pp_f16_kernel<6>, hipFuncAttributeMaxDynamicSharedMemorySize, G128S_LDS));
  attr = 1;
  return RARC_OK;
}

// Split-K factor for a GEMM that feeds a LayerNorm: with few output tiles (small batches) the k loop is cut into
// S slices so that about one workgroup per CU is in flight; each slice keeps >= 8 k-tiles.
static int enc_split(int m, int n, int k) {
  const int tiles = (m / GM) * (n / GN);
  int S = 1;
  while (S < 4 && tiles * S * 2 <= 256 && k % (S * 2 * GK) == 0 && k / (S * 2) >= 8 * GK) S *= 2;
  return S;
}

static int enc_gemm_splitk(const uint16_t* d_a, const uint16_t* d_w, float* d_parts, int m, int n, int k, int S,
                           hipStream_t s) {
  if (int rc = gemm_attrs()) return rc;
  constexpr size_t lds_deep = 4 * (128 * GK * 2 + GN * GK * 2);
  static const bool nopp = getenv("RARC_GEMM_PP") && atoi(getenv("RARC_GEMM_PP")) == 0;
  if (!nopp && (k / S) >= 4 * GK)
    hipLaunchKernelGGL((rarc_gemm128pp_f16_kernel<2>), dim3((m / GM) * (n / GN), S), dim3(512), G128S_LDS, s,
                       (const half_t*)d_a, (const half_t*)d_w, (const half_t*)nullptr, (half_t*)d_parts, m, n, k / S, k,
                       m < n ? 1 : 0);
  else
  hipLaunchKernelGGL((rarc_gemm_f16_kernel<2, 128, 4>), dim3((m / GM) * (n / GN), S), dim3(256), lds_deep, s,
                     (const half_t*)d_a, (const half_t*)d_w, (const half_t*)nullptr, (half_t*)d_parts, m, n, k / S, k, m < n ? 1 : 0);
  RARC_HIP_CHECK(hipGetLastError());
  return RARC_OK;
}

// 256 x 256 or 256 x 128 tiles?  One workgroup per CU either way, so the time is rounds x time per round, and a
// 256 x 256 tile takes about 1.5x a 256 x 128 one for twice the area (measured 1.44x at K = 3072): the larger tile
// wins whenever its ragged last round costs less than that (51 200 x 1024 x 3072: 800 tiles in 4 rounds, 339 us,
// against 1600 tiles in 7 rounds, 415 us; 8192 x 3072 x 1024: 384 tiles in 2 rounds ties with 768 in 3 and stays).
static bool gemm_prefers_256x256(int t256, int t128) {
  constexpr int CUS = 256;
  if (t256 < CUS) return false;
  return 3 * ((t256 + CUS - 1) / CUS) < 2 * ((t128 + CUS - 1) / CUS);
}

// true when rarc_enc_gemm(m, n, k) runs one of the 256-row ping-pong kernels, i.e. when act = 3 (fused SwiGLU) is available
bool rarc_gemm_swiglu_fused(int m, int n, int k) {
  static const int force = getenv("RARC_GEMM_PP") ? atoi(getenv("RARC_GEMM_PP")) : -1;
  if (force == 0 || m <= 0 || m % 256 || n % GN || k % GK) return false;
  const int t256 = n % 256 == 0 ? (m / 256) * (n / 256) : 0, t128 = (m / 256) * (n / GN);
  if (force != 1 && gemm_prefers_256x256(t256, t128)) return true;
  return t128 >= 256 && k >= 3 * GK;
}

// true when rarc_enc_gemm(m, n, k) AND its cut-off tail run kernels that have the row-scale / residual epilogues (act | 16,
// act | 32): the 256-row ping-pong kernels for the bulk, the 128 x 128 ping-pong kernel for a tail
bool rarc_gemm_norm_fusable(int m, int n, int k) {
  static const int force = getenv("RARC_GEMM_PP") ? atoi(getenv("RARC_GEMM_PP")) : -1;
  if (force == 0 || m <= 0 || m % 256 || n % GN || k % GK || k < 4 * GK) return false;
  const int t256 = n % 256 == 0 ? (m / 256) * (n / 256) : 0, t128 = (m / 256) * (n / GN);
  return (force != 1 && gemm_prefers_256x256(t256, t128)) || t128 >= 256;
}

// act 0 / 1 / 3 as rarc_enc_gemm; act 4: raw fp32 products, no bias, d_c is float [M][N] (encoder_f32.hip)
// zero_bias: the caller guarantees d_bias holds zeros (the reranker LM's projections) — the seamless 256 x 256 kernel, which
// has no bias path, may take the large shapes
static int enc_gemm_impl(const uint16_t* d_a, const uint16_t* d_w, const uint16_t* d_bias, uint16_t* d_c, int m, int n, int k,
                         int act, void* stream, bool zero_bias = false, float* d_ssq = nullptr, bool is_tail = false) {
  hipStream_t s = (hipStream_t)stream;
  const half_t *a = (const half_t*)d_a, *w = (const half_t*)d_w, *bs = (const half_t*)d_bias;
  half_t* c = (half_t*)d_c;
  // 256-row tiles (3-stage pipeline, one workgroup per CU) when M allows it and there are enough tiles to
  // fill the chip; the 128-row kernel otherwise
  const bool big = (m % 256 == 0) && ((m / 256) * (n / GN) >= 256);
  // XCD-aware grouped tile order (gemm_tile_of); RARC_GEMM_SWZ=0: the plain walk, the smaller operand crossing the fabric 8 times
  static const bool swz = !(getenv("RARC_GEMM_SWZ") && atoi(getenv("RARC_GEMM_SWZ")) == 0);
  const int order = swz ? 2 : (m < n ? 1 : 0);
  // few tiles (small batches): at most one workgroup per CU, nothing else to hide the HBM/L2 latency of the
  // k loop behind -> four stages (three tiles in flight) instead of two
  const bool deep = !big && ((m / GM) * (n / GN) <= 256) && k >= 4 * GK;
  constexpr size_t lds_small = 2 * (128 * GK * 2 + GN * GK * 2), lds_big = 3 * (256 * GK * 2 + GN * GK * 2);
  constexpr size_t lds_deep = 4 * (128 * GK * 2 + GN * GK * 2);
  if (int rc = gemm_attrs()) return rc;
  // ping-pong kernels for the big shapes, 256 x 256 or 256 x 128 tiles by gemm_prefers_256x256; the older kernels
  // below serve small batches and odd shapes
  static const int force = getenv("RARC_GEMM_PP") ? atoi(getenv("RARC_GEMM_PP")) : -1;  // 0 off, 1 = 256x128 only
  if (force != 0 && m % 256 == 0) {
    const int t256 = n % 256 == 0 ? (m / 256) * (n / 256) : 0, t128 = (m / 256) * (n / GN);
    if (force != 1 && gemm_prefers_256x256(t256, t128)) {
      static const bool persist = !(getenv("RARC_GEMM_PERSIST") && atoi(getenv("RARC_GEMM_PERSIST")) == 0);
      // A short last round is given away: when the tiles beyond the last whole round of 256 are at most a quarter
      // round (51 200 x 1024: 800 tiles = 3 rounds + 32), the whole rounds run here on the leading rows and the
      // trailing rows go through the dispatcher again as a problem of their own — 128 x 128 tiles on half the CUs
      // finish those 32 tiles' worth in 0.4 of a tile time instead of holding 224 CUs idle for a whole one.
      // (The cut is by rows, so the whole rounds must cover whole tile rows: a multiple of lcm(256, tiles_n) tiles.)
      static const bool cut_tail = !(getenv("RARC_GEMM_TAIL") && atoi(getenv("RARC_GEMM_TAIL")) == 0);
      const int tiles_n = n / 256;
      int lcm = tiles_n;
      while (lcm % 256) lcm += tiles_n;
      const int full = t256 / lcm * lcm, rem = t256 - full;
      int m_main = m;
      // (round 3: up to HALF a round is cut off — between a quarter and a half round the tail runs as ONE round of 256 x 128
      //  tiles, 2 rem <= 256 workgroups, about two thirds of a 256 x 256 round; that kernel has every epilogue)
      if (cut_tail && persist && full > 0 && rem > 0 && 2 * rem <= 256 &&
          (4 * rem <= 256 ? ((act & 15) != 3 || rarc_gemm_swiglu_fused(m - full / tiles_n * 256, n, k)) : k >= 3 * GK))
        m_main = full / tiles_n * 256;
      const int t_main = (m_main / 256) * tiles_n;
      const int g256 = persist && t_main > 256 ? 256 : t_main;
      // The seamless stream of k tiles (rarc_gemm256s_f16_kernel) measured within +-2 % of the kernel with a seam at
      // K >= 1024 and +3..6 % at K = 256 / 512 (profiles/r03_gemm_seamless.txt): it takes the short streams, where the seam
      // is a larger share; RARC_GEMM_SEAM=1 / 0 forces it on / off for every eligible shape (A/B runs, tests).
      const char* seam_s = getenv("RARC_GEMM_SEAM");   // (read per call: a test switches it inside one process)
      const int seam_env = seam_s ? atoi(seam_s) : -1;
      const bool seamless = seam_env < 0 ? k <= 8 * GK : seam_env != 0;
      if (seamless && zero_bias && (act == 0 || act == 3) && k >= 4 * GK) {
        if (act == 3) hipLaunchKernelGGL((rarc_gemm256s_f16_kernel<3>), dim3(g256), dim3(512), G256S_LDS, s, a, w, c, m_main, n, k, order);
        else hipLaunchKernelGGL((rarc_gemm256s_f16_kernel<0>), dim3(g256), dim3(512), G256S_LDS, s, a, w, c, m_main, n, k, order);
      } else
      if (act == 4) hipLaunchKernelGGL((rarc_gemm256_f16_kernel<4>), dim3(g256), dim3(512), G256_LDS, s, a, w, bs, c, m_main, n, k, order);
      else if (act == 3) hipLaunchKernelGGL((rarc_gemm256_f16_kernel<3>), dim3(g256), dim3(512), G256_LDS, s, a, w, bs, c, m_main, n, k, order);
      else if (act == 1) hipLaunchKernelGGL((rarc_gemm256_f16_kernel<1>), dim3(g256), dim3(512), G256_LDS, s, a, w, bs, c, m_main, n, k, order);
      else if (act == 16) hipLaunchKernelGGL((rarc_gemm256_f16_kernel<16>), dim3(g256), dim3(512), G256_LDS, s, a, w, bs, c, m_main, n, k, order);
      else if (act == 19) hipLaunchKernelGGL((rarc_gemm256_f16_kernel<19>), dim3(g256), dim3(512), G256_LDS, s, a, w, bs, c, m_main, n, k, order);
      else if (act == 32) hipLaunchKernelGGL((rarc_gemm256_f16_kernel<32>), dim3(g256), dim3(512), G256_LDS, s, a, w, bs, c, m_main, n, k, order);
      else if (act == 96) hipLaunchKernelGGL((rarc_gemm256_f16_kernel<96>), dim3(g256), dim3(512), G256_LDS, s, a, w, bs, c, m_main, n, k, order, d_ssq);
      else hipLaunchKernelGGL((rarc_gemm256_f16_kernel<0>), dim3(g256), dim3(512), G256_LDS, s, a, w, bs, c, m_main, n, k, order);
      RARC_HIP_CHECK(hipGetLastError());
      if (m_main < m)   // (d_c counts 2-byte elements: an fp32 row is 2n of them)
        return enc_gemm_impl(d_a + (size_t)m_main * k, d_w, (act & 16) ? d_bias + (size_t)2 * m_main : d_bias,   // (RS: `bias` is float rowscale[M])
                             d_c + (size_t)m_main * ((act & 15) == 3 ? n / 2 : (act == 4 ? 2 * n : n)), m - m_main, n, k, act, stream, zero_bias,
                             d_ssq ? d_ssq + (size_t)m_main * (n / 32) : nullptr, true);
      return RARC_OK;
    }
    if ((t128 >= 256 || (is_tail && t128 > 128)) && k >= 3 * GK) {
      if (act == 4) hipLaunchKernelGGL((rarc_gemm256x128_f16_kernel<4>), dim3(t128), dim3(512), G128_LDS, s, a, w, bs, c, m, n, k, order);
      else if (act == 3) hipLaunchKernelGGL((rarc_gemm256x128_f16_kernel<3>), dim3(t128), dim3(512), G128_LDS, s, a, w, bs, c, m, n, k, order);
      else if (act == 1) hipLaunchKernelGGL((rarc_gemm256x128_f16_kernel<1>), dim3(t128), dim3(512), G128_LDS, s, a, w, bs, c, m, n, k, order);
      else if (act == 16) hipLaunchKernelGGL((rarc_gemm256x128_f16_kernel<16>), dim3(t128), dim3(512), G128_LDS, s, a, w, bs, c, m, n, k, order);
      else if (act == 19) hipLaunchKernelGGL((rarc_gemm256x128_f16_kernel<19>), dim3(t128), dim3(512), G128_LDS, s, a, w, bs, c, m, n, k, order);
      else if (act == 32) hipLaunchKernelGGL((rarc_gemm256x128_f16_kernel<32>), dim3(t128), dim3(512), G128_LDS, s, a, w, bs, c, m, n, k, order);
      else if (act == 96) hipLaunchKernelGGL((rarc_gemm256x128_f16_kernel<96>), dim3(t128), dim3(512), G128_LDS, s, a, w, bs, c, m, n, k, order, d_ssq);
      else hipLaunchKernelGGL((rarc_gemm256x128_f16_kernel<0>), dim3(t128), dim3(512), G128_LDS, s, a, w, bs, c, m, n, k, order);
      RARC_HIP_CHECK(hipGetLastError());
      return RARC_OK;
    }
  }
  RARC_REQUIRE((act & 15) != 3, RARC_E_UNSUPPORTED, "rarc_enc_gemm: the fused SwiGLU epilogue needs a shape the 256-row kernels take "
               "(ask rarc_gemm_swiglu_fused first)");
  RARC_REQUIRE(!(act & 112) || (deep && force != 0 && k >= 4 * GK), RARC_E_UNSUPPORTED,
               "rarc_enc_gemm: the row-scale / residual epilogues need a shape the ping-pong kernels take (ask rarc_gemm_norm_fusable first)");
  if (act == 4) return enc_gemm_splitk(d_a, d_w, (float*)d_c, m, n, k, 1, s);   // small / odd shapes: the split-K kernels, one slice
  if (deep && force != 0 && k >= 4 * GK) {
    const int grid = (m / GM) * (n / GN);
    if (act == 1) hipLaunchKernelGGL((rarc_gemm128pp_f16_kernel<1>), dim3(grid), dim3(512), G128S_LDS, s, a, w, bs, c, m, n, k, k, order);
    else if (act == 16) hipLaunchKernelGGL((rarc_gemm128pp_f16_kernel<16>), dim3(grid), dim3(512), G128S_LDS, s, a, w, bs, c, m, n, k, k, order);
    else if (act == 32) hipLaunchKernelGGL((rarc_gemm128pp_f16_kernel<32>), dim3(grid), dim3(512), G128S_LDS, s, a, w, bs, c, m, n, k, k, order);
    else if (act == 96) hipLaunchKernelGGL((rarc_gemm128pp_f16_kernel<96>), dim3(grid), dim3(512), G128S_LDS, s, a, w, bs, c, m, n, k, k, order, d_ssq);
    else hipLaunchKernelGGL((rarc_gemm128pp_f16_kernel<0>), dim3(grid), dim3(512), G128S_LDS, s, a, w, bs, c, m, n, k, k, order);
  } else if (deep) {
    const int grid = (m / GM) * (n / GN);
    if (act == 1) hipLaunchKernelGGL((rarc_gemm_f16_kernel<1, 128, 4>), dim3(grid), dim3(256), lds_deep, s, a, w, bs, c, m, n, k, k, order);
    else hipLaunchKernelGGL((rarc_gemm_f16_kernel<0, 128, 4>), dim3(grid), dim3(256), lds_deep, s, a, w, bs, c, m, n, k, k, order);
  } else if (big) {
    const int grid = (m / 256) * (n / GN);
    if (act == 1) hipLaunchKernelGGL((rarc_gemm_f16_kernel<1, 256, 3>), dim3(grid), dim3(512), lds_big, s, a, w, bs, c, m, n, k, k, order);
    else hipLaunchKernelGGL((rarc_gemm_f16_kernel<0, 256, 3>), dim3(grid), dim3(512), lds_big, s, a, w, bs, c, m, n, k, k, order);
  } else {
    const int grid = (m / GM) * (n / GN);
    if (act == 1) hipLaunchKernelGGL((rarc_gemm_f16_kernel<1, 128, 2>), dim3(grid), dim3(256), lds_small, s, a, w, bs, c, m, n, k, k, order);
    else hipLaunchKernelGGL((rarc_gemm_f16_kernel<0, 128, 2>), dim3(grid), dim3(256), lds_small, s, a, w, bs, c, m, n, k, k, order);
  }
  RARC_HIP_CHECK(hipGetLastError());
  return RARC_OK;
}

extern "C" int rarc_enc_gemm(const uint16_t* d_a, const uint16_t* d_w, const uint16_t* d_bias, uint16_t* d_c, int m,
                             int n, int k, int act, void* stream) {
  RARC_REQUIRE(d_a && d_w && d_bias && d_c, RARC_E_INVALID, "rarc_enc_gemm: null pointer");
  RARC_REQUIRE(m > 0 && n > 0 && k > 0 && m % GM == 0 && n % GN == 0 && k % GK == 0, RARC_E_UNSUPPORTED,
               "rarc_enc_gemm: need M,N multiples of 128 and K multiple of 64 (got %d,%d,%d)", m, n, k);
  RARC_REQUIRE(act == 0 || act == 1 || act == 3, RARC_E_INVALID, "rarc_enc_gemm: act must be 0, 1 or 3");
  return enc_gemm_impl(d_a, d_w, d_bias, d_c, m, n, k, act, stream);
}

// rarc_enc_gemm for callers whose bias is known to be zero (decoder.hip): d_zero_bias must hold max(n) zeros
extern "C" int rarc_enc_gemm_zero_bias(const uint16_t* d_a, const uint16_t* d_w, const uint16_t* d_zero_bias, uint16_t* d_c, int m,
                                       int n, int k, int act, void* stream) {
  RARC_REQUIRE(d_a && d_w && d_zero_bias && d_c, RARC_E_INVALID, "rarc_enc_gemm: null pointer");
  RARC_REQUIRE(m > 0 && n > 0 && k > 0 && m % GM == 0 && n % GN == 0 && k % GK == 0, RARC_E_UNSUPPORTED,
               "rarc_enc_gemm: need M,N multiples of 128 and K multiple of 64 (got %d,%d,%d)", m, n, k);
  RARC_REQUIRE(act == 0 || act == 3, RARC_E_INVALID, "rarc_enc_gemm_zero_bias: act must be 0 or 3");
  return enc_gemm_impl(d_a, d_w, d_zero_bias, d_c, m, n, k, act, stream, true);
}

// the LM's norm-fused projections (decoder.hip): act 16 / 19 with d_rowscale float [m] in the bias slot, act 32 in place on d_c
// (act 96 = 32 + statistics: d_ssq float [m][n / 32] receives the sums of squares of the new rows' 32-column segments)
int rarc_gemm_fused_norm(const uint16_t* d_a, const uint16_t* d_w, const void* d_rowscale_or_zero, uint16_t* d_c, int m, int n, int k,
                         int act, void* stream, float* d_ssq) {
  RARC_REQUIRE(d_a && d_w && d_rowscale_or_zero && d_c && (act == 16 || act == 19 || act == 32 || (act == 96 && d_ssq)), RARC_E_INVALID,
               "rarc_gemm_fused_norm: bad argument");
  RARC_REQUIRE(rarc_gemm_norm_fusable(m, n, k), RARC_E_UNSUPPORTED, "rarc_gemm_fused_norm: shape %d x %d x %d not fusable", m, n, k);
  return enc_gemm_impl(d_a, d_w, (const uint16_t*)d_rowscale_or_zero, d_c, m, n, k, act, stream, false, d_ssq);
}

// The score GEMM of the wide search path with its select in the epilogue (wide.hip, round 5): S = A[M][K]·W[256][K]ᵀ in fp32,
// never stored — S[r][q] >= thr[q] sends (score, row0 + r) to query q's candidate list.  M a multiple of 256, K of 64, >= 256.
bool rarc_gemm_f16_select_takes(int m, int k) { return m > 0 && m % 256 == 0 && k % GK == 0 && k >= 4 * GK; }
int rarc_gemm_f16_select(const uint16_t* a, const uint16_t* w, int m, int k, const float* thr, unsigned long long* cand,
                         uint32_t* count, uint32_t* status, uint32_t cap, uint32_t row0, uint32_t n_valid, uint32_t shards,
                         hipStream_t s) {
  RARC_REQUIRE(a && w && thr && cand && count && status && rarc_gemm_f16_select_takes(m, k) && (shards == 1 || shards == 8) &&
                   cap % shards == 0,
               RARC_E_INVALID, "rarc_gemm_f16_select: bad arguments (m=%d k=%d shards=%u cap=%u)", m, k, shards, cap);
  if (int rc = gemm_attrs()) return rc;
  static const bool swz = !(getenv("RARC_GEMM_SWZ") && atoi(getenv("RARC_GEMM_SWZ")) == 0);
  const int order = swz ? 2 : 0;
  GemmSplitEpi fx;
  fx.thr = thr; fx.cand = cand; fx.count = count; fx.status = status; fx.cap = cap; fx.row0 = row0; fx.n_valid = n_valid;
  fx.shards = shards;
  const int tiles = m / 256;
  // up to one 256 x 256 tile per CU: the tiles start in lockstep, every CU fetches, then every CU multiplies — a lone tile took
  // 169 us where a streamed one takes 53.  The 128 x 128 ping-pong kernel (four tiles per CU's worth of work, a four-deep
  // operand ring) takes those chunks: the ramp of every search and all of a small shard.
  static const int small_max = getenv("RARC_WIDE_SMALL_TILES") ? atoi(getenv("RARC_WIDE_SMALL_TILES")) : 256;   // (A/B: 0 = never)
  if (tiles <= small_max) {
    hipLaunchKernelGGL((rarc_gemm128pp_f16_kernel<6>), dim3((m / 128) * 2, 1), dim3(512), G128S_LDS, s, (const half_t*)a, (const half_t*)w,
                       (const half_t*)nullptr, (half_t*)nullptr, m, 256, k, k, order, (float*)nullptr, fx);
    RARC_HIP_CHECK(hipGetLastError());
    return RARC_OK;
  }
  hipLaunchKernelGGL((rarc_gemm256_f16_kernel<6>), dim3(tiles > 256 ? 256 : tiles), dim3(512), G256_LDS, s, (const half_t*)a,
                     (const half_t*)w, (const half_t*)nullptr, (half_t*)nullptr, m, 256, k, order, (float*)nullptr, fx);
  RARC_HIP_CHECK(hipGetLastError());
  return RARC_OK;
}

// The same with N columns (a multiple of 256; thr / count / status are [N], cand is [N][cap]): the all-pairs cosine of pairs.hip,
// whose columns are a block of the rows themselves.
int rarc_gemm_f16_select_n(const uint16_t* a, const uint16_t* w, int m, int n, int k, const float* thr, unsigned long long* cand,
                           uint32_t* count, uint32_t* status, uint32_t cap, uint32_t row0, uint32_t n_valid, hipStream_t s) {
  RARC_REQUIRE(a && w && thr && cand && count && status && rarc_gemm_f16_select_takes(m, k) && n > 0 && n % 256 == 0, RARC_E_INVALID,
               "rarc_gemm_f16_select_n: bad arguments (m=%d n=%d k=%d)", m, n, k);
  if (int rc = gemm_attrs()) return rc;
  static const bool swz = !(getenv("RARC_GEMM_SWZ") && atoi(getenv("RARC_GEMM_SWZ")) == 0);
  const int order = swz ? 2 : 0;
  GemmSplitEpi fx;
  fx.thr = thr; fx.cand = cand; fx.count = count; fx.status = status; fx.cap = cap; fx.row0 = row0; fx.n_valid = n_valid;
  const long long tiles = (long long)(m / 256) * (n / 256);
  hipLaunchKernelGGL((rarc_gemm256_f16_kernel<6>), dim3(tiles > 256 ? 256 : (int)tiles), dim3(512), G256_LDS, s, (const half_t*)a,
                     (const half_t*)w, (const half_t*)nullptr, (half_t*)nullptr, m, n, k, order, (float*)nullptr, fx);
  RARC_HIP_CHECK(hipGetLastError());
  return RARC_OK;
}

// fp32 product of fp16 operands, no bias: C32[M][N] = A[M][K]·W[N][K]ᵀ — the GEMM of the fp32-class forward
// (encoder_f32.hip: split operands, K = 3x the model's k), on the same tile kernels with an fp32 epilogue.
int rarc_gemm_f16_f32out(const uint16_t* a, const uint16_t* w, float* c, int m, int n, int k, hipStream_t s) {
  static const bool small_only = getenv("RARC_GEMM32_BIG") && atoi(getenv("RARC_GEMM32_BIG")) == 0;
  if (small_only) return enc_gemm_splitk(a, w, c, m, n, k, 1, s);
  return enc_gemm_impl(a, w, nullptr, (uint16_t*)c, m, n, k, 4, (void*)s);
}

// The same product as up to `max_parts` fp32 PARTIAL slabs c[part][M][N] (k cut into equal slices; *parts = how many, the
// caller's epilogue sums them in order).  Small batches only: with at most one output tile per CU (a single query is 128
// tokens: 8 tiles of an N = 1024 projection) a tile's whole contraction — K' = 3K, 192 k tiles for the FFN's second
// projection — runs on one CU while 200 others idle; cut four ways it is a quarter as long.  Shapes that fill the chip
// keep one slab (the 256-row kernels).
int rarc_gemm_f16_f32out_parts(const uint16_t* a, const uint16_t* w, float* c, int m, int n, int k, int max_parts, int* parts,
                               hipStream_t s) {
  static const bool no_split = getenv("RARC_GEMM32_SPLIT") && atoi(getenv("RARC_GEMM32_SPLIT")) == 0;
  const int tiles = (m / GM) * (n / GN);
  int S = 1;
  static const int min_kt = getenv("RARC_GEMM32_MIN_KT") ? atoi(getenv("RARC_GEMM32_MIN_KT")) : 6;      // (tools/enc_small_split_sweep.sh: 4 x 32 tokens 2.26 -> 2.14 ms with 6 / 16 instead of 8 / 8)
  static const int max_s = getenv("RARC_GEMM32_MAX_S") ? atoi(getenv("RARC_GEMM32_MAX_S")) : 16;
  if (!no_split && !(m % 256 == 0 && (m / 256) * (n / GN) >= 256))
    while (2 * S <= max_parts && 2 * S <= max_s && tiles * S * 2 <= 256 && k % (S * 2 * GK) == 0 && k / (S * 2) >= min_kt * GK) S *= 2;
  *parts = S;
  if (S == 1) return rarc_gemm_f16_f32out(a, w, c, m, n, k, s);
  return enc_gemm_splitk(a, w, c, m, n, k, S, s);
}

// FFN1 of the fp32-class encoder with the GELU and the split of its output fused into the GEMM's epilogue (ACT 5).  Only
// shapes the 256 x 256 kernel takes WHOLE (no tail given to another kernel) and that fill the chip: returns 1 ("not taken",
// nothing launched) otherwise and the caller runs the unfused pair (fp32 product + rarc_e32_epi_kernel<., 1>).
// ... and, round 5, batches whose 128 x 128 tiles number 192..256 (FFN1 at 768..1024 tokens): the small-batch ping-pong kernel
// with the same epilogue, one tile per CU, no partial slabs, no row pass after it (bge-large, 32 x 32 tokens: 3.81 -> 3.68 ms;
// at 128 tiles the unfused pair wins — split-K x2 fills the chip, the fused form half of it: 2.78 vs 2.93 ms at 16 x 32).
static bool gemm_gelu_split_small(int m, int n, int k3) {
  const int t = (m / 128) * (n / 128);
  return m % 128 == 0 && n % 128 == 0 && t >= 192 && t <= 256 && k3 >= 4 * GK && k3 % GK == 0;
}
bool rarc_gemm_f16_gelu_split_takes(int m, int n, int k3) {
  const char* e = getenv("RARC_E32_FUSE_GELU");   // (read per call: tests switch it between forwards)
  if (e && atoi(e) == 0) return false;
  if (gemm_gelu_split_small(m, n, k3)) return !(e && atoi(e) == 2);      // (2: the big-batch form only — A/B)
  if (m % 256 || n % 256 || k3 < 4 * GK || k3 % GK) return false;
  const int t256 = (m / 256) * (n / 256), t128 = (m / 256) * (n / GN);
  return t256 >= 256 && gemm_prefers_256x256(t256, t128) && (t256 % 256 == 0 || t256 >= 1024);
}
int rarc_gemm_f16_gelu_split(const uint16_t* a3, const uint16_t* w3, const float* ra, const float* rw, const float* bias,
                             const float* sg, uint16_t* out3, int m, int n, int k3, hipStream_t s) {
  if (!rarc_gemm_f16_gelu_split_takes(m, n, k3)) return 1;
  const int t256 = (m / 256) * (n / 256);
  if (int rc = gemm_attrs()) return rc;
  static const bool swz = !(getenv("RARC_GEMM_SWZ") && atoi(getenv("RARC_GEMM_SWZ")) == 0);
  const int order = swz ? 2 : (m < n ? 1 : 0);
  GemmSplitEpi fx;
  fx.ra = ra; fx.rw = rw; fx.bias = bias; fx.sg = sg;
  if (gemm_gelu_split_small(m, n, k3)) {
    hipLaunchKernelGGL((rarc_gemm128pp_f16_kernel<5>), dim3((m / 128) * (n / 128), 1), dim3(512), G128S_LDS, s, (const half_t*)a3,
                       (const half_t*)w3, (const half_t*)nullptr, (half_t*)out3, m, n, k3, k3, order, (float*)nullptr, fx);
    RARC_HIP_CHECK(hipGetLastError());
    return RARC_OK;
  }
  hipLaunchKernelGGL((rarc_gemm256_f16_kernel<5>), dim3(t256 > 256 ? 256 : t256), dim3(512), G256_LDS, s, (const half_t*)a3,
                     (const half_t*)w3, (const half_t*)nullptr, (half_t*)out3, m, n, k3, order, (float*)nullptr, fx);
  RARC_HIP_CHECK(hipGetLastError());
  return RARC_OK;
}

extern "C" int rarc_enc_embed_ln(const int32_t* d_ids, const uint16_t* d_word, const uint16_t* d_pos,
                                 const uint16_t* d_type0, const uint16_t* d_gamma, const uint16_t* d_beta, float eps,
                                 int n_tokens, int seq_len, int hidden, int vocab, uint16_t* d_out, void* stream) {
  RARC_REQUIRE(d_ids && d_word && d_pos && d_type0 && d_gamma && d_beta && d_out, RARC_E_INVALID, "rarc_enc_embed_ln: null pointer");
  RARC_REQUIRE(hidden % 64 == 0 && hidden <= 1024 && n_tokens > 0 && seq_len > 0, RARC_E_UNSUPPORTED,
               "rarc_enc_embed_ln: hidden must be a multiple of 64, <= 1024");
  RARC_REQUIRE(vocab > 0, RARC_E_INVALID, "rarc_enc_embed_ln: vocab (rows of the word table) must be given");
  hipLaunchKernelGGL(rarc_embed_ln_kernel, dim3((n_tokens + 3) / 4), dim3(256), 0, (hipStream_t)stream, d_ids,
                     (const half_t*)d_word, (const half_t*)d_pos, (const half_t*)d_type0, (const half_t*)d_gamma,
                     (const half_t*)d_beta, eps, n_tokens, seq_len, hidden, vocab, (half_t*)d_out);
  RARC_HIP_CHECK(hipGetLastError());
  return RARC_OK;
}

extern "C" int rarc_enc_add_ln(const uint16_t* d_x, const uint16_t* d_resid, const uint16_t* d_gamma,
                               const uint16_t* d_beta, float eps, int n_rows, int hidden, uint16_t* d_out,
                               void* stream) {
  RARC_REQUIRE(d_x && d_resid && d_gamma && d_beta && d_out, RARC_E_INVALID, "rarc_enc_add_ln: null pointer");
  RARC_REQUIRE(hidden % 64 == 0 && hidden <= 1024 && n_rows > 0, RARC_E_UNSUPPORTED,
               "rarc_enc_add_ln: hidden must be a multiple of 64, <= 1024");
  hipLaunchKernelGGL(rarc_add_ln_kernel, dim3((n_rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const half_t*)d_x,
                     (const float*)nullptr, 0, (const half_t*)nullptr, (const half_t*)d_resid, (const half_t*)d_gamma,
                     (const half_t*)d_beta, eps, n_rows, hidden, (half_t*)d_out);
  RARC_HIP_CHECK(hipGetLastError());
  return RARC_OK;
}

static int enc_attention_impl(const uint16_t* d_qkv, const int32_t* d_lens, int n_seq, int seq_len, int hidden, int n_heads,
                              uint16_t* d_ctx, const float* d_rel, int rel_span, void* stream) {
  RARC_REQUIRE(d_qkv && d_lens && d_ctx, RARC_E_INVALID, "rarc_enc_attention: null pointer");
  RARC_REQUIRE(n_heads > 0 && (hidden == n_heads * 64 || hidden == n_heads * 32) && seq_len > 0 && seq_len <= 512 &&
                   n_seq > 0,
               RARC_E_UNSUPPORTED, "rarc_enc_attention: head_dim must be 32 or 64 and seq_len <= 512");
  RARC_REQUIRE(!d_rel || rel_span >= seq_len, RARC_E_INVALID,
               "rarc_enc_forward: the relative-position bias spans %d positions, the batch is %d long", rel_span, seq_len);
  const int q_blocks = (seq_len + 31) / 32;
  const int n_units = n_seq * n_heads * q_blocks;  // one wave each, four per workgroup
#define ENC_ATTN_LAUNCH(DHV, RELV)                                                                                  \
  hipLaunchKernelGGL((rarc_attention_mfma_kernel<DHV, RELV>), dim3((n_units + 3) / 4), dim3(256), 0, (hipStream_t)stream, \
                     (const half_t*)d_qkv, d_lens, seq_len, hidden, n_heads, q_blocks, n_units, (half_t*)d_ctx, d_rel, rel_span)
  // more than one query block per sequence: the workgroup-shared form (round 4); RARC_ENC_ATTN=wave keeps the per-wave kernel (A/B)
  const char* attn_env = getenv("RARC_ENC_ATTN");
  const bool shared = q_blocks >= 2 && !(attn_env && !strcmp(attn_env, "wave"));
  const int q_groups = (q_blocks + 3) / 4;
#define ENC_ATTN_SHARED(DHV, RELV)                                                                                           \
  hipLaunchKernelGGL((rarc_attention_mfma_shared_kernel<DHV, RELV>), dim3(n_seq * n_heads * q_groups), dim3(256), 0,         \
                     (hipStream_t)stream, (const half_t*)d_qkv, d_lens, seq_len, hidden, n_heads, q_blocks, q_groups,        \
                     (half_t*)d_ctx, d_rel, rel_span)
  if (shared) {
    if (hidden == n_heads * 64) {
      if (d_rel) ENC_ATTN_SHARED(64, true); else ENC_ATTN_SHARED(64, false);
    } else {
      if (d_rel) ENC_ATTN_SHARED(32, true); else ENC_ATTN_SHARED(32, false);
    }
  } else if (hidden == n_heads * 64) {
    if (d_rel) ENC_ATTN_LAUNCH(64, true); else ENC_ATTN_LAUNCH(64, false);
  } else {
    if (d_rel) ENC_ATTN_LAUNCH(32, true); else ENC_ATTN_LAUNCH(32, false);
  }
#undef ENC_ATTN_SHARED
#undef ENC_ATTN_LAUNCH
  RARC_HIP_CHECK(hipGetLastError());
  return RARC_OK;
}

extern "C" int rarc_enc_attention(const uint16_t* d_qkv, const int32_t* d_lens, int n_seq, int seq_len, int hidden,
                                  int n_heads, uint16_t* d_ctx, void* stream) {
  return enc_attention_impl(d_qkv, d_lens, n_seq, seq_len, hidden, n_heads, d_ctx, nullptr, 0, stream);
}

extern "C" int rarc_enc_pool(const uint16_t* d_hidden, int n_seq, int seq_len, int hidden, int normalize,
                             float* d_out, void* stream) {
  RARC_REQUIRE(d_hidden && d_out && n_seq > 0 && hidden % 8 == 0, RARC_E_INVALID, "rarc_enc_pool: bad arguments");
  hipLaunchKernelGGL(rarc_pool_kernel, dim3(n_seq), dim3(64), 0, (hipStream_t)stream, (const half_t*)d_hidden, seq_len,
                     hidden, normalize, d_out);
  RARC_HIP_CHECK(hipGetLastError());
  return RARC_OK;
}

extern "C" int rarc_enc_pool_mean(const uint16_t* d_hidden, const int32_t* d_lens, int n_seq, int seq_len, int hidden,
                                  int normalize, float* d_out, void* stream) {
  RARC_REQUIRE(d_hidden && d_lens && d_out && n_seq > 0 && seq_len > 0 && hidden % 8 == 0 && hidden <= 1024, RARC_E_INVALID,
               "rarc_enc_pool_mean: bad arguments");
  hipLaunchKernelGGL(rarc_pool_mean_kernel, dim3(n_seq), dim3(256), 0, (hipStream_t)stream, (const half_t*)d_hidden, d_lens,
                     seq_len, hidden, normalize, d_out);
  RARC_HIP_CHECK(hipGetLastError());
  return RARC_OK;
}

// ------------------------------------------------------------------------------------------
// Whole forward: the per-layer launch loop, issued from here so that a caller pays one foreign call
// ------------------------------------------------------------------------------------------
static inline size_t enc_align(size_t v) { return (v + 255) & ~(size_t)255; }

// split-K partials: S*M*N floats with (M/128)*(N/128)*S <= 256 workgroups (enc_split)
static inline size_t enc_parts_bytes(int hidden, int n_tokens) {
  const size_t mn = (size_t)n_tokens * hidden;
  return enc_align((mn * 4 < (size_t)256 * GM * GN ? mn * 4 : (size_t)256 * GM * GN) * sizeof(float));
}

extern "C" size_t rarc_enc_workspace_bytes(int hidden, int inter, int n_tokens) {
  if (hidden <= 0 || inter <= 0 || n_tokens <= 0) return 0;
  const size_t mh = enc_align((size_t)n_tokens * hidden * 2);
  return 3 * mh + enc_align((size_t)n_tokens * 3 * hidden * 2) + enc_align((size_t)n_tokens * inter * 2) +
         enc_parts_bytes(hidden, n_tokens);
}

extern "C" int rarc_enc_forward(const RarcEncModel* model, const int32_t* d_ids, const int32_t* d_lens, int n_seq,
                                int seq_len, int normalize, void* d_ws, size_t ws_bytes, float* d_out, void* stream) {
  RARC_RANGE();
  RARC_REQUIRE(model && model->layers && d_ids && d_lens && d_ws && d_out, RARC_E_INVALID, "rarc_enc_forward: null pointer");
  const int H = model->hidden, I = model->inter;
  RARC_REQUIRE(n_seq > 0 && seq_len > 0 && model->n_layers > 0, RARC_E_INVALID, "rarc_enc_forward: empty batch or model");
  RARC_REQUIRE(model->vocab > 0 && model->max_pos >= seq_len, RARC_E_INVALID,
               "rarc_enc_forward: model->vocab (%d) must be set and model->max_pos (%d) must cover seq_len (%d)",
               model->vocab, model->max_pos, seq_len);
  const long long m_ll = (long long)n_seq * seq_len;
  RARC_REQUIRE(m_ll % GM == 0 && m_ll < (1ll << 31), RARC_E_UNSUPPORTED,
               "rarc_enc_forward: n_seq*seq_len must be a multiple of 128 (got %lld)", m_ll);
  const int M = (int)m_ll;
  RARC_REQUIRE(ws_bytes >= rarc_enc_workspace_bytes(H, I, M), RARC_E_INVALID, "rarc_enc_forward: workspace too small");
  char* w = (char*)d_ws;
  const size_t mh = enc_align((size_t)M * H * 2);
  uint16_t* x = (uint16_t*)w;
  uint16_t* y = (uint16_t*)(w + mh);
  uint16_t* ctx = (uint16_t*)(w + 2 * mh);
  uint16_t* qkv = (uint16_t*)(w + 3 * mh);
  uint16_t* mid = (uint16_t*)(w + 3 * mh + enc_align((size_t)M * 3 * H * 2));
  float* parts = (float*)(w + 3 * mh + enc_align((size_t)M * 3 * H * 2) + enc_align((size_t)M * I * 2));
  hipStream_t hs = (hipStream_t)stream;
  const int s_o = enc_split(M, H, H), s_f2 = enc_split(M, H, I);
  // y = x·Wᵀ + b then x = LayerNorm(y + x); with S > 1 the GEMM leaves S fp32 partials and the LayerNorm sums them
  auto proj_ln = [&](const uint16_t* a, const uint16_t* wt, const uint16_t* b, int k, int S, const uint16_t* g,
                     const uint16_t* be) -> int {
    if (S == 1) {
      if (int r = rarc_enc_gemm(a, wt, b, y, M, H, k, 0, stream)) return r;
      return rarc_enc_add_ln(y, x, g, be, model->ln_eps, M, H, x, stream);
    }
    if (int r = enc_gemm_splitk(a, wt, parts, M, H, k, S, hs)) return r;
    hipLaunchKernelGGL(rarc_add_ln_kernel, dim3((M + 3) / 4), dim3(256), 0, hs, (const half_t*)nullptr, (const float*)parts,
                       S, (const half_t*)b, (const half_t*)x, (const half_t*)g, (const half_t*)be, model->ln_eps, M, H,
                       (half_t*)x);
    RARC_HIP_CHECK(hipGetLastError());
    return RARC_OK;
  };
  int rc = rarc_enc_embed_ln(d_ids, model->word, model->pos, model->type0, model->emb_g, model->emb_b, model->ln_eps, M,
                             seq_len, H, model->vocab, x, stream);
  for (int l = 0; l < model->n_layers && rc == RARC_OK; ++l) {
    const RarcEncLayer& L = model->layers[l];
    if ((rc = rarc_enc_gemm(x, L.qkv_w, L.qkv_b, qkv, M, 3 * H, H, 0, stream)) != RARC_OK) break;
    if ((rc = enc_attention_impl(qkv, d_lens, n_seq, seq_len, H, model->heads, ctx, model->rel_bias, model->rel_span, stream)) != RARC_OK) break;
    if ((rc = proj_ln(ctx, L.o_w, L.o_b, H, s_o, L.ln1_g, L.ln1_b)) != RARC_OK) break;
    if ((rc = rarc_enc_gemm(x, L.f1_w, L.f1_b, mid, M, I, H, 1, stream)) != RARC_OK) break;
    rc = proj_ln(mid, L.f2_w, L.f2_b, I, s_f2, L.ln2_g, L.ln2_b);
  }
  if (rc != RARC_OK) return rc;
  if (normalize & 2) return rarc_enc_pool_mean(x, d_lens, n_seq, seq_len, H, normalize & 1, d_out, stream);
  return rarc_enc_pool(x, n_seq, seq_len, H, normalize & 1, d_out, stream);
}
