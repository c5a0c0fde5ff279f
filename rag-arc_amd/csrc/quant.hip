// quant.hip — quantisation metadata for the int8 prefilter of the flat scan.
//
// The scan (scan_q8.hip) never trusts int8 scores: it uses them to DISCARD rows, and a row may only
// be discarded when   approx + eps < (a lower bound of the k-th best canonical score),   where
//   approx = <q8, d8> / (s_q * s_t)             (exact integer dot product, int8 MFMA)
//   |<q, d> - approx| <= ||q8/s_q|| * R + ||q - q8/s_q|| * max||d||        (Cauchy-Schwarz)
//   R = max over stored rows of ||d - d8/s_t||_2.
// This file computes s_t (one scale per 32-row tile: 127 / max|x| rounded DOWN to fp16, so that
// |x * s_t| <= 127) and R, with the very same rarc_quant8_chunk() the scan uses.  Runs once per
// ingest (index.add, VectorStore_Faiss.py:199-202): one extra streaming pass over the new rows.
#include "rarc_common.h"

// smallest fp16 value >= v (v >= 0); +inf bits when v exceeds the fp16 range
__device__ __forceinline__ uint16_t half_bits_round_up(float v) {
  if (!(v < 65504.f)) return 0x7c00u;
  half_t h = (half_t)v;  // RNE
  uint16_t b = __builtin_bit_cast(uint16_t, h);
  if ((float)h < v) b += 1;  // non-negative: next representable value up
  return b;
}

// largest fp16 value <= v (v > 0, finite)
__device__ __forceinline__ half_t half_round_down(float v) {
  half_t h = (half_t)v;  // RNE
  if ((float)h > v) {
    uint16_t b = __builtin_bit_cast(uint16_t, h);
    b -= 1;  // positive, non-zero: previous representable value
    h = __builtin_bit_cast(half_t, b);
  }
  return h;
}

// one workgroup (256 threads) per tile of 32 rows x d_pad fp16 (contiguous 64*d_pad bytes)
__global__ __launch_bounds__(256) void rarc_quant_meta_kernel(const uint4* __restrict__ corpus, int d_pad,
                                                              uint32_t first_tile, uint32_t n_tiles,
                                                              float* __restrict__ meta,
                                                              uint2* __restrict__ shadow) {
  __shared__ uint32_t s_max;
  __shared__ uint32_t s_res[32];
  const int tid = threadIdx.x;
  const int cpr = d_pad / 8;       // chunks per row
  const int nchunk = 32 * cpr;     // chunks per tile
  for (uint32_t t = first_tile + blockIdx.x; t < first_tile + n_tiles; t += gridDim.x) {
    if (tid == 0) s_max = 0;
    if (tid < 32) s_res[tid] = 0;
    __syncthreads();
    const uint4* src = corpus + (size_t)t * nchunk;
    uint32_t m = 0;
    for (int c = tid; c < nchunk; c += 256) {
      const uint4 v = src[c];
      const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const uint32_t lo = w[i] & 0x7fffu, hi = (w[i] >> 16) & 0x7fffu;  // |x| bit patterns: monotone
        m = lo > m ? lo : m;
        m = hi > m ? hi : m;
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const uint32_t x = __shfl_xor(m, o, 64);
      m = x > m ? x : m;
    }
    if ((tid & 63) == 0) atomicMax(&s_max, m);
    __syncthreads();
    uint32_t mb = s_max;
    if (mb > 0x7bffu) mb = 0x7bffu;  // inf/nan rows: undefined results, but no undefined behaviour
    const float mx = (float)__builtin_bit_cast(half_t, (uint16_t)mb);
    half_t s = (half_t)1.f;
    if (mx > 0.f) {
      float sv = 127.f / mx;
      if (sv > 32768.f) sv = 32768.f;
      s = half_round_down(sv);
      // 127/mx was rounded to fp32 first: make sure mx * s <= 127 holds exactly
      while ((double)mx * (double)(float)s > 127.0) s = half_round_down((float)s * 0.999f);
    }
    const float sf = (float)s;
    // residuals in quantised units u = x*s - d8 (|u| <= 0.5, exact in fp32), accumulated per row as
    // integers (order free): ui = ceil(|u| * 1024) <= 512, sum over a row < 2^32 for d_pad <= 16384
    for (int c = tid; c < nchunk; c += 256) {
      const uint4 v = src[c];
      const uint2 q = rarc_quant8_chunk(v, s);
      if (shadow) shadow[(size_t)t * nchunk + c] = q;  // the int8 image itself (row-major int8 [row][d_pad])
      const uint32_t w[4] = {v.x, v.y, v.z, v.w};
      const uint32_t qb[2] = {q.x, q.y};
      uint32_t acc = 0;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const uint16_t hb = (uint16_t)(w[e >> 1] >> (16 * (e & 1)));
        const float x = (float)__builtin_bit_cast(half_t, hb);
        const int d8 = (int)(int8_t)(qb[e >> 2] >> (8 * (e & 3)));
        const float u = __builtin_fabsf(__builtin_fmaf(x, sf, -(float)d8));
        const uint32_t ui = (uint32_t)__builtin_ceilf(u * 1024.f);
        acc += ui * ui;
      }
      atomicAdd(&s_res[c / cpr], acc);
    }
    __syncthreads();
    if (tid < 64) {
      // ||d - d8/s|| <= sqrt(sum ui^2) / (1024 * s); rounded up.  R_t = the tile's largest (lanes 32-63 mirror 0-31)
      float r = (float)(sqrt((double)s_res[tid & 31]) / (1024.0 * (double)sf) * 1.000001);
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) r = fmaxf(r, __shfl_xor(r, o, 64));
      if (tid == 0) {
        atomicMax((uint32_t*)meta, __float_as_uint(r));  // non-negative floats order like their bits
        float* mt = meta + RARC_QMETA_HDR + RARC_QMETA_STRIDE * (size_t)t;
        mt[0] = rarc_tmeta_pack(__builtin_bit_cast(uint16_t, s), half_bits_round_up(r));
        mt[1] = 1.0f / sf;
      }
    }
    __syncthreads();
  }
}

extern "C" size_t rarc_quant_meta_floats(int64_t n_rows) {
  return (size_t)RARC_QMETA_HDR + (size_t)RARC_QMETA_STRIDE * (size_t)((n_rows + 31) / 32);
}

extern "C" int rarc_quant_meta_f16(const uint16_t* d_corpus_f16, int64_t n_rows, int d_pad, int64_t first_row,
                                   float* d_qmeta, void* stream) {
  RARC_REQUIRE(d_qmeta && (d_corpus_f16 || n_rows == 0), RARC_E_INVALID, "rarc_quant_meta_f16: null pointer");
  RARC_REQUIRE(d_pad > 0 && d_pad % RARC_DIM_ALIGN == 0 && n_rows >= 0 && first_row >= 0 && first_row <= n_rows &&
                   n_rows < (int64_t)0xffffffe0ll,
               RARC_E_INVALID, "rarc_quant_meta_f16: bad arguments (n_rows=%lld first_row=%lld d_pad=%d)",
               (long long)n_rows, (long long)first_row, d_pad);
  const uint32_t t0 = (uint32_t)(first_row / 32), t1 = (uint32_t)((n_rows + 31) / 32);
  if (t1 <= t0) return RARC_OK;
  const uint32_t nt = t1 - t0;
  const int grid = nt < 8192u ? (int)nt : 8192;
  hipLaunchKernelGGL(rarc_quant_meta_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream,
                     (const uint4*)d_corpus_f16, d_pad, t0, nt, d_qmeta, (uint2*)nullptr);
  RARC_HIP_CHECK(hipGetLastError());
  return RARC_OK;
}

// Same pass, also keeping the int8 image ("shadow") the prefilter scan would otherwise recompute from
// the fp16 rows on every search: d_shadow8 is int8 [ceil32(n_rows)][d_pad], caller-owned.
extern "C" int rarc_quant_shadow_f16(const uint16_t* d_corpus_f16, int64_t n_rows, int d_pad, int64_t first_row,
                                     float* d_qmeta, int8_t* d_shadow8, void* stream) {
  RARC_REQUIRE(d_qmeta && d_shadow8 && (d_corpus_f16 || n_rows == 0), RARC_E_INVALID,
               "rarc_quant_shadow_f16: null pointer");
  RARC_REQUIRE(d_pad > 0 && d_pad % 256 == 0 && n_rows >= 0 && first_row >= 0 && first_row <= n_rows &&
                   n_rows < (int64_t)0xffffffe0ll,
               RARC_E_INVALID, "rarc_quant_shadow_f16: bad arguments (d_pad must be a multiple of 256; got %d)", d_pad);
  const uint32_t t0 = (uint32_t)(first_row / 32), t1 = (uint32_t)((n_rows + 31) / 32);
  if (t1 <= t0) return RARC_OK;
  const uint32_t nt = t1 - t0;
  const int grid = nt < 8192u ? (int)nt : 8192;
  hipLaunchKernelGGL(rarc_quant_meta_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream,
                     (const uint4*)d_corpus_f16, d_pad, t0, nt, d_qmeta, (uint2*)d_shadow8);
  RARC_HIP_CHECK(hipGetLastError());
  return RARC_OK;
}

// ---- fp8 corpus: tile scales, per-row multipliers, residual bound ----------------------------------
// A stored value is x = rowscale * val (val = decoded e4m3fn byte).  Tile scale s_t = 127 / max|x| over the
// tile's 32 rows (rounded down to fp16); row multiplier mul_r = rowscale_r * s_t rounded down to fp16, so
// that d8 = RNE(val * mul_r) is one fp16 fma per pair of values in the scan and |val * mul_r| <= 127.
// Residual of a row: x - d8/s_t = (val*mul_r - d8)/s_t + x*(1 - mul_r/(rowscale_r*s_t)); the first part is
// summed exactly in integers as for fp16, the second is at most 2^-10 ||x||.  One workgroup per tile, 8
// lanes per row.
__global__ __launch_bounds__(256) void rarc_quant_meta_f8_kernel(const uint4* __restrict__ corpus,
                                                                 const float* __restrict__ rowscale, int d_pad,
                                                                 uint32_t n_rows, uint32_t first_tile,
                                                                 uint32_t n_tiles, float* __restrict__ meta) {
  __shared__ float s_rowmax[32];
  __shared__ float s_rowres[32];
  __shared__ float s_tmax;
  const int tid = threadIdx.x, lane = tid & 63, j = tid & 7, rr = tid >> 3;  // 32 rows x 8 lanes
  const int cpr = d_pad / 16;  // 16-byte chunks per row
  for (uint32_t t = first_tile + blockIdx.x; t < first_tile + n_tiles; t += gridDim.x) {
    const uint32_t row = t * 32 + rr;
    const float rs = row < n_rows ? rowscale[row] : 0.f;  // rows past the end count as zero rows
    const uint4* src = corpus + (size_t)row * cpr;
    // pass 1: max |val| of the row -> max |x| of the tile
    float vmax = 0.f;
    double vsq = 0.0;
    if (row < n_rows) {
      for (int c = j; c < cpr; c += 8) {
        const uint4 v = src[c];
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float f[4];
          rarc_f8x4_to_f32(w[i], f);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            vmax = fmaxf(vmax, __builtin_fabsf(f[e]));
            vsq += (double)f[e] * (double)f[e];
          }
        }
      }
    }
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) {
      vmax = fmaxf(vmax, __shfl_xor(vmax, o, 64));
      vsq += __shfl_xor(vsq, o, 64);
    }
    if (j == 0) s_rowmax[rr] = vmax * rs;
    __syncthreads();
    if (tid == 0) {
      float m = 0.f;
      for (int i = 0; i < 32; ++i) m = fmaxf(m, s_rowmax[i]);
      s_tmax = m;
    }
    __syncthreads();
    const float mx = s_tmax;
    half_t s = (half_t)1.f;
    if (mx > 0.f && mx < INFINITY) {
      float sv = 127.f / mx;
      if (sv > 32768.f) sv = 32768.f;
      s = half_round_down(sv);
      while ((double)mx * (double)(float)s > 127.0) s = half_round_down((float)s * 0.999f);
    }
    const float sf = (float)s;
    // row multiplier: rowscale * s_t rounded down to fp16 (0 when it underflows: the row quantises to zeros)
    float mulf = rs * sf;
    half_t mul = (half_t)0.f;
    if (mulf > 0.f) {
      mul = half_round_down(mulf < 65504.f ? mulf : 65504.f);
      while ((double)vmax * (double)(float)mul > 127.0) mul = half_round_down((float)mul * 0.999f);
    }
    const float mulq = (float)mul;
    // pass 2: quantisation residual in units of 1/mul, exactly as the scan quantises
    uint32_t acc = 0;
    if (row < n_rows && mulq > 0.f) {
      for (int c = j; c < cpr; c += 8) {
        const uint4 v = src[c];
        const uint4 q = rarc_quant8_chunk_f8(v, mul);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w}, qb[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float f[4];
          rarc_f8x4_to_f32(w[i], f);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int d8 = (int)(int8_t)(qb[i] >> (8 * e));
            const float u = __builtin_fabsf(__builtin_fmaf(f[e], mulq, -(float)d8));
            const uint32_t ui = (uint32_t)__builtin_ceilf(u * 1024.f);
            acc += ui * ui;
          }
        }
      }
    }
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) acc += __shfl_xor(acc, o, 64);
    if (j == 0) {
      float resf = 0.f;
      if (row < n_rows) {
        const double xn = (double)rs * sqrt(vsq);  // ||x||
        double res;
        if (mulq > 0.f) {
          const double delta = 1.0 - (double)mulq / ((double)rs * (double)sf);  // in [0, 2^-10]
          res = sqrt((double)acc) / (1024.0 * (double)sf) + (delta > 0 ? delta : 0.0) * xn;
        } else {
          res = xn;  // row quantised to zeros
        }
        resf = (float)(res * 1.000001) * 1.000001f;
      }
      s_rowres[rr] = resf;
      meta[RARC_QMETA_HDR + RARC_QMETA_F8_STRIDE * (size_t)t + 2 + rr] = row < n_rows ? mulq : 0.f;
    }
    __syncthreads();
    if (tid == 0) {
      float rt = 0.f;
      for (int i = 0; i < 32; ++i) rt = fmaxf(rt, s_rowres[i]);
      atomicMax((uint32_t*)meta, __float_as_uint(rt));
      float* mt = meta + RARC_QMETA_HDR + RARC_QMETA_F8_STRIDE * (size_t)t;
      mt[0] = rarc_tmeta_pack(__builtin_bit_cast(uint16_t, s), half_bits_round_up(rt));  // (s_t, R_t)
      mt[1] = 1.0f / sf;
    }
    __syncthreads();
  }
}

extern "C" size_t rarc_quant_meta_floats_f8(int64_t n_rows) {
  return (size_t)RARC_QMETA_HDR + (size_t)RARC_QMETA_F8_STRIDE * (size_t)((n_rows + 31) / 32);
}

extern "C" int rarc_quant_meta_f8(const uint8_t* d_corpus_f8, const float* d_row_scale, int64_t n_rows, int d_pad,
                                  int64_t first_row, float* d_qmeta, void* stream) {
  RARC_REQUIRE(d_qmeta && ((d_corpus_f8 && d_row_scale) || n_rows == 0), RARC_E_INVALID,
               "rarc_quant_meta_f8: null pointer");
  RARC_REQUIRE(d_pad > 0 && d_pad % RARC_DIM_ALIGN_F8 == 0 && n_rows >= 0 && first_row >= 0 && first_row <= n_rows &&
                   n_rows < (int64_t)0xffffffe0ll,
               RARC_E_INVALID, "rarc_quant_meta_f8: bad arguments (n_rows=%lld first_row=%lld d_pad=%d)",
               (long long)n_rows, (long long)first_row, d_pad);
  const uint32_t t0 = (uint32_t)(first_row / 32), t1 = (uint32_t)((n_rows + 31) / 32);
  if (t1 <= t0) return RARC_OK;
  const uint32_t nt = t1 - t0;
  const int grid = nt < 8192u ? (int)nt : 8192;
  hipLaunchKernelGGL(rarc_quant_meta_f8_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream,
                     (const uint4*)d_corpus_f8, d_row_scale, d_pad, (uint32_t)n_rows, t0, nt, d_qmeta);
  RARC_HIP_CHECK(hipGetLastError());
  return RARC_OK;
}

// ---- test hook: the dense int8 score matrix of a small shard -------------------------------------
// approx[q][r] = <q8[q], d8[r]> * qinv[q] / s_tile(r), computed with the same rarc_quant8_chunk() as
// the scan (one thread per (query, row); no MFMA).  tests/ compare it with canonical scores to check
// the error bound eps8 on adversarial data: |canonical - approx| <= eps8[q] for EVERY pair.
__global__ __launch_bounds__(256) void rarc_q8_scores_kernel(const uint4* __restrict__ corpus, int d_pad,
                                                             uint32_t n_rows, const float* __restrict__ meta,
                                                             const int8_t* __restrict__ q8,
                                                             const float* __restrict__ qinv, int nq,
                                                             float* __restrict__ out) {
  const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  const int q = blockIdx.y;
  if (r >= n_rows || q >= nq) return;
  const uint32_t t = r / 32;
  const half_t s = __builtin_bit_cast(half_t, (uint16_t)__float_as_uint(meta[RARC_QMETA_HDR + RARC_QMETA_STRIDE * (size_t)t]));
  const float tinv = meta[RARC_QMETA_HDR + RARC_QMETA_STRIDE * (size_t)t + 1];
  const uint4* row = corpus + (size_t)r * (d_pad / 8);
  const int8_t* qp = q8 + (size_t)q * d_pad;
  int acc = 0;
  for (int c = 0; c < d_pad / 8; ++c) {
    const uint2 o = rarc_quant8_chunk(row[c], s);
    const uint32_t w[2] = {o.x, o.y};
#pragma unroll
    for (int e = 0; e < 8; ++e) acc += (int)(int8_t)(w[e >> 2] >> (8 * (e & 3))) * (int)qp[8 * c + e];
  }
  out[(size_t)q * n_rows + r] = (float)acc * (qinv[q] * tinv);
}

extern "C" int rarc_debug_q8_scores(const uint16_t* d_corpus_f16, int64_t n_rows, int d_pad, const float* d_qmeta,
                                    const void* d_qblock, int nq, float* d_out, void* stream) {
  RARC_REQUIRE(d_corpus_f16 && d_qmeta && d_qblock && d_out && n_rows > 0 && n_rows <= (1 << 22) && nq >= 1 &&
                   nq <= RARC_MAX_QUERIES && d_pad > 0 && d_pad % RARC_DIM_ALIGN == 0,
               RARC_E_INVALID, "rarc_debug_q8_scores: bad arguments");
  const RarcQb qb = rarc_qb_carve(d_qblock, d_pad);
  hipLaunchKernelGGL(rarc_q8_scores_kernel, dim3((unsigned)((n_rows + 255) / 256), (unsigned)nq), dim3(256), 0,
                     (hipStream_t)stream, (const uint4*)d_corpus_f16, d_pad, (uint32_t)n_rows, d_qmeta, qb.q8, qb.qinv,
                     nq, d_out);
  RARC_HIP_CHECK(hipGetLastError());
  return RARC_OK;
}
