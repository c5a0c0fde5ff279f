// pairs.hip — all pairs (i < j) of n embeddings whose cosine reaches a threshold.
//
// The reference deduplicates graph entities with it (encapsulation/database/graph_db/Base_Neo4j.py:538-583): sklearn's
// cosine_similarity over every entity embedding — the full n x n float64 matrix — then a python double loop over i < j that
// keeps the pairs with similarity >= 0.95.  The matrix is n^2 doubles and the loop n^2 / 2 interpreter steps; neither is the
// point: what is wanted is the handful of pairs above the threshold.
//
// Here the matrix is never formed.  The rows are normalised once (double-precision norms) into an fp16 image; the score GEMM
// the wide search already has (encoder.hip: 256 x 256 MFMA tiles, fp32 accumulate, the select in its epilogue —
// rarc_gemm256_f16_kernel<6>) multiplies the image with a super-block of up to 4096 of its own rows as the columns and
// nominates every (row, column) whose approximate cosine reaches threshold - eps, eps bounding what the fp16 rounding of both
// operands and the fp32 accumulation can cost (pairs_eps below); only rows up to the super-block's last column are multiplied
// (i < j needs no others).  Each nominated pair with i < j is then scored exactly — double-precision dot product of the fp32
// rows, times the two inverse norms, fixed summation order — and kept if that reaches the threshold.  So the pairs returned
// are exactly those whose float64 cosine is >= threshold (up to the last bits of a float64 dot product: sklearn's BLAS sums in
// another order), at the cost of n^2 d / 2 MFMA flops: 100,000 entities x 1024 dimensions are 10 TFLOP — tens of milliseconds.
#include "rarc_common.h"

int rarc_gemm_f16_select_n(const uint16_t* a, const uint16_t* w, int m, int n, int k, const float* thr, unsigned long long* cand,
                           uint32_t* count, uint32_t* status, uint32_t cap, uint32_t row0, uint32_t n_valid, hipStream_t s);

namespace {
constexpr int PAIRS_COLS = 4096;     // columns per GEMM launch (16 tiles wide: with >= 16 row tiles the chip is full)

struct PairsWs {
  uint16_t* image;       // [n_pad][d_pad] fp16, rows normalised, zero padded
  double* inv_norm;      // [n_pad]
  float* thr;            // [PAIRS_COLS]
  uint32_t* count;       // [PAIRS_COLS]
  uint32_t* status;      // [PAIRS_COLS]
  unsigned long long* cand;   // [PAIRS_COLS][cap]
};
inline size_t a256(size_t v) { return (v + 255) & ~(size_t)255; }
inline int64_t pad256(int64_t n) { return (n + 255) / 256 * 256; }
size_t pairs_ws_bytes(int64_t n, int d_pad, int cap) {
  const int64_t n_pad = pad256(n);
  return a256((size_t)n_pad * d_pad * 2) + a256((size_t)n_pad * 8) + 3 * a256((size_t)PAIRS_COLS * 4) +
         a256((size_t)PAIRS_COLS * cap * 8) + 256;
}
PairsWs pairs_carve(void* base, int64_t n, int d_pad, int cap) {
  const int64_t n_pad = pad256(n);
  char* b = (char*)(((uintptr_t)base + 255) & ~(uintptr_t)255);
  PairsWs w;
  w.image = (uint16_t*)b;    b += a256((size_t)n_pad * d_pad * 2);
  w.inv_norm = (double*)b;   b += a256((size_t)n_pad * 8);
  w.thr = (float*)b;         b += a256((size_t)PAIRS_COLS * 4);
  w.count = (uint32_t*)b;    b += a256((size_t)PAIRS_COLS * 4);
  w.status = (uint32_t*)b;   b += a256((size_t)PAIRS_COLS * 4);
  w.cand = (unsigned long long*)b;
  (void)cap;
  return w;
}

// sum over a wave in a FIXED order (lane l holds the partial of elements l, l + 64, ...): a tree over the lane index, so the
// exact score of a pair does not depend on where it was computed
__device__ __forceinline__ double pairs_wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
}  // namespace

// |fp32-accumulated x16·y16 - x·y| for unit vectors x, y rounded to fp16 elementwise: each factor is off by 2^-11 relative
// (or 2^-25 absolute under the normal range), the products' sum by d_pad·2^-24 at most (fp32 accumulation, MFMA order unknown).
static inline float pairs_eps(int d_pad) { return 0.0009765625f + (float)d_pad * 1.1920928955078125e-07f + 1e-6f; }

// one workgroup (4 waves) per row: double-precision norm, inverse norm, the normalised row as fp16 (zero row -> zeros, as
// sklearn's normalize leaves it); rows [n, n_pad) and columns [d, d_pad) are written as zeros
__global__ __launch_bounds__(256) void pairs_prepare_kernel(const float* __restrict__ rows, int64_t ld, int64_t n, int d, int d_pad,
                                                            uint16_t* __restrict__ image, double* __restrict__ inv_norm) {
  __shared__ double s_part[4];
  const int64_t r = blockIdx.x;
  const int tid = threadIdx.x;
  half_t* out = (half_t*)image + (size_t)r * d_pad;
  if (r >= n) {
    for (int m = tid; m < d_pad; m += 256) out[m] = (half_t)0.f;
    if (tid == 0) inv_norm[r] = 0.0;
    return;
  }
  const float* x = rows + (size_t)r * ld;
  double acc = 0.0;
  for (int m = tid; m < d; m += 256) acc += (double)x[m] * (double)x[m];
  acc = pairs_wave_sum(acc);
  if ((tid & 63) == 0) s_part[tid >> 6] = acc;
  __syncthreads();
  const double ss = (s_part[0] + s_part[1]) + (s_part[2] + s_part[3]);
  const double inv = ss > 0.0 ? 1.0 / sqrt(ss) : 0.0;
  for (int m = tid; m < d_pad; m += 256) out[m] = m < d ? (half_t)(float)((double)x[m] * inv) : (half_t)0.f;
  if (tid == 0) inv_norm[r] = inv;
}

__global__ void pairs_reset_kernel(float* thr, uint32_t* count, uint32_t* status, float t, int64_t col0, int64_t n) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q < PAIRS_COLS) {
    thr[q] = col0 + q < n ? t : INFINITY;     // padding columns never nominate
    count[q] = 0;
    status[q] = 0;
  }
}

// one workgroup per column j = col0 + q: every nominated row i < j is scored exactly (a wave per pair) and kept if it reaches
// the threshold.  The kept pairs of up to 1024 candidates are collected in LDS and take their places in the output with ONE
// atomic (a counter update per pair queues on its single address at ~350 ns each: half a million duplicate pairs would
// cost 0.16 s next to a 10 ms GEMM).  out_count counts every kept pair, also those beyond out_cap.
__global__ __launch_bounds__(256) void pairs_finalize_kernel(const float* __restrict__ rows, int64_t ld, int d,
                                                             const double* __restrict__ inv_norm, const unsigned long long* cand,
                                                             const uint32_t* count, uint32_t cap, int64_t col0, int64_t n,
                                                             double threshold, int64_t* out_pairs, double* out_scores,
                                                             unsigned long long out_cap, unsigned long long* out_count,
                                                             uint32_t* flags) {
  constexpr uint32_t BLOCK = 1024;
  __shared__ double s_cos[BLOCK];
  __shared__ uint32_t s_row[BLOCK];
  __shared__ uint32_t s_keep;
  __shared__ unsigned long long s_base;
  const int q = blockIdx.x;
  const int64_t j = col0 + q;
  if (j >= n) return;
  const uint32_t c_all = count[q], c = c_all < cap ? c_all : cap;
  if (c_all > cap && threadIdx.x == 0) atomicOr(flags, 1u);          // a list overflowed: the caller repeats with a larger cap
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const float* y = rows + (size_t)j * ld;
  const double inv_j = inv_norm[j];
  for (uint32_t e0 = 0; e0 < c; e0 += BLOCK) {
    if (threadIdx.x == 0) s_keep = 0;
    __syncthreads();
    const uint32_t e1 = e0 + BLOCK < c ? e0 + BLOCK : c;
    for (uint32_t e = e0 + wave; e < e1; e += 4) {
      const int64_t i = (int64_t)rarc_candrow(cand[(size_t)q * cap + e]);
      if (i >= j) continue;                                          // (wave-uniform: the whole wave reads the same entry)
      const float* x = rows + (size_t)i * ld;
      double acc = 0.0;
      for (int m = lane; m < d; m += 64) acc += (double)x[m] * (double)y[m];
      acc = pairs_wave_sum(acc);
      const double cosv = acc * inv_norm[i] * inv_j;
      if (lane == 0 && cosv >= threshold) {
        const uint32_t slot = atomicAdd(&s_keep, 1u);
        s_row[slot] = (uint32_t)i;
        s_cos[slot] = cosv;
      }
    }
    __syncthreads();
    const uint32_t kept = s_keep;
    if (threadIdx.x == 0 && kept) s_base = atomicAdd(out_count, (unsigned long long)kept);
    __syncthreads();
    for (uint32_t t = threadIdx.x; t < kept; t += blockDim.x) {
      const unsigned long long pos = s_base + t;
      if (pos < out_cap) {
        out_pairs[2 * pos] = (int64_t)s_row[t];
        out_pairs[2 * pos + 1] = j;
        out_scores[pos] = s_cos[t];
      } else {
        atomicOr(flags, 2u);                                         // more pairs than the output holds
      }
    }
    __syncthreads();
  }
}

extern "C" size_t rarc_similar_pairs_workspace_bytes(int64_t n_rows, int d, int cand_cap) {
  if (n_rows <= 0 || d <= 0 || cand_cap <= 0) return 0;
  const int d_pad = (d + 63) / 64 * 64 < 256 ? 256 : (d + 63) / 64 * 64;
  return pairs_ws_bytes(n_rows, d_pad, cand_cap);
}

// d_rows: fp32 [n_rows][ld] on the device, any scale (they are normalised here).  Pairs come out in no particular order
// (the caller sorts them: they are few); *d_out_count = how many reached the threshold, d_flags bit 0 = a column's
// nomination list of cand_cap entries overflowed (call again with a larger cand_cap; cand_cap >= n_rows cannot), bit 1 =
// more pairs than out_cap (call again with room for *d_out_count).  d_out_count and d_flags are zeroed here.
extern "C" int rarc_similar_pairs(const float* d_rows, int64_t ld, int64_t n_rows, int d, double threshold, void* d_ws, size_t ws_bytes,
                                  int cand_cap, int64_t* d_out_pairs, double* d_out_scores, int64_t out_cap,
                                  unsigned long long* d_out_count, uint32_t* d_flags, void* stream) {
  RARC_RANGE();
  RARC_REQUIRE(d_rows && d_ws && d_out_count && d_flags && (out_cap == 0 || (d_out_pairs && d_out_scores)), RARC_E_INVALID,
               "rarc_similar_pairs: null pointer");
  RARC_REQUIRE(n_rows >= 0 && n_rows < (int64_t)0x7fffff00ll && d >= 1 && d <= 4096 && ld >= d && out_cap >= 0 && cand_cap >= 1,
               RARC_E_INVALID, "rarc_similar_pairs: bad shape (n=%lld d=%d ld=%lld)", (long long)n_rows, d, (long long)ld);
  RARC_REQUIRE(threshold > 0.0 && threshold <= 1.0000001, RARC_E_INVALID, "rarc_similar_pairs: threshold %g outside (0, 1]", threshold);
  hipStream_t s = (hipStream_t)stream;
  RARC_HIP_CHECK(hipMemsetAsync(d_out_count, 0, 8, s));
  RARC_HIP_CHECK(hipMemsetAsync(d_flags, 0, 4, s));
  if (n_rows < 2) return RARC_OK;
  const int d_pad = (d + 63) / 64 * 64 < 256 ? 256 : (d + 63) / 64 * 64;
  RARC_REQUIRE(ws_bytes >= pairs_ws_bytes(n_rows, d_pad, cand_cap), RARC_E_WORKSPACE, "rarc_similar_pairs: workspace of %zu bytes, %zu needed",
               ws_bytes, pairs_ws_bytes(n_rows, d_pad, cand_cap));
  const PairsWs w = pairs_carve(d_ws, n_rows, d_pad, cand_cap);
  const int64_t n_pad = pad256(n_rows);
  hipLaunchKernelGGL(pairs_prepare_kernel, dim3((unsigned)n_pad), dim3(256), 0, s, d_rows, ld, n_rows, d, d_pad, w.image, w.inv_norm);
  RARC_HIP_CHECK(hipGetLastError());
  const float t_approx = (float)threshold - pairs_eps(d_pad);
  for (int64_t col0 = 0; col0 < n_rows; col0 += PAIRS_COLS) {
    const int64_t cols = n_pad - col0 < PAIRS_COLS ? n_pad - col0 : PAIRS_COLS;      // a multiple of 256
    const int64_t m = col0 + cols;                                                    // rows [0, m): every i < j of these columns
    hipLaunchKernelGGL(pairs_reset_kernel, dim3(PAIRS_COLS / 256), dim3(256), 0, s, w.thr, w.count, w.status, t_approx, col0, n_rows);
    RARC_HIP_CHECK(hipGetLastError());
    if (int rc = rarc_gemm_f16_select_n(w.image, w.image + (size_t)col0 * d_pad, (int)m, (int)cols, d_pad, w.thr, w.cand, w.count,
                                        w.status, (uint32_t)cand_cap, 0u, (uint32_t)n_rows, s))
      return rc;
    hipLaunchKernelGGL(pairs_finalize_kernel, dim3((unsigned)cols), dim3(256), 0, s, d_rows, ld, d, w.inv_norm, w.cand, w.count,
                       (uint32_t)cand_cap, col0, n_rows, threshold, d_out_pairs, d_out_scores, (unsigned long long)out_cap,
                       d_out_count, d_flags);
    RARC_HIP_CHECK(hipGetLastError());
  }
  return RARC_OK;
}
