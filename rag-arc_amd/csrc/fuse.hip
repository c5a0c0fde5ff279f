// fuse.hip — rank-level kernels after retrieval.
//   rarc_rrf_fuse      RRFusion.fuse                 core/utils/Fusion.py:45-76
//   rarc_rerank_order  Qwen3Reranker score -> order  core/rerank/Reranker_Qwen3.py:41-49, :70-74
//
// Both are tiny per-query problems (<= a few hundred items); one workgroup per query, O(n^2)
// all-pairs in LDS.  What matters here is bit-exactness, not bandwidth:
//   * RRF adds 1.0/(k + rank) in fp64 in the reference's order: lists in order, positions in
//     order, per key (Python float += is fp64; `sorted(..., reverse=True)` is stable, so ties
//     keep first-insertion order).
//   * rerank: p_yes = exp(log_softmax([z_no, z_yes])[1]) through fp16 tensors, then Python's
//     stable sort descending (ties keep retrieval order).
#include "rarc_common.h"

constexpr int RRF_MAX_ITEMS = 4096;

__global__ __launch_bounds__(256) void rarc_rrf_kernel(const int64_t* keys, const int32_t* lens, int n_lists,
                                                       int max_len, double rrf_k, int top_k, int64_t* out_keys,
                                                       double* out_scores, int32_t* out_n) {
  __shared__ int64_t s_key[RRF_MAX_ITEMS];
  __shared__ double s_score[RRF_MAX_ITEMS];   // valid where s_first[t] == t
  __shared__ int32_t s_rank[RRF_MAX_ITEMS];   // 1-based rank inside its list
  __shared__ int32_t s_first[RRF_MAX_ITEMS];
  __shared__ int32_t s_off[64 + 1];
  __shared__ int32_t s_nuniq;
  const int b = blockIdx.x;
  if (threadIdx.x == 0) {
    int o = 0;
    for (int r = 0; r < n_lists; ++r) {
      s_off[r] = o;
      int l = lens[b * n_lists + r];
      l = l < 0 ? 0 : (l > max_len ? max_len : l);
      o += l;
    }
    s_off[n_lists] = o;
    s_nuniq = 0;
  }
  __syncthreads();
  const int n = s_off[n_lists];
  for (int r = 0; r < n_lists; ++r) {
    const int o = s_off[r], l = s_off[r + 1] - o;
    for (int i = threadIdx.x; i < l; i += blockDim.x) {
      s_key[o + i] = keys[((size_t)b * n_lists + r) * max_len + i];
      s_rank[o + i] = i + 1;
    }
  }
  __syncthreads();
  // The three all-pairs sweeps below are branch-free on purpose: with early exits and conditional divisions every
  // LDS read waited for the previous one (57 us per 256 x 200 items); as straight unrolled loops the reads pipeline.
  // first occurrence of every key in (list, position) order; each item's own term 1.0 / (k + rank), once
  for (int t = threadIdx.x; t < n; t += blockDim.x) {
    const int64_t k = s_key[t];
    int f = t;
#pragma unroll 8
    for (int u = 0; u < t; ++u) f = (s_key[u] == k && u < f) ? u : f;
    s_first[t] = f;
    s_score[t] = 1.0 / (rrf_k + (double)s_rank[t]);   // (becomes the key's sum below, for first occurrences)
  }
  __syncthreads();
  // score of each distinct key: sequential fp64 sum over its occurrences, in order (adding 0.0 for the others
  // leaves every partial sum bit for bit what the reference's `+=` over the occurrences alone produces)
  double my_sum[(RRF_MAX_ITEMS + 255) / 256];
  {
    int j = 0;
    for (int t = threadIdx.x; t < n; t += blockDim.x, ++j) {
      const int64_t k = s_key[t];
      double s = 0.0;
#pragma unroll 8
      for (int u = 0; u < n; ++u) s += (u >= t && s_key[u] == k) ? s_score[u] : 0.0;
      my_sum[j] = s;
    }
  }
  __syncthreads();  // every term has been read before any is overwritten by a sum
  {
    int j = 0;
    for (int t = threadIdx.x; t < n; t += blockDim.x, ++j) {
      if (s_first[t] != t) continue;
      s_score[t] = my_sum[j];
      atomicAdd(&s_nuniq, 1);
    }
  }
  __syncthreads();
  // stable descending order by counting: position = #{distinct u better than t}
  for (int t = threadIdx.x; t < n; t += blockDim.x) {
    if (s_first[t] != t) continue;
    const double s = s_score[t];
    int pos = 0;
#pragma unroll 8
    for (int u = 0; u < n; ++u) {
      const double su = s_score[u];
      pos += (s_first[u] == u && u != t && ((su > s) || (su == s && u < t))) ? 1 : 0;
    }
    if (pos < top_k) {
      out_keys[(size_t)b * top_k + pos] = s_key[t];
      out_scores[(size_t)b * top_k + pos] = s;
    }
  }
  if (threadIdx.x == 0) out_n[b] = s_nuniq < top_k ? s_nuniq : top_k;
}

extern "C" int rarc_rrf_fuse(const int64_t* d_keys, const int32_t* d_len, int nq, int n_lists, int max_len,
                             double rrf_k, int top_k, int64_t* d_out_keys, double* d_out_scores,
                             int32_t* d_out_n, void* stream) {
  RARC_REQUIRE(d_keys && d_len && d_out_keys && d_out_scores && d_out_n, RARC_E_INVALID, "rarc_rrf_fuse: null pointer");
  RARC_REQUIRE(nq >= 0 && n_lists >= 1 && n_lists <= 64 && max_len >= 0 && top_k >= 0, RARC_E_INVALID,
               "rarc_rrf_fuse: bad sizes (nq=%d lists=%d max_len=%d top_k=%d)", nq, n_lists, max_len, top_k);
  RARC_REQUIRE((int64_t)n_lists * max_len <= RRF_MAX_ITEMS, RARC_E_UNSUPPORTED,
               "rarc_rrf_fuse: %d lists x %d items exceeds %d items per query", n_lists, max_len, RRF_MAX_ITEMS);
  if (nq == 0) return RARC_OK;
  hipLaunchKernelGGL(rarc_rrf_kernel, dim3(nq), dim3(256), 0, (hipStream_t)stream, d_keys, d_len, n_lists,
                     max_len, rrf_k, top_k, d_out_keys, d_out_scores, d_out_n);
  RARC_HIP_CHECK(hipGetLastError());
  return RARC_OK;
}

// ---- reranker: fp16 logits -> fp16 p_yes -> stable descending permutation -----------------------
constexpr int RERANK_MAX = 4096;

__global__ __launch_bounds__(256) void rarc_rerank_kernel(const half_t* z_no, const half_t* z_yes, int n,
                                                          half_t* out_scores, int32_t* out_perm) {
  __shared__ float s_p[RERANK_MAX];
  const int b = blockIdx.x;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const float zn = (float)z_no[(size_t)b * n + i], zy = (float)z_yes[(size_t)b * n + i];
    const float m = fmaxf(zn, zy);
    const float sum = expf(zn - m) + expf(zy - m);
    const half_t ls = (half_t)((zy - m) - logf(sum));  // log_softmax output tensor is fp16
    const half_t pr = (half_t)expf((float)ls);         // .exp() on the fp16 tensor
    out_scores[(size_t)b * n + i] = pr;
    s_p[i] = (float)pr;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const float s = s_p[i];
    int pos = 0;
    for (int u = 0; u < n; ++u) {
      const float su = s_p[u];
      pos += (su > s) || (su == s && u < i);
    }
    out_perm[(size_t)b * n + pos] = i;
  }
}

extern "C" int rarc_rerank_order(const uint16_t* d_z_no, const uint16_t* d_z_yes, int nq, int n,
                                 uint16_t* d_out_scores_f16, int32_t* d_out_perm, void* stream) {
  RARC_REQUIRE(d_z_no && d_z_yes && d_out_scores_f16 && d_out_perm, RARC_E_INVALID, "rarc_rerank_order: null pointer");
  RARC_REQUIRE(nq >= 0 && n >= 0 && n <= RERANK_MAX, RARC_E_UNSUPPORTED, "rarc_rerank_order: n=%d exceeds %d", n,
               RERANK_MAX);
  if (nq == 0 || n == 0) return RARC_OK;
  hipLaunchKernelGGL(rarc_rerank_kernel, dim3(nq), dim3(256), 0, (hipStream_t)stream, (const half_t*)d_z_no,
                     (const half_t*)d_z_yes, n, (half_t*)d_out_scores_f16, d_out_perm);
  RARC_HIP_CHECK(hipGetLastError());
  return RARC_OK;
}
