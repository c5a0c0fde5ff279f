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

// 1024 threads per query: FOUR lanes per item (item t = tid / 4 + 256 m, lane part s = tid & 3 takes the partners
// u = s, s + 4, ...), so an all-pairs sweep is n / 4 steps per thread and 16 waves per CU hide each other's latency —
// with one thread per item every step of the three sweeps waited out its own dependency chain (135 cycles per step,
// 45 us per call whatever the number of queries).  Predicates are combined with bitwise operators (written with
// && / || the compiler emitted a chain of exec-masked branches per step).  The order-free parts (first occurrence =
// a minimum, next occurrence = a minimum, output position = a count) are reduced over the four lanes; the fp64 sum of
// a key's terms — whose ORDER is the reference's — walks the key's occurrence chain sequentially (usually two links).
constexpr int RRF_THREADS = 1024;
__global__ __launch_bounds__(RRF_THREADS) void rarc_rrf_kernel(const int64_t* keys, const int32_t* lens, int n_lists,
                                                               int max_len, double rrf_k, int top_k, int64_t* out_keys,
                                                               double* out_scores, int32_t* out_n) {
  __shared__ int64_t s_key[RRF_MAX_ITEMS];
  __shared__ double s_score[RRF_MAX_ITEMS];   // each item's own term, then (first occurrences) the key's sum
  __shared__ int32_t s_first[RRF_MAX_ITEMS];  // first occurrence of the item's key
  __shared__ int32_t s_next[RRF_MAX_ITEMS];   // next occurrence of the same key after this item, or n
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
      s_score[o + i] = 1.0 / (rrf_k + (double)(i + 1));   // rank = position + 1 inside its list
    }
  }
  __syncthreads();
  const int part = threadIdx.x & 3, item0 = threadIdx.x >> 2, per = RRF_THREADS / 4;
  const int n_round = (n + per - 1) / per * per;   // whole waves run every trip (the lane-group reductions need all four)
  // first occurrence of every key in (list, position) order
  for (int t = item0; t < n_round; t += per) {
    const bool live = t < n;
    const int64_t k = live ? s_key[t] : 0;
    int f = live ? t : 0;
#pragma unroll 4
    for (int u = part; u < t && u < n; u += 4) f = ((int)(s_key[u] == k) & (int)(u < f)) ? u : f;
    f = min(f, __shfl_xor(f, 1, 64));
    f = min(f, __shfl_xor(f, 2, 64));
    if (live && part == 0) s_first[t] = f;
  }
  __syncthreads();
  // next occurrence of the same key (chains the occurrences of a key in order)
  for (int t = item0; t < n_round; t += per) {
    const bool live = t < n;
    const int f = live ? s_first[t] : -1;
    int nx = n;
    const int u0 = t + 1 + ((part - (t + 1)) & 3);   // smallest u > t with u = part (mod 4)
#pragma unroll 4
    for (int u = u0; u < n; u += 4) nx = ((int)(s_first[u] == f) & (int)(u < nx)) ? u : nx;
    nx = min(nx, __shfl_xor(nx, 1, 64));
    nx = min(nx, __shfl_xor(nx, 2, 64));
    if (live && part == 0) s_next[t] = nx;
  }
  __syncthreads();
  // score of each distinct key: sequential fp64 sum over its occurrences, in order (what the reference's `+=` does)
  double my_sum[(RRF_MAX_ITEMS + RRF_THREADS - 1) / RRF_THREADS];
  {
    int m = 0;
    for (int t = threadIdx.x; t < n; t += blockDim.x, ++m) {
      double sum = 0.0;
      if (s_first[t] == t)
        for (int u = t; u < n; u = s_next[u]) sum += s_score[u];
      my_sum[m] = sum;
    }
  }
  __syncthreads();  // every term has been read before any is overwritten by a sum
  {
    int m = 0;
    for (int t = threadIdx.x; t < n; t += blockDim.x, ++m) {
      if (s_first[t] != t) continue;
      s_score[t] = my_sum[m];
      atomicAdd(&s_nuniq, 1);
    }
  }
  __syncthreads();
  // stable descending order by counting: position = #{distinct u better than t}
  for (int t = item0; t < n_round; t += per) {
    const bool mine = t < n && s_first[t] == t;
    const double sc = mine ? s_score[t] : 0.0;
    int pos = 0;
#pragma unroll 4
    for (int u = part; u < n; u += 4) {
      const double su = s_score[u];
      pos += (int)(s_first[u] == u) & (int)(u != t) & ((int)(su > sc) | ((int)(su == sc) & (int)(u < t)));
    }
    pos += __shfl_xor(pos, 1, 64);
    pos += __shfl_xor(pos, 2, 64);
    if (mine && part == 0 && pos < top_k) {
      out_keys[(size_t)b * top_k + pos] = s_key[t];
      out_scores[(size_t)b * top_k + pos] = sc;
    }
  }
  if (threadIdx.x == 0) out_n[b] = s_nuniq < top_k ? s_nuniq : top_k;
}

extern "C" int rarc_rrf_fuse(const int64_t* d_keys, const int32_t* d_len, int nq, int n_lists, int max_len,
                             double rrf_k, int top_k, int64_t* d_out_keys, double* d_out_scores,
                             int32_t* d_out_n, void* stream) {
  RARC_RANGE();
  RARC_REQUIRE(d_keys && d_len && d_out_keys && d_out_scores && d_out_n, RARC_E_INVALID, "rarc_rrf_fuse: null pointer");
  RARC_REQUIRE(nq >= 0 && n_lists >= 1 && n_lists <= 64 && max_len >= 0 && top_k >= 0, RARC_E_INVALID,
               "rarc_rrf_fuse: bad sizes (nq=%d lists=%d max_len=%d top_k=%d)", nq, n_lists, max_len, top_k);
  RARC_REQUIRE((int64_t)n_lists * max_len <= RRF_MAX_ITEMS, RARC_E_UNSUPPORTED,
               "rarc_rrf_fuse: %d lists x %d items exceeds %d items per query", n_lists, max_len, RRF_MAX_ITEMS);
  if (nq == 0) return RARC_OK;
  hipLaunchKernelGGL(rarc_rrf_kernel, dim3(nq), dim3(RRF_THREADS), 0, (hipStream_t)stream, d_keys, d_len, n_lists,
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
  RARC_RANGE();
  RARC_REQUIRE(d_z_no && d_z_yes && d_out_scores_f16 && d_out_perm, RARC_E_INVALID, "rarc_rerank_order: null pointer");
  RARC_REQUIRE(nq >= 0 && n >= 0 && n <= RERANK_MAX, RARC_E_UNSUPPORTED, "rarc_rerank_order: n=%d exceeds %d", n,
               RERANK_MAX);
  if (nq == 0 || n == 0) return RARC_OK;
  hipLaunchKernelGGL(rarc_rerank_kernel, dim3(nq), dim3(256), 0, (hipStream_t)stream, (const half_t*)d_z_no,
                     (const half_t*)d_z_yes, n, (half_t*)d_out_scores_f16, d_out_perm);
  RARC_HIP_CHECK(hipGetLastError());
  return RARC_OK;
}

// ---- maximal marginal relevance: the greedy selection of _mmr_select (VectorStore_Faiss.py:16-62) --------------
// n candidates (fetch_k, a few dozen) of dimension d, float64 arithmetic like the reference's numpy on python floats:
//   pick candidate 0; then k - 1 times: score_i = lambda·<q, e_i> − (1 − lambda)·max(0, max_{s selected} <e_s, e_i>),
//   take the FIRST candidate with the largest score among the remaining ones (python's max).
// Candidates come as fp32 (the resident rows, gathered) and are widened; `normalize` divides them (and the query) by
// their float64 norms first, as max_marginal_relevance_search_by_vector does for cosine stores (:308-312, :355-359).
// One workgroup; instead of an n x n similarity matrix the running maximum per candidate is updated with the newest
// pick each round: O(k·n·d).  d_work: n·d + d doubles.
constexpr int MMR_MAX_N = 1024;
__global__ __launch_bounds__(256) void rarc_mmr_kernel(const float* cand, int64_t ld, const double* query, int n, int d,
                                                       int normalize, int k, double lambda, double* work, int32_t* out) {
  __shared__ double s_qsim[MMR_MAX_N];
  __shared__ double s_maxsim[MMR_MAX_N];
  __shared__ int32_t s_taken[MMR_MAX_N];
  __shared__ double s_red[256];
  __shared__ int32_t s_best;
  const int tid = threadIdx.x;
  double* e = work;                 // [n][d] widened (normalised) candidates
  double* qn = work + (size_t)n * d;  // [d]
  // sequential float64 sums per vector: one thread per vector (n is small); numpy's dot / norm use a blocked order,
  // which can differ in the last place — selections differ from the reference's only across ties that close
  for (int i = tid; i <= n; i += blockDim.x) {
    const bool isq = i == n;
    double ss = 0.0;
    if (normalize) {
      for (int m = 0; m < d; ++m) {
        const double v = isq ? query[m] : (double)cand[(size_t)i * ld + m];
        ss += v * v;
      }
    }
    const double nr = normalize ? sqrt(ss) : 1.0;
    for (int m = 0; m < d; ++m) {
      const double v = isq ? query[m] : (double)cand[(size_t)i * ld + m];
      (isq ? qn : e + (size_t)i * d)[m] = normalize ? v / nr : v;
    }
  }
  __syncthreads();
  for (int i = tid; i < n; i += blockDim.x) {
    double s = 0.0;
    for (int m = 0; m < d; ++m) s += qn[m] * e[(size_t)i * d + m];
    s_qsim[i] = s;
    s_maxsim[i] = 0.0;   // (the reference's max_sim starts from 0)
    s_taken[i] = i == 0;
  }
  if (tid == 0) out[0] = 0;
  __syncthreads();
  int last = 0;
  for (int step = 1; step < k && step < n; ++step) {
    double best = -INFINITY;
    int best_i = 0x7fffffff;
    for (int i = tid; i < n; i += blockDim.x) {
      if (s_taken[i]) continue;
      double s = 0.0;
      for (int m = 0; m < d; ++m) s += e[(size_t)last * d + m] * e[(size_t)i * d + m];
      const double mx = s > s_maxsim[i] ? s : s_maxsim[i];
      s_maxsim[i] = mx;
      const double val = lambda * s_qsim[i] - (1.0 - lambda) * mx;
      if (val > best) { best = val; best_i = i; }   // (i ascends per thread: the first maximum is kept)
    }
    s_red[tid] = best;
    __syncthreads();
    if (tid == 0) {
      double b = -INFINITY;
      for (int t = 0; t < (int)blockDim.x; ++t) b = s_red[t] > b ? s_red[t] : b;
      s_red[0] = b;
      s_best = 0x7fffffff;
    }
    __syncthreads();
    if (best_i != 0x7fffffff && best == s_red[0]) atomicMin(&s_best, best_i);  // the lowest index among equal scores
    __syncthreads();
    last = s_best;
    if (tid == 0) { out[step] = last; s_taken[last] = 1; }
    __syncthreads();
  }
}

extern "C" size_t rarc_mmr_workspace_doubles(int n, int d) { return n > 0 && d > 0 ? (size_t)n * d + d : 0; }

extern "C" int rarc_mmr_select(const float* d_cand, int64_t ld, const double* d_query, int n, int d, int normalize, int k,
                               double lambda, double* d_work, int32_t* d_out, void* stream) {
  RARC_RANGE();
  RARC_REQUIRE(d_cand && d_query && d_work && d_out, RARC_E_INVALID, "rarc_mmr_select: null pointer");
  RARC_REQUIRE(n >= 1 && n <= MMR_MAX_N && d >= 1 && ld >= d && k >= 1, RARC_E_INVALID,
               "rarc_mmr_select: bad sizes (n=%d of at most %d, d=%d, k=%d)", n, MMR_MAX_N, d, k);
  hipLaunchKernelGGL(rarc_mmr_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, d_cand, ld, d_query, n, d, normalize, k,
                     lambda, d_work, d_out);
  RARC_HIP_CHECK(hipGetLastError());
  return RARC_OK;
}
