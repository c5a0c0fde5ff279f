// compact.hip — rarc_compact_rows: delete rows of a resident buffer by stable in-place compaction (include/rarc.h).
//
// The reference deletes by rebuilding: it resets the index and re-embeds every surviving text
// (encapsulation/database/vector_db/VectorStore_Faiss.py:374-415).  The surviving embeddings are already in HBM; this
// moves them down over the holes instead.  Row i of the result is the i-th surviving row of the input (order kept: doc
// ids stay "row index" and ties stay ordered by insertion).
//
// holes h_0 < h_1 < ... (rows to drop) arrive as adj_j = h_j - j = the number of survivors before hole j (non-decreasing);
// the source of destination row i is  src(i) = i + #{j : adj_j <= i}  (one upper-bound search per row).  src(i) >= i and
// src is increasing, so a destination chunk [a, b) only ever needs rows at or after a: chunk by chunk, ascending, each
// chunk gathered into a scratch buffer and then copied into place — nothing a later chunk needs is overwritten.
// HBM-bound: 2 reads + 2 writes of the rows after the first hole.
#include "rarc_common.h"

template <typename V>
__global__ __launch_bounds__(256) void rarc_compact_gather_kernel(const V* __restrict__ rows, int64_t row_vecs,
                                                                  const int64_t* __restrict__ adj, int64_t n_holes,
                                                                  int64_t first, int64_t count, V* __restrict__ tmp) {
  // one 64-lane wave per row (4 rows per workgroup per step); the wave's lanes stride over the row's vectors
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int64_t r = (int64_t)blockIdx.x * 4 + wave; r < count; r += (int64_t)gridDim.x * 4) {
    const int64_t i = first + r;
    int64_t lo = 0, hi = n_holes;            // upper bound of i in adj
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if (adj[mid] <= i) lo = mid + 1; else hi = mid;
    }
    const V* src = rows + (i + lo) * row_vecs;
    V* dst = tmp + r * row_vecs;
    for (int64_t v = lane; v < row_vecs; v += 64) dst[v] = src[v];
  }
}

// d_rows: [n_rows] rows of row_bytes each (row_bytes % 4 == 0).  d_adj: device int64 [n_holes], adj_j = h_j - j for the
// sorted distinct holes h_j.  d_tmp: scratch of tmp_bytes (>= one row; larger = fewer launches).  On return (stream
// order) rows [0, n_rows - n_holes) are the survivors in their old order and rows [n_rows - n_holes, n_rows) are zero.
extern "C" int rarc_compact_rows(void* d_rows, int64_t row_bytes, int64_t n_rows, const int64_t* d_adj, int64_t n_holes,
                                 int64_t first_hole, void* d_tmp, size_t tmp_bytes, void* stream) {
  RARC_REQUIRE(d_rows && row_bytes > 0 && row_bytes % 4 == 0 && n_rows >= 0 && n_holes >= 0 && n_holes <= n_rows,
               RARC_E_INVALID, "rarc_compact_rows: bad arguments (row_bytes=%lld n_rows=%lld n_holes=%lld)",
               (long long)row_bytes, (long long)n_rows, (long long)n_holes);
  if (n_holes == 0) return RARC_OK;
  RARC_REQUIRE(d_adj && d_tmp && first_hole >= 0 && first_hole < n_rows && tmp_bytes >= (size_t)row_bytes, RARC_E_INVALID,
               "rarc_compact_rows: null / too small scratch or first hole outside the rows");
  hipStream_t s = (hipStream_t)stream;
  const int64_t n_new = n_rows - n_holes;
  const int64_t chunk = (int64_t)(tmp_bytes / (size_t)row_bytes);
  const bool wide = row_bytes % 16 == 0 && ((uintptr_t)d_rows % 16 == 0) && ((uintptr_t)d_tmp % 16 == 0);
  for (int64_t a = first_hole; a < n_new; a += chunk) {
    const int64_t m = n_new - a < chunk ? n_new - a : chunk;
    const int grid = (int)((m + 3) / 4 < 16384 ? (m + 3) / 4 : 16384);
    if (wide)
      hipLaunchKernelGGL(rarc_compact_gather_kernel<uint4>, dim3(grid), dim3(256), 0, s, (const uint4*)d_rows,
                         row_bytes / 16, d_adj, n_holes, a, m, (uint4*)d_tmp);
    else
      hipLaunchKernelGGL(rarc_compact_gather_kernel<uint32_t>, dim3(grid), dim3(256), 0, s, (const uint32_t*)d_rows,
                         row_bytes / 4, d_adj, n_holes, a, m, (uint32_t*)d_tmp);
    RARC_HIP_CHECK(hipGetLastError());
    RARC_HIP_CHECK(hipMemcpyAsync((char*)d_rows + a * row_bytes, d_tmp, (size_t)(m * row_bytes), hipMemcpyDeviceToDevice, s));
  }
  RARC_HIP_CHECK(hipMemsetAsync((char*)d_rows + n_new * row_bytes, 0, (size_t)(n_holes * row_bytes), s));
  return RARC_OK;
}
