// wide.hip — exact top-k for what the register-resident scans do not take: rows wider than 1024 (padded) dimensions
// and k beyond 1024.  faiss.IndexFlatIP takes any d and any k (encapsulation/database/vector_db/VectorStore_Faiss.py:101-115,
// :262-263), and the reference's other embedding source returns 1536- / 3072-d vectors (encapsulation/llm/openai_llm.py:139-161).
//
// The fast scans (scan_q8.hip, scan_f16.hip) keep the whole query block in registers — 256 x D int8 is 256 KB at D = 1024, a
// CU's register file is 512 KB — so beyond that the score matrix goes through the GEMM the encoder already has
// (encoder.hip: 256 x 256 x 64 ping-pong MFMA tiles, fp16 in, fp32 accumulate), one chunk of rows at a time:
//
//   first chunk (16384 rows + the shard's ragged end; no threshold yet):
//                                 S16[rows][256] = rows16 · Q16^T         rarc_enc_gemm_zero_bias   (MFMA; fp16 scores)
//                                 every row a candidate of every query    wide_select_kernel        (written by row number)
//   every later chunk (x4, up to 1,048,576 rows):
//                                 candidates += {(row, s) : s >= thr[q]}  rarc_gemm_f16_select      (the same GEMM, fp32 scores compared in its
//                                                                                                    epilogue, nothing stored: encoder.hip ACT 6)
//   after a chunk:                thr[q] = (k-th best s so far) - 2 eps   wide_tighten_kernel       (radix select; drops what fell under)
//   at the end, per query:        canonical fp32 score of every candidate, exact order, top-k      wide_finalize_kernel
//   (a chunk the fused GEMM does not take — RARC_WIDE_FUSE=0, odd shapes — goes through the first chunk's kernel pair with
//    the select pass comparing against thr)
//
// Exactness.  eps[q] bounds |s - canonical| for every row (fp16 rounding of the query and of the stored score, fp32
// accumulation: wide_eps_kernel).  If a_k is the k-th best approximate score seen so far, k rows have canonical scores
// >= a_k - eps, so the true k-th best canonical score L >= a_k - eps, and a row with canonical score >= L has s >= a_k - 2 eps:
// nothing below thr = a_k - 2 eps can be in the answer, ever (a_k only rises).  The candidates that reach the finalize
// therefore contain every row with canonical score >= L, ties included; their canonical scores are computed in the
// oracle's order (8 chains + tree) and ranked by (score desc, id asc).  The chunks grow geometrically from a first one of
// max(16384, 2k) rows, so the list of a query holds its k best plus a margin of rows, not a share of the corpus.  A candidate list
// that fills up sets the query's status word (the caller re-runs with a larger capacity; capacity >= n cannot overflow).
//
// HBM traffic per row: the 2·d_pad bytes of the row, once (PMC: 31.7 GB fetched per search of 30.7 GB of rows).
// Bound: the MFMA pipe (2·256·n·d_pad flops on the encoder's GEMM, which runs at the board's power limit: 10M x 1536 rows in
// 8.3 ms where the GEMM alone takes 8.28).
#include "rarc_common.h"

bool rarc_gemm_f16_select_takes(int m, int k);   // encoder.hip: the same GEMM with the select in its epilogue (no score matrix)
int rarc_gemm_f16_select(const uint16_t* a, const uint16_t* w, int m, int k, const float* thr, unsigned long long* cand,
                         uint32_t* count, uint32_t* status, uint32_t cap, uint32_t row0, uint32_t n_valid, uint32_t shards,
                         hipStream_t s);
extern "C" int rarc_enc_gemm_zero_bias(const uint16_t* d_a, const uint16_t* d_w, const uint16_t* d_zero_bias, uint16_t* d_c, int m,
                                       int n, int k, int act, void* stream);

namespace {
constexpr int WIDE_NQ = RARC_MAX_QUERIES;     // score columns per row (queries beyond nq are padding)
constexpr int WIDE_CHUNK = 131072;            // rows per GEMM that stores its scores (64 MB of fp16)
constexpr int WIDE_FUSED_CHUNK = 1 << 20;     // rows per GEMM that nominates in its epilogue (no buffer to size it by)
constexpr int WIDE_KMAX = 8192;               // largest k (the finalize sorts its answer in LDS: 64 KB)
// A query's candidate list is WIDE_SHARDS sub-lists of cap / WIDE_SHARDS entries with a counter each; a workgroup nominates
// into sub-list blockIdx.x % WIDE_SHARDS.  With one counter per query the ~400 nominations a growing chunk makes per query came
// from a thousand workgroups at once and queued on one address: 100 of the 150 us such a chunk took (in the fused GEMM and in
// the select pass alike).  The tighten reads the sub-lists one after the other and deals its survivors round-robin over them
// (flat, into sub-list 0's place onward, before the finalize).
constexpr int WIDE_SHARDS = 8;

struct WideWs {
  uint16_t* scores;     // [WIDE_CHUNK][256] fp16
  uint16_t* zero_bias;  // [256]
  uint16_t* tail;       // [128][d_pad] fp16: the last, partial 128-row block of a shard, zero padded
  float* thr;           // [256]
  float* eps;           // [256]
  uint32_t* count;      // [256][WIDE_SHARDS]
  uint32_t* count2;     // [256][WIDE_SHARDS]
  uint64_t* cand;       // [256][cap]
  uint64_t* cand2;      // [256][cap]  (compaction target; the two swap roles)
};
size_t wide_ws_bytes(int d_pad, int cap) {
  return (size_t)WIDE_CHUNK * WIDE_NQ * 2 + 4096 + (size_t)128 * d_pad * 2 + 8 * 4096 + 2 * (size_t)WIDE_NQ * cap * 8;
}
WideWs wide_carve(void* base, int d_pad, int cap) {
  char* b = (char*)base;
  WideWs w;
  w.scores = (uint16_t*)b; b += (size_t)WIDE_CHUNK * WIDE_NQ * 2;
  w.zero_bias = (uint16_t*)b; b += 4096;
  w.tail = (uint16_t*)b; b += (size_t)128 * d_pad * 2;
  w.thr = (float*)b; b += 1024;
  w.eps = (float*)b; b += 1024;
  w.count = (uint32_t*)b; b += 8192;
  w.count2 = (uint32_t*)b; b += 8192;
  b += 8 * 4096 - (2 * 1024 + 2 * 8192);
  w.cand = (uint64_t*)b; b += (size_t)WIDE_NQ * cap * 8;
  w.cand2 = (uint64_t*)b;
  return w;
}
}  // namespace

// eps[q] >= |fp16(fp32-accumulated q16·d) - canonical(q32·d)| for every stored row d:
//   query rounding ||q32 - q16||·||d||, both fp32 accumulations (d_pad·2^-23·||q||·||d||, Cauchy-Schwarz over the
//   |q_i d_i|), the fp16 rounding of the stored score (2^-11·||q||·||d||, + 2^-24 absolute under the normal range), and for
//   fp32 storage the image's distance rho = qmeta[1] from the rows the canonical score is taken on.
__global__ __launch_bounds__(256) void wide_eps_kernel(const float* q32, const uint16_t* q16, int d_pad, int nq,
                                                       float max_norm, float rho, float* eps, float* thr, uint32_t* count,
                                                       uint32_t* count2, uint32_t* status, uint16_t* zero_bias) {
  __shared__ double s_red[4][2];
  const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  double dn = 0.0, qn = 0.0;
  for (int m = tid; m < d_pad; m += 256) {
    const float v = q32[(size_t)q * d_pad + m];
    const float h = (float)__builtin_bit_cast(half_t, q16[(size_t)q * d_pad + m]);
    dn += ((double)v - (double)h) * ((double)v - (double)h);
    qn += (double)v * (double)v;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    dn += __shfl_xor(dn, o, 64);
    qn += __shfl_xor(qn, o, 64);
  }
  if (lane == 0) { s_red[tid >> 6][0] = dn; s_red[tid >> 6][1] = qn; }
  __syncthreads();
  if (tid == 0) {
    dn = (s_red[0][0] + s_red[1][0]) + (s_red[2][0] + s_red[3][0]);
    qn = (s_red[0][1] + s_red[1][1]) + (s_red[2][1] + s_red[3][1]);
    const double nq2 = sqrt(qn), mn = (double)max_norm;
    const double e = (sqrt(dn) + ((double)d_pad * 1.1920928955078125e-07 + 4.8828125e-04) * nq2) * mn * 1.01 + (double)rho * nq2 * 1.0001 +
                     6.0e-08 + 1e-30;
    eps[q] = (float)e * 1.0001f;
    thr[q] = q < nq ? -INFINITY : INFINITY;     // padding queries never nominate anything
    for (int sh = 0; sh < WIDE_SHARDS; ++sh) {
      count[q * WIDE_SHARDS + sh] = 0;
      count2[q * WIDE_SHARDS + sh] = 0;
    }
    status[q] = 0;
    zero_bias[q] = 0;
  }
}

// One pass over a chunk's scores: thread t of a 32-thread group takes 8 queries of a row (one 16-byte load); a workgroup
// walks 8 rows per step.  A survivor goes to its query's list with one global atomic (survivors are k per shard plus a
// margin — and the whole first chunk, which is what sizes it).
__global__ __launch_bounds__(256) void wide_select_kernel(const uint16_t* __restrict__ scores, uint32_t m_rows, uint32_t row0,
                                                          uint32_t n_valid, const float* __restrict__ thr, uint64_t* cand,
                                                          uint32_t* count, uint32_t cap, uint32_t* status, uint32_t nq, int first) {
  const int tid = threadIdx.x;
  const int qg = tid & 31;                       // queries [8 qg, 8 qg + 8)
  if (first) {
    // the shard's first chunk: no threshold yet, every row is a candidate of every live query — its place in the list is
    // its row number (n_valid <= cap), no counter to fight over
    // sub-list s takes the rows [s R, (s + 1) R), R = ceil(n_valid / 8) <= cap / 8: each one written front to back
    const uint32_t cap_f = cap / WIDE_SHARDS, per = (n_valid + WIDE_SHARDS - 1u) / WIDE_SHARDS;
    for (uint32_t r = blockIdx.x * 8 + (tid >> 5); r < n_valid; r += gridDim.x * 8) {
      const half8 s8 = *(const half8*)(scores + (size_t)r * WIDE_NQ + 8 * qg);
      const uint32_t sh = r / per;
      const size_t at = (size_t)sh * cap_f + (r - sh * per);
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if ((uint32_t)(8 * qg + j) < nq) cand[(size_t)(8 * qg + j) * cap + at] = rarc_candkey((float)s8[j], row0 + r);
    }
    if (blockIdx.x == 0 && (uint32_t)tid < nq)
      for (uint32_t sh = 0; sh < WIDE_SHARDS; ++sh)
        count[tid * WIDE_SHARDS + sh] = sh * per >= n_valid ? 0u : (n_valid - sh * per < per ? n_valid - sh * per : per);
    return;
  }
  const uint32_t shard = blockIdx.x % WIDE_SHARDS, cap_s = cap / WIDE_SHARDS;
  half_t t[8];                                   // the thresholds rounded DOWN to fp16: a pre-screen on the raw halves (never rejects what the fp32 threshold takes)
  half_t tmin = (half_t)65504.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float tf = thr[8 * qg + j];
    half_t h = (half_t)tf;                       // round to nearest ...
    if ((float)h > tf && tf > -65504.f) h = __builtin_bit_cast(half_t, (uint16_t)(__builtin_bit_cast(uint16_t, h) + ((float)h > 0.f ? -1 : 1)));   // ... then down to <= tf
    if (!(tf > -65504.f)) h = (half_t)-65504.f;  // -inf: everything passes (scores are finite)
    if (tf > 65504.f) h = (half_t)65504.f;       // +inf (padding queries): nothing finite reaches it — handled below
    t[j] = h;
    tmin = h < tmin ? h : tmin;
  }
  const bool dead = thr[8 * qg] > 65504.f && thr[8 * qg + 7] > 65504.f;   // (a whole group of padding queries)
  auto take = [&](uint32_t r, const half8& s8) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (s8[j] >= t[j]) {
        const float s = (float)s8[j];
        const uint32_t q = 8 * qg + j;
        if (s >= thr[q]) {                       // the exact (fp32) threshold decides: the fp16 one only pre-screens
          const uint32_t pos = atomicAdd(&count[q * WIDE_SHARDS + shard], 1u);
          if (pos < cap_s) cand[(size_t)q * cap + (size_t)shard * cap_s + pos] = rarc_candkey(s, row0 + r);
          else atomicOr(&status[q], RARC_Q_OVERFLOW | RARC_Q_WHY_SEGMENT);
        }
      }
    }
  };
  // four rows in flight per thread (the pass is a pure stream of 16-byte loads: 64 MB per chunk)
  const uint32_t stride = gridDim.x * 8;
  uint32_t r = blockIdx.x * 8 + (tid >> 5);
  for (; r + 3 * stride < n_valid; r += 4 * stride) {
    half8 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = *(const half8*)(scores + (size_t)(r + u * stride) * WIDE_NQ + 8 * qg);
    if (dead) continue;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      half_t mx = v[u][0];
#pragma unroll
      for (int j = 1; j < 8; ++j) mx = v[u][j] > mx ? v[u][j] : mx;
      if (mx >= tmin) take(r + u * stride, v[u]);
    }
  }
  for (; r < n_valid; r += stride) {
    const half8 s8 = *(const half8*)(scores + (size_t)r * WIDE_NQ + 8 * qg);
    if (!dead) take(r, s8);
  }
  (void)m_rows;
}

// k-th largest 32-bit value among n words read through `at(i)` by the whole block: four rounds of an 8-bit radix
// histogram in LDS.  Returns the value (every thread); n >= k >= 1.
template <typename At>
__device__ uint32_t wide_kth_largest_u32(At at, uint32_t n, uint32_t k, uint32_t* s_hist, uint32_t* s_pick) {
  uint32_t prefix = 0, mask = 0, need = k;
  for (int shift = 24; shift >= 0; shift -= 8) {
    for (uint32_t i = threadIdx.x; i < 256; i += blockDim.x) s_hist[i] = 0;
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
      const uint32_t v = at(i);
      if ((v & mask) == prefix) atomicAdd(&s_hist[(v >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (threadIdx.x < 64) {      // the bin holding the need-th largest, by one wave (a serial walk of the 256 counters by one
      uint32_t above = 0;        //  thread was 7 us a round — 28 of the 40 us a tighten took)
      int b = rarc_wave_find_from_top(s_hist, 256, need, &above);
      if (b < 0) { b = 0; above = 0; }           // (n >= need: cannot happen)
      if (threadIdx.x == 0) {
        s_pick[0] = (uint32_t)b;
        s_pick[1] = need - above;
      }
    }
    __syncthreads();
    prefix |= s_pick[0] << shift;
    mask |= 255u << shift;
    need = s_pick[1];
    __syncthreads();
  }
  return prefix;
}

// The same over entries a caller ENUMERATES (each(f): f(value) for every entry of this thread's share): the tighten's sub-lists.
template <typename Each>
__device__ uint32_t wide_kth_largest_each(Each each, uint32_t k, uint32_t* s_hist, uint32_t* s_pick) {
  uint32_t prefix = 0, mask = 0, need = k;
  for (int shift = 24; shift >= 0; shift -= 8) {
    for (uint32_t i = threadIdx.x; i < 256; i += blockDim.x) s_hist[i] = 0;
    __syncthreads();
    each([&](uint32_t v) {
      if ((v & mask) == prefix) atomicAdd(&s_hist[(v >> shift) & 255u], 1u);
    });
    __syncthreads();
    if (threadIdx.x < 64) {
      uint32_t above = 0;
      int b = rarc_wave_find_from_top(s_hist, 256, need, &above);
      if (b < 0) { b = 0; above = 0; }
      if (threadIdx.x == 0) {
        s_pick[0] = (uint32_t)b;
        s_pick[1] = need - above;
      }
    }
    __syncthreads();
    prefix |= s_pick[0] << shift;
    mask |= 255u << shift;
    need = s_pick[1];
    __syncthreads();
  }
  return prefix;
}

// thr[q] = max(thr[q], (k-th best approximate score so far) - 2 eps[q]); the list is copied without what fell under it: the
// sub-lists are read one after the other (entry L of the query = entry L - first[s] of sub-list s), the survivors dealt
// round-robin over the target's sub-lists — or laid out flat from its start (flat_out: the last pass, before the finalize).
__global__ __launch_bounds__(1024) void wide_tighten_kernel(const uint64_t* __restrict__ cand, uint64_t* __restrict__ cand2,
                                                            uint32_t* count, uint32_t* count2, uint32_t cap, uint32_t k,
                                                            const float* __restrict__ eps, float* thr, int flat_out) {
  __shared__ uint32_t s_hist[256];
  __shared__ uint32_t s_pick[2];
  __shared__ uint32_t s_n;
  const uint32_t q = blockIdx.x;
  const uint32_t cap_s = cap / WIDE_SHARDS;
  // 128 threads per sub-list (blockDim = 1024 = 8 x 128): my_first = where this thread's sub-list starts in the query's entries
  static_assert(WIDE_SHARDS == 8, "wide_tighten_kernel: 1024 threads = 8 sub-lists x 128");
  const uint32_t my_sh = threadIdx.x >> 7, my_lane = threadIdx.x & 127u;
  uint32_t c = 0, my_first = 0, my_n = 0;
#pragma unroll
  for (int sh = 0; sh < WIDE_SHARDS; ++sh) {
    const uint32_t c_raw = count[q * WIDE_SHARDS + sh], c_sh = c_raw < cap_s ? c_raw : cap_s;
    if ((uint32_t)sh < my_sh) my_first += c_sh;
    if ((uint32_t)sh == my_sh) my_n = c_sh;
    c += c_sh;
  }
  const uint64_t* src = cand + (size_t)q * cap;
  uint64_t* dst = cand2 + (size_t)q * cap;
  // every entry of the query once: each thread walks its own sub-list front to back
  const uint64_t* my_src = src + (size_t)my_sh * cap_s;
  auto each_key = [&](auto&& f) {
    for (uint32_t i = my_lane; i < my_n; i += 128u) f(my_src[i]);
  };
  float t = thr[q];
  // the list after a search's FIRST chunk is that whole chunk (16,640 entries x 256 queries = 34 MB): four select rounds and
  // the compaction read it five times — 30 of that pass's 37 us.  Up to TIGHTEN_LDS_KEYS entries the score words are read
  // once into LDS and the rounds run there.
  constexpr uint32_t TIGHTEN_LDS_KEYS = 20480;
  __shared__ uint32_t s_keys[TIGHTEN_LDS_KEYS];
  if (c >= k) {
    uint32_t kth;
    if (c <= TIGHTEN_LDS_KEYS) {
      for (uint32_t i = my_lane; i < my_n; i += 128u) s_keys[my_first + i] = (uint32_t)(my_src[i] >> 32);
      __syncthreads();
      kth = wide_kth_largest_u32([&](uint32_t i) { return s_keys[i]; }, c, k, s_hist, s_pick);
    } else {
      kth = wide_kth_largest_each([&](auto&& g) { each_key([&](uint64_t key) { g((uint32_t)(key >> 32)); }); }, k, s_hist, s_pick);
    }
    const float a_k = rarc_unordkey(kth);
    float nt = a_k - 2.0f * eps[q] * 1.000001f;
    nt = nt - fabsf(nt) * 1.2e-7f - 1e-37f;        // (rounded down: the bound must not be overstated by the subtraction)
    if (nt > t) t = nt;
  }
  if (threadIdx.x == 0) s_n = 0;
  __syncthreads();
  each_key([&](uint64_t key) {
    if (rarc_candscore(key) >= t) {
      const uint32_t p = atomicAdd(&s_n, 1u);
      dst[flat_out ? (size_t)p : (size_t)(p % WIDE_SHARDS) * cap_s + p / WIDE_SHARDS] = key;
    }
  });
  __syncthreads();
  if (threadIdx.x < WIDE_SHARDS) {
    const uint32_t sh = threadIdx.x, n_out = s_n;
    if (sh == 0) thr[q] = t;
    count2[q * WIDE_SHARDS + sh] = flat_out ? (sh == 0 ? n_out : 0u) : (n_out + WIDE_SHARDS - 1u - sh) / WIDE_SHARDS;
    count[q * WIDE_SHARDS + sh] = 0;            // this buffer is the next compaction's target
  }
}

// canonical fp32 score of a query with one stored row, by 8 lanes: lane j runs chain j (elements 8m + j, m ascending), the
// tree of rarc_canon_tree joins them — the same arithmetic, in the same order, as canon_dot_f16 / oracle canon_dot.
// fp16 rows: the group fetches 128 contiguous bytes per step (lane j the 16 bytes of elements 8(8b + j) .. + 7) and passes them
// through its 128 bytes of LDS, from which lane j picks element j of each of the eight pieces in ascending order — one
// 16-byte load per 64 elements and lane instead of eight 2-byte ones (the finalize was 100 us of a 490 us search of 100,000
// rows, 2 ms of 13 at k = 2000: all of it these loads).  A wave's LDS operations execute in order, so the lanes of a group
// (always inside one wave) see each other's writes without a barrier.  q: the query in LDS (fp32).
template <bool F32ROWS>
__device__ __forceinline__ float wide_canon_dot8(const float* q, const void* __restrict__ rows, size_t row, int d_pad, int j,
                                                 uint4* stage) {
  float a = 0.f;
  if (F32ROWS) {
    const float* r = (const float*)rows + row * (size_t)d_pad;
    for (int m = j; m < d_pad; m += 8) a = __builtin_fmaf(q[m], r[m], a);
  } else {
    const uint4* r = (const uint4*)((const half_t*)rows + row * (size_t)d_pad) + j;
    const half_t* sh = (const half_t*)stage + j;
    const int nblk = d_pad >> 6;                    // 64 elements per step (d_pad is a multiple of 64)
    uint4 v0 = r[0], v1 = nblk > 1 ? r[8] : v0;     // two steps in flight
    for (int b = 0; b < nblk; ++b) {
      const uint4 vn = b + 2 < nblk ? r[(b + 2) * 8] : v1;
      stage[j] = v0;
      asm volatile("" ::: "memory");
      const float* qb = q + 64 * b + j;
#pragma unroll
      for (int mm = 0; mm < 8; ++mm) a = __builtin_fmaf(qb[8 * mm], (float)sh[8 * mm], a);
      asm volatile("" ::: "memory");
      v0 = v1;
      v1 = vn;
    }
  }
  // ((a0 + a4) + (a2 + a6)) + ((a1 + a5) + (a3 + a7)): lanes j and j ^ 4, then j ^ 2, then j ^ 1
  a = a + __shfl_xor(a, 4, 8);
  a = a + __shfl_xor(a, 2, 8);
  a = a + __shfl_xor(a, 1, 8);
  return a;
}

// One workgroup per query: canonical scores of its candidates, the k best by (score desc, id asc), written out.
template <bool F32ROWS>
__global__ __launch_bounds__(1024) void wide_finalize_kernel(const void* __restrict__ rows, int d_pad, const float* __restrict__ q32,
                                                             uint64_t* cand, const uint32_t* count, uint32_t cap, uint32_t k,
                                                             uint32_t nq, int64_t id_base, int64_t* out_ids, float* out_scores,
                                                             uint32_t* status) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint64_t* s_top = (uint64_t*)smem;                       // [pow2 >= k]
  __shared__ uint32_t s_hist[256];
  __shared__ uint32_t s_pick[2];
  __shared__ uint32_t s_n;
  const uint32_t q = blockIdx.x;
  if (q >= nq) return;
  const uint32_t c_all = count[q * WIDE_SHARDS], c = c_all < cap ? c_all : cap;     // (laid out flat by the last tighten pass)
  if (c_all > cap && threadIdx.x == 0) atomicOr(&status[q], RARC_Q_OVERFLOW | RARC_Q_WHY_SEGMENT);
  uint64_t* keys = cand + (size_t)q * cap;
  __shared__ float s_q[4096];                               // the query (d_pad <= 4096)
  __shared__ uint4 s_stage[1024];                           // 128 bytes per 8-lane group
  for (int m = threadIdx.x; m < d_pad; m += blockDim.x) s_q[m] = q32[(size_t)q * d_pad + m];
  __syncthreads();
  // 1. canonical keys, in place (8 lanes per candidate)
  {
    const int j = threadIdx.x & 7;
    uint4* stage = s_stage + (threadIdx.x & ~7);
    for (uint32_t i = threadIdx.x >> 3; i < ((c + 127u) & ~127u); i += blockDim.x >> 3) {
      const bool live = i < c;
      const uint32_t row = live ? rarc_candrow(keys[i]) : 0u;
      const float s = wide_canon_dot8<F32ROWS>(s_q, rows, row, d_pad, j, stage);
      if (live && j == 0) keys[i] = rarc_candkey(s, row);
    }
  }
  __syncthreads();
  // 2. the k-th largest KEY (64 bits: score, then ~row = id ascending): radix select on the high word, then on the low word
  //    among the keys that share it
  const uint32_t kk = k < c ? k : c;
  uint32_t pow2 = 1;
  while (pow2 < kk) pow2 <<= 1;
  for (uint32_t i = threadIdx.x; i < pow2; i += blockDim.x) s_top[i] = 0;
  if (threadIdx.x == 0) s_n = 0;
  __syncthreads();
  if (kk > 0) {
    const uint32_t hi = wide_kth_largest_u32([&](uint32_t i) { return (uint32_t)(keys[i] >> 32); }, c, kk, s_hist, s_pick);
    // how many keys lie strictly above `hi` in the high word; the rest of the k come from the ties on it, by low word
    __shared__ uint32_t s_above;
    if (threadIdx.x == 0) s_above = 0;
    __syncthreads();
    uint32_t mine = 0;
    for (uint32_t i = threadIdx.x; i < c; i += blockDim.x) mine += ((uint32_t)(keys[i] >> 32) > hi);
    if (mine) atomicAdd(&s_above, mine);
    __syncthreads();
    const uint32_t need_lo = kk - s_above;   // >= 1
    // (the low word is ~row: larger = smaller id; ties on the high word are few except for duplicate rows)
    const uint32_t lo = wide_kth_largest_u32(
        [&](uint32_t i) { return (uint32_t)(keys[i] >> 32) == hi ? (uint32_t)keys[i] : 0u; }, c, need_lo, s_hist, s_pick);
    const uint64_t kth = ((uint64_t)hi << 32) | lo;
    for (uint32_t i = threadIdx.x; i < c; i += blockDim.x)
      if (keys[i] >= kth) {
        const uint32_t pos = atomicAdd(&s_n, 1u);
        if (pos < pow2) s_top[pos] = keys[i];
      }
    __syncthreads();
    // 3. exact order: bitonic sort, descending (zeros — below every real key — pad to the power of two)
    for (uint32_t kb = 2; kb <= pow2; kb <<= 1)
      for (uint32_t jb = kb >> 1; jb > 0; jb >>= 1) {
        for (uint32_t i = threadIdx.x; i < pow2; i += blockDim.x) {
          const uint32_t ixj = i ^ jb;
          if (ixj > i) {
            const uint64_t a = s_top[i], b = s_top[ixj];
            const bool desc = (i & kb) == 0;
            if (desc ? a < b : a > b) { s_top[i] = b; s_top[ixj] = a; }
          }
        }
        __syncthreads();
      }
  }
  for (uint32_t i = threadIdx.x; i < k; i += blockDim.x) {
    const bool have = i < kk;
    const uint64_t key = have ? s_top[i] : 0;
    out_ids[(size_t)q * k + i] = have ? id_base + (int64_t)rarc_candrow(key) : -1;
    out_scores[(size_t)q * k + i] = have ? rarc_candscore(key) : -INFINITY;
  }
}

extern "C" size_t rarc_wide_workspace_bytes(int d_pad, int cand_cap) {
  return (d_pad > 0 && cand_cap > 0) ? wide_ws_bytes(d_pad, cand_cap) + 256 : 0;
}

// d_rows: the stored rows (fp16 [n][d_pad]; fmt 2: fp32 [n][d_pad] with d_image16 their fp16 image, what the GEMM reads).
// d_qblock: written by rarc_prep_queries (q32, q16).  max_norm: the largest stored row norm; rho: fmt 2, >= ||row32 - image16||.
// d_status: uint32 [256], per-query flag words (RARC_Q_OVERFLOW: re-run with a larger cand_cap; cand_cap >= n_rows cannot
// overflow).  cand_cap >= max(16384, 2k) rounded up to 256, + 256.
extern "C" int rarc_search_wide(const void* d_rows, const uint16_t* d_image16, int fmt, int64_t n_rows, int d_pad, float max_norm,
                                float rho, const void* d_qblock, int nq, int k, int64_t id_base, int64_t* d_out_ids,
                                float* d_out_scores, uint32_t* d_status, void* d_ws, size_t ws_bytes, int cand_cap, void* stream) {
  RARC_RANGE();
  RARC_REQUIRE(d_rows && d_qblock && d_out_ids && d_out_scores && d_status && d_ws, RARC_E_INVALID, "rarc_search_wide: null pointer");
  RARC_REQUIRE((fmt == 0 && !d_image16) || (fmt == 2 && d_image16), RARC_E_INVALID,
               "rarc_search_wide: fmt 0 (fp16 rows) or 2 (fp32 rows + their fp16 image)");
  RARC_REQUIRE(d_pad > 0 && d_pad % RARC_DIM_ALIGN == 0 && d_pad <= 4096, RARC_E_UNSUPPORTED,
               "rarc_search_wide: padded dim %d unsupported (multiple of %d, <= 4096)", d_pad, RARC_DIM_ALIGN);
  RARC_REQUIRE(nq >= 1 && nq <= RARC_MAX_QUERIES && k >= 1 && k <= WIDE_KMAX && n_rows >= 0 && n_rows < (int64_t)0xffffff00ll,
               RARC_E_INVALID, "rarc_search_wide: need 1 <= nq <= %d, 1 <= k <= %d (nq=%d k=%d)", RARC_MAX_QUERIES, WIDE_KMAX, nq, k);
  // The first chunk — no threshold yet: every one of its rows is a candidate of every query, written by row number — is
  // max(16384, 2k) rows plus whatever makes the REST of the shard a whole number of 256-row blocks (the fused GEMM's unit; the
  // ragged end is dealt with here, once, by the kernel pair that takes any row count).  It was 2048 rows and the chunks
  // doubled: a 100,000-row shard then went through five fused GEMMs of 16..142 tiles, each a tile's full latency (60-100 us)
  // on a corner of the chip — 0.65 ms per batch for 0.08 ms of matrix work; 16384 rows fill the chip on the 128 x 128 kernel.
  int first = 2 * k > 16384 ? 2 * k : 16384;
  first = (first + 255) / 256 * 256;
  RARC_REQUIRE(cand_cap >= first + 256, RARC_E_WORKSPACE, "rarc_search_wide: cand_cap %d < %d (the first chunk of rows)", cand_cap,
               first + 256);
  RARC_REQUIRE(cand_cap % WIDE_SHARDS == 0, RARC_E_INVALID, "rarc_search_wide: cand_cap %d must be a multiple of %d", cand_cap, WIDE_SHARDS);
  char* wsb = (char*)(((uintptr_t)d_ws + 255) & ~(uintptr_t)255);
  RARC_REQUIRE(wsb + wide_ws_bytes(d_pad, cand_cap) <= (char*)d_ws + ws_bytes, RARC_E_WORKSPACE,
               "rarc_search_wide: workspace of %zu bytes, %zu needed", ws_bytes, wide_ws_bytes(d_pad, cand_cap) + 256);
  hipStream_t s = (hipStream_t)stream;
  WideWs w = wide_carve(wsb, d_pad, cand_cap);
  const RarcQb qb = rarc_qb_carve(d_qblock, d_pad);
  hipLaunchKernelGGL(wide_eps_kernel, dim3(WIDE_NQ), dim3(256), 0, s, qb.q32, qb.q16, d_pad, nq, max_norm, rho, w.eps, w.thr,
                     w.count, w.count2, d_status, w.zero_bias);
  RARC_HIP_CHECK(hipGetLastError());
  const uint16_t* a16 = fmt == 2 ? d_image16 : (const uint16_t*)d_rows;
  uint64_t *cur = w.cand, *other = w.cand2;
  uint32_t *ccur = w.count, *cother = w.count2;
  int64_t at = 0;
  int64_t chunk = n_rows > first ? first + (n_rows - first) % 256 : n_rows;
  int64_t nominal = first;                    // the chunk sizes' geometric ladder: first, first * g, ... up to fused_max
  const int64_t growth = k <= 1024 ? 4 : 2;   // (a chunk of g x the rows seen lets ~g k rows per query through: lists of 16384)
  int n_chunk = 0;
  int64_t fused_max = WIDE_FUSED_CHUNK;
  if (const char* e = getenv("RARC_WIDE_FUSED_ROWS")) {   // (experiments: tools/wide_chunk_sweep.sh)
    if (atoll(e) >= WIDE_CHUNK) fused_max = atoll(e) / WIDE_CHUNK * WIDE_CHUNK;
  }
  while (at < n_rows) {
    int64_t m = n_rows - at < chunk ? n_rows - at : chunk;
    int rc;
    // every chunk but the first: the GEMM nominates in its epilogue (fp32 scores against the thresholds, no 64 MB of
    // scores written and read back) for the whole 256-row blocks; what is left of the chunk takes the two-kernel form below
    static const bool fuse = !(getenv("RARC_WIDE_FUSE") && atoi(getenv("RARC_WIDE_FUSE")) == 0);
    if (fuse && at > 0 && m >= 256 && rarc_gemm_f16_select_takes((int)(m / 256 * 256), d_pad)) {
      const int64_t mf = m / 256 * 256;
      if ((rc = rarc_gemm_f16_select(a16 + (size_t)at * d_pad, qb.q16, (int)mf, d_pad, w.thr, (unsigned long long*)cur, ccur, d_status,
                                     (uint32_t)cand_cap, (uint32_t)at, (uint32_t)mf, (uint32_t)WIDE_SHARDS, s)) != RARC_OK)
        return rc;
      at += mf;
      m -= mf;
      if (m == 0) {
        ++n_chunk;
        if (chunk < fused_max || chunk >= 4 * WIDE_CHUNK || n_chunk % 4 == 0 || at >= n_rows) {   // (every chunk while they still grow: a stale threshold lets chunk/seen x k rows through)
          hipLaunchKernelGGL(wide_tighten_kernel, dim3(WIDE_NQ), dim3(1024), 0, s, cur, other, ccur, cother, (uint32_t)cand_cap,
                             (uint32_t)k, w.eps, w.thr, at >= n_rows ? 1 : 0);
          RARC_HIP_CHECK(hipGetLastError());
          { uint64_t* t = cur; cur = other; other = t; }
          { uint32_t* t = ccur; ccur = cother; cother = t; }
        }
        // (nothing is stored per row any more: past the ramp a fused chunk may be eight times the score buffer's rows —
        //  fewer launches, a tighten per ~1M rows.  k in the thousands was 2x slower with them at first, for two reasons since
        //  removed: a counter update per nomination inside the GEMM (now one per 16-lane group), and no tighten between the
        //  growing chunks past 131072 rows — a threshold from 126K rows let 8 x k rows of a 524K-row chunk through and the lists
        //  overflowed, i.e. every search ran twice.  10M x 1536, k = 2000: 16.8 -> 13.3 ms.)
        nominal = nominal * growth < fused_max ? nominal * growth : fused_max;
        chunk = nominal;
        continue;
      }
    }
    if (m > WIDE_CHUNK) {        // (the remainder of a fused-size chunk that could not be fused: back to the buffer's size)
      chunk = WIDE_CHUNK;
      m = WIDE_CHUNK;
    }
    const int64_t m_full = m / 128 * 128;
    if (m_full > 0) {
      if ((rc = rarc_enc_gemm_zero_bias(a16 + (size_t)at * d_pad, qb.q16, w.zero_bias, w.scores, (int)m_full, WIDE_NQ, d_pad, 0,
                                        stream)) != RARC_OK)
        return rc;
    }
    if (m_full < m) {      // the shard's last rows: a zero-padded 128-row block of their own
      const int64_t rest = m - m_full;
      RARC_HIP_CHECK(hipMemsetAsync(w.tail, 0, (size_t)128 * d_pad * 2, s));
      RARC_HIP_CHECK(hipMemcpyAsync(w.tail, a16 + (size_t)(at + m_full) * d_pad, (size_t)rest * d_pad * 2, hipMemcpyDeviceToDevice, s));
      if ((rc = rarc_enc_gemm_zero_bias(w.tail, qb.q16, w.zero_bias, w.scores + (size_t)m_full * WIDE_NQ, 128, WIDE_NQ, d_pad, 0,
                                        stream)) != RARC_OK)
        return rc;
    }
    const uint32_t m_sel = (uint32_t)((m + 127) / 128 * 128);
    const int grid = (int)((m + 7) / 8 < 2048 ? (m + 7) / 8 : 2048);
    hipLaunchKernelGGL(wide_select_kernel, dim3(grid), dim3(256), 0, s, w.scores, m_sel, (uint32_t)at, (uint32_t)m, w.thr, cur, ccur,
                       (uint32_t)cand_cap, d_status, (uint32_t)nq, at == 0 ? 1 : 0);
    RARC_HIP_CHECK(hipGetLastError());
    at += m;
    // the threshold is raised (and the lists cut back) after every chunk while the chunks still grow, then after every
    // fourth: a full-size chunk adds a handful of rows to a list (k·131072/n), the radix select over it costs what a
    // third of the chunk's GEMM does
    ++n_chunk;
    if (chunk < WIDE_CHUNK || n_chunk % 4 == 0 || at >= n_rows) {
      hipLaunchKernelGGL(wide_tighten_kernel, dim3(WIDE_NQ), dim3(1024), 0, s, cur, other, ccur, cother, (uint32_t)cand_cap,
                         (uint32_t)k, w.eps, w.thr, at >= n_rows ? 1 : 0);
      RARC_HIP_CHECK(hipGetLastError());
      { uint64_t* t = cur; cur = other; other = t; }
      { uint32_t* t = ccur; ccur = cother; cother = t; }
    }
    // chunks grow geometrically (x4, x2 for k in the thousands) up to the full size: under the k-th best score of the S rows
    // seen so far a chunk of g S more rows lets about g k of them through (growing 8x it was 7k: for k in the thousands that
    // filled the lists).  (The fused form above takes over after the first chunk wherever the GEMM takes the shape; this
    // branch continues the ladder for what it leaves.)
    nominal = nominal * growth < fused_max ? nominal * growth : fused_max;
    chunk = nominal < WIDE_CHUNK ? nominal : WIDE_CHUNK;
  }
  uint32_t pow2 = 1;
  while (pow2 < (uint32_t)k) pow2 <<= 1;
  const size_t lds = (size_t)pow2 * 8;
  if (fmt == 2) {
    static RarcPerDevice attr_done;
    if (size_t& done = attr_done.cur(); !done) {
      RARC_HIP_CHECK(hipFuncSetAttribute((const void*)wide_finalize_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
      done = 1;
    }
    hipLaunchKernelGGL(wide_finalize_kernel<true>, dim3(WIDE_NQ), dim3(1024), lds, s, d_rows, d_pad, qb.q32, cur, ccur,
                       (uint32_t)cand_cap, (uint32_t)k, (uint32_t)nq, id_base, d_out_ids, d_out_scores, d_status);
  } else {
    static RarcPerDevice attr_done;
    if (size_t& done = attr_done.cur(); !done) {
      RARC_HIP_CHECK(hipFuncSetAttribute((const void*)wide_finalize_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
      done = 1;
    }
    hipLaunchKernelGGL(wide_finalize_kernel<false>, dim3(WIDE_NQ), dim3(1024), lds, s, d_rows, d_pad, qb.q32, cur, ccur,
                       (uint32_t)cand_cap, (uint32_t)k, (uint32_t)nq, id_base, d_out_ids, d_out_scores, d_status);
  }
  RARC_HIP_CHECK(hipGetLastError());
  return RARC_OK;
}
