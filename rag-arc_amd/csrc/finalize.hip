// finalize.hip — per-query: select the k' best scan candidates, rescore them with the
// canonical fp32 scorer, certify exactness, sort (score desc, id asc), emit top-k.
//
// Completes faiss.IndexFlatIP.search (VectorStore_Faiss.py:263): the MFMA scan only
// NOMINATES rows; the scores and order returned to the caller come from the canonical fp32
// inner product of DESIGN.md §canonical-score, which oracle/rarc_oracle.c evaluates in the
// same order, so ids and scores are bit-identical to the oracle's.
//
// Also: rarc_repair (exact single-query rescan) and rarc_topk_merge (multi-shard merge).
#include "rarc_common.h"
#include <type_traits>

constexpr int FIN_THREADS = 256;

// canonical fp32 dot of a fp32 query with an fp16 row (d multiple of 8)
__device__ __forceinline__ float canon_dot_f16(const float* __restrict__ q,
                                               const half_t* __restrict__ row, int d) {
  float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll 16
  for (int m = 0; m < d; m += 8) {
    const half8 x = *(const half8*)(row + m);
    const float4 q0 = *(const float4*)(q + m);
    const float4 q1 = *(const float4*)(q + m + 4);
    a[0] = __builtin_fmaf(q0.x, (float)x[0], a[0]);
    a[1] = __builtin_fmaf(q0.y, (float)x[1], a[1]);
    a[2] = __builtin_fmaf(q0.z, (float)x[2], a[2]);
    a[3] = __builtin_fmaf(q0.w, (float)x[3], a[3]);
    a[4] = __builtin_fmaf(q1.x, (float)x[4], a[4]);
    a[5] = __builtin_fmaf(q1.y, (float)x[5], a[5]);
    a[6] = __builtin_fmaf(q1.z, (float)x[6], a[6]);
    a[7] = __builtin_fmaf(q1.w, (float)x[7], a[7]);
  }
  return rarc_canon_tree(a);
}

// canonical fp32 dot of a fp32 query with an fp32 row (d multiple of 8): the same 8 chains and tree
__device__ __forceinline__ float canon_dot_f32(const float* __restrict__ q, const float* __restrict__ row, int d) {
  float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
  for (int m = 0; m < d; m += 8) {
    const float4 x0 = *(const float4*)(row + m), x1 = *(const float4*)(row + m + 4);
    const float4 q0 = *(const float4*)(q + m), q1 = *(const float4*)(q + m + 4);
    a[0] = __builtin_fmaf(q0.x, x0.x, a[0]);
    a[1] = __builtin_fmaf(q0.y, x0.y, a[1]);
    a[2] = __builtin_fmaf(q0.z, x0.z, a[2]);
    a[3] = __builtin_fmaf(q0.w, x0.w, a[3]);
    a[4] = __builtin_fmaf(q1.x, x1.x, a[4]);
    a[5] = __builtin_fmaf(q1.y, x1.y, a[5]);
    a[6] = __builtin_fmaf(q1.z, x1.z, a[6]);
    a[7] = __builtin_fmaf(q1.w, x1.w, a[7]);
  }
  return rarc_canon_tree(a);
}

// in-LDS bitonic sort, descending, n = power of two, all threads of the block participate
__device__ __forceinline__ void bitonic_desc(uint64_t* a, int n) {
  for (int k = 2; k <= n; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const int ixj = i ^ j;
        if (ixj > i) {
          const uint64_t x = a[i], y = a[ixj];
          const bool up = ((i & k) == 0);  // descending block
          if (up ? (x < y) : (x > y)) {
            a[i] = y;
            a[ixj] = x;
          }
        }
      }
      __syncthreads();
    }
  }
}

struct FinParams {
  const half_t* corpus;
  const float* q32;  // [256][d]
  const float* eps;  // [256]
  const uint32_t* cnt2;  // [256 wg][256 q]
  const uint64_t* cand;  // [256 q][256 wg][seg]
  const uint32_t* hist;
  const float* binlo;
  const float* bininv;
  uint32_t* flags;
  uint32_t seg;
  uint32_t n_wg;
  int d;
  int k, kprime;
  int64_t id_base;
  int64_t* out_ids;
  float* out_scores;
  uint32_t* status;
  uint32_t* flag_host;   // pinned host word (or null): set non-zero when a query is flagged (RarcLaunchExtras)
};

constexpr int FIN_SURV = 4096;  // survivors of the final-threshold compaction that get ranked

// one workgroup per query
__global__ __launch_bounds__(FIN_THREADS) void rarc_finalize_kernel(const FinParams p) {
  __shared__ uint64_t keys[FIN_SURV];
  __shared__ uint64_t sel[RARC_MAX_K];
  __shared__ uint64_t fin[RARC_MAX_K];
  __shared__ uint32_t s_hist[RARC_NB];
  __shared__ __attribute__((aligned(16))) float s_q[768];
  __shared__ float s_thr;
  __shared__ uint32_t s_ns, s_total, s_over, s_maxn;
  const int q = blockIdx.x, tid = threadIdx.x;

  // ---- tightest valid threshold from the final histogram (same rule as the scan's owners) ----
  for (int i = tid; i < RARC_NB; i += blockDim.x) s_hist[i] = p.hist[(size_t)q * RARC_NB + i];
  for (int i = tid; i < p.d; i += blockDim.x) s_q[i] = p.q32[(size_t)q * p.d + i];
  if (tid == 0) { s_ns = 0; s_total = 0; s_over = 0; s_maxn = 0; }
  __syncthreads();
  if (tid < 64) {
    uint32_t above;
    const int b = rarc_wave_find_from_top(s_hist, RARC_NB, (uint32_t)p.kprime, &above);
    if (tid == 0) s_thr = rarc_bin_threshold(b, p.binlo[q], p.bininv[q]);
  }
  __syncthreads();
  const float thr = s_thr;

  // ---- gather this query's segments (thread w <-> scan workgroup w), keep what clears the threshold ----
  {
    uint32_t n = 0;
    const uint64_t* src = p.cand;
    if ((uint32_t)tid < p.n_wg) {
      const uint32_t c = p.cnt2[(size_t)tid * RARC_MAX_QUERIES + q];
      n = c < p.seg ? c : p.seg;
      if (c > p.seg) atomicOr(&s_over, 1u);
      atomicAdd(&s_total, c);
      src = p.cand + ((size_t)q * RARC_MAX_WG + tid) * p.seg;
    }
    // first 16 slots of the segment in one memory round trip (8 x 16 B), the rest (rare) one by one
    uint64_t first[16];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const ulonglong2 v = ((const ulonglong2*)src)[i];
      first[2 * i] = v.x;
      first[2 * i + 1] = v.y;
    }
#pragma unroll
    for (uint32_t i = 0; i < 16; ++i) {
      if (i < n && rarc_candscore(first[i]) >= thr) {
        const uint32_t pos = atomicAdd(&s_ns, 1u);
        if (pos < FIN_SURV) keys[pos] = first[i];
      }
    }
    for (uint32_t i = 16; i < n; ++i) {
      const uint64_t key = src[i];
      if (rarc_candscore(key) >= thr) {
        const uint32_t pos = atomicAdd(&s_ns, 1u);
        if (pos < FIN_SURV) keys[pos] = key;
      }
    }
  }
  __syncthreads();
  const uint32_t ns_all = s_ns, total = s_total;
  const int ns = ns_all < FIN_SURV ? (int)ns_all : FIN_SURV;

  // ---- the kp best by approximate score: rank by counting (keys are distinct) ----
  const int kp = ns < p.kprime ? ns : p.kprime;
  for (int i = tid; i < ns; i += blockDim.x) {
    const uint64_t mine = keys[i];
    int rank = 0;
    for (int j = 0; j < ns; ++j) rank += (keys[j] > mine);
    if (rank < kp) sel[rank] = mine;
  }
  __syncthreads();

  // ---- canonical rescore of the selected rows (query vector staged in LDS: shared by all rows) ----
  const float* qv = s_q;
  for (int i = tid; i < kp; i += blockDim.x) {
    const uint32_t row = rarc_candrow(sel[i]);
    fin[i] = rarc_candkey(canon_dot_f16(qv, p.corpus + (size_t)row * p.d, p.d), row);
  }
  __syncthreads();
  // final order (canonical score desc, id asc) by counting; written straight to the outputs
  for (int i = tid; i < p.k; i += blockDim.x) {
    if (i >= kp) {
      p.out_ids[(size_t)q * p.k + i] = -1;
      p.out_scores[(size_t)q * p.k + i] = -INFINITY;
    }
  }
  __shared__ float s_sk;
  for (int i = tid; i < kp; i += blockDim.x) {
    const uint64_t mine = fin[i];
    int rank = 0;
    for (int j = 0; j < kp; ++j) rank += (fin[j] > mine);
    if (rank < p.k) {
      p.out_ids[(size_t)q * p.k + rank] = p.id_base + (int64_t)rarc_candrow(mine);
      p.out_scores[(size_t)q * p.k + rank] = rarc_candscore(mine);
    }
    if (rank == (p.k < kp ? p.k : kp) - 1) s_sk = rarc_candscore(mine);
  }
  __syncthreads();

  // ---- certificate: no unselected row can reach the k-th canonical score ----
  // every unselected row has approximate score <= t_min, hence canonical <= t_min + eps.
  if (tid == 0) {
    uint32_t st = RARC_Q_OK;
    if (s_over || ns_all > (uint32_t)FIN_SURV) st |= RARC_Q_OVERFLOW;
    if (total >= (uint32_t)p.kprime && kp >= 1) {  // rows may have been left out: need the margin
      const float tmin = rarc_candscore(sel[kp - 1]);
      if (!(tmin + p.eps[q] < s_sk)) st |= RARC_Q_UNCERTAIN;
    }
    p.status[q] = st;
    if (st) {  // one word the host can poll instead of scanning d_status
      atomicOr(&p.flags[1], st);
      atomicOr(&p.status[RARC_MAX_QUERIES], st);
      if (p.flag_host) __hip_atomic_store(p.flag_host, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

int rarc_finalize_launch(const uint16_t* corpus, int d_pad, const float* q32, const float* eps, int nq,
                         int k, int kprime, int64_t id_base, const RarcWs& ws, int cap, int n_wg,
                         int64_t* out_ids, float* out_scores, uint32_t* status, hipStream_t s) {
  FinParams p;
  p.corpus = (const half_t*)corpus;
  p.q32 = q32;
  p.eps = eps;
  p.cnt2 = ws.cnt2;
  p.cand = ws.cand;
  p.hist = ws.hist;
  p.binlo = ws.binlo;
  p.bininv = ws.bininv;
  p.flags = ws.flags;
  p.seg = (uint32_t)(cap / RARC_MAX_WG);
  p.n_wg = (uint32_t)n_wg;
  p.d = d_pad;
  p.k = k;
  p.kprime = kprime;
  p.id_base = id_base;
  p.out_ids = out_ids;
  p.out_scores = out_scores;
  p.status = status;
  p.flag_host = rarc_launch_extras().flag_host;
  hipLaunchKernelGGL(rarc_finalize_kernel, dim3(nq), dim3(FIN_THREADS), 0, s, p);
  RARC_HIP_CHECK(hipGetLastError());
  return RARC_OK;
}

// ============================================================================================
// Finalize for the int8-prefilter scan (scan_q8.hip).  The candidate list of a query holds every row
// whose int8 score a satisfies a >= thr, with thr <= a_(k) − 2·eps (a_(k): k-th best int8 score,
// eps: the query's bound on |canonical − a|).  Per query:
//   1. T1 = histogram edge with >= k candidates at or above it; G1 = {a >= T1}: rescore canonically,
//      L = k-th best canonical score in G1  (a lower bound of the true k-th best score, no eps in it;
//      L >= T1 − eps)
//   2. G2 = {a + eps >= L} \ G1: rescore canonically.  Every other row has canonical <= a + eps < L,
//      so it is not in the top-k; and a >= L − eps >= T1 − 2·eps >= thr, so G2 is inside the list.
//   3. exact top-k of G1 ∪ G2 by (canonical score desc, id asc).
// Nothing here is probabilistic: the only failure mode is running out of buffer space, which sets
// RARC_Q_OVERFLOW and sends the query to rarc_repair_f16.
// ============================================================================================
constexpr int FIN8_THREADS = 512;
constexpr int FIN8_RS = 6144;     // rows rescored canonically per query (G1 ∪ G2); 48 KiB of LDS beside the 97 KiB row staging
constexpr int FIN8_MAXD = 1024;
#ifndef FIN8_COLLECT_U
#define FIN8_COLLECT_U 8
#endif

struct Fin8Params {
  const void* corpus;      // the rows the canonical rescore reads: fp16 (fmt 0), fp8 (fmt 1) or fp32 (fmt 2)
  const float* rowscale;   // fmt 1: per-row scales
  int fmt;
  const float* q32;   // [256][d]
  const float* eps8;  // [256]
  const uint32_t* cnt2;
  const uint64_t* cand;
  const uint32_t* hist;
  const float* binlo;
  const float* bininv;
  uint32_t* flags;
  uint32_t seg;
  uint32_t n_wg;
  int d;
  int k;
  int64_t id_base;
  int64_t* out_ids;
  float* out_scores;
  uint32_t* status;
  const float* tmeta;   // per-tile metadata of the scanned image (qmeta + RARC_QMETA_HDR): word 0 carries R_t
  int mstride;          // floats per tile in tmeta (2, or 34 for fp8 rows)
  const float* hq;      // [256] ||q8/s_q||: inside tile t the error bound is eps8 - hq·(R - R_t)  (scan_q8.hip)
  unsigned long long* dbg;  // tools: phase stamps of block 0 (100 MHz ticks), else null
  uint32_t* tighten_thr;    // non-null: stop after step 1 and raise thr[q] to L - eps8 (split scan, scan_q8.hip)
  int tighten_mode;         // 1: raise only.  2: the candidates so far came from the fp16 STAGE of the hybrid search, whose
                            // thresholds live under an eps16 margin — too high for the int8 stage that follows: thr[q] is
                            // SET to L - eps8 (no k-th score yet: lowered by eps8 - eps16, which is valid too)
  const float* eps16;       // hybrid search: some candidates carry fp16 scores (error <= eps16[q]); null otherwise
  uint32_t* flag_host;      // pinned host word (or null): set non-zero when a query is flagged (RarcLaunchExtras)
};

__global__ __launch_bounds__(FIN8_THREADS) void rarc_finalize_q8_kernel(const Fin8Params p) {
  extern __shared__ __attribute__((aligned(16))) char fsm[];  // row staging: [waves][8 rows][2d + 16]
  __shared__ uint64_t ex[FIN8_RS];  // candidate keys; rescored in place (approx key -> canonical key)
  __shared__ __attribute__((aligned(16))) float s_q[FIN8_MAXD];
  __shared__ uint32_t s_hist[RARC_NB];
  __shared__ uint32_t s_cnt[RARC_MAX_WG];  // candidates each scan workgroup produced for this query
  __shared__ float s_t1, s_L;
  __shared__ uint32_t s_ne, s_over, s_cmp;
  __shared__ int s_selb;
  __shared__ uint32_t s_selabove;
  const int q = blockIdx.x, tid = threadIdx.x;
  const float eps = p.eps8[q];
#define FIN8_STAMP(i) if (p.dbg && q == 0 && tid == 0) p.dbg[i] = __builtin_amdgcn_s_memrealtime();
  FIN8_STAMP(0)

  for (int i = tid; i < RARC_NB; i += blockDim.x) s_hist[i] = p.hist[(size_t)q * RARC_NB + i];
  for (int i = tid; i < p.d; i += blockDim.x) s_q[i] = p.q32[(size_t)q * p.d + i];
  if (tid == 0) { s_ne = 0; s_over = 0; s_L = -INFINITY; }
  __syncthreads();
  for (int i = tid; i < RARC_MAX_WG; i += blockDim.x) {
    const uint32_t c = (uint32_t)i < p.n_wg ? p.cnt2[(size_t)i * RARC_MAX_QUERIES + q] : 0u;
    s_cnt[i] = c;
    if (c > p.seg) atomicOr(&s_over, 1u);
  }
  if (tid < 64) {
    uint32_t above;
    const int b = rarc_wave_find_from_top(s_hist, RARC_NB, (uint32_t)p.k, &above);
    if (tid == 0) s_t1 = rarc_bin_threshold(b, p.binlo[q], p.bininv[q]);
  }
  __syncthreads();
  float t1 = s_t1;  // -inf when fewer than k candidates exist: then G1 is everything

  // walk the query's segments and append the keys `want` accepts to ex[].  One wave per segment at a
  // time, 64 consecutive keys per load (coalesced); a wave issues the loads of four segments before it
  // looks at any of them, so a sweep over ~30K keys costs a handful of memory round trips.
  auto collect = [&](auto want) {
    // segments in flight per wave: the sweep is a chain of dependent memory round trips, (256 / waves / U) x
    // (longest segment / 64) of them — 4: 22 us per sweep of ~20 K keys; 8, 16: 17-18 us; 32: 20 us (41 MB of
    // 512-byte pieces for the 256 queries together: the sweep is at what cold HBM gives such reads).
    // (One flat index space over all segments with a per-key binary search of the prefix sums: 32 us.)
    constexpr int U = FIN8_COLLECT_U;
    const int lane = tid & 63, wv = tid >> 6, nwv = blockDim.x >> 6;
    for (uint32_t wg0 = wv * U; wg0 < p.n_wg; wg0 += nwv * U) {
      uint32_t n[U];
      uint32_t nmax = 0;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint32_t wg = wg0 + u;
        const uint32_t c = wg < p.n_wg ? s_cnt[wg] : 0u;
        n[u] = c < p.seg ? c : p.seg;
        nmax = n[u] > nmax ? n[u] : nmax;
      }
      for (uint32_t base = 0; base < nmax; base += 64) {
        uint64_t key[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const uint32_t i = base + lane;
          key[u] = i < n[u] ? p.cand[((size_t)q * RARC_MAX_WG + wg0 + u) * p.seg + i] : 0ull;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          if (base + lane < n[u] && want(rarc_candscore(key[u]), rarc_candrow(key[u]))) {
            const uint32_t e = atomicAdd(&s_ne, 1u);
            if (e < FIN8_RS) ex[e] = key[u];
          }
        }
      }
    }
  };
  // Canonical rescore of ex[from..to): 8 lanes per row, lane j runs chain j (elements 8m+j, m
  // ascending, one fma each — the order the oracle uses), then the canonical tree over the 8 lanes.
  // A wave stages its 8 rows in LDS with coalesced loads (one instruction covers 8 rows x 128
  // contiguous bytes) and reads them back transposed, 2 bytes per lane per step; the loads of the
  // next 64 rows are in flight while the current ones are reduced.  (One row per THREAD, 16 bytes at
  // a time from 64 different rows per instruction, took 260 µs for 670 rows per query.)
  // NT = 16-byte loads per lane per row (8 lanes x 16 B = 128 B per step), a compile-time constant: with a
  // run-time count the prefetch registers `pre` were demoted to scratch memory, every global load was waited
  // for and copied there at once, and the prefetch hid nothing (17 us per round of 64 rows instead of 3)
  auto rescore_n = [&](auto nt_c, int from, int to) __attribute__((always_inline)) {
    constexpr int NT = decltype(nt_c)::value;
    const int lane = tid & 63, wv = tid >> 6, j = lane & 7, rr = lane >> 3;
    constexpr int rbytes = NT * 128;                       // bytes of one stored row
    constexpr int rstride = rbytes + 16;                   // +16: the 8 rows of a wave start in 8 different bank groups
    char* stage = fsm + (size_t)wv * 8 * rstride;          // this wave's 8 rows
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));  // (HIP's uint4 struct in a conditionally written array stays in memory)
    u32x4 pre[NT];
    auto issue = [&](int i) {                              // lane fetches chunks t*8 + j of row ex[i]
      if (i < to) {
        const u32x4* src = (const u32x4*)((const char*)p.corpus + (size_t)rarc_candrow(ex[i]) * rbytes) + j;
#pragma unroll
        for (int t = 0; t < NT; ++t) pre[t] = src[t * 8];
      }
    };
    int i = from + wv * 8 + rr;
    issue(i);
    for (; i - rr - wv * 8 < to; i += (blockDim.x >> 6) * 8) {  // wave-uniform trip count
#pragma unroll
      for (int t = 0; t < NT; ++t) *(u32x4*)(stage + rr * rstride + (t * 8 + j) * 16) = pre[t];
      const int cur = i;
      issue(i + (blockDim.x >> 6) * 8);
      __builtin_amdgcn_wave_barrier();
      float acc = 0.f;
      if (cur < to) {
        if (p.fmt == 2) {  // fp32 rows: element 8m + j is the float at byte 32m + 4j
          const char* rowp = stage + rr * rstride + 4 * j;
          for (int m0 = 0; m0 < NT * 4; m0 += 16) {  // d = rbytes / 4, a multiple of 128
            float xs[16], qs[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
              xs[u] = *(const float*)(rowp + 32 * (m0 + u));
              qs[u] = s_q[8 * (m0 + u) + j];
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) acc = __builtin_fmaf(qs[u], xs[u], acc);
          }
        } else if (p.fmt) {  // fp8: element 8m + j is byte 8m + j of the row
          const char* rowp = stage + rr * rstride + j;
          for (int m0 = 0; m0 < NT * 16; m0 += 16) {  // d = rbytes
            float xs[16], qs[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
              xs[u] = __builtin_amdgcn_cvt_f32_fp8((uint32_t) * (const uint8_t*)(rowp + 8 * (m0 + u)), 0);
              qs[u] = s_q[8 * (m0 + u) + j];
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) acc = __builtin_fmaf(qs[u], xs[u], acc);
          }
        } else {
          const char* rowp = stage + rr * rstride + 2 * j;
          for (int m0 = 0; m0 < NT * 8; m0 += 16) {  // d = rbytes / 2, a multiple of 128: 16 chain steps per trip
            float xs[16], qs[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {  // all 32 LDS reads of the trip issued before the first fma
              xs[u] = (float)*(const half_t*)(rowp + 16 * (m0 + u));
              qs[u] = s_q[8 * (m0 + u) + j];
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) acc = __builtin_fmaf(qs[u], xs[u], acc);
          }
        }
      }
      float a8[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) a8[u] = __shfl(acc, (lane & ~7) + u, 64);
      if (cur < to && j == 0) {
        const uint32_t row = rarc_candrow(ex[cur]);
        float sc = rarc_canon_tree(a8);
        if (p.fmt == 1) sc = p.rowscale[row] * sc;
        ex[cur] = rarc_candkey(sc, row);
      }
      __builtin_amdgcn_wave_barrier();
    }
  };
  auto rescore = [&](int from, int to) __attribute__((always_inline)) {
    switch ((p.fmt == 2 ? p.d * 4 : (p.fmt ? p.d : p.d * 2)) / 128) {  // 128-byte steps per stored row; block-uniform
#define FIN8_NT(N) case N: rescore_n(std::integral_constant<int, N>{}, from, to); break;
      FIN8_NT(2) FIN8_NT(4) FIN8_NT(6) FIN8_NT(8) FIN8_NT(10) FIN8_NT(12) FIN8_NT(14) FIN8_NT(16)
      FIN8_NT(20) FIN8_NT(24) FIN8_NT(28) FIN8_NT(32)   // fp32 rows of 640 ... 1024 elements
#undef FIN8_NT
      default: break;  // the launcher admits only d multiples of 128 (fp16) / 256 (fp8), <= 1024
    }
  };

  // The k-th largest of the n distinct keys ex[0..n) (1 <= kth <= n).  Up to FIN8_SMALL keys: rank by counting (n²/threads
  // compares: nothing at the few hundred rows an isotropic corpus leaves).  Beyond: radix select, one byte per pass from the
  // top, 256 LDS counters (s_hist is free once T1 is known) — O(n) per pass.  On clustered corpora the whole cluster of a
  // query sits inside the int8 margin, the buffer fills (6144 keys) and ranking by counting was 38M compares per query and
  // ranking: 5.5 of the 8.8 ms a batch took on 10M clustered rows (profiles/r04_clustered_*).
  constexpr int FIN8_SMALL = 1024;
  auto select_kth = [&](int n, int kth) __attribute__((always_inline)) -> uint64_t {
    if (n <= FIN8_SMALL) {
      __syncthreads();
      for (int i = tid; i < n; i += blockDim.x) {
        const uint64_t mine = ex[i];
        int rank = 0;
        for (int j = 0; j < n; ++j) rank += (ex[j] > mine);
        if (rank == kth - 1) { s_selb = (int)(mine >> 32); s_selabove = (uint32_t)mine; }
      }
      __syncthreads();
      const uint64_t r = ((uint64_t)(uint32_t)s_selb << 32) | s_selabove;
      __syncthreads();
      return r;
    }
    uint64_t prefix = 0, mask = 0;
    uint32_t need = (uint32_t)kth;
    for (int shift = 56; shift >= 0; shift -= 8) {
      __syncthreads();
      for (int i = tid; i < RARC_NB; i += blockDim.x) s_hist[i] = 0;
      __syncthreads();
      for (int i = tid; i < n; i += blockDim.x) {
        const uint64_t key = ex[i];
        if ((key & mask) == prefix) atomicAdd(&s_hist[(uint32_t)(key >> shift) & 255u], 1u);
      }
      __syncthreads();
      if (tid < 64) {
        uint32_t above = 0;
        const int b = rarc_wave_find_from_top(s_hist, RARC_NB, need, &above);
        if (tid == 0) { s_selb = b; s_selabove = above; }
      }
      __syncthreads();
      prefix |= (uint64_t)(uint32_t)s_selb << shift;
      mask |= 0xFFull << shift;
      need -= s_selabove;
    }
    __syncthreads();
    return prefix;
  };
  static_assert(RARC_NB == 256, "the radix select counts one byte per pass in s_hist");

  // ---- step 1: G1 = {a >= T1}: canonical scores, L = k-th best of them ----
  FIN8_STAMP(1)
  collect([&](float a, uint32_t) { return a >= t1; });
  __syncthreads();
  if (s_ne > (uint32_t)FIN8_RS && t1 > -INFINITY) {
    // The histogram's bins were too coarse for this query (a window sized from a sample whose k-th best lies far
    // below the final one): G1 does not fit.  Any cut with k <= |{a >= cut}| <= buffer will do: bisect for one
    // between the histogram's edge and the top of its window, one sweep over the candidate keys per probe.
    // (the upper end is the best approximate score actually present, found by one more sweep — NOT the top of the
    //  histogram's window: on clustered corpora a query's whole cluster, thousands of rows, can lie above a window that
    //  was sized from a sample the cluster was barely in, and then no cut inside the window holds few enough rows:
    //  the one query the round-3 tree flagged on every clustered batch, at the price of a second scan of the shard)
    __syncthreads();
    if (tid == 0) s_cmp = 0;
    __syncthreads();
    collect([&](float a, uint32_t) { atomicMax(&s_cmp, rarc_ordkey(a)); return false; });
    __syncthreads();
    float a_lo = t1, a_hi = rarc_unordkey(s_cmp + 1u);   // (the next float up: a cut must be able to sit above the best score)
    __syncthreads();
    for (int probe = 0; probe < 40 && a_hi > a_lo; ++probe) {
      const float mid = a_lo + 0.5f * (a_hi - a_lo);
      if (!(mid > a_lo && mid < a_hi)) break;
      __syncthreads();
      if (tid == 0) s_ne = 0;
      __syncthreads();
      collect([&](float a, uint32_t) { return a >= mid; });
      __syncthreads();
      const uint32_t c = s_ne;
      if (c > (uint32_t)FIN8_RS) a_lo = mid;
      else if (c < (uint32_t)p.k) a_hi = mid;
      else { t1 = mid; break; }
    }
    if (s_ne > (uint32_t)FIN8_RS || s_ne < (uint32_t)p.k) {  // no cut found (ties): leave the overflow for the flag
      __syncthreads();
      if (tid == 0) s_ne = 0;
      __syncthreads();
      collect([&](float a, uint32_t) { return a >= t1; });
      __syncthreads();
    }
  }
  FIN8_STAMP(2)
  const uint32_t ne1_all = s_ne;
  const int ne1 = ne1_all < FIN8_RS ? (int)ne1_all : FIN8_RS;
  rescore(0, ne1);
  __syncthreads();
  FIN8_STAMP(3)
  if (ne1 >= p.k) {  // keys are distinct
    const uint64_t kth = select_kth(ne1, p.k);
    if (tid == 0) s_L = rarc_candscore(kth);
  }
  __syncthreads();
  const float L = s_L;  // -inf when fewer than k candidates exist (then G1 already holds them all)
  FIN8_STAMP(4)
  if (p.tighten_thr) {
    // Mid-scan pass: L is the k-th best CANONICAL score among rows already scanned, so it bounds the final k-th
    // best from below, and a row that good scores at least L - eps8 in int8: the rest of the shard runs under that.
    if (tid == 0 && L > -INFINITY) {
      const float t = L - (1.0002f * eps + 1e-7f);
      if (p.tighten_mode == 2 || t > __uint_as_float(p.tighten_thr[q])) p.tighten_thr[q] = __float_as_uint(t);
    } else if (tid == 0 && p.tighten_mode == 2) {
      // fewer than k candidates so far: the stage's own threshold thr16 <= L_final - eps16 becomes
      // thr16 - (eps8 - eps16) <= L_final - eps8
      const float e16 = p.eps16 ? p.eps16[q] : 0.f;
      const float cur = __uint_as_float(p.tighten_thr[q]);
      if (cur > -INFINITY && cur < INFINITY) p.tighten_thr[q] = __float_as_uint(cur - (1.0002f * fmaxf(eps - e16, 0.f) + 1e-7f));
    }
    return;
  }

  // ---- step 2: G2 = {a < T1, a + eps >= L}: canonical scores ----
  // Could the row still reach `ref` canonically?  canonical <= a + eps8 for every row, and inside its own tile
  // <= a + eps8 - hq·(R - R_t) (the bound the scan discarded with): the coarse test first, the tile's metadata word
  // (a 4-byte gather) only for the few thousand rows that pass it — each row it then drops is a whole row not rescored
  const float hqv = p.hq ? p.hq[q] : 0.f;
  const float e16 = p.eps16 ? p.eps16[q] : 0.f;
  const float r_max = p.tmeta ? p.tmeta[0 - RARC_QMETA_HDR] : 0.f;
  auto reaches = [&](float a, uint32_t row, float ref) {
    if (!(a + eps * 1.0001f >= ref)) return false;
    if (!p.tmeta) return true;
    const uint32_t w = __float_as_uint(p.tmeta[(size_t)(row >> 5) * p.mstride]);
    const float rt = (float)__builtin_bit_cast(half_t, (uint16_t)(w >> 16));
    // (hybrid search: a candidate of the fp16 stage is only known to eps16, which the tile bound must not undercut)
    return a + fmaxf(eps - hqv * fmaxf(r_max - rt, 0.f), e16) * 1.0001f >= ref;
  };
  if (ne1_all <= (uint32_t)FIN8_RS && t1 > -INFINITY) collect([&](float a, uint32_t row) { return a < t1 && reaches(a, row, L); });
  __syncthreads();
  FIN8_STAMP(5)
  uint32_t ne_all = s_ne;
  int ne = ne_all < FIN8_RS ? (int)ne_all : FIN8_RS;
  bool band_fail = false;
  uint32_t why = 0;   // which limit flagged the query (RARC_Q_WHY_*, the status word's second byte): diagnostics only
  if (ne1_all > (uint32_t)FIN8_RS) why |= RARC_Q_WHY_G1;
  if (ne_all > (uint32_t)FIN8_RS && ne1_all <= (uint32_t)FIN8_RS && ne1 >= p.k && L > -INFINITY) {
    // G2 does not fit (large k on a large shard: k = 996 at 100M rows wants ~7000 rows).  Take it in bands of
    // approximate score, best band first: keep the k best canonical keys so far in ex[0..k), collect the band's
    // candidates behind them, rescore, keep the k best again.  The k-th best canonical score Lc only rises, so
    // the floor Lc - eps below which nothing can matter rises with it; a band that does not fit is halved.
    uint64_t* tmp = (uint64_t*)fsm;  // (the row staging area is idle while keys are ranked)
    auto keep_top_k = [&](int n) {   // ex[0..n) -> its k best keys (any order) in ex[0..k); s_L = the k-th
      const uint64_t kth = select_kth(n, p.k);
      if (tid == 0) { s_cmp = 0; s_L = rarc_candscore(kth); }
      __syncthreads();
      for (int i = tid; i < n; i += blockDim.x) {
        const uint64_t mine = ex[i];
        if (mine >= kth) tmp[atomicAdd(&s_cmp, 1u)] = mine;     // exactly k of them: the keys are distinct
      }
      __syncthreads();
      for (int i = tid; i < p.k; i += blockDim.x) ex[i] = tmp[i];
      __syncthreads();
    };
    keep_top_k(ne1);
    // Only "the band's top reached the floor" ends the loop as a success: running out of iterations flags the
    // query for repair.  A consumed band hands its width to the next one (dense G2: about one iteration per band
    // instead of a fresh run of halvings from the whole remaining range).
    float hi = t1, lo_try = -INFINITY;
    bool reached_floor = false;
    for (int guard = 0; guard < 512; ++guard) {
      const float floor_a = s_L - eps * 1.0001f;
      if (!(hi > floor_a)) { reached_floor = true; break; }
      const float lo = lo_try > floor_a ? lo_try : floor_a;
      __syncthreads();
      if (tid == 0) s_ne = (uint32_t)p.k;
      __syncthreads();
      const float lc = s_L;
      collect([&](float a, uint32_t row) { return a >= lo && a < hi && reaches(a, row, lc); });
      __syncthreads();
      const uint32_t cnt = s_ne;
      if (cnt > (uint32_t)FIN8_RS) {  // halve the band from below
        const float mid = lo + 0.5f * (hi - lo);
        if (!(mid > lo && mid < hi)) { band_fail = true; why |= RARC_Q_WHY_BAND_TIES; break; }
        lo_try = mid;
        continue;
      }
      rescore(p.k, (int)cnt);
      __syncthreads();
      keep_top_k((int)cnt);
      const float width = hi - lo;
      hi = lo;
      lo_try = (cnt > (uint32_t)(FIN8_RS / 2)) ? hi - width : hi - 2.0f * width;  // sparse band: try twice the width next
    }
    if (!reached_floor && !band_fail) why |= RARC_Q_WHY_BAND_GUARD;
    if (!reached_floor) band_fail = true;
    ne = p.k;
    ne_all = band_fail ? (uint32_t)FIN8_RS + 1u : (uint32_t)p.k;
  } else {
    rescore(ne1, ne);
    __syncthreads();
  }
  FIN8_STAMP(6)

  // ---- step 3: exact order (canonical score desc, id asc) by counting ----
  const int kk = ne < p.k ? ne : p.k;
  for (int i = tid; i < p.k; i += blockDim.x) {
    if (i >= kk) {
      p.out_ids[(size_t)q * p.k + i] = -1;
      p.out_scores[(size_t)q * p.k + i] = -INFINITY;
    }
  }
  if (ne <= FIN8_SMALL) {
    for (int i = tid; i < ne; i += blockDim.x) {
      const uint64_t mine = ex[i];
      if (rarc_candscore(mine) < L) continue;  // cannot rank inside the top k (k rows of G1 are >= L)
      int rank = 0;
      for (int j = 0; j < ne; ++j) rank += (ex[j] > mine);
      if (rank < p.k) {
        p.out_ids[(size_t)q * p.k + rank] = p.id_base + (int64_t)rarc_candrow(mine);
        p.out_scores[(size_t)q * p.k + rank] = rarc_candscore(mine);
      }
    }
  } else {   // thousands of rescored rows: cut at the k-th key first, rank only what is above it
    uint64_t* top = (uint64_t*)fsm;  // (the row staging area is idle now)
    const uint64_t kth = select_kth(ne, kk);
    if (tid == 0) s_cmp = 0;
    __syncthreads();
    for (int i = tid; i < ne; i += blockDim.x) {
      const uint64_t mine = ex[i];
      if (mine >= kth) top[atomicAdd(&s_cmp, 1u)] = mine;        // exactly kk keys
    }
    __syncthreads();
    for (int i = tid; i < kk; i += blockDim.x) {
      const uint64_t mine = top[i];
      int rank = 0;
      for (int j = 0; j < kk; ++j) rank += (top[j] > mine);
      p.out_ids[(size_t)q * p.k + rank] = p.id_base + (int64_t)rarc_candrow(mine);
      p.out_scores[(size_t)q * p.k + rank] = rarc_candscore(mine);
    }
  }
  FIN8_STAMP(7)
  if (p.dbg && q == 0 && tid == 0) { p.dbg[8] = ne1_all; p.dbg[9] = ne_all; }
  if (p.dbg && tid == 0) { p.dbg[16 + 4 * q] = ne1_all; p.dbg[17 + 4 * q] = ne_all; p.dbg[18 + 4 * q] = __builtin_amdgcn_s_memrealtime(); p.dbg[19 + 4 * q] = ((unsigned long long)__float_as_uint(t1) << 32) | __float_as_uint(L); }
  if (tid == 0) {
    uint32_t st = RARC_Q_OK;
    if (s_over || ne_all > (uint32_t)FIN8_RS) st |= RARC_Q_OVERFLOW;
    if (st) st |= why | (s_over ? RARC_Q_WHY_SEGMENT : 0u) | ((ne_all > (uint32_t)FIN8_RS && !why) ? RARC_Q_WHY_G2 : 0u);
    p.status[q] = st;
    if (st) {
      atomicOr(&p.flags[1], st);
      atomicOr(&p.status[RARC_MAX_QUERIES], st);
      if (p.flag_host) __hip_atomic_store(p.flag_host, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

unsigned long long* g_fin8_dbg = nullptr;  // set by tools only
int rarc_finalize_q8_launch(const void* corpus, const float* rowscale, int fmt, int d_pad, const float* q32,
                            const float* eps8, int nq, int k, int64_t id_base, const RarcWs& ws, int cap, int n_wg,
                            int64_t* out_ids, float* out_scores, uint32_t* status, hipStream_t s, int tighten,
                            const float* qmeta, const float* hq, const float* eps16) {
  RARC_REQUIRE(d_pad <= FIN8_MAXD, RARC_E_UNSUPPORTED, "rarc_finalize_q8: d_pad %d > %d", d_pad, FIN8_MAXD);
  Fin8Params p;
  p.corpus = corpus;
  p.rowscale = rowscale;
  p.fmt = fmt;
  p.q32 = q32;
  p.eps8 = eps8;
  p.cnt2 = ws.cnt2;
  p.cand = ws.cand;
  p.hist = ws.hist;
  p.binlo = ws.binlo;
  p.bininv = ws.bininv;
  p.flags = ws.flags;
  p.seg = (uint32_t)(cap / RARC_MAX_WG);
  p.n_wg = (uint32_t)n_wg;
  p.d = d_pad;
  p.k = k;
  p.id_base = id_base;
  p.out_ids = out_ids;
  p.out_scores = out_scores;
  p.status = status;
  p.tmeta = qmeta ? qmeta + RARC_QMETA_HDR : nullptr;
  p.mstride = fmt == 1 ? RARC_QMETA_F8_STRIDE : RARC_QMETA_STRIDE;
  p.hq = hq;
  p.dbg = tighten ? nullptr : g_fin8_dbg;
  p.tighten_thr = tighten ? (uint32_t*)ws.thr : nullptr;
  p.tighten_mode = tighten;
  p.eps16 = eps16;
  p.flag_host = tighten ? nullptr : rarc_launch_extras().flag_host;
  // 8 waves stage 8 rows each up to 1536 bytes per row (97 KB); up to 3072 bytes: 4 waves; fp32 rows of 1024: 2 waves
  const size_t rbytes = (size_t)d_pad * (fmt == 2 ? 4 : (fmt ? 1 : 2));
  const int threads = rbytes <= 1536 ? FIN8_THREADS : (rbytes <= 3072 ? FIN8_THREADS / 2 : FIN8_THREADS / 4);
  const size_t lds = (size_t)(threads / 64) * 8 * (rbytes + 16);
  static RarcPerDevice lds_attr_dev;
  size_t& lds_attr = lds_attr_dev.cur();
  if (lds > lds_attr) {
    RARC_HIP_CHECK(hipFuncSetAttribute((const void*)rarc_finalize_q8_kernel,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    lds_attr = lds;
  }
  hipLaunchKernelGGL(rarc_finalize_q8_kernel, dim3(nq), dim3(threads), lds, s, p);
  RARC_HIP_CHECK(hipGetLastError());
  return RARC_OK;
}

// ============================================================================================
// Exact repair of one query: full canonical scan of the shard; rows that beat the current k-th
// entry are appended; then the row is re-sorted.  O(n_rows * d) reads — the rare path.
// ============================================================================================
// canonical fp32 dot of a fp32 query with an fp8 (e4m3fn) row, times the row's scale (d multiple of 16)
__device__ __forceinline__ float canon_dot_f8(const float* __restrict__ q, const uint8_t* __restrict__ row, int d,
                                              float scale) {
  float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
  for (int m = 0; m < d; m += 16) {
    const uint4 v = *(const uint4*)(row + m);
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float f[4];
      rarc_f8x4_to_f32(w[i], f);
      const int e0 = 4 * i;  // elements m + e0 .. m + e0 + 3: chains (e0 & 7) ..
#pragma unroll
      for (int e = 0; e < 4; ++e) a[(e0 + e) & 7] = __builtin_fmaf(q[m + e0 + e], f[e], a[(e0 + e) & 7]);
    }
  }
  return scale * rarc_canon_tree(a);
}

struct RepairParams {
  const half_t* corpus;
  const uint8_t* corpus8;   // fp8 rows (then corpus is null)
  const float* corpus32;    // fp32 rows (then corpus and corpus8 are null)
  const float* rowscale;
  const float* qv;  // [d]
  uint32_t n_rows;
  int d, k;
  int64_t id_base;
  int64_t* ids;   // [k] in/out
  float* scores;  // [k] in/out
  uint64_t* list; // scratch [cap]
  uint32_t* count;
  uint32_t cap;
};

__global__ __launch_bounds__(256) void rarc_repair_scan_kernel(const RepairParams p) {
  // current k-th entry (worst kept).  -1 id == "fewer than k rows so far": everything beats it.
  const int64_t kid = p.ids[p.k - 1];
  const float ks = p.scores[p.k - 1];
  const uint64_t kth = (kid < 0) ? 0ull : rarc_candkey(ks, (uint32_t)(kid - p.id_base));
  for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < p.n_rows; r += gridDim.x * blockDim.x) {
    const float c = p.corpus32 ? canon_dot_f32(p.qv, p.corpus32 + (size_t)r * p.d, p.d)
                  : p.corpus8 ? canon_dot_f8(p.qv, p.corpus8 + (size_t)r * p.d, p.d, p.rowscale[r])
                              : canon_dot_f16(p.qv, p.corpus + (size_t)r * p.d, p.d);
    const uint64_t key = rarc_candkey(c, r);
    if (key > kth) {
      const uint32_t pos = atomicAdd(p.count, 1u);
      if (pos < p.cap) p.list[pos] = key;
    }
  }
}

// merge: list holds every row strictly better than the old k-th (it includes the old top k-1
// themselves, because they also beat the old k-th), so top-k(list ∪ {old k-th}) is exact.
__global__ __launch_bounds__(256) void rarc_repair_merge_kernel(const RepairParams p, uint32_t* found) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint64_t* keys = (uint64_t*)smem;
  const uint32_t total = *p.count;
  const uint32_t n = total < p.cap ? total : p.cap;
  int np2 = 2;
  while (np2 < (int)n + 1) np2 <<= 1;
  const int64_t kid = p.ids[p.k - 1];
  const uint64_t kth = (kid < 0) ? 0ull : rarc_candkey(p.scores[p.k - 1], (uint32_t)(kid - p.id_base));
  for (int i = threadIdx.x; i < np2; i += blockDim.x)
    keys[i] = (i < (int)n) ? p.list[i] : (i == (int)n ? kth : 0ull);
  __syncthreads();
  bitonic_desc(keys, np2);
  // rows found beyond the k-1 that were already listed
  if (threadIdx.x == 0) {
    uint32_t old_better = 0;  // old entries 0..k-2 that are real
    for (int i = 0; i < p.k - 1; ++i) old_better += (p.ids[i] >= 0);
    *found = (total > old_better) ? (total - old_better) : 0u;
    if (total > p.cap) *found |= 0x80000000u;  // scratch overflow: result not trustworthy
  }
  __syncthreads();
  for (int i = threadIdx.x; i < p.k; i += blockDim.x) {
    const uint64_t key = keys[i];
    if (key != 0ull) {
      p.ids[i] = p.id_base + (int64_t)rarc_candrow(key);
      p.scores[i] = rarc_candscore(key);
    } else {
      p.ids[i] = -1;
      p.scores[i] = -INFINITY;
    }
  }
}

// ============================================================================================
// Batched exact verification: for up to 8 queries at once, count the rows of the shard whose canonical key beats
// the k-th entry of the query's answer.  Same canonical arithmetic as the repair scan, but a row is read ONCE for the
// eight queries (the single-query scan reads the whole shard per query: 62 ms per query at 100M rows; this: the same
// time per EIGHT queries), so a full 256-query batch can be checked against an exact scan in a couple of seconds.
// One thread per row; the queries sit in LDS and are read as wave-wide broadcasts.
// ============================================================================================
constexpr int VERIFY_NQ = 8;
struct VerifyParams {
  const void* corpus;
  const float* rowscale;
  int fmt;                 // 0 fp16, 1 fp8, 2 fp32 rows
  const float* q32;        // the query block's fp32 rows [256][d]
  int q_first, nq;         // queries q_first .. q_first + nq - 1  (nq <= VERIFY_NQ)
  uint32_t n_rows;
  int d, k;
  int64_t id_base;
  const int64_t* ids;      // [*][k] answers (row q of the batch at ids + q*k)
  const float* scores;
  uint32_t* counts;        // [3 * VERIFY_NQ] out: [i] rows with key > k-th key; [VERIFY_NQ + i] rows with key >= k-th key whose
                           // (id, canonical score) pair is an entry of the answer, bit for bit; [2 * VERIFY_NQ + i] rows at or
                           // above the k-th key that were NOT looked up because a thread had used its lookups up
};

__global__ __launch_bounds__(256) void rarc_verify_kernel(const VerifyParams p) {
  __shared__ __attribute__((aligned(16))) float s_q[VERIFY_NQ][1024];
  __shared__ uint64_t s_kth[VERIFY_NQ];
  __shared__ uint32_t s_cnt[3 * VERIFY_NQ];
  const int tid = threadIdx.x;
  for (int i = tid; i < VERIFY_NQ * p.d; i += blockDim.x) {
    const int qi = i / p.d, m = i % p.d;
    s_q[qi][m] = qi < p.nq ? p.q32[(size_t)(p.q_first + qi) * p.d + m] : 0.f;
  }
  if (tid < VERIFY_NQ) {
    uint64_t kth = ~0ull;  // padding queries: nothing beats them
    if (tid < p.nq) {
      const int64_t kid = p.ids[(size_t)(p.q_first + tid) * p.k + p.k - 1];
      kth = kid < 0 ? 0ull : rarc_candkey(p.scores[(size_t)(p.q_first + tid) * p.k + p.k - 1], (uint32_t)(kid - p.id_base));
    }
    s_kth[tid] = kth;
    s_cnt[tid] = 0;
    s_cnt[VERIFY_NQ + tid] = 0;
    s_cnt[2 * VERIFY_NQ + tid] = 0;
  }
  __syncthreads();
  uint32_t mine[VERIFY_NQ];
#pragma unroll
  for (int qi = 0; qi < VERIFY_NQ; ++qi) mine[qi] = 0;
  // a thread looks at most 64 rows up PER QUERY: an exact answer needs ~k per query over the WHOLE grid; a corrupted one (k-th
  // entry far too low: every row passes) must not turn the scan into n_rows x k loads.  A row that is skipped for that reason
  // is counted in counts[2 * VERIFY_NQ + qi], so the host can tell "check truncated" from "answer wrong" (a
  // thread owns the rows congruent to it modulo 524,288 — a tiled synthetic corpus can put a query's whole answer there)
  uint8_t lookups[VERIFY_NQ];
#pragma unroll
  for (int qi = 0; qi < VERIFY_NQ; ++qi) lookups[qi] = 0;
  for (uint32_t r = blockIdx.x * blockDim.x + tid; r < p.n_rows; r += gridDim.x * blockDim.x) {
    float a[VERIFY_NQ][8];
#pragma unroll
    for (int qi = 0; qi < VERIFY_NQ; ++qi)
#pragma unroll
      for (int j = 0; j < 8; ++j) a[qi][j] = 0.f;
    float scale = 1.f;
    if (p.fmt == 0) {
      const half_t* row = (const half_t*)p.corpus + (size_t)r * p.d;
      for (int m = 0; m < p.d; m += 8) {
        const half8 x = *(const half8*)(row + m);
#pragma unroll
        for (int qi = 0; qi < VERIFY_NQ; ++qi) {
          const float4 q0 = *(const float4*)(&s_q[qi][m]), q1 = *(const float4*)(&s_q[qi][m + 4]);
          a[qi][0] = __builtin_fmaf(q0.x, (float)x[0], a[qi][0]);
          a[qi][1] = __builtin_fmaf(q0.y, (float)x[1], a[qi][1]);
          a[qi][2] = __builtin_fmaf(q0.z, (float)x[2], a[qi][2]);
          a[qi][3] = __builtin_fmaf(q0.w, (float)x[3], a[qi][3]);
          a[qi][4] = __builtin_fmaf(q1.x, (float)x[4], a[qi][4]);
          a[qi][5] = __builtin_fmaf(q1.y, (float)x[5], a[qi][5]);
          a[qi][6] = __builtin_fmaf(q1.z, (float)x[6], a[qi][6]);
          a[qi][7] = __builtin_fmaf(q1.w, (float)x[7], a[qi][7]);
        }
      }
    } else if (p.fmt == 2) {
      const float* row = (const float*)p.corpus + (size_t)r * p.d;
      for (int m = 0; m < p.d; m += 8) {
        const float4 x0 = *(const float4*)(row + m), x1 = *(const float4*)(row + m + 4);
#pragma unroll
        for (int qi = 0; qi < VERIFY_NQ; ++qi) {
          const float4 q0 = *(const float4*)(&s_q[qi][m]), q1 = *(const float4*)(&s_q[qi][m + 4]);
          a[qi][0] = __builtin_fmaf(q0.x, x0.x, a[qi][0]);
          a[qi][1] = __builtin_fmaf(q0.y, x0.y, a[qi][1]);
          a[qi][2] = __builtin_fmaf(q0.z, x0.z, a[qi][2]);
          a[qi][3] = __builtin_fmaf(q0.w, x0.w, a[qi][3]);
          a[qi][4] = __builtin_fmaf(q1.x, x1.x, a[qi][4]);
          a[qi][5] = __builtin_fmaf(q1.y, x1.y, a[qi][5]);
          a[qi][6] = __builtin_fmaf(q1.z, x1.z, a[qi][6]);
          a[qi][7] = __builtin_fmaf(q1.w, x1.w, a[qi][7]);
        }
      }
    } else {
      const uint8_t* row = (const uint8_t*)p.corpus + (size_t)r * p.d;
      scale = p.rowscale[r];
      for (int m = 0; m < p.d; m += 8) {
        const uint2 v = *(const uint2*)(row + m);
        float xa[4], xb[4];
        rarc_f8x4_to_f32(v.x, xa);
        rarc_f8x4_to_f32(v.y, xb);
        const float x[8] = {xa[0], xa[1], xa[2], xa[3], xb[0], xb[1], xb[2], xb[3]};
#pragma unroll
        for (int qi = 0; qi < VERIFY_NQ; ++qi) {
          const float4 q0 = *(const float4*)(&s_q[qi][m]), q1 = *(const float4*)(&s_q[qi][m + 4]);
          a[qi][0] = __builtin_fmaf(q0.x, x[0], a[qi][0]);
          a[qi][1] = __builtin_fmaf(q0.y, x[1], a[qi][1]);
          a[qi][2] = __builtin_fmaf(q0.z, x[2], a[qi][2]);
          a[qi][3] = __builtin_fmaf(q0.w, x[3], a[qi][3]);
          a[qi][4] = __builtin_fmaf(q1.x, x[4], a[qi][4]);
          a[qi][5] = __builtin_fmaf(q1.y, x[5], a[qi][5]);
          a[qi][6] = __builtin_fmaf(q1.z, x[6], a[qi][6]);
          a[qi][7] = __builtin_fmaf(q1.w, x[7], a[qi][7]);
        }
      }
    }
#pragma unroll
    for (int qi = 0; qi < VERIFY_NQ; ++qi) {
      float c = rarc_canon_tree(a[qi]);
      if (p.fmt == 1) c = scale * c;
      const uint64_t key = rarc_candkey(c, r);
      mine[qi] += key > s_kth[qi] ? 1u : 0u;
      // a row at or above the k-th entry belongs in the answer: look its (id, score) pair up there (about k rows per
      // query in a whole scan take this branch).  The host wants as many exact pairs as the answer has valid entries —
      // an entry whose score was not the row's canonical score, or whose row does not reach the k-th key, is missed.
      if (qi < p.nq && key >= s_kth[qi] && s_kth[qi] != 0ull && lookups[qi] >= 64) atomicAdd(&s_cnt[2 * VERIFY_NQ + qi], 1u);
      if (qi < p.nq && key >= s_kth[qi] && s_kth[qi] != 0ull && lookups[qi] < 64) {
        ++lookups[qi];
        const int64_t* ai = p.ids + (size_t)(p.q_first + qi) * p.k;
        const float* as = p.scores + (size_t)(p.q_first + qi) * p.k;
        const int64_t want = (int64_t)r + p.id_base;
        for (int j = 0; j < p.k; ++j)
          if (ai[j] == want) {
            if (__builtin_bit_cast(uint32_t, as[j]) == __builtin_bit_cast(uint32_t, c)) atomicAdd(&s_cnt[VERIFY_NQ + qi], 1u);
            break;
          }
      }
    }
  }
#pragma unroll
  for (int qi = 0; qi < VERIFY_NQ; ++qi)
    if (mine[qi]) atomicAdd(&s_cnt[qi], mine[qi]);
  __syncthreads();
  if (tid < p.nq && s_cnt[tid]) atomicAdd(&p.counts[tid], s_cnt[tid]);
  if (tid < p.nq && s_cnt[VERIFY_NQ + tid]) atomicAdd(&p.counts[VERIFY_NQ + tid], s_cnt[VERIFY_NQ + tid]);
  if (tid < p.nq && s_cnt[2 * VERIFY_NQ + tid]) atomicAdd(&p.counts[2 * VERIFY_NQ + tid], s_cnt[2 * VERIFY_NQ + tid]);
}

int rarc_verify_launch(const void* corpus, const float* rowscale, int fmt, int64_t n_rows, int d_pad, const float* q32,
                       int q_first, int nq, int k, int64_t id_base, const int64_t* ids, const float* scores,
                       uint32_t* counts, hipStream_t s) {
  VerifyParams p{corpus, rowscale, fmt, q32, q_first, nq, (uint32_t)n_rows, d_pad, k, id_base, ids, scores, counts};
  RARC_HIP_CHECK(hipMemsetAsync(counts, 0, sizeof(uint32_t) * 3 * VERIFY_NQ, s));
  hipLaunchKernelGGL(rarc_verify_kernel, dim3(2048), dim3(256), 0, s, p);
  RARC_HIP_CHECK(hipGetLastError());
  return RARC_OK;
}

int rarc_repair_launch(const void* corpus, const float* rowscale, int fmt, int64_t n_rows, int d_pad,
                       const float* qv, int k, int64_t id_base, int64_t* ids, float* scores, uint32_t* found,
                       const RarcWs& ws, int cap, hipStream_t s) {
  RepairParams p;
  p.corpus = fmt ? nullptr : (const half_t*)corpus;
  p.corpus8 = fmt == 1 ? (const uint8_t*)corpus : nullptr;
  p.corpus32 = fmt == 2 ? (const float*)corpus : nullptr;
  p.rowscale = rowscale;
  p.qv = qv;
  p.n_rows = (uint32_t)n_rows;
  p.d = d_pad;
  p.k = k;
  p.id_base = id_base;
  p.ids = ids;
  p.scores = scores;
  p.list = ws.cand;
  p.count = ws.cnt;  // cnt[0] reused as the append counter
  p.cap = (uint32_t)(cap < 8192 ? cap : 8192);
  RARC_HIP_CHECK(hipMemsetAsync(p.count, 0, 4, s));
  const int grid = 2048;
  hipLaunchKernelGGL(rarc_repair_scan_kernel, dim3(grid), dim3(256), 0, s, p);
  RARC_HIP_CHECK(hipGetLastError());
  int np2 = 2;
  while (np2 < (int)p.cap + 1) np2 <<= 1;
  const size_t lds = (size_t)np2 * 8;
  static RarcPerDevice lds_attr_dev;
  size_t& lds_attr = lds_attr_dev.cur();
  if (lds > lds_attr) {
    RARC_HIP_CHECK(hipFuncSetAttribute((const void*)rarc_repair_merge_kernel,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    lds_attr = lds;
  }
  hipLaunchKernelGGL(rarc_repair_merge_kernel, dim3(1), dim3(256), lds, s, p, found);
  RARC_HIP_CHECK(hipGetLastError());
  return RARC_OK;
}

// ============================================================================================
// Multi-shard merge: [G][nq][k] sorted lists -> [nq][k]  (score desc, id asc)
// ============================================================================================
// PACKED: the lists arrive as int32 [G][nq][k][3] = (id low, id high, score bits) — the form that crosses the
// all-gather as ONE tensor (rarc_pack_results writes it) — and `ids` points at them, `scores` is unused.
template <bool PACKED>
__global__ __launch_bounds__(256) void rarc_merge_kernel(const int64_t* ids, const float* scores, int G,
                                                         int nq, int k, int64_t* out_ids,
                                                         float* out_scores) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // 128-bit sort key split in two arrays: (ordkey(score), ~id) compared lexicographically
  uint32_t* sk = (uint32_t*)smem;
  const int q = blockIdx.x;
  const int n = G * k;
  int np2 = 2;
  while (np2 < n) np2 <<= 1;
  int64_t* sid = (int64_t*)(smem + (size_t)np2 * 4);
  for (int i = threadIdx.x; i < np2; i += blockDim.x) {
    uint32_t key = 0;
    int64_t id = INT64_MAX;
    if (i < n) {
      const int g = i / k, j = i % k;
      const size_t o = ((size_t)g * nq + q) * k + j;
      float sc;
      if (PACKED) {
        const uint32_t* pk = (const uint32_t*)ids + o * 3;
        id = (int64_t)((uint64_t)pk[0] | ((uint64_t)pk[1] << 32));
        sc = __uint_as_float(pk[2]);
      } else {
        id = ids[o];
        sc = scores[o];
      }
      if (id >= 0) key = rarc_ordkey(sc);
      else id = INT64_MAX;
    }
    sk[i] = key;
    sid[i] = id;
  }
  __syncthreads();
  for (int kk = 2; kk <= np2; kk <<= 1) {
    for (int j = kk >> 1; j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < np2; i += blockDim.x) {
        const int ixj = i ^ j;
        if (ixj > i) {
          const uint32_t ka = sk[i], kb = sk[ixj];
          const int64_t ia = sid[i], ib = sid[ixj];
          // "x better than y": higher score, then lower id
          const bool a_better = (ka > kb) || (ka == kb && ia < ib);
          const bool b_better = (kb > ka) || (ka == kb && ib < ia);
          const bool up = ((i & kk) == 0);
          if (up ? b_better : a_better) {
            sk[i] = kb; sk[ixj] = ka;
            sid[i] = ib; sid[ixj] = ia;
          }
        }
      }
      __syncthreads();
    }
  }
  for (int i = threadIdx.x; i < k; i += blockDim.x) {
    const bool valid = (i < n) && sid[i] != INT64_MAX;
    out_ids[(size_t)q * k + i] = valid ? sid[i] : -1;
    out_scores[(size_t)q * k + i] = valid ? rarc_unordkey(sk[i]) : -INFINITY;
  }
}

int rarc_merge_launch(const int64_t* ids, const float* scores, int G, int nq, int k, int64_t* out_ids,
                      float* out_scores, hipStream_t s, bool packed) {
  int np2 = 2;
  while (np2 < G * k) np2 <<= 1;
  const size_t lds = (size_t)np2 * 12;
  RARC_REQUIRE(lds <= 160 * 1024, RARC_E_UNSUPPORTED, "rarc_topk_merge: %d lists x k=%d too large", G, k);
  static RarcPerDevice lds_attr_dev;
  size_t& lds_attr = lds_attr_dev.cur();
  if (lds > lds_attr) {
    RARC_HIP_CHECK(hipFuncSetAttribute((const void*)rarc_merge_kernel<false>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    RARC_HIP_CHECK(hipFuncSetAttribute((const void*)rarc_merge_kernel<true>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    lds_attr = lds;
  }
  if (packed)
    hipLaunchKernelGGL(rarc_merge_kernel<true>, dim3(nq), dim3(256), lds, s, ids, scores, G, nq, k, out_ids, out_scores);
  else
    hipLaunchKernelGGL(rarc_merge_kernel<false>, dim3(nq), dim3(256), lds, s, ids, scores, G, nq, k, out_ids, out_scores);
  RARC_HIP_CHECK(hipGetLastError());
  return RARC_OK;
}

__global__ __launch_bounds__(256) void rarc_pack_kernel(const int64_t* ids, const float* scores, int n, uint32_t* out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const uint64_t id = (uint64_t)ids[i];
  out[3 * i] = (uint32_t)id;
  out[3 * i + 1] = (uint32_t)(id >> 32);
  out[3 * i + 2] = __float_as_uint(scores[i]);
}

int rarc_pack_launch(const int64_t* ids, const float* scores, int n, uint32_t* out, hipStream_t s) {
  hipLaunchKernelGGL(rarc_pack_kernel, dim3((n + 255) / 256), dim3(256), 0, s, ids, scores, n, out);
  RARC_HIP_CHECK(hipGetLastError());
  return RARC_OK;
}
