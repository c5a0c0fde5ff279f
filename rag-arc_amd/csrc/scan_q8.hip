// scan_q8.hip — flat corpus scan with an int8 prefilter:  HBM fp16 rows -> int8 in registers ->
// v_mfma_i32_32x32x32_i8 against register-resident int8 queries -> exact pruning.
//
// Replaces the inner loop of faiss.IndexFlatIP.search reached from
//   encapsulation/database/vector_db/VectorStore_Faiss.py:263
// for up to 256 queries at once.  The B×N score matrix is never materialised.
//
// Why int8: at batch 256 the fp16 MFMA formulation of Q·Dᵀ needs 256 flop per corpus byte, which is
// the chip's matrix/HBM balance point, and on real (toggling) data the matrix pipe clocks down under
// its power limit before HBM saturates (measured: 47-51 % of HBM peak).  The int8
// MFMA has twice the rate at a fraction of the energy, so with it the scan is bound by what it must
// be bound by — reading every fp16 row once.  Exactness is not given up: int8 scores only DISCARD
// rows, under a Cauchy-Schwarz bound on the quantisation error (quant.hip); every row that could
// possibly be in the top-k is rescored with the canonical fp32 inner product by the finalize kernel.
//
// Shape (one persistent workgroup per CU, 8 waves):
//   * Q8 (256 × D int8) lives in REGISTERS: wave w owns queries [32w, 32w+32) as the MFMA B operand
//     (D/32 fragments × 4 VGPRs = 96 VGPRs at D = 768).
//   * a tile = 32 corpus rows = 64·D contiguous bytes.  Each thread fetches D/128 16-byte chunks with
//     plain global_load_dwordx4 (a wave reads 1 KiB contiguous), two tiles ahead, converts them to
//     int8 with the tile's scale (4 v_pk_fma_f16 + 2 v_perm_b32 per chunk) and writes 8 bytes to the
//     LDS int8 tile (row stride D+16: the 16-lane groups of the A-fragment ds_read_b128 then hit 16
//     distinct bank groups).  Two LDS tiles: convert(t+1) overlaps MFMA(t); one barrier per tile.
//   * per tile and wave: D/32 v_mfma_i32_32x32x32_i8; lane l ends with 16 integer scores of query
//     l&31.  Prune: integer max, one compare against the lane's threshold; survivors are appended to
//     the workgroup's private segment of the query's candidate list and bump the query's histogram.
//   * thresholds: thr[q] is a lower bound of (k-th best approx score) − 2·eps8[q], or of
//     (a lower bound of the k-th best canonical score) − eps8[q]; both mean "a row below it cannot be
//     in the top-k".  The owner workgroup of a query turns its histogram into a higher threshold.
//
// Algorithmic HBM bytes per launch: n_rows × D × 2 (+ 8 bytes per tile of metadata).
#include "rarc_common.h"

struct ScanQ8Params {
  const uint4* corpus;   // fp16 rows [ceil32(n_rows)][D], as 16-byte chunks
  const float2* tmeta;   // per tile: (scale, 1/scale)
  const int8_t* q8;      // [256][D]
  const float* qinv;     // [256]  1 / s_q
  const float* eps8;     // [256]
  uint32_t n_rows;
  uint32_t n_tiles;
  uint32_t* thr;          // float bits [256]
  const float* binlo;     // [256]
  const float* binscale;  // [256]
  const float* bininv;    // [256]
  uint32_t* hist;         // [256][RARC_NB]
  uint32_t* cnt2;         // [256 wg][256 q]
  uint64_t* cand;         // [256 q][256 wg][seg]
  uint32_t seg;
  uint32_t kprime;
  uint32_t nq;
};

constexpr int Q8_WAVES = 8;
constexpr int Q8_THREADS = Q8_WAVES * 64;

constexpr int Q8_STAGE = 6144;  // staged (query, key) appends per workgroup between flushes

template <int D>
struct ScanQ8Lds {
  static constexpr int RS = D + 16;         // row stride of the int8 tile (bytes)
  static constexpr int TILE = 32 * RS;      // one int8 tile
  static constexpr int CNT = 2 * TILE;      // uint32 [256] slot counters of the private segments
  static constexpr int BINLO = CNT + 1024;
  static constexpr int BINSCALE = BINLO + 1024;
  static constexpr int BININV = BINSCALE + 1024;
  static constexpr int EPS8 = BININV + 1024;
  static constexpr int HLAND = EPS8 + 1024;     // uint32 [256]: the owned query's histogram, as last fetched
  static constexpr int MISC = HLAND + 1024;     // [0] staged count
  static constexpr int SKEY = MISC + 64;        // uint64 [Q8_STAGE]
  static constexpr int SQ = SKEY + 8 * Q8_STAGE;  // uint8 [Q8_STAGE]
  static constexpr int TOTAL = SQ + Q8_STAGE;
};

// barrier that orders LDS traffic only: global loads stay in flight across it (a __syncthreads()
// would drain vmcnt and with it the two-tile prefetch)
__device__ __forceinline__ void q8_lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// ABL (tools/scan_q8_bench): 1 = no pruning, 2 = no global loads after the prologue, 4 = no MFMA,
// 8 = no threshold refresh, 16 = no conversion, 32 = prune fast path only, 64 = never flush
//
// Vector-memory discipline.  The prefetched tile registers are consumed with counted waits
// ("all but the newest N operations have returned"), which the compiler derives per program path and
// merges to the minimum over paths.  So (1) every iteration issues the SAME sequence of loads on every
// path — tile chunks, tile scale, refreshed threshold, one histogram word — with out-of-range tiles
// redirected to tile 0 instead of skipped; and (2) nothing else touches vector memory in the steady
// state: survivors are staged in LDS and flushed (stores + histogram atomics) only now and then, the
// owner publishes a threshold only when it rose.  An extra store in the queue would not break
// anything, but the counted wait behind it would sit until that store is acknowledged — microseconds
// under a saturated HBM — which measured as +50 % kernel time when every survivor went out directly.
template <int D, int ABL = 0>
__global__ __launch_bounds__(Q8_THREADS) void rarc_scan_q8_kernel(const ScanQ8Params p) {
  static_assert(D % 128 == 0 && D >= 128 && D <= 1024, "D must be a multiple of 128, <= 1024");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using L = ScanQ8Lds<D>;
  constexpr int KS = D / 32;          // MFMA k-steps
  constexpr int CPR = D / 8;          // 16-byte fp16 chunks per row
  constexpr int CPT = D / 128;        // chunks per thread per tile (32*CPR / 512)
  constexpr int TCH = 32 * CPR;       // chunks per tile

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int row = lane & 31, h = lane >> 5;
  const uint32_t qidx = wave * 32 + row;  // this lane's query

  uint32_t* s_cnt = (uint32_t*)(smem + L::CNT);
  float* s_binlo = (float*)(smem + L::BINLO);
  float* s_binscale = (float*)(smem + L::BINSCALE);
  float* s_bininv = (float*)(smem + L::BININV);
  float* s_eps8 = (float*)(smem + L::EPS8);
  uint32_t* s_hland = (uint32_t*)(smem + L::HLAND);
  uint32_t* s_nstage = (uint32_t*)(smem + L::MISC);
  uint64_t* s_skey = (uint64_t*)(smem + L::SKEY);
  uint8_t* s_sq = (uint8_t*)(smem + L::SQ);
  if (tid < RARC_MAX_QUERIES) {
    s_cnt[tid] = 0;
    s_binlo[tid] = p.binlo[tid];
    s_binscale[tid] = p.binscale[tid];
    s_bininv[tid] = p.bininv[tid];
    s_eps8[tid] = p.eps8[tid];
    s_hland[tid] = 0;
  }
  if (tid == 0) *s_nstage = 0;

  // resident query fragments (B operand): lane holds Q8[qidx][32*ks + 16*h .. +16)
  i32x4 qf[KS];
  {
    const int8_t* qp = p.q8 + (size_t)qidx * D + 16 * h;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[ks] = *(const i32x4*)(qp + 32 * ks);
  }
  const float my_qinv = p.qinv[qidx];
  float thr = __uint_as_float(p.thr[qidx]);  // seed threshold; +inf for padding queries
  // everything fetched so far has landed before the first tile load is issued: the compiler's wait
  // counters then never tie a query fragment to the (deliberately long-lived) tile prefetches
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)

  // this thread's chunks of a tile: c = j*512 + tid -> LDS byte offset of its 8 int8 values
  uint32_t woff[CPT];
#pragma unroll
  for (int j = 0; j < CPT; ++j) {
    const uint32_t c = j * Q8_THREADS + tid;
    woff[j] = (c / CPR) * L::RS + (c % CPR) * 8;
  }
  const uint32_t aoff = row * L::RS + 16 * h;  // A fragment of k-step ks: + 32*ks

  const uint32_t t0 = blockIdx.x, stride = gridDim.x;
  // the query this workgroup owns (publishes thresholds for), and the histogram word this lane
  // fetches each iteration: wave w covers bins [32w, 32w+32) (lanes 32-63 duplicate lanes 0-31)
  const bool has_own = blockIdx.x < p.nq;
  const uint32_t own_q = has_own ? blockIdx.x : 0u;  // (queries beyond gridDim.x keep their seed threshold)
  const uint32_t* hword = p.hist + (size_t)own_q * RARC_NB + 32 * wave + row;

  // One "fetch group" per tile, identical on every path: the tile's chunks, its (scale, 1/scale)
  // pair, this lane's refreshed threshold, one word of the owned query's histogram.  Everything in a
  // group lands together (vector-memory returns are in order), two iterations after it was issued.
  struct Fetch {
    uint4 c[CPT];
    float2 meta;
    uint32_t thr;
    uint32_t hw;
  };
  auto clamp_tile = [&](uint32_t t) { return t < p.n_tiles ? t : 0u; };  // past the end: tile 0 (an L2 hit)
  auto fetch = [&](Fetch& f, uint32_t tile) {
    const uint4* src = p.corpus + (size_t)tile * TCH + tid;
#pragma unroll
    for (int j = 0; j < CPT; ++j) f.c[j] = src[j * Q8_THREADS];
    f.meta = p.tmeta[tile];
    f.thr = __hip_atomic_load(&p.thr[qidx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    f.hw = __hip_atomic_load(hword, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  auto convert_tile = [&](const Fetch& f, int buf) {
    const half_t s = (half_t)f.meta.x;
    char* dst = smem + buf * L::TILE;
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
      uint2 o;
      if (ABL & 16) { o.x = f.c[j].x ^ f.c[j].z; o.y = f.c[j].y ^ f.c[j].w; }
      else o = rarc_quant8_chunk(f.c[j], s);
      *(uint2*)(dst + woff[j]) = o;
    }
  };

  // one survivor straight to global memory (flush, and the overflow path of the staging buffer)
  auto emit = [&](uint32_t q, uint64_t key) {
    const uint32_t slot = atomicAdd(&s_cnt[q], 1u);
    if (slot < p.seg) p.cand[((size_t)q * RARC_MAX_WG + blockIdx.x) * p.seg + slot] = key;
    atomicAdd(&p.hist[q * RARC_NB + rarc_bin_of(rarc_candscore(key), s_binlo[q], s_binscale[q])], 1u);
  };
  // lane holds 16 integer scores of its query: rows 8*(r>>2) + 4*h + (r&3) of the tile.
  // Fast path: integer max, one compare.  Slow path (some lane has a survivor): the lane counts its
  // survivors, reserves that many staging entries with one LDS atomic and writes (query, key) pairs
  // to LDS — no vector-memory traffic.
  auto prune = [&](const i32x16& acc, uint32_t tile, float tinv) {
    int m = acc[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) m = acc[r] > m ? acc[r] : m;
    const float sc = my_qinv * tinv;
    if (ABL & 32) {
      if (__builtin_amdgcn_ballot_w64((float)m * sc >= 1e30f) != 0) p.cnt2[1] = 1;
      return;
    }
    if (__builtin_amdgcn_ballot_w64((float)m * sc >= thr) != 0) {
      const uint32_t row0 = tile * 32 + 4 * h;
      uint32_t mask = 0;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const uint32_t doc = row0 + (r & 3) + 8 * (r >> 2);
        mask |= ((float)acc[r] * sc >= thr && doc < p.n_rows) ? (1u << r) : 0u;
      }
      if (mask) {
        uint32_t pos = atomicAdd(s_nstage, (uint32_t)__builtin_popcount(mask));
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          if (mask & (1u << r)) {
            const uint64_t key = rarc_candkey((float)acc[r] * sc, row0 + (r & 3) + 8 * (r >> 2));
            if (pos < (uint32_t)Q8_STAGE) {
              s_skey[pos] = key;
              s_sq[pos] = (uint8_t)qidx;
            } else {
              emit(qidx, key);  // staging full (a tile where almost everything passes): go direct
            }
            ++pos;
          }
        }
      }
    }
  };
  // all threads; staged entries -> private candidate segments + global histogram
  auto flush = [&]() {
    uint32_t n = *s_nstage;
    n = n < (uint32_t)Q8_STAGE ? n : (uint32_t)Q8_STAGE;
    for (uint32_t e = tid; e < n; e += Q8_THREADS) emit(s_sq[e], s_skey[e]);
    q8_lds_barrier();
    if (tid == 0) *s_nstage = 0;
    q8_lds_barrier();
  };

  // ---- prologue: tile t0 straight into LDS buffer 0; tiles t0+stride, t0+2·stride in flight ----
  // (the launch guarantees gridDim.x <= n_tiles, so tile t0 exists)
  Fetch f[2];
  fetch(f[0], t0);
  convert_tile(f[0], 0);
  float tinv_cur = f[0].meta.y;  // 1/scale of the tile in the LDS buffer about to be read
  // (scheduling fences: the groups must be ISSUED in this order, or the counted waits the compiler
  // derives for the first loop iteration assume the wrong group is the newest)
  __builtin_amdgcn_sched_barrier(0);
  fetch(f[1], clamp_tile(t0 + stride));
  __builtin_amdgcn_sched_barrier(0);
  fetch(f[0], clamp_tile(t0 + 2 * stride));
  __builtin_amdgcn_sched_barrier(0);
  float last_pub = -INFINITY;  // owner lane: last threshold it published
  q8_lds_barrier();

  uint32_t it = 0;
  // one iteration = one tile; PAR (its parity) names the LDS buffer read and the fetch group consumed
#define Q8_ITER(PAR)                                                                                          \
  {                                                                                                           \
    const float tinv = tinv_cur;                                                                              \
    i32x16 acc = {0};                                                                                         \
    if (!(ABL & 4)) {                                                                                         \
      const char* a_base = smem + (PAR) * L::TILE + aoff;                                                     \
      _Pragma("unroll") for (int ks = 0; ks < KS; ++ks) {                                                     \
        const i32x4 a = *(const i32x4*)(a_base + 32 * ks);                                                    \
        acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, qf[ks], acc, 0, 0, 0);                                 \
      }                                                                                                       \
    }                                                                                                         \
    /* fetch group of tile cur+stride (issued two iterations ago): chunks -> int8 -> the other buffer */      \
    if (cur + stride < p.n_tiles) {                                                                           \
      convert_tile(f[(PAR) ^ 1], (PAR) ^ 1);                                                                  \
      tinv_cur = f[(PAR) ^ 1].meta.y;                                                                         \
    }                                                                                                         \
    if (!(ABL & 8)) {                                                                                         \
      thr = fmaxf(thr, __uint_as_float(f[(PAR) ^ 1].thr));                                                    \
      if (lane < 32) s_hland[32 * wave + lane] = f[(PAR) ^ 1].hw;                                             \
    }                                                                                                         \
    if (!(ABL & 1) && live) prune(acc, cur, tinv);                                                            \
    else if (acc[0] == 0x7fffffff) p.cnt2[0] = 1; /* keep the MFMAs alive */                                  \
    /* refill that group with tile cur+3·stride */                                                            \
    if (!(ABL & 2)) fetch(f[(PAR) ^ 1], clamp_tile(cur + 3 * stride));                                        \
    ++it;                                                                                                     \
    q8_lds_barrier();                                                                                         \
    /* owner: the histogram landed by all waves before this barrier -> a higher threshold */                  \
    if (!(ABL & 8) && wave == 0 && has_own && (it <= 16 || (it & 3) == 0)) {                                  \
      const int b = rarc_wave_find_from_top_256(s_hland[4 * lane], s_hland[4 * lane + 1],                     \
                                                s_hland[4 * lane + 2], s_hland[4 * lane + 3], p.kprime);      \
      if (b >= 2 && lane == 0) {                                                                              \
        const float lo = s_binlo[own_q];                                                                      \
        const float t = fmaxf(rarc_bin_threshold(b, lo, s_bininv[own_q]) - 2.0002f * s_eps8[own_q], lo);      \
        if (t > last_pub) { /* counts only grow: t is monotone; publish only real progress */                 \
          last_pub = t;                                                                                       \
          __hip_atomic_store(&p.thr[own_q], __float_as_uint(t), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  \
        }                                                                                                     \
      }                                                                                                       \
    }                                                                                                         \
    /* flush the staged survivors: every tile while thresholds are still forming, then rarely */              \
    if (!(ABL & 64)) {                                                                                        \
      const uint32_t ns = *s_nstage; /* same value in every thread: read after the barrier */                 \
      if (ns > 0 && (it <= 16 || (it & (it - 1)) == 0 || (it & 127) == 0 || ns > (uint32_t)Q8_STAGE / 2))     \
        flush();                                                                                              \
    }                                                                                                         \
  }

  // Always whole pairs of iterations, no exit from the middle of the loop body (a mid-body break makes
  // the compiler's wait-counter merge treat the wrong fetch group as the newest one and drain the
  // queue every iteration).  When a workgroup's tile count is odd the last half-iteration runs on a
  // tile index past the end: its scores are ignored (`live`), its fetch is the usual tile-0 dummy.
  for (uint32_t base = t0; base < p.n_tiles; base += 2 * stride) {
    {
      const uint32_t cur = base;
      const bool live = true;
      Q8_ITER(0)
    }
    {
      const uint32_t cur = base + stride;
      const bool live = cur < p.n_tiles;
      Q8_ITER(1)
    }
  }
#undef Q8_ITER
  __syncthreads();
  flush();
  if (tid < RARC_MAX_QUERIES) p.cnt2[(size_t)blockIdx.x * RARC_MAX_QUERIES + tid] = s_cnt[tid];
}

// ---- host side -----------------------------------------------------------------------------------
bool rarc_prof_next(hipEvent_t* start, hipEvent_t* stop);  // rarc_api.hip
int rarc_seed_launch(const uint16_t* corpus, int64_t n_rows, int d_pad, const uint16_t* q16, int nq, int kprime,
                     float bin_lo, float bin_hi, const float* sub_a, const float* sub_b, const RarcWs& ws,
                     hipStream_t s);  // scan_f16.hip

template <int D>
static int launch_scan_q8(const ScanQ8Params& p, int grid, hipStream_t s) {
  constexpr size_t lds = ScanQ8Lds<D>::TOTAL;
  static bool attr_done = false;
  if (!attr_done) {
    RARC_HIP_CHECK(hipFuncSetAttribute((const void*)rarc_scan_q8_kernel<D>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_done = true;
  }
  hipEvent_t e0, e1;
  const bool prof = rarc_prof_next(&e0, &e1);
  if (prof) RARC_HIP_CHECK(hipEventRecord(e0, s));
  hipLaunchKernelGGL(rarc_scan_q8_kernel<D>, dim3(grid), dim3(Q8_THREADS), lds, s, p);
  RARC_HIP_CHECK(hipGetLastError());
  if (prof) RARC_HIP_CHECK(hipEventRecord(e1, s));
  return RARC_OK;
}

// Host entry used by rarc_api.hip.  *grid_out = workgroups launched (owners of candidate segments).
int rarc_scan_q8_launch(const uint16_t* corpus, int64_t n_rows, int d_pad, const float* qmeta,
                        const uint16_t* q16, const int8_t* q8, const float* qinv, const float* eps16,
                        const float* eps8, int nq, int kprime, float bin_lo, float bin_hi, const RarcWs& ws,
                        int cap, int* grid_out, hipStream_t s) {
  ScanQ8Params p;
  p.corpus = (const uint4*)corpus;
  p.tmeta = (const float2*)(qmeta + RARC_QMETA_HDR);
  p.q8 = q8;
  p.qinv = qinv;
  p.eps8 = eps8;
  p.n_rows = (uint32_t)n_rows;
  p.n_tiles = (uint32_t)((n_rows + 31) / 32);
  p.thr = (uint32_t*)ws.thr;
  p.binlo = ws.binlo;
  p.binscale = ws.binscale;
  p.bininv = ws.bininv;
  p.hist = ws.hist;
  p.cnt2 = ws.cnt2;
  p.cand = ws.cand;
  p.seg = (uint32_t)(cap / RARC_MAX_WG);
  p.kprime = (uint32_t)kprime;
  p.nq = (uint32_t)nq;

  // seed pass (fp16 MFMA on a strided sample): t = k'-th best sample score, accurate to eps16, so
  // t − eps16 bounds the k-th best canonical score from below; rows whose int8 score is under
  // t − eps16 − eps8 are out
  int rc = rarc_seed_launch(corpus, n_rows, d_pad, q16, nq, kprime, bin_lo, bin_hi, eps16, eps8, ws, s);
  if (rc) return rc;

  int dev = 0, cus = 256;
  RARC_HIP_CHECK(hipGetDevice(&dev));
  RARC_HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  int grid = cus < RARC_MAX_WG ? cus : RARC_MAX_WG;
  if ((uint32_t)grid > p.n_tiles) grid = (int)p.n_tiles;
  *grid_out = grid;
  if (p.n_tiles == 0) return RARC_OK;
  switch (d_pad) {
    case 128: return launch_scan_q8<128>(p, grid, s);
    case 256: return launch_scan_q8<256>(p, grid, s);
    case 384: return launch_scan_q8<384>(p, grid, s);
    case 512: return launch_scan_q8<512>(p, grid, s);
    case 640: return launch_scan_q8<640>(p, grid, s);
    case 768: return launch_scan_q8<768>(p, grid, s);
    case 896: return launch_scan_q8<896>(p, grid, s);
    case 1024: return launch_scan_q8<1024>(p, grid, s);
    default:
      rarc_set_error("rarc_scan_q8: padded dim %d unsupported (multiple of 128, <= 1024)", d_pad);
      return RARC_E_UNSUPPORTED;
  }
}
