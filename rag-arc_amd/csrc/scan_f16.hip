// scan_f16.hip — fused  scores = Q · Dᵀ (fp16 MFMA, fp32 accumulate)  +  per-query pruning.
//
// Replaces the inner loop of faiss.IndexFlatIP.search reached from
//   encapsulation/database/vector_db/VectorStore_Faiss.py:263
// for up to 256 queries at once.  The B×N score matrix is never materialised.
//
// Shape of the computation (one persistent workgroup per CU, 8 waves):
//   * Q (256 × D fp16) lives in REGISTERS for the whole kernel: wave w owns queries
//     [32w, 32w+32) as the MFMA B operand (D/16 fragments × 4 VGPRs = 192 VGPRs at D=768).
//   * the corpus streams HBM → LDS in 32-row tiles by LDS-DMA (global_load_lds_dwordx4),
//     triple buffered, every DMA instruction covering 8 rows × one full 128-B line.
//     The LDS image is XOR-swizzled on the SOURCE side so that the MFMA A-fragment
//     ds_read_b128 of every 16-lane group hits 16 distinct 16-B slots (conflict free).
//   * per tile each wave issues D/16 × v_mfma_f32_32x32x16_f16 (A = 32 corpus rows from LDS,
//     B = its resident query fragments): lane l ends with 16 scores of query (l & 31).
//   * epilogue: one max + compare against the lane's query threshold; survivors (rare) go
//     through a per-wave LDS queue to global candidate lists, and a per-query histogram lets
//     the owning workgroup raise that query's threshold (always a valid lower bound of the
//     k'-th best score, so pruning never drops a true top-k' row).
//
// Algorithmic HBM bytes per launch: n_rows × D × 2  (DESIGN.md §kernels).
#include "rarc_common.h"

struct ScanParams {
  const half_t* corpus;  // [ceil32(n_rows)][D]
  const half_t* q16;     // [256][D]
  uint32_t n_rows;
  uint32_t n_tiles;
  uint32_t* thr;   // float bits [256]
  uint32_t* hist;  // [256][RARC_NB]
  uint32_t* cnt;   // [256]
  uint64_t* cand;  // [256][cap]
  uint32_t* flags;
  uint32_t cap;
  uint32_t kprime;
  uint32_t nq;
  float bin_lo, bin_scale, bin_inv_scale;
};

constexpr int SCAN_WAVES = 8;
constexpr int SCAN_PF = 4;     // A-fragment ring depth (ds_read_b128 in flight per wave)
constexpr int SCAN_WQ = 128;   // per-wave candidate queue entries

// ---- inline-asm pipeline steps (hipcc will not software-pipeline this loop at 240+ VGPRs) ----
// Every step names the registers it touches, so ordering between steps is by data flow.
#define RARC_DSREAD(dst, addr, off) \
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
// wait for the oldest ring slot, multiply, refill the slot with fragment s+PF
#define RARC_STEP_FIRST(acc, r, q, addr, off)                                       \
  asm volatile("s_waitcnt lgkmcnt(3)\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, 0\n\t"  \
               "ds_read_b128 %1, %3 offset:%4"                                      \
               : "=&v"(acc), "+v"(r) : "v"(q), "v"(addr), "n"(off))
#define RARC_STEP_MID(acc, r, q, addr, off)                                          \
  asm volatile("s_waitcnt lgkmcnt(3)\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0\n\t"  \
               "ds_read_b128 %1, %3 offset:%4"                                       \
               : "+v"(acc), "+v"(r) : "v"(q), "v"(addr), "n"(off))
#define RARC_STEP_TAIL(acc, r, q, n)                                                  \
  asm volatile("s_waitcnt lgkmcnt(%3)\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0"      \
               : "+v"(acc) : "v"(r), "v"(q), "n"(n))
// MFMA result -> VALU read needs software wait states when the producer is inside asm
#define RARC_MFMA_DRAIN(acc) asm volatile("s_nop 15\n\ts_nop 3" : "+v"(acc))

template <int D>
__global__ __launch_bounds__(SCAN_WAVES * 64, 2) void rarc_scan_f16_kernel(const ScanParams p) {
  static_assert(D % 128 == 0 && D >= 128 && D <= 768, "D must be a multiple of 128, <= 768");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NP = D / 64;            // 64-element (128-B) panels per row
  constexpr int KS = NP * 4;            // MFMA k-steps (K = 16 each)
  constexpr int TILE_BYTES = 32 * D * 2;
  constexpr int NDMA = TILE_BYTES / 1024;  // 1-KiB DMA wave-instructions per tile
  constexpr int DPW = NDMA / SCAN_WAVES;   // ... per wave
  static_assert(NDMA % SCAN_WAVES == 0, "tile must split evenly over the waves");
  constexpr int WQ_OFF = 3 * TILE_BYTES;

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int row = lane & 31, h = lane >> 5;
  const uint32_t qidx = wave * 32 + row;  // this lane's query

  // resident query fragments: B operand, lane holds Q[qidx][16*ks + 8*h .. +8)
  half8 qf[KS];
  {
    const half_t* qp = p.q16 + (size_t)qidx * D + 8 * h;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[ks] = *(const half8*)(qp + 16 * ks);
  }

  // A-fragment LDS offsets: element (r, chunk c) of a panel sits at r*128 + ((c ^ sw(r))<<4),
  // sw(r) = (r >> 1) & 7; k-step kk of a panel reads chunks 2kk (lanes 0-31) / 2kk+1 (32-63).
  int xk[4];
  {
    const int sw = (row >> 1) & 7;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) xk[kk] = row * 128 + (((2 * kk + h) ^ sw) << 4);
  }
  // DMA source offsets: instruction (panel pn, row-block b) moves rows 8b..8b+7 × 128 B;
  // lane = 8*(row in block) + slot, and fetches chunk slot ^ sw(row).
  const int drow = lane >> 3, dslot = lane & 7;
  const uint32_t voff_even = drow * (D * 2) + ((dslot ^ ((drow >> 1) & 7)) << 4);
  const uint32_t voff_odd = drow * (D * 2) + ((dslot ^ ((4 + (drow >> 1)) & 7)) << 4);

  uint64_t* wq_key = (uint64_t*)(smem + WQ_OFF) + wave * SCAN_WQ;
  uint32_t* wq_q = (uint32_t*)(smem + WQ_OFF + SCAN_WAVES * SCAN_WQ * 8) + wave * SCAN_WQ;
  int wq_n = 0;

  float thr = __uint_as_float(p.thr[qidx]);  // -inf for live queries, +inf for padding
  uint32_t thr_pend = __float_as_uint(thr);
  uint32_t hp0 = 0, hp1 = 0, hp2 = 0, hp3 = 0;  // owner: pending histogram words (wave 0)
  int own_q = -1;                                // owner: query whose histogram is in flight
  int event = 0;

  auto issue = [&](int buf, uint32_t tile) {
    const char* gbase = (const char*)p.corpus + (size_t)tile * TILE_BYTES;
#pragma unroll
    for (int j = 0; j < DPW; ++j) {
      const int i = wave * DPW + j;
      const int pn = i >> 2, b = i & 3;
      const char* g = gbase + (size_t)(8 * b) * (D * 2) + pn * 128 + ((b & 1) ? voff_odd : voff_even);
      __builtin_amdgcn_global_load_lds(RARC_GPTR(g), RARC_LPTR(smem + buf * TILE_BYTES + i * 1024),
                                       16, 0, 0);
    }
  };

  auto flush = [&]() {
    __builtin_amdgcn_wave_barrier();
    for (int i = lane; i < wq_n; i += 64) {
      const uint64_t key = wq_key[i];
      const uint32_t q = wq_q[i];
      const uint32_t pos = atomicAdd(&p.cnt[q], 1u);
      if (pos < p.cap) p.cand[(size_t)q * p.cap + pos] = key;
      else atomicOr(&p.flags[0], 1u);
      const float s = rarc_candscore(key);
      int bin = (int)floorf((s - p.bin_lo) * p.bin_scale);
      bin = bin < 0 ? 0 : (bin > RARC_NB - 1 ? RARC_NB - 1 : bin);
      atomicAdd(&p.hist[q * RARC_NB + bin], 1u);
    }
    // keep the DMA vmcnt bookkeeping exact: nothing but DMAs may stay in flight
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
  };

  const uint32_t t0 = blockIdx.x, stride = gridDim.x;
  if (t0 < p.n_tiles) issue(0, t0);
  if (t0 + stride < p.n_tiles) issue(1, t0 + stride);
  int buf = 0;
  uint32_t it = 0;
  for (uint32_t cur = t0; cur < p.n_tiles; cur += stride) {
    if (cur + stride < p.n_tiles) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DPW) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    {
      int nb = buf + 2;
      if (nb >= 3) nb -= 3;
      if (cur + 2 * stride < p.n_tiles) issue(nb, cur + 2 * stride);
    }

    // ---- 32 rows × 32 queries per wave: KS chained MFMAs fed by a 4-deep LDS read ring ----
    f32x16 acc;
    {
      const int tb = buf * TILE_BYTES;
      const int a0 = tb + xk[0], a1 = tb + xk[1], a2 = tb + xk[2], a3 = tb + xk[3];
      half8 r0, r1, r2, r3;
      RARC_DSREAD(r0, a0, 0);
      RARC_DSREAD(r1, a1, 0);
      RARC_DSREAD(r2, a2, 0);
      RARC_DSREAD(r3, a3, 0);
      RARC_STEP_FIRST(acc, r0, qf[0], a0, 4096);
      RARC_STEP_MID(acc, r1, qf[1], a1, 4096);
      RARC_STEP_MID(acc, r2, qf[2], a2, 4096);
      RARC_STEP_MID(acc, r3, qf[3], a3, 4096);
#pragma unroll
      for (int pn = 1; pn < NP - 1; ++pn) {
        RARC_STEP_MID(acc, r0, qf[4 * pn + 0], a0, (pn + 1) * 4096);
        RARC_STEP_MID(acc, r1, qf[4 * pn + 1], a1, (pn + 1) * 4096);
        RARC_STEP_MID(acc, r2, qf[4 * pn + 2], a2, (pn + 1) * 4096);
        RARC_STEP_MID(acc, r3, qf[4 * pn + 3], a3, (pn + 1) * 4096);
      }
      RARC_STEP_TAIL(acc, r0, qf[KS - 4], 3);
      RARC_STEP_TAIL(acc, r1, qf[KS - 3], 2);
      RARC_STEP_TAIL(acc, r2, qf[KS - 2], 1);
      RARC_STEP_TAIL(acc, r3, qf[KS - 1], 0);
      RARC_MFMA_DRAIN(acc);
    }

    // ---- pending threshold work issued one iteration ago (latency already covered) ----
    if (own_q >= 0) {  // wave 0 only: suffix-scan the owned query's histogram
      const uint32_t mine = hp0 + hp1 + hp2 + hp3;  // bins 4*lane .. 4*lane+3
      uint32_t suf = mine;                           // inclusive suffix sum over lanes >= lane
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_down(suf, d, 64);
        if (lane + d < 64) suf += o;
      }
      const uint64_t ge = __builtin_amdgcn_ballot_w64(suf >= p.kprime);
      if (ge) {
        const int hl = 63 - __builtin_clzll(ge);  // highest lane whose suffix reaches k'
        const uint32_t above = __shfl(suf - mine, hl, 64);
        const uint32_t c3 = __shfl(hp3, hl, 64), c2 = __shfl(hp2, hl, 64), c1 = __shfl(hp1, hl, 64);
        int b = 4 * hl;
        if (above + c3 >= p.kprime) b += 3;
        else if (above + c3 + c2 >= p.kprime) b += 2;
        else if (above + c3 + c2 + c1 >= p.kprime) b += 1;
        // every row dropped below lo + (b-1)/scale is provably in a bin < b
        if (b >= 2 && lane == 0) {
          const float t = p.bin_lo + (float)(b - 1) * p.bin_inv_scale;
          const float old = __uint_as_float(
              __hip_atomic_load(&p.thr[own_q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
          if (t > old)
            __hip_atomic_store(&p.thr[own_q], __float_as_uint(t), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      own_q = -1;
    }
    thr = fmaxf(thr, __uint_as_float(thr_pend));

    // ---- prune: lane holds 16 scores of query qidx (rows 8*(r>>2) + 4*h + (r&3)) ----
    float m = fmaxf(fmaxf(fmaxf(acc[0], acc[1]), fmaxf(acc[2], acc[3])),
                    fmaxf(fmaxf(acc[4], acc[5]), fmaxf(acc[6], acc[7])));
    m = fmaxf(m, fmaxf(fmaxf(fmaxf(acc[8], acc[9]), fmaxf(acc[10], acc[11])),
                       fmaxf(fmaxf(acc[12], acc[13]), fmaxf(acc[14], acc[15]))));
    if (__builtin_amdgcn_ballot_w64(m >= thr) != 0) {
      const uint32_t row0 = cur * 32 + 4 * h;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const uint32_t doc = row0 + (r & 3) + 8 * (r >> 2);
        const bool pass = (acc[r] >= thr) && (doc < p.n_rows);
        const uint64_t mask = __builtin_amdgcn_ballot_w64(pass);
        if (mask) {
          const int n = __builtin_popcountll(mask);
          if (wq_n + n > SCAN_WQ) {
            flush();
            wq_n = 0;
          }
          if (pass) {
            const int off = wq_n + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32),
                                       __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0));
            wq_key[off] = rarc_candkey(acc[r], doc);
            wq_q[off] = qidx;
          }
          wq_n += n;
        }
      }
    }

    // ---- threshold refresh events: it = 1,2,4,8,... then every 64 tiles ----
    ++it;
    if ((it & (it - 1)) == 0 || (it & 63) == 0) {
      if (wq_n > 0) {  // publish what we have so the histograms see it
        flush();
        wq_n = 0;
      }
      if (wave == 0) {
        const uint32_t owned = (p.nq + stride - 1 - blockIdx.x) / stride;  // queries ≡ blockIdx (mod grid)
        if (owned > 0) {
          own_q = blockIdx.x + stride * (event % owned);
          const uint32_t* hq = p.hist + (size_t)own_q * RARC_NB + 4 * lane;
          hp0 = __hip_atomic_load(hq + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          hp1 = __hip_atomic_load(hq + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          hp2 = __hip_atomic_load(hq + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          hp3 = __hip_atomic_load(hq + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        ++event;
      }
      thr_pend = __hip_atomic_load(&p.thr[qidx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }

    buf = buf + 1;
    if (buf >= 3) buf = 0;
  }
  if (wq_n > 0) flush();
}

// ---- init: thresholds, counters, histograms ------------------------------------------------
__global__ void rarc_scan_init_kernel(uint32_t* thr, uint32_t* cnt, uint32_t* flags, uint32_t* hist,
                                      uint32_t nq) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < RARC_MAX_QUERIES) {
    thr[i] = (i < nq) ? 0xff800000u : 0x7f800000u;  // -inf : +inf (padding never passes)
    cnt[i] = 0;
  }
  if (i < 64) flags[i] = 0;
  for (uint32_t j = i; j < RARC_MAX_QUERIES * RARC_NB; j += gridDim.x * blockDim.x) hist[j] = 0;
}

template <int D>
static int launch_scan(const ScanParams& p, int grid, hipStream_t s) {
  constexpr size_t lds = 3 * 32 * D * 2 + SCAN_WAVES * SCAN_WQ * 12;
  static bool attr_done = false;
  if (!attr_done) {
    RARC_HIP_CHECK(hipFuncSetAttribute((const void*)rarc_scan_f16_kernel<D>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_done = true;
  }
  hipLaunchKernelGGL(rarc_scan_f16_kernel<D>, dim3(grid), dim3(SCAN_WAVES * 64), lds, s, p);
  RARC_HIP_CHECK(hipGetLastError());
  return RARC_OK;
}

// Host entry used by rarc_api.cpp.
int rarc_scan_f16_launch(const uint16_t* corpus, int64_t n_rows, int d_pad, const uint16_t* q16, int nq,
                         int kprime, float bin_lo, float bin_hi, const RarcWs& ws, int cap,
                         hipStream_t s) {
  ScanParams p;
  p.corpus = (const half_t*)corpus;
  p.q16 = (const half_t*)q16;
  p.n_rows = (uint32_t)n_rows;
  p.n_tiles = (uint32_t)((n_rows + 31) / 32);
  p.thr = (uint32_t*)ws.thr;
  p.hist = ws.hist;
  p.cnt = ws.cnt;
  p.cand = ws.cand;
  p.flags = ws.flags;
  p.cap = (uint32_t)cap;
  p.kprime = (uint32_t)kprime;
  p.nq = (uint32_t)nq;
  p.bin_lo = bin_lo;
  p.bin_scale = (float)RARC_NB / (bin_hi - bin_lo);
  p.bin_inv_scale = (bin_hi - bin_lo) / (float)RARC_NB;

  hipLaunchKernelGGL(rarc_scan_init_kernel, dim3(64), dim3(256), 0, s, (uint32_t*)ws.thr, ws.cnt,
                     ws.flags, ws.hist, (uint32_t)nq);
  RARC_HIP_CHECK(hipGetLastError());
  if (p.n_tiles == 0) return RARC_OK;

  int dev = 0, cus = 256;
  RARC_HIP_CHECK(hipGetDevice(&dev));
  RARC_HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  int grid = cus;
  if ((uint32_t)grid > p.n_tiles) grid = (int)p.n_tiles;
  switch (d_pad) {
    case 128: return launch_scan<128>(p, grid, s);
    case 256: return launch_scan<256>(p, grid, s);
    case 384: return launch_scan<384>(p, grid, s);
    case 512: return launch_scan<512>(p, grid, s);
    case 640: return launch_scan<640>(p, grid, s);
    case 768: return launch_scan<768>(p, grid, s);
    default:
      rarc_set_error("rarc_scan_f16: padded dim %d unsupported (multiple of 128, <= 768)", d_pad);
      return RARC_E_UNSUPPORTED;
  }
}
