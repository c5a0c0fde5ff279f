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
//   * per tile each wave issues D/32 × 4 v_mfma_f32_16x16x32_f16 on four accumulators (A = 16 corpus rows
//     from LDS, B = 16 of its resident queries): lane l ends with 2 x 4 scores of each of the queries
//     (l & 15) and 16 + (l & 15).  (SCAN_MMA16=0 builds the first form: D/16 chained 32x32x16 on one
//     accumulator, which issue every ~51 cycles whenever the SIMD's other wave is outside its own chain;
//     measured 0.370 -> 0.364 ms per 1M x 768 scan in steady state — the board sits at its power limit with
//     the shader clock near 1.4 GHz under this kernel, so matrix-pipe bubbles removed come back as clock.)
//   * epilogue: one max + compare against the lane's query threshold.  Survivors (≈1 per
//     wave-tile) take a slot in the workgroup's private segment of that query's candidate list
//     (slot counter in LDS — no returning global atomic, nothing to wait for) and bump the
//     query's global histogram; the workgroup that owns the query turns the histogram into a
//     higher threshold (always a valid lower bound of the k'-th best score, so pruning never
//     drops a true top-k' row).
//
// Algorithmic HBM bytes per launch: n_rows × D × 2  (DESIGN.md §kernels).
#include "rarc_common.h"
#include <stdlib.h>

struct ScanParams {
  const half_t* corpus;  // [ceil32(n_rows)][D]
  const half_t* q16;     // [256][D]
  uint32_t n_rows;
  uint32_t n_tiles;
  uint32_t* thr;          // float bits [256]
  const float* binlo;     // [256]
  const float* binscale;  // [256]
  const float* bininv;    // [256]
  uint32_t* hist;         // [256][RARC_NB]
  uint32_t* cnt2;         // [256 wg][256 q]
  uint64_t* cand;         // [256 q][256 wg][seg]
  uint32_t seg;           // slots per (query, workgroup)
  uint32_t kprime;
  uint32_t nq;
  unsigned long long* dbg;  // tools/scan_bench timeline (ABL & 64), else null
  // Rigorous-margin mode (the fp16 STAGE of the hybrid small-shard search, scan_q8.hip): the owner publishes
  // edge - margin_scale * margin[q] instead of the bin edge.  With margin = eps16 (|fp16 score - canonical| <= eps16) and
  // scale 2: >= k' rows score >= edge, so the k'-th best canonical score L >= edge - eps16, and a row as good as L scores
  // >= L - eps16 >= edge - 2 eps16 — discarding below that loses nothing, with no certificate needed afterwards.
  const float* margin;      // null: the classic path (k' > k candidates + exactness certificate in rarc_finalize_kernel)
  float margin_scale;
};

constexpr int SCAN_WAVES = 8;

// ---- inline-asm MFMA pipeline (hipcc will not software-pipeline this loop at 240+ VGPRs) ----
// One tile = KS chained v_mfma_f32_32x32x16_f16 fed by an R-deep ring of ds_read_b128.
// Step S: wait until ring slot S%R has landed (in-order LDS returns: lgkmcnt(R-1)), multiply,
// refill the slot with fragment S+R.  Every asm statement names the registers it touches, so the
// order between steps is fixed by data flow; immediates come from template parameters.
constexpr int SCAN_RING = 6;
#ifdef SCAN_HALFREAD
constexpr int SCAN_LGK = SCAN_RING / 2 - 1;
#else
constexpr int SCAN_LGK = SCAN_RING - 1;
#endif
#ifndef SCAN_DMA_B_INLOOP
#define SCAN_DMA_B_INLOOP 0  // group B's pieces: 0 = all right after the barrier, 1 = spread through its MFMA chain too
#endif
#ifndef SCAN_DMA_EVERY
#define SCAN_DMA_EVERY 8  // MFMA steps between two LDS-DMA issues of a wave (1 = all in the first steps)
#endif

template <int S, int KS, int R>
struct ScanSteps {
  // `dma(j)` issues this wave's j-th LDS-DMA piece of a later tile (group A only): one call per step
  // during the first steps, so the (slow, CU-serialised) VMEM issue hides in MFMA shadows instead of
  // delaying the tile's first MFMA, yet the pieces still get two tile periods to land
  template <class Dma>
  static __device__ __forceinline__ void run(f32x16& acc, half8 (&rg)[R], const half8 (&qf)[KS], int a0, int a1,
                                             int a2, int a3, Dma& dma) {
    constexpr int slot = S % R;
    if constexpr (S % SCAN_DMA_EVERY == 1 && S / SCAN_DMA_EVERY < KS / 8) dma(S / SCAN_DMA_EVERY);
#ifdef SCAN_HALFREAD  // timing experiment (wrong results): every second A fragment is not read, its MFMA reuses the neighbour's
    if constexpr (S + R < KS && ((S + R) & 1)) {
      if constexpr (S == 0) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(acc) : "v"(rg[slot ^ 1]), "v"(qf[S]));
      else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(rg[slot ^ 1]), "v"(qf[S]));
    } else
#endif
    if constexpr (S + R < KS) {
      constexpr int S2 = S + R;  // fragment that refills this slot
      const int addr = (S2 & 3) == 0 ? a0 : (S2 & 3) == 1 ? a1 : (S2 & 3) == 2 ? a2 : a3;
      if constexpr (S == 0)
        asm volatile("s_waitcnt lgkmcnt(%5)\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, 0\n\t"
                     "ds_read_b128 %1, %3 offset:%4"
                     : "=&v"(acc), "+v"(rg[slot]) : "v"(qf[S]), "v"(addr), "n"((S2 >> 2) * 4096), "n"(SCAN_LGK));
      else
        asm volatile("s_waitcnt lgkmcnt(%5)\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0\n\t"
                     "ds_read_b128 %1, %3 offset:%4"
                     : "+v"(acc), "+v"(rg[slot]) : "v"(qf[S]), "v"(addr), "n"((S2 >> 2) * 4096), "n"(SCAN_LGK));
    } else if constexpr (S == 0) {
      asm volatile("s_waitcnt lgkmcnt(%3)\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, 0"
                   : "=&v"(acc) : "v"(rg[slot]), "v"(qf[S]), "n"(KS - 1 - S));
    } else {
      asm volatile("s_waitcnt lgkmcnt(%3)\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0"
                   : "+v"(acc) : "v"(rg[slot]), "v"(qf[S]), "n"(KS - 1 - S));
    }
    if constexpr (S + 1 < KS) ScanSteps<S + 1, KS, R>::run(acc, rg, qf, a0, a1, a2, a3, dma);
  }
};
template <int S, int R>
struct ScanPrologue {
  static __device__ __forceinline__ void run(half8 (&rg)[R], int a0, int a1, int a2, int a3) {
    const int addr = (S & 3) == 0 ? a0 : (S & 3) == 1 ? a1 : (S & 3) == 2 ? a2 : a3;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(rg[S]) : "v"(addr), "n"((S >> 2) * 4096));
    if constexpr (S + 1 < R) ScanPrologue<S + 1, R>::run(rg, a0, a1, a2, a3);
  }
};
// MFMA result -> VALU read needs software wait states when the producer is inside asm
#define RARC_MFMA_DRAIN(acc) asm volatile("s_nop 15\n\ts_nop 3" : "+v"(acc))
// ---- the same tile with v_mfma_f32_16x16x32_f16: 2 x 2 blocks of 16 rows x 16 queries per wave (SCAN_MMA16) ----
// A chain of MFMAs on ONE accumulator with the ring's s_waitcnt / ds_read between them issues every ~51 cycles while the
// SIMD's other wave is outside its own chain (s_memtime timeline of tools/scan_bench: that is ~40 % of a tile period);
// four accumulators taken in turn never wait for each other.  Step S covers K = 32: the fragments of rows 0-15 and 16-31
// (a ring of R steps), four MFMAs, refill.  Same LDS image, same bytes read from LDS, same registers.
#ifndef SCAN_MMA16
#define SCAN_MMA16 1
#endif
#ifndef SCAN_DMA_OUTSIDE
#define SCAN_DMA_OUTSIDE 0  // 1: no DMA issue from inside an MFMA chain (measured: 0.3675 vs 0.3636 ms per 1M x 768 scan with it inside)
#endif
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int SCAN_RING16 = 3;
template <int S, int KS2, int R>
struct ScanSteps16 {
  template <class Dma>
  static __device__ __forceinline__ void run(f32x4 (&c)[4], half8 (&ra)[R], half8 (&rb)[R], const half8 (&qf)[2 * KS2],
                                             int ae, int ao, Dma& dma) {
    constexpr int slot = S % R;
    if constexpr (S % 4 == 1 && S / 4 < KS2 / 4) dma(S / 4);
    if constexpr (S + R < KS2) {
      constexpr int S2 = S + R;  // step that refills this slot
      const int addr = (S2 & 1) ? ao : ae;
      if constexpr (S == 0)
        asm volatile("s_waitcnt lgkmcnt(%11)\n\t"
                     "v_mfma_f32_16x16x32_f16 %0, %4, %6, 0\n\tv_mfma_f32_16x16x32_f16 %1, %4, %7, 0\n\t"
                     "v_mfma_f32_16x16x32_f16 %2, %5, %6, 0\n\tv_mfma_f32_16x16x32_f16 %3, %5, %7, 0\n\t"
                     "ds_read_b128 %4, %8 offset:%9\n\tds_read_b128 %5, %8 offset:%10"
                     : "=&v"(c[0]), "=&v"(c[1]), "=&v"(c[2]), "=&v"(c[3]), "+v"(ra[slot]), "+v"(rb[slot])
                     : "v"(qf[2 * S]), "v"(qf[2 * S + 1]), "v"(addr), "n"((S2 >> 1) * 4096), "n"((S2 >> 1) * 4096 + 2048),
                       "n"(2 * R - 2));
      else
        asm volatile("s_waitcnt lgkmcnt(%11)\n\t"
                     "v_mfma_f32_16x16x32_f16 %0, %4, %6, %0\n\tv_mfma_f32_16x16x32_f16 %1, %4, %7, %1\n\t"
                     "v_mfma_f32_16x16x32_f16 %2, %5, %6, %2\n\tv_mfma_f32_16x16x32_f16 %3, %5, %7, %3\n\t"
                     "ds_read_b128 %4, %8 offset:%9\n\tds_read_b128 %5, %8 offset:%10"
                     : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(ra[slot]), "+v"(rb[slot])
                     : "v"(qf[2 * S]), "v"(qf[2 * S + 1]), "v"(addr), "n"((S2 >> 1) * 4096), "n"((S2 >> 1) * 4096 + 2048),
                       "n"(2 * R - 2));
    } else if constexpr (S == 0) {
      asm volatile("s_waitcnt lgkmcnt(%8)\n\t"
                   "v_mfma_f32_16x16x32_f16 %0, %4, %6, 0\n\tv_mfma_f32_16x16x32_f16 %1, %4, %7, 0\n\t"
                   "v_mfma_f32_16x16x32_f16 %2, %5, %6, 0\n\tv_mfma_f32_16x16x32_f16 %3, %5, %7, 0"
                   : "=&v"(c[0]), "=&v"(c[1]), "=&v"(c[2]), "=&v"(c[3])
                   : "v"(ra[slot]), "v"(rb[slot]), "v"(qf[2 * S]), "v"(qf[2 * S + 1]), "n"(2 * (KS2 - 1 - S)));
    } else {
      asm volatile("s_waitcnt lgkmcnt(%8)\n\t"
                   "v_mfma_f32_16x16x32_f16 %0, %4, %6, %0\n\tv_mfma_f32_16x16x32_f16 %1, %4, %7, %1\n\t"
                   "v_mfma_f32_16x16x32_f16 %2, %5, %6, %2\n\tv_mfma_f32_16x16x32_f16 %3, %5, %7, %3"
                   : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3])
                   : "v"(ra[slot]), "v"(rb[slot]), "v"(qf[2 * S]), "v"(qf[2 * S + 1]), "n"(2 * (KS2 - 1 - S)));
    }
    if constexpr (S + 1 < KS2) ScanSteps16<S + 1, KS2, R>::run(c, ra, rb, qf, ae, ao, dma);
  }
};
template <int S, int R>
struct ScanPrologue16 {
  static __device__ __forceinline__ void run(half8 (&ra)[R], half8 (&rb)[R], int ae, int ao) {
    const int addr = (S & 1) ? ao : ae;
    asm volatile("ds_read_b128 %0, %2 offset:%3\n\tds_read_b128 %1, %2 offset:%4"
                 : "=&v"(ra[S]), "=&v"(rb[S]) : "v"(addr), "n"((S >> 1) * 4096), "n"((S >> 1) * 4096 + 2048));
    if constexpr (S + 1 < R) ScanPrologue16<S + 1, R>::run(ra, rb, ae, ao);
  }
};
#define RARC_MFMA_DRAIN4(c) asm volatile("s_nop 15\n\ts_nop 3" : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]))

// LDS carve (bytes): 3 tile buffers | slot counters [256] | binlo | binscale | bininv | owner landing [256]
template <int D>
struct ScanLds {
  static constexpr int TILE_BYTES = 32 * D * 2;
  static constexpr int CNT = 3 * TILE_BYTES;
  static constexpr int BINLO = CNT + 1024;
  static constexpr int BINSCALE = BINLO + 1024;
  static constexpr int BININV = BINSCALE + 1024;
  static constexpr int HLAND = BININV + 1024;  // owner: histogram of the owned query, filled by DMA
  static constexpr int TLAND = HLAND + 1024;   // per wave 64 x 4 B: refreshed thresholds, filled by DMA
  static constexpr int THOFF = TLAND + 2048;   // per query: what the owner takes off a bin edge before publishing it (0, or 2 eps16)
  static constexpr int TOTAL = THOFF + 1024;
};

// ABL: ablation bits for tools/scan_bench (0 in the product build)
//   1 = no pruning epilogue, 2 = no DMA after the prologue, 4 = no MFMA loop, 8 = no refresh events
//   16 = non-temporal tile DMA, 32 = s_setprio 1 during the MFMA phase (experiments)
//   64 = record s_memtime stamps of workgroup 0 into p.dbg[iter][wave][8]
template <int D, int ABL = 0>
__global__ __launch_bounds__(SCAN_WAVES * 64, 2) void rarc_scan_f16_kernel(const ScanParams p) {
  static_assert(D % 128 == 0 && D >= 128 && D <= 768, "D must be a multiple of 128, <= 768");
  static_assert(D / 16 >= SCAN_RING, "ring deeper than the k-loop");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using L = ScanLds<D>;
  constexpr int NP = D / 64;            // 64-element (128-B) panels per row
  constexpr int KS = NP * 4;            // MFMA k-steps (K = 16 each)
  constexpr int TILE_BYTES = L::TILE_BYTES;
  constexpr int NDMA = TILE_BYTES / 1024;  // 1-KiB DMA wave-instructions per tile
  constexpr int DPW = NDMA / SCAN_WAVES;   // ... per wave
  static_assert(NDMA % SCAN_WAVES == 0, "tile must split evenly over the waves");

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int row = lane & 31, h = lane >> 5;
  const uint32_t qidx = wave * 32 + row;  // this lane's query

  uint32_t* s_cnt = (uint32_t*)(smem + L::CNT);
  float* s_binlo = (float*)(smem + L::BINLO);
  float* s_binscale = (float*)(smem + L::BINSCALE);
  float* s_bininv = (float*)(smem + L::BININV);
  if (tid < RARC_MAX_QUERIES) {
    s_cnt[tid] = 0;
    s_binlo[tid] = p.binlo[tid];
    s_binscale[tid] = p.binscale[tid];
    s_bininv[tid] = p.bininv[tid];
    ((float*)(smem + L::THOFF))[tid] = p.margin ? p.margin_scale * p.margin[tid] + 1e-30f : 0.f;
  }

#if SCAN_MMA16
  // resident query fragments (16x16x32 B operands): lane holds Q[q_lo + 16 b][32 s + 8 (lane >> 4) .. +8) in qf[2 s + b]
  constexpr int KS2 = D / 32;
  static_assert(KS2 >= SCAN_RING16, "ring deeper than the k-loop");
  const uint32_t q_lo = wave * 32 + (lane & 15);  // this lane's two queries: q_lo and q_lo + 16
  half8 qf[2 * KS2];
  {
    const half_t* qp = p.q16 + (size_t)q_lo * D + 8 * (lane >> 4);
#pragma unroll
    for (int ks = 0; ks < KS2; ++ks) {
      qf[2 * ks] = *(const half8*)(qp + 32 * ks);
      qf[2 * ks + 1] = *(const half8*)(qp + 16 * D + 32 * ks);
    }
  }
#else
  // resident query fragments: B operand, lane holds Q[qidx][16*ks + 8*h .. +8)
  half8 qf[KS];
  {
    const half_t* qp = p.q16 + (size_t)qidx * D + 8 * h;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[ks] = *(const half8*)(qp + 16 * ks);
  }
#endif

  // A-fragment LDS offsets: element (r, chunk c) of a panel sits at r*128 + ((c ^ sw(r))<<4),
  // sw(r) = (r >> 1) & 7; k-step kk of a panel reads chunks 2kk (lanes 0-31) / 2kk+1 (32-63).
  // a0..a3 always point into the CURRENT tile buffer (advanced by one buffer per iteration).
#if SCAN_MMA16
  // 16x16x32: lane reads row (lane & 15) [+16 through the offset field: same swizzle], chunk 4 (s & 1) + (lane >> 4) of
  // panel s >> 1; (4 a + g) ^ sw = (g ^ sw) ^ 4 a, so odd steps are the even steps' address ^ 64.  A 16-lane group still
  // covers 16 distinct 16-B slots (8 (r & 1) + (c ^ (r >> 1))).
  int a0, a1;
  {
    const int r16 = lane & 15, sw = (r16 >> 1) & 7;
    a0 = r16 * 128 + ((((lane >> 4)) ^ sw) << 4);
    a1 = a0 ^ 64;
  }
#else
  int a0, a1, a2, a3;
  {
    const int sw = (row >> 1) & 7;
    a0 = row * 128 + (((0 + h) ^ sw) << 4);
    a1 = row * 128 + (((2 + h) ^ sw) << 4);
    a2 = row * 128 + (((4 + h) ^ sw) << 4);
    a3 = row * 128 + (((6 + h) ^ sw) << 4);
  }
#endif
  // DMA source offsets: instruction (panel pn, row-block b) moves rows 8b..8b+7 × 128 B;
  // lane = 8*(row in block) + slot, and fetches chunk slot ^ sw(row).
  const int drow = lane >> 3, dslot = lane & 7;
  // (odd row-blocks: sw differs by 4, i.e. the chunk offset by 64 bytes -> voff ^ 64)
  const uint32_t voff_even = drow * (D * 2) + ((dslot ^ ((drow >> 1) & 7)) << 4);

#if SCAN_MMA16
  float thr = __uint_as_float(p.thr[q_lo]), thr1 = __uint_as_float(p.thr[q_lo + 16]);  // seed thresholds; +inf for padding
#else
  float thr = __uint_as_float(p.thr[qidx]);  // seed threshold; +inf for padding queries
  const float my_binlo = p.binlo[qidx], my_binscale = p.binscale[qidx];  // this lane's query window
#endif
  bool pend = false;  // a threshold refresh (and, on wave 0, an owned histogram) is in flight to LDS
  int own_q = -1;     // owner (wave 0): query whose histogram is in flight
  int event = 0;

  // Per-wave constants of its DPW DMA instructions: source offset inside the tile (SGPR soffset),
  // LDS landing offset, and which of the two per-lane source patterns (even / odd row-block) applies.
  const uint32_t voff_odd = voff_even ^ 64u;
  int dma_soff[DPW], dma_loff[DPW];
  bool dma_odd[DPW];
#pragma unroll
  for (int j = 0; j < DPW; ++j) {
    const int i = wave * DPW + j;
    const int pn = i >> 2, b = i & 3;
    dma_soff[j] = 8 * b * (D * 2) + pn * 128;
    dma_loff[j] = i * 1024;
    dma_odd[j] = (b & 1) != 0;
  }
  // one tile = DPW buffer_load_dwordx4 ... lds per wave: descriptor rebased per tile (2 SALU), the
  // rest of the address is an SGPR offset + a per-lane VGPR offset -> ~2 scalar ops per instruction
  auto issue_piece = [&](int buf, uint32_t tile, int j) {
#if defined(__HIP_DEVICE_COMPILE__)  // buffer builtins exist only in the device pass of this file
    auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.corpus + (size_t)tile * TILE_BYTES), 0,
                                                  TILE_BYTES, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, RARC_LPTR(smem + buf * TILE_BYTES + dma_loff[j]), 16,
                                             dma_odd[j] ? voff_odd : voff_even, dma_soff[j], 0, (ABL & 16) ? 2 : 0);
#else
    (void)buf; (void)tile; (void)j;
#endif
  };
  auto issue = [&](int buf, uint32_t tile) {
#pragma unroll
    for (int j = 0; j < DPW; ++j) issue_piece(buf, tile, j);
  };

  // one surviving score: slot from the LDS counter, key to the private segment, histogram bump.
  // Nothing here returns data through the vector-memory queue, so nothing has to be waited for.
  auto append = [&](float s, uint32_t doc, uint32_t q) {
    const uint32_t slot = rarc_lds_add_rtn(L::CNT + 4 * q, 1u);
    if (slot < p.seg)
      p.cand[((size_t)q * RARC_MAX_WG + blockIdx.x) * p.seg + slot] = rarc_candkey(s, doc);
#if SCAN_MMA16
    const int bin = rarc_bin_of(s, rarc_lds_read_f32(L::BINLO + 4 * q), rarc_lds_read_f32(L::BINSCALE + 4 * q));
#else
    const int bin = rarc_bin_of(s, my_binlo, my_binscale);
#endif
    atomicAdd(&p.hist[q * RARC_NB + bin], 1u);
  };

#if SCAN_MMA16
  // prune one tile's scores: c[2 rb + qb][i] = score of row 16 rb + 4 (lane >> 4) + i against query q_lo + 16 qb
  auto prune = [&](const f32x4 (&c)[4], uint32_t tile) {
    float mb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) mb[j] = fmaxf(fmaxf(c[j][0], c[j][1]), fmaxf(c[j][2], c[j][3]));
    const bool hit = fmaxf(mb[0], mb[2]) >= thr || fmaxf(mb[1], mb[3]) >= thr1;
    if (__builtin_amdgcn_ballot_w64(hit) != 0) {
      const int ln = rarc_fresh_lane();
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float t = (j & 1) ? thr1 : thr;
        if (__builtin_amdgcn_ballot_w64(mb[j] >= t) != 0) {
          const uint32_t q = wave * 32 + 16 * (j & 1) + (ln & 15);
          const uint32_t row0 = tile * 32 + 16 * (j >> 1) + 4 * (ln >> 4);
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (c[j][i] >= t && row0 + i < p.n_rows) append(c[j][i], row0 + i, q);
        }
      }
    }
  };
#else
  // prune one tile's scores: lane holds 16 scores of its query (rows 8*(r>>2) + 4*h + (r&3))
  auto prune = [&](const f32x16& acc, uint32_t tile) {
      const float m0 = fmaxf(fmaxf(acc[0], acc[1]), fmaxf(acc[2], acc[3]));
      const float m1 = fmaxf(fmaxf(acc[4], acc[5]), fmaxf(acc[6], acc[7]));
      const float m2 = fmaxf(fmaxf(acc[8], acc[9]), fmaxf(acc[10], acc[11]));
      const float m3 = fmaxf(fmaxf(acc[12], acc[13]), fmaxf(acc[14], acc[15]));
      const float m = fmaxf(fmaxf(m0, m1), fmaxf(m2, m3));
      if (__builtin_amdgcn_ballot_w64(m >= thr) != 0) {
        const int ln = rarc_fresh_lane();
        const uint32_t q = wave * 32 + (ln & 31);
        const uint32_t row0 = tile * 32 + 4 * (ln >> 5);
        const float mg[4] = {m0, m1, m2, m3};
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          if (__builtin_amdgcn_ballot_w64(mg[g] >= thr) != 0) {
#pragma unroll
            for (int r = 4 * g; r < 4 * g + 4; ++r) {
              const uint32_t doc = row0 + (r & 3) + 8 * (r >> 2);
              if (acc[r] >= thr && doc < p.n_rows) append(acc[r], doc, q);
            }
          }
        }
      }
  };
#endif

  const uint32_t t0 = blockIdx.x, stride = gridDim.x;
  if (t0 < p.n_tiles) issue(0, t0);
  if (t0 + stride < p.n_tiles) issue(1, t0 + stride);
  int buf = 0;
  uint32_t it = 0;
  // Ping-pong: waves w and w+4 share a SIMD.  Group A (waves 0-3) runs MFMA(t) then prune(t);
  // group B (waves 4-7) runs prune(t-1) then MFMA(t).  Each wave's VALU/LDS/VMEM epilogue thus
  // overlaps its SIMD partner's MFMA stream instead of idling the matrix pipe.
  const bool grp_b = wave >= SCAN_WAVES / 2;
  // Round 6 — a wave whose 32 queries all lie at or beyond nq (zero rows of the query block) has nothing to multiply: its
  // scores are zeros whatever the tile holds, and nobody reads them (the finalize runs nq workgroups).  It skips the MFMA
  // chain and the pruning and keeps its part of the staging — DMA pieces, barriers, threshold traffic — exactly as it was, so
  // the vector-memory order the waits count on does not change.  The reference's own call is ONE query (VectorStore_Faiss.py:
  // 258-263): seven of the eight waves then leave the matrix pipe alone and the kernel runs at what HBM delivers
  // instead of at the power-limited MFMA rate of a 256-query batch.  Same bits for every real query.
  const bool idle_wave = (uint32_t)wave * 32u >= p.nq;
  uint32_t prev = 0xffffffffu;  // group B: tile whose scores are still in acc
#if SCAN_MMA16
  f32x4 acc[4];
#else
  f32x16 acc;
#endif
// stamps go to LDS (8 KiB behind the carve, iterations 32..63) and are copied out at the end: a global store per stamp would sit
// in the vector-memory queue and make every counted vmcnt wait of the instrumented workgroup stricter than the product's
#define RARC_STAMP_AT(iter, slot)                                                                          \
  if ((ABL & 64) && blockIdx.x == 0 && (iter) >= 32u && (iter) < 64u) {                                     \
    const uint32_t t_ = (uint32_t)__builtin_amdgcn_s_memtime();                                             \
    asm volatile("ds_write_b32 %0, %1" ::"v"((uint32_t)(L::TOTAL + ((((iter)-32u) * SCAN_WAVES + wave) * 8 + (slot)) * 4)), \
                 "v"(t_) : "memory");                                                                       \
  }
#define RARC_STAMP(slot) RARC_STAMP_AT(it, slot)
  for (uint32_t cur = t0; cur < p.n_tiles; cur += stride) {
    // DMA(cur) landed?  Loads return in order, so "at most N outstanding" == everything older than
    // the newest N vector-memory ops has completed.  The newest ops of this wave are, by
    // construction, the DPW pieces of the next tile followed by the refresh-event DMAs (0, 1 or 2)
    // issued at the end of the previous iteration (group A's prune stores may sit in between: they
    // only make the wait stricter, and have long been acknowledged by the time we get here).
    {
      const int ev = (pend ? 1 : 0) + (own_q >= 0 ? 1 : 0);
      if ((ABL & 2) || cur + stride >= p.n_tiles) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else if (ev == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DPW) : "memory");
      else if (ev == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DPW + 1) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DPW + 2) : "memory");
    }
    RARC_STAMP_AT(it - 1u, 6)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    RARC_STAMP(0)
    const bool issued = !(ABL & 2) && cur + 2 * stride < p.n_tiles;
    int nb = buf + 2;
    if (nb >= 3) nb -= 3;
    // the DMA of tile cur+2 is issued piecewise from inside the MFMA loop (see ScanSteps)
    const uint32_t dma_tile = cur + 2 * stride;
    auto dma = [&](int j) {
      if (SCAN_DMA_OUTSIDE) return;
      if (issued && (SCAN_DMA_B_INLOOP || !grp_b)) {  // (else group B issued all its pieces right after the barrier)
        asm volatile("" ::: "memory");
        issue_piece(nb, dma_tile, j);
        asm volatile("" ::: "memory");
      }
    };
#if SCAN_DMA_OUTSIDE
    // A wave never issues DMA from inside its MFMA chain: with four accumulators one wave keeps its SIMD's matrix pipe full
    // on its own, so the wave that is OUTSIDE its chain does the vector-memory issue (it may block on a full queue without
    // costing a matrix slot).  Group B: prune(t-1), DMA(t+2), MFMA(t); group A: MFMA(t), prune(t), DMA(t+2).  Pruning first
    // keeps its stores OLDER than the pieces, so "at most DPW outstanding" at the next barrier asks for nothing but DMA(t+1).
    if (!(ABL & 1) && !idle_wave && grp_b && prev != 0xffffffffu) prune(acc, prev);
    if (issued && (grp_b || (ABL & 4))) issue(nb, dma_tile);
    RARC_STAMP(1)
#else
    if (issued && ((grp_b && !SCAN_DMA_B_INLOOP) || (ABL & 4))) issue(nb, dma_tile);  // group B (it prunes first anyway)
    RARC_STAMP(1)
    // group B prunes the PREVIOUS tile now, while group A (same SIMDs) already streams MFMAs
    if (!(ABL & 1) && !idle_wave && grp_b && prev != 0xffffffffu) prune(acc, prev);
#endif

    // ---- 32 rows x 32 queries per wave: KS chained MFMAs fed by a 6-deep LDS read ring ----
#if SCAN_MMA16
    if ((ABL & 4) || idle_wave) {
#pragma unroll
      for (int j = 0; j < 4; ++j) { acc[j] = (f32x4){0}; asm volatile("" : "+v"(acc[j])); }
      // (an idle wave of group A still owes the tile its DMA pieces: a working wave issues them from inside its MFMA chain)
      if (idle_wave && !(ABL & 4) && !SCAN_DMA_OUTSIDE && issued && (SCAN_DMA_B_INLOOP || !grp_b)) issue(nb, dma_tile);
    } else {
      half8 ra[SCAN_RING16], rb[SCAN_RING16];
      if (ABL & 32) __builtin_amdgcn_s_setprio(1);
      RARC_STAMP(2)
      ScanPrologue16<0, SCAN_RING16>::run(ra, rb, a0, a1);
      ScanSteps16<0, KS2, SCAN_RING16>::run(acc, ra, rb, qf, a0, a1, dma);
      if (ABL & 32) __builtin_amdgcn_s_setprio(0);
      RARC_MFMA_DRAIN4(acc);
      RARC_STAMP(3)
    }
#else
    if ((ABL & 4) || idle_wave) {
      acc = (f32x16){0};
      asm volatile("" : "+v"(acc));
      if (idle_wave && !(ABL & 4) && !SCAN_DMA_OUTSIDE && issued && (SCAN_DMA_B_INLOOP || !grp_b)) issue(nb, dma_tile);
    } else {
      half8 rg[SCAN_RING];
      if (ABL & 32) __builtin_amdgcn_s_setprio(1);
      RARC_STAMP(2)
      ScanPrologue<0, SCAN_RING>::run(rg, a0, a1, a2, a3);
      ScanSteps<0, KS, SCAN_RING>::run(acc, rg, qf, a0, a1, a2, a3, dma);
      if (ABL & 32) __builtin_amdgcn_s_setprio(0);
      RARC_MFMA_DRAIN(acc);
      RARC_STAMP(3)
    }
#endif
    if (!(ABL & 1) && !idle_wave) {  // group A prunes this tile right away; group B defers it to the next iteration
      if (!grp_b) prune(acc, cur);
      else prev = cur;
    }
    if (SCAN_DMA_OUTSIDE && issued && !grp_b && !(ABL & 4)) issue(nb, dma_tile);

    RARC_STAMP(4)
    // ---- threshold refresh issued one iteration ago: its DMAs are older than this iteration's
    // tile DMAs, so "at most DPW outstanding" (or 0 if none were issued) means they have landed ----
    if (pend) {
      if (issued) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DPW) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#if SCAN_MMA16
      thr = fmaxf(thr, rarc_lds_read_f32(L::TLAND + wave * 256 + 4 * (rarc_fresh_lane() & 15)));
      thr1 = fmaxf(thr1, rarc_lds_read_f32(L::TLAND + wave * 256 + 64 + 4 * (rarc_fresh_lane() & 15)));
#else
      thr = fmaxf(thr, rarc_lds_read_f32(L::TLAND + wave * 256 + 4 * rarc_fresh_lane()));
#endif
      if (own_q >= 0) {  // wave 0: turn the owned query's histogram into a threshold
        uint32_t c0, c1, c2, c3;
        rarc_lds_read_u32x4(L::HLAND + 16 * rarc_fresh_lane(), c0, c1, c2, c3);
        const int b = rarc_wave_find_from_top_256(c0, c1, c2, c3, p.kprime);
        if (b >= 2 && rarc_fresh_lane() == 0) {
          // counts only grow, so b (hence t) never decreases; t > lo >= the seed threshold
          float t = rarc_bin_threshold(b, rarc_lds_read_f32(L::BINLO + 4 * own_q),
                                       rarc_lds_read_f32(L::BININV + 4 * own_q));
          t -= rarc_lds_read_f32(L::THOFF + 4 * own_q);   // (from LDS: a global load here would make hipcc drain the DMA queue)
          __hip_atomic_store(&p.thr[own_q], __float_as_uint(t), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        own_q = -1;
      }
      pend = false;
    }

    // ---- threshold refresh events: every tile up to 8, then every 4th ----
    ++it;
    if (!(ABL & 8) && (it <= 8 || (it & 3) == 0)) {
      if (wave == 0) {
        const uint32_t owned = (p.nq + stride - 1 - blockIdx.x) / stride;  // queries ≡ blockIdx (mod grid)
        if (owned > 0) {
          own_q = blockIdx.x + stride * (event % owned);
          // 256 bins x 4 B = one 1-KiB LDS-DMA (sc1: served coherently from L2), consumed next tile
          __builtin_amdgcn_global_load_lds(RARC_GPTR(p.hist + (size_t)own_q * RARC_NB + 4 * rarc_fresh_lane()),
                                           RARC_LPTR(smem + L::HLAND), 16, 0, 16);
        }
        ++event;
      }
      __builtin_amdgcn_global_load_lds(RARC_GPTR(p.thr + wave * 32 + (rarc_fresh_lane() & 31)),
                                       RARC_LPTR(smem + L::TLAND + wave * 256), 4, 0, 16);
      pend = true;
    }

    RARC_STAMP_AT(it - 1u, 5)
#if SCAN_MMA16
    if (buf == 2) {
      buf = 0;
      a0 -= 2 * TILE_BYTES; a1 -= 2 * TILE_BYTES;
    } else {
      ++buf;
      a0 += TILE_BYTES; a1 += TILE_BYTES;
    }
#else
    if (buf == 2) {
      buf = 0;
      a0 -= 2 * TILE_BYTES; a1 -= 2 * TILE_BYTES; a2 -= 2 * TILE_BYTES; a3 -= 2 * TILE_BYTES;
    } else {
      ++buf;
      a0 += TILE_BYTES; a1 += TILE_BYTES; a2 += TILE_BYTES; a3 += TILE_BYTES;
    }
#endif
  }
  if (!(ABL & 1) && !idle_wave && grp_b && prev != 0xffffffffu) prune(acc, prev);
  __syncthreads();
  if ((ABL & 64) && blockIdx.x == 0) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    for (int i = tid; i < 32 * SCAN_WAVES * 8; i += SCAN_WAVES * 64)
      p.dbg[32 * SCAN_WAVES * 8 + i] = __float_as_uint(rarc_lds_read_f32(L::TOTAL + 4 * i));
  }
  {
    const int t2 = wave * 64 + rarc_fresh_lane();
    if (t2 < RARC_MAX_QUERIES) p.cnt2[(size_t)blockIdx.x * RARC_MAX_QUERIES + t2] = s_cnt[t2];
  }
}

// ---- seed pass: score a strided sample of tiles so the scan starts with a real threshold ----
// Without it every workgroup passes everything until the histogram feedback loop closes
// (~4 tiles x 256 workgroups x 32 rows = 32K candidates per query).  grid = (8 query blocks,
// seed_tiles); one block scores one 32-row tile against 32 queries, its 4 waves splitting the
// k range (all fragment loads of a wave are issued at once: one memory round trip), partial
// accumulators summed through LDS.  Rows come straight from global memory (6 MB, L2-shared by
// the 8 query blocks).
// FMT 1: fp8 (e4m3fn) rows with per-row scales — bytes decode exactly to fp16, the score is multiplied
// by the row's scale on the way out.
template <int D, int FMT = 0>
__global__ __launch_bounds__(256) void rarc_seed_kernel(const void* __restrict__ corpus_v,
                                                        const float* __restrict__ rowscale,
                                                        const half_t* __restrict__ q16, uint32_t n_rows,
                                                        uint32_t n_tiles, uint32_t seed_tiles,
                                                        float* __restrict__ seed) {
  constexpr int KS = D / 16, KW = KS / 4;  // k-steps per wave (D multiple of 128 -> KS multiple of 8)
  __shared__ float part[3][16][64];
  __shared__ float tr[32 * 33];  // [32 queries][32 rows + 1]: output transposition
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = lane & 31, h = lane >> 5;
  const uint32_t step = n_tiles / seed_tiles;  // >= 1 (seed_tiles <= n_tiles)
  const uint32_t qidx = blockIdx.x * 32 + row;
  // the block's 32 queries stay in registers (this wave's quarter of K) while it walks its share of the
  // sample tiles: re-reading them per tile was most of this kernel's L2 traffic
  const half_t* qp = q16 + (size_t)qidx * D + 8 * h + 16 * KW * wave;
  half8 bf[KW];
#pragma unroll
  for (int ks = 0; ks < KW; ++ks) bf[ks] = *(const half8*)(qp + 16 * ks);
  // the next tile's fragments are fetched before this tile's partial sums go through LDS (two barriers): the
  // walk was one exposed memory round trip per tile
  auto load_tile = [&](uint32_t ti, half8 (&dst)[KW]) __attribute__((always_inline)) {
    uint32_t arow = ti * step * 32 + row;
    if (arow >= n_rows) arow = n_rows - 1;
    if constexpr (FMT == 1) {
      const uint8_t* ap = (const uint8_t*)corpus_v + (size_t)arow * D + 8 * h + 16 * KW * wave;
#pragma unroll
      for (int ks = 0; ks < KW; ++ks) {
        const uint2 v = *(const uint2*)(ap + 16 * ks);  // 8 fp8 values
        const half2_t p0 = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(v.x, 1.0f, false);
        const half2_t p1 = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(v.x, 1.0f, true);
        const half2_t p2 = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(v.y, 1.0f, false);
        const half2_t p3 = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(v.y, 1.0f, true);
        dst[ks] = (half8){p0.x, p0.y, p1.x, p1.y, p2.x, p2.y, p3.x, p3.y};
      }
    } else {
      const half_t* ap = (const half_t*)corpus_v + (size_t)arow * D + 8 * h + 16 * KW * wave;
#pragma unroll
      for (int ks = 0; ks < KW; ++ks) dst[ks] = *(const half8*)(ap + 16 * ks);
    }
  };
  half8 af[KW], an[KW];
  if (blockIdx.y < seed_tiles) load_tile(blockIdx.y, af);
  for (uint32_t ti = blockIdx.y; ti < seed_tiles; ti += gridDim.y) {
    const uint32_t tile = ti * step;
    const uint32_t tn = ti + gridDim.y < seed_tiles ? ti + gridDim.y : ti;  // (the last trip re-reads its own tile)
    load_tile(tn, an);
    f32x16 acc = {0};
#pragma unroll
    for (int ks = 0; ks < KW; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[ks], bf[ks], acc, 0, 0, 0);
    __syncthreads();  // the previous tile's partial sums have been read
    if (wave > 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) part[wave - 1][r][lane] = acc[r];
    }
    __syncthreads();
    if (wave == 0) {
      // a lane holds 16 scores of ONE query, four bytes each in 16 different cache lines of the seed buffer:
      // transposed through LDS, a query's 32 scores of the tile leave as one 128-byte line
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v = (acc[r] + part[0][r][lane]) + (part[1][r][lane] + part[2][r][lane]);
        const int rr = (r & 3) + 8 * (r >> 2) + 4 * h;
        const bool live = tile * 32 + rr < n_rows;
        if (FMT == 1 && live) v *= rowscale[tile * 32 + rr];
        tr[row * 33 + rr] = live ? v : -INFINITY;
      }
      __builtin_amdgcn_wave_barrier();
      float* out = seed + (size_t)(blockIdx.x * 32) * (RARC_SEED_MAX_TILES * 32) + ti * 32;
#pragma unroll
      for (int t = 0; t < 16; ++t) {  // 2 queries x 32 rows per instruction
        const int ql = 2 * t + (lane >> 5), rr = lane & 31;
        out[(size_t)ql * (RARC_SEED_MAX_TILES * 32) + rr] = tr[ql * 33 + rr];
      }
      __builtin_amdgcn_wave_barrier();
    }
#pragma unroll
    for (int ks = 0; ks < KW; ++ks) af[ks] = an[ks];
  }
}

// One workgroup per query slot.  thr[q] = (a hair below) the k'-th largest seed score — a lower
// bound of the k'-th best score of the whole shard.  Three sweeps over the query's seed scores (up to
// 65536 floats, L2 resident): min/max, a 2048-bin histogram of the order-preserving keys over
// [min, max], then the few keys of the bin holding the k'-th largest are ranked directly.  Also sets
// the query's histogram window and clears its histogram.
__global__ __launch_bounds__(1024) void rarc_seed_thr_kernel(const float* seed, uint32_t seed_rows, uint32_t kprime,
                                                            uint32_t nq, float bin_lo_dflt, float bin_hi_dflt,
                                                            const float* sub_a, const float* sub_b,
                                                            uint32_t* thr, float* binlo, float* binscale,
                                                            float* bininv, uint32_t* flags, uint32_t* hist,
                                                            const float* floor, const float* floor_eps) {
  constexpr int LIST = 4096;
  __shared__ uint32_t s_hist[2048];
  __shared__ uint32_t s_list[LIST];
  __shared__ uint32_t s_bin, s_need, s_max, s_min, s_nlist, s_key;
  const uint32_t q = blockIdx.x, tid = threadIdx.x;
  for (uint32_t i = tid; i < RARC_NB; i += blockDim.x) hist[q * RARC_NB + i] = 0;
  if (q == 0 && tid < 64) flags[tid] = 0;
  float t = -INFINITY, mx = -INFINITY;
  if (q < nq && seed_rows > 0) {  // block-uniform
    const float* sq = seed + (size_t)q * (RARC_SEED_MAX_TILES * 32);
    auto keys4 = [&](uint32_t j4, uint32_t (&k4)[4]) {  // 16-byte loads: the sweeps were latency-bound at 4 bytes
      const float4 v4 = ((const float4*)sq)[j4];
      const float v[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) k4[e] = rarc_ordkey(v[e] == v[e] ? v[e] : -INFINITY);  // NaN never becomes a threshold
    };
    for (uint32_t i = tid; i < 2048; i += blockDim.x) s_hist[i] = 0;
    if (tid == 0) { s_max = 0; s_min = 0xffffffffu; s_nlist = 0; s_key = 0; s_bin = 0; s_need = kprime; }
    __syncthreads();
    {  // block min / max of the keys: wave reduction first, one LDS atomic per wave
      uint32_t lmax = 0, lmin = 0xffffffffu;
      for (uint32_t j4 = tid; j4 < seed_rows / 4; j4 += blockDim.x) {  // (seed_rows is a multiple of 32)
        uint32_t k4[4];
        keys4(j4, k4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          lmax = k4[e] > lmax ? k4[e] : lmax;
          lmin = (k4[e] > 0x007fffffu && k4[e] < lmin) ? k4[e] : lmin;  // skip -inf padding
        }
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const uint32_t a = __shfl_xor(lmax, o, 64), b2 = __shfl_xor(lmin, o, 64);
        lmax = a > lmax ? a : lmax;
        lmin = b2 < lmin ? b2 : lmin;
      }
      if ((tid & 63) == 0) { atomicMax(&s_max, lmax); atomicMin(&s_min, lmin); }
    }
    __syncthreads();
    // 2048 linear bins over [min, max] of the (monotone) keys: spreads the values, so the LDS
    // atomics do not pile up on a few addresses the way the keys' top bits would
    const uint32_t kmin = s_min < s_max ? s_min : s_max, krange = s_max - kmin;
    const float kscale = krange ? 2047.f / (float)krange : 0.f;
    auto bin_of_key = [&](uint32_t k) {
      uint32_t b2 = k > kmin ? (uint32_t)((float)(k - kmin) * kscale) : 0u;  // monotone in key
      return b2 > 2047u ? 2047u : b2;
    };
    for (uint32_t j4 = tid; j4 < seed_rows / 4; j4 += blockDim.x) {
      uint32_t k4[4];
      keys4(j4, k4);
#pragma unroll
      for (int e = 0; e < 4; ++e) atomicAdd(&s_hist[bin_of_key(k4[e])], 1u);
    }
    __syncthreads();
    if (tid < 64) {  // wave 0: bin holding the k'-th largest, count above it
      uint32_t above = 0;
      const int b = rarc_wave_find_from_top(s_hist, 2048, kprime, &above);
      if (tid == 0) { s_bin = b < 0 ? 0u : (uint32_t)b; s_need = kprime - above; }
    }
    __syncthreads();
    const uint32_t bin = s_bin, need = s_need;
    for (uint32_t j4 = tid; j4 < seed_rows / 4; j4 += blockDim.x) {
      uint32_t k4[4];
      keys4(j4, k4);
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (bin_of_key(k4[e]) == bin) {
          const uint32_t pos = atomicAdd(&s_nlist, 1u);
          if (pos < (uint32_t)LIST) s_list[pos] = k4[e];
        }
    }
    __syncthreads();
    const uint32_t nl_all = s_nlist, nl = nl_all < (uint32_t)LIST ? nl_all : (uint32_t)LIST;
    if (nl_all <= (uint32_t)LIST) {
      for (uint32_t i = tid; i < nl; i += blockDim.x) {  // the need-th largest of the bin (ties share a value)
        const uint32_t mine = s_list[i];
        uint32_t gt = 0, ge = 0;
        for (uint32_t j = 0; j < nl; ++j) { gt += s_list[j] > mine; ge += s_list[j] >= mine; }
        if (gt < need && need <= ge) s_key = mine;
      }
    } else if (tid == 0) {
      // (a bin too crowded to rank — near-identical scores: its lower edge is still a valid bound)
      // (minus a margin covering the fp32 rounding of the bin map: 24-bit floats of 32-bit key offsets)
      const uint32_t edge = (uint32_t)((float)bin / (kscale > 0.f ? kscale : 1.f));
      const uint32_t margin = (krange >> 10) + 1024u;
      s_key = edge > margin ? kmin + (edge - margin) : kmin;
    }
    __syncthreads();
    if (kprime <= seed_rows) {
      t = rarc_unordkey(s_key);
      t -= fabsf(t) * 1e-5f + 1e-7f;  // the seed's 4-way k split rounds differently from the scan's chain
    }
    mx = rarc_unordkey(s_max);
  }
  if (tid == 0) {
    float lo = bin_lo_dflt, hi = bin_hi_dflt;
    // a re-run search knows a lower bound of the k-th best canonical score from its first attempt (the k-th entry of
    // the incomplete answer): far better than anything a sample can give
    if (floor && q < nq) {
      // (floor_eps: the scorer's own error bound when no sub_a / sub_b is taken off below — the fp16 scan)
      const float f = floor[q] - (floor_eps ? floor_eps[q] * 1.0001f : 0.f);
      if (f > t) t = f - (fabsf(f) * 1e-6f + 1e-30f);
    }
    if (q >= nq) t = INFINITY;  // padding queries never pass
    else if (t > -INFINITY) {
      // int8 prefilter: the threshold lives in approx-score space, below the sample statistic by the
      // error bounds of both scorers (scan_q8.hip); the histogram window starts there
      float ts = t;
      if (sub_a) ts -= sub_a[q] * 1.0001f;
      if (sub_b) ts -= sub_b[q] * 1.0001f;
      {
        float w = mx > t ? 4.f * (mx - t) : 0.f;  // (a floor above the sample's best: the width floor below sizes the window)
        // (k = 1: the sample's k-th best IS its maximum and 4·(mx − t) collapses — every later, better row would
        //  fall into the open top bin, the owner could never raise the threshold and the finalize would start
        //  from thousands of rows; a quarter of the way to the score bound is what the rule gives at k = 100)
        const float wfloor = 0.25f * (bin_hi_dflt - t);
        if (w < wfloor) w = wfloor;
        const float wmin = 1e-3f * fabsf(t) + 1e-20f;  // keeps fp32 rounding of the bin map below one bin
        if (w < wmin) w = wmin;
        lo = ts;
        hi = t + w;
        // (no score exceeds the caller's bound: a window reaching past it only makes the bins coarser)
        if (hi > bin_hi_dflt && bin_hi_dflt > t + wmin) hi = bin_hi_dflt;
      }
      t = ts;
    }
    thr[q] = __float_as_uint(t);
    binlo[q] = lo;
    binscale[q] = (float)RARC_NB / (hi - lo);
    bininv[q] = (hi - lo) / (float)RARC_NB;
  }
}

template <int D, int FMT = 0>
static int launch_seed(const ScanParams& p, uint32_t seed_tiles, float* seed, hipStream_t s,
                       const float* rowscale = nullptr) {
  // 8 query blocks x up to 128 tile walkers (1024 workgroups of 4 waves fill the chip); a walker takes at least
  // four tiles, so that its 48 KB of query fragments are fetched once per four tiles and the next tile's loads run
  // behind the current tile's MFMAs (128-tile sample of a small shard: 128 walkers 23.4 us, 64: 15.1, 32: 13.5, 16: 20.8)
  static const uint32_t walk_env = getenv("RARC_SEED_WALKERS") ? (uint32_t)atoi(getenv("RARC_SEED_WALKERS")) : 0u;  // (experiments)
  uint32_t walkers = walk_env ? walk_env : (seed_tiles + 3u) / 4u;
  if (walkers > 128u) walkers = 128u;
  if (walkers > seed_tiles) walkers = seed_tiles;
  if (walkers < 1u) walkers = 1u;
  hipLaunchKernelGGL((rarc_seed_kernel<D, FMT>), dim3(RARC_MAX_QUERIES / 32, walkers), dim3(256), 0, s,
                     (const void*)p.corpus, rowscale, p.q16, p.n_rows, p.n_tiles, seed_tiles, seed);
  RARC_HIP_CHECK(hipGetLastError());
  return RARC_OK;
}

bool rarc_prof_next(hipEvent_t* start, hipEvent_t* stop);  // rarc_api.hip

template <int D>
static int launch_scan(const ScanParams& p, int grid, hipStream_t s) {
  constexpr size_t lds = ScanLds<D>::TOTAL;
  static RarcPerDevice attr_done;
  if (size_t& done = attr_done.cur(); !done) {
    RARC_HIP_CHECK(hipFuncSetAttribute((const void*)rarc_scan_f16_kernel<D>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    done = 1;
  }
  hipEvent_t e0, e1;
  const bool prof = rarc_prof_next(&e0, &e1);
  if (prof) RARC_HIP_CHECK(hipEventRecord(e0, s));
  hipLaunchKernelGGL(rarc_scan_f16_kernel<D>, dim3(grid), dim3(SCAN_WAVES * 64), lds, s, p);
  RARC_HIP_CHECK(hipGetLastError());
  if (prof) RARC_HIP_CHECK(hipEventRecord(e1, s));
  return RARC_OK;
}

// Seed pass alone (used by the int8 scan, scan_q8.hip): thr[q] = sample statistic − sub_a[q] − sub_b[q].
int rarc_seed_launch(const void* corpus, const float* rowscale, int fmt, int64_t n_rows, int d_pad,
                     const uint16_t* q16, int nq, int kprime, float bin_lo, float bin_hi, const float* sub_a,
                     const float* sub_b, const RarcWs& ws, hipStream_t s, int64_t rows_covered, const float* floor) {
  ScanParams p;
  p.corpus = (const half_t*)corpus;
  p.q16 = (const half_t*)q16;
  p.n_rows = (uint32_t)n_rows;
  p.n_tiles = (uint32_t)((n_rows + 31) / 32);
  // sample size grows with the shard: 1/1024 of the tiles, at least 4096 and at most 65536 rows.  The
  // scan starts from the k'-th best of the sample, and with the int8 error margin taken off it a small
  // sample lets several per cent of a large shard through before feedback takes over (100M rows: 42K
  // candidates per query and a 3 % longer scan with 4096 sample rows, 33K with 65536) — but the sample
  // costs the same 8 µs per 1024 rows whatever the shard size, which a 3.7 ms scan of a 12.5M-row shard
  // (100M rows over 8 GPUs) cannot afford: it gets 12K rows.
  // (rows_covered: the part of the shard that will run under this threshold — the first launch of a split scan)
  uint32_t seed_tiles = (uint32_t)(((rows_covered > 0 ? rows_covered : n_rows) + 31) / 32) / 1024;
  if (seed_tiles < (uint32_t)RARC_SEED_TILES) seed_tiles = (uint32_t)RARC_SEED_TILES;
  if (seed_tiles > (uint32_t)RARC_SEED_MAX_TILES) seed_tiles = (uint32_t)RARC_SEED_MAX_TILES;
  if (seed_tiles > p.n_tiles) seed_tiles = p.n_tiles;
  int rc = RARC_OK;
  if (seed_tiles > 0) {
    if (fmt == 1) {
      switch (d_pad) {
        case 256: rc = launch_seed<256, 1>(p, seed_tiles, ws.seed, s, rowscale); break;
        case 512: rc = launch_seed<512, 1>(p, seed_tiles, ws.seed, s, rowscale); break;
        case 768: rc = launch_seed<768, 1>(p, seed_tiles, ws.seed, s, rowscale); break;
        case 1024: rc = launch_seed<1024, 1>(p, seed_tiles, ws.seed, s, rowscale); break;
        default:
          rarc_set_error("rarc_seed (fp8): padded dim %d unsupported (multiple of 256, <= 1024)", d_pad);
          rc = RARC_E_UNSUPPORTED;
      }
    } else
    switch (d_pad) {
#define SEED_CASE(DD) case DD: rc = launch_seed<DD>(p, seed_tiles, ws.seed, s); break;
      SEED_CASE(128) SEED_CASE(256) SEED_CASE(384) SEED_CASE(512) SEED_CASE(640) SEED_CASE(768) SEED_CASE(896)
      SEED_CASE(1024)
#undef SEED_CASE
      default:
        rarc_set_error("rarc_seed: padded dim %d unsupported (multiple of 128, <= 1024)", d_pad);
        rc = RARC_E_UNSUPPORTED;
    }
    if (rc) return rc;
  }
  hipLaunchKernelGGL(rarc_seed_thr_kernel, dim3(RARC_MAX_QUERIES), dim3(1024), 0, s, ws.seed, seed_tiles * 32,
                     (uint32_t)kprime, (uint32_t)nq, bin_lo, bin_hi, sub_a, sub_b, (uint32_t*)ws.thr, ws.binlo,
                     ws.binscale, ws.bininv, ws.flags, ws.hist, floor, (const float*)nullptr);
  RARC_HIP_CHECK(hipGetLastError());
  return RARC_OK;
}

// Host entry used by rarc_api.hip.  *grid_out = workgroups launched (owners of candidate segments).
int rarc_scan_f16_launch(const uint16_t* corpus, int64_t n_rows, int d_pad, const uint16_t* q16, int nq,
                         int kprime, float bin_lo, float bin_hi, const RarcWs& ws, int cap, int* grid_out,
                         hipStream_t s, const float* floor, const float* floor_eps) {
  ScanParams p;
  p.corpus = (const half_t*)corpus;
  p.q16 = (const half_t*)q16;
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
  p.dbg = nullptr;
  p.margin = nullptr;
  p.margin_scale = 0.f;

  const uint32_t seed_tiles = p.n_tiles < (uint32_t)RARC_SEED_TILES ? p.n_tiles : (uint32_t)RARC_SEED_TILES;
#define RARC_DISPATCH_D(CALL)                                                                         \
  switch (d_pad) {                                                                                    \
    case 128: rc = CALL(128); break;                                                                  \
    case 256: rc = CALL(256); break;                                                                  \
    case 384: rc = CALL(384); break;                                                                  \
    case 512: rc = CALL(512); break;                                                                  \
    case 640: rc = CALL(640); break;                                                                  \
    case 768: rc = CALL(768); break;                                                                  \
    default:                                                                                          \
      rarc_set_error("rarc_scan_f16: padded dim %d unsupported (multiple of 128, <= 768)", d_pad);    \
      rc = RARC_E_UNSUPPORTED;                                                                        \
  }
  int rc = RARC_OK;
  if (seed_tiles > 0) {
#define SEED_CALL(DD) launch_seed<DD>(p, seed_tiles, ws.seed, s)
    RARC_DISPATCH_D(SEED_CALL)
    if (rc) return rc;
  }
  hipLaunchKernelGGL(rarc_seed_thr_kernel, dim3(RARC_MAX_QUERIES), dim3(1024), 0, s, ws.seed, seed_tiles * 32,
                     (uint32_t)kprime, (uint32_t)nq, bin_lo, bin_hi, (const float*)nullptr, (const float*)nullptr,
                     (uint32_t*)ws.thr, ws.binlo, ws.binscale, ws.bininv, ws.flags, ws.hist, floor, floor_eps);
  RARC_HIP_CHECK(hipGetLastError());

  int dev = 0, cus = 256;
  RARC_HIP_CHECK(hipGetDevice(&dev));
  RARC_HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  int grid = cus < RARC_MAX_WG ? cus : RARC_MAX_WG;
  // RARC_SCAN_WGS=n: fewer persistent workgroups than CUs (measurement: do the small kernels of a neighbouring batch on
  // a second stream find room on the CUs this leaves free?  DESIGN 8 / 11)
  if (const char* e = getenv("RARC_SCAN_WGS")) { const int g = atoi(e); if (g >= 8 && g < grid) grid = g; }
  if ((uint32_t)grid > p.n_tiles) grid = (int)p.n_tiles;
  *grid_out = grid;
  if (p.n_tiles == 0) return RARC_OK;
  RARC_REQUIRE(rarc_gate_scan(s) == 0, RARC_E_HIP, "rarc_scan_f16: hipStreamWaitEvent on the gate event failed");
#define SCAN_CALL(DD) launch_scan<DD>(p, grid, s)
  RARC_DISPATCH_D(SCAN_CALL)
  return rc;
}


// The fp16 MFMA scan as the FIRST STAGE of the hybrid small-shard search (scan_q8.hip: rarc_scan_q8_launch): tiles
// [0, stage_tiles) of the shard, thresholds / histogram windows as the seed pass left them, candidates into the same
// private segments the int8 stage continues (resume), published thresholds under the rigorous 2·eps16 margin.
int rarc_scan_f16_stage_launch(const uint16_t* corpus, int64_t n_rows, uint32_t stage_tiles, int d_pad, const uint16_t* q16,
                               const float* eps16, int nq, int kprime, const RarcWs& ws, int cap, int grid, hipStream_t s) {
  ScanParams p;
  p.corpus = (const half_t*)corpus;
  p.q16 = (const half_t*)q16;
  p.n_rows = (uint32_t)n_rows;
  p.n_tiles = stage_tiles;
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
  p.dbg = nullptr;
  p.margin = eps16;
  p.margin_scale = 2.0002f;
  int rc = RARC_OK;
  RARC_DISPATCH_D(SCAN_CALL)
  return rc;
}
