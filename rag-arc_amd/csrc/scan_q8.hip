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
//     l&31.  (fp8 and int8-shadow rows: D/64 x 4 v_mfma_i32_16x16x64_i8 and a regrouping of the scores, see M16 in the kernel.)  Prune: integer max, one compare against the lane's threshold; survivors are appended to
//     the workgroup's private segment of the query's candidate list and bump the query's histogram.
//   * thresholds: thr[q] is a lower bound of (k-th best approx score) − 2·eps8[q], or of
//     (a lower bound of the k-th best canonical score) − eps8[q]; both mean "a row below it cannot be
//     in the top-k".  The owner workgroup of a query turns its histogram into a higher threshold.
//
// Algorithmic HBM bytes per launch: n_rows × D × 2 (+ 8 bytes per tile of metadata).
#include "rarc_common.h"
#include <stdio.h>

struct ScanQ8Params {
  const uint4* corpus;   // fp16 (FMT 0), fp8 (FMT 1) or int8-shadow (FMT 2) rows [ceil32(n_rows)][D], 16-byte chunks
  const float* tmeta;    // per tile: (packed fp16 scale | fp16 R_t, 1/scale) [+ 32 row multipliers for fp8]; qmeta + RARC_QMETA_HDR
  const float* hq;       // [256]  ||q8/s_q|| (rounded down): tile t's threshold is raised by hq·(R − R_t)
  const int8_t* q8;      // [256][D]
  const float* qinv;     // [256]  1 / s_q
  const float* eps8;     // [256]
  uint32_t n_rows;
  uint32_t n_tiles;       // END of this launch's tile range (the shard's tile count unless the scan is split)
  uint32_t t_begin;       // first tile of this launch's range
  uint32_t resume;        // 1: second launch of a split scan — the private segments continue from cnt2
  float hot_margin;       // survivors within hot_margin·eps8 of the threshold stay out of the global histogram
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
  unsigned long long* dbg;  // tools/scan_q8_bench: {shader cycles, 100 MHz ticks} of workgroup 0; else null
};

#ifndef Q8_M16_FP16_ROWS
#define Q8_M16_FP16_ROWS 0  // 1: fp16 rows on the 16x16x64 form as well (A/B builds)
#endif
// build-time knobs of the instantiations without the ping-pong (D > 768; A/B builds: tools/build_variant_any.sh scan_q8 <name> -D...)
#ifndef Q8_BIG_PF
#define Q8_BIG_PF 1     // k steps of LDS fragments read ahead of the 16x16x64 MFMAs
#endif
#ifndef Q8_BIG_NG
#define Q8_BIG_NG 2     // fetch groups (tiles of global loads in flight per thread), fp8 / shadow rows
#endif
#ifndef Q8_EB
#define Q8_EB 0         // 1: the iteration's barrier right behind the matrix phase (see EB in the kernel)
#endif
constexpr int Q8_WAVES = 8;
constexpr int Q8_THREADS = Q8_WAVES * 64;

constexpr int Q8_WSTAGE = 768;            // staged (query, key) survivors per wave between flushes
constexpr int Q8_STAGE = 8 * Q8_WSTAGE;   // ... per workgroup
constexpr int Q8_TB_SLOTS = 16;           // passing lanes per wave and tile handled by the transposed survivor walk
constexpr int Q8_TB_STRIDE = 80;          // 16 int32 scores | threshold | scale | query + row-half

template <bool B>
struct Q8Flag { static constexpr bool value = B; };
constexpr int Q8_DEEP_D = 384;  // rows up to this many dimensions run with four fetch groups in flight (see NG)

// Row stride of the int8 tile.  32x32x32 form (fp16 rows): D + 16 — the 16-lane groups of the A-fragment ds_read_b128 hit
// 16 distinct bank groups.  16x16x64 form (fp8 / shadow rows, round 3): lane l reads row l & 15, 16-byte k quarter l >> 4;
// with D + 16 the hardware's ds_read_b128 lane groups ({0-3, 12-15, 20-27}, ...) put two lanes on one bank group in EVERY
// group — each fragment read took 8 LDS cycles instead of 4 (round 4 PMC, 100M x 1024 fp8: SQ_LDS_BANK_CONFLICT 43 % of
// SQ_LDS_IDX_ACTIVE, the LDS busy 59 % of the kernel).  D + 32 is conflict-free for that form: with D a multiple of 256 the
// bank group of a lane is (2·row + quarter) mod 16, a bijection on each of the four lane groups (checked exhaustively).
template <int D, bool M16V = false>
struct ScanQ8Lds {
  static constexpr int RS = D + (M16V ? 32 : 16);   // row stride of the int8 tile (bytes)
  static constexpr int TILE = 32 * RS;      // one int8 tile
  static constexpr int CNT = 2 * TILE;      // uint32 [256] slot counters of the private segments
  static constexpr int BINLO = CNT + 1024;
  static constexpr int BINSCALE = BINLO + 1024;
  static constexpr int BININV = BINSCALE + 1024;
  static constexpr int EPS8 = BININV + 1024;
  static constexpr int HLAND = EPS8 + 1024;     // uint32 [256]: the owned query's histogram, as last fetched
  static constexpr int SKEY = HLAND + 1024;     // uint64 [Q8_STAGE]
  static constexpr int SQ = SKEY + 8 * Q8_STAGE;  // uint8 [Q8_STAGE]
  static constexpr int TB = (SQ + Q8_STAGE + 15) & ~15;  // per wave: Q8_TB_SLOTS x 80 B transposition slots (prune)
  static constexpr int TOTAL = TB + Q8_WAVES * Q8_TB_SLOTS * Q8_TB_STRIDE;
};

// the tile's metadata word: low half = fp16 scale, high half = fp16 R_t (rounded up; +inf when out of range)
__device__ __forceinline__ half_t q8_tile_scale(float w) { return __builtin_bit_cast(half_t, (uint16_t)__float_as_uint(w)); }
__device__ __forceinline__ float q8_tile_rt(float w) { return (float)__builtin_bit_cast(half_t, (uint16_t)(__float_as_uint(w) >> 16)); }

// barrier that orders LDS traffic only: global loads stay in flight across it (a __syncthreads()
// would drain vmcnt and with it the two-tile prefetch)
__device__ __forceinline__ void q8_lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// (ablation builds: a value that depends on every accumulator element — round 3's "no pruning" build looked at element 0
//  only, and for the 16x16x64 form hipcc then dropped the three accumulator blocks nobody read: 32 MFMAs per tile, not 128)
typedef int q8_i32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ int q8_all16(const q8_i32x16& a) {
  int x = a[0];
#pragma unroll
  for (int i = 1; i < 16; ++i) x ^= a[i];
  return x;
}

// ABL (tools/scan_q8_bench): 1 = no pruning, 2 = no global loads after the prologue, 4 = no MFMA,
// 8 = no threshold refresh, 16 = no conversion, 32768 = fp8: converted but not written to LDS, 32 = prune fast path only, 64 = never flush,
// 256 = no ping-pong between the wave groups, 512 = barrier at the end of the iteration where EB would put it behind the matrix phase, 1024 = s_memtime timeline of workgroup 0 into p.dbg,
// 131072 = fp8 / shadow form: only the first two k steps' A fragments are read from LDS (the MFMAs reuse them: what the other 7/8 of the ds_read_b128 cost),
// 65536 = waves 4-7 issue no MFMAs (fp8 / shadow form: half the matrix work per CU, everything else unchanged — round 5's overlap question),
// 16384 = every survivor updates the histogram, 4096 = survivors walked per lane (no LDS transposition), 8192 = parked scores walked at once (no batching), 2048 = fast path carries the position of the best score along (the earlier form; 0.5 % slower at 100M rows)
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
template <int D, int FMT = 0, int ABL = 0>
__global__ __launch_bounds__(Q8_THREADS) void rarc_scan_q8_kernel(const ScanQ8Params p) {
  static_assert(D % 128 == 0 && D >= 128 && D <= 1024, "D must be a multiple of 128, <= 1024");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using L = ScanQ8Lds<D, (FMT != 0) || Q8_M16_FP16_ROWS>;
  constexpr int KS = D / 32;          // MFMA k-steps
  static_assert(FMT == 0 || D % 256 == 0, "fp8 rows are padded to a multiple of 256");
  constexpr int EPC = FMT ? 16 : 8;   // values per 16-byte chunk (fp8 / int8 : fp16)
  constexpr int CPR = D / EPC;        // 16-byte chunks per row
  constexpr int CPT = 32 * CPR / Q8_THREADS;  // chunks per thread per tile
  constexpr int MSTRIDE = FMT == 1 ? RARC_QMETA_F8_STRIDE : RARC_QMETA_STRIDE;  // floats of metadata per tile
  constexpr int TCH = 32 * CPR;       // chunks per tile

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int row = lane & 31, h = lane >> 5;
  // Round 3, fp8 and int8-shadow rows (M16): the matrix work is v_mfma_i32_16x16x64_i8 on a 2 x 2 grid of 16 rows x 16 queries
  // per wave instead of one chain of v_mfma_i32_32x32x32_i8.  Per MAC the smaller shape moves half the accumulator bytes for
  // twice the (one-byte) operand bytes — 0.25 B instead of 0.31 B of register traffic — and these kernels sit at the board's
  // power limit with the matrix pipe as their bound: same box, alternated, 100M rows: fp8 x 1024 30.4-30.7 -> 29.0-29.1 ms,
  // int8 shadow x 768 22.1 -> 21.3-21.4 ms (profiles/r03_q8_mma16.txt).  fp16 rows keep the 32x32x32 chain: their scan is
  // bound by the row fetch and the pruning, the regrouping below costs what the cooler MFMAs give back (27.2-27.3 vs 26.6-26.7 ms).
  // The 16x16 MFMAs leave a lane with 2 x 4 scores of EACH of the queries q_lo and q_lo + 16; eight v_permlane32_swap
  // (lane l <-> l + 32) regroup them so that a lane ends, as before, with 16 scores of ONE query — lanes 0-31 serve q_lo, lanes
  // 32-63 q_lo + 16 — and everything after the MFMAs keeps one query's state per lane (two cost the registers that spilled, or
  // LDS round trips in the pruning fast path: both were built and measured slower).
  constexpr bool M16 = (FMT != 0) || Q8_M16_FP16_ROWS;
  const uint32_t q_lo = wave * 32 + (lane & 15);      // M16: the MFMA's two query blocks are q_lo (b = 0) and q_lo + 16 (b = 1)
  const uint32_t qidx = M16 ? q_lo + 16 * (lane >> 5) : wave * 32 + row;  // this lane's query (M16: once the scores are regrouped)
  const int rq = lane >> 4;                           // M16, MFMA layout: the lane's rows of a 16-row block are 4 rq .. 4 rq + 3
  const int rql = rq & 1;                             // M16, regrouped: element 8 g + r is row 16 (r >> 2) + 4 (rql + 2 g) + (r & 3)
  // position of accumulator element r inside the tile, minus the lane's first row (row0 below)
  auto row_of = [&](uint32_t r) -> uint32_t { return M16 ? (r & 3) + 16 * ((r >> 2) & 1) + 8 * (r >> 3) : (r & 3) + 8 * (r >> 2); };
  const unsigned long long dbg_c0 = p.dbg ? __builtin_amdgcn_s_memtime() : 0ull;
  const unsigned long long dbg_r0 = p.dbg ? __builtin_amdgcn_s_memrealtime() : 0ull;

  uint32_t* s_cnt = (uint32_t*)(smem + L::CNT);
  float* s_binlo = (float*)(smem + L::BINLO);
  float* s_binscale = (float*)(smem + L::BINSCALE);
  float* s_bininv = (float*)(smem + L::BININV);
  float* s_eps8 = (float*)(smem + L::EPS8);
  uint32_t* s_hland = (uint32_t*)(smem + L::HLAND);
  uint64_t* s_skey = (uint64_t*)(smem + L::SKEY);
  uint8_t* s_sq = (uint8_t*)(smem + L::SQ);
  if (tid < RARC_MAX_QUERIES) {
    s_cnt[tid] = p.resume ? p.cnt2[(size_t)blockIdx.x * RARC_MAX_QUERIES + tid] : 0u;
    s_binlo[tid] = p.binlo[tid];
    s_binscale[tid] = p.binscale[tid];
    s_bininv[tid] = p.bininv[tid];
    s_eps8[tid] = p.eps8[tid];
    s_hland[tid] = 0;
  }

  // resident query fragments (B operands).  32x32x32: lane holds Q8[qidx][32 ks + 16 h .. +16) in qf[ks];
  // M16: Q8[q_lo + 16 b][64 s + 16 rq .. +16) in qf[2 s + b]
  constexpr int KS2 = D / 64;
  i32x4 qf[KS];
  if constexpr (M16) {
    const int8_t* qp = p.q8 + (size_t)q_lo * D + 16 * rq;
#pragma unroll
    for (int ks = 0; ks < KS2; ++ks) {
      qf[2 * ks] = *(const i32x4*)(qp + 64 * ks);
      qf[2 * ks + 1] = *(const i32x4*)(qp + 16 * D + 64 * ks);
    }
  } else {
    const int8_t* qp = p.q8 + (size_t)qidx * D + 16 * h;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[ks] = *(const i32x4*)(qp + 32 * ks);
  }
  const float my_qinv = p.qinv[qidx];
  const float my_hq = p.hq[qidx];
  const float r_max = p.tmeta[0 - RARC_QMETA_HDR];  // R = max R_t (qmeta[0]), what eps8 was built on
  float thr = __uint_as_float(p.thr[qidx]);  // seed threshold; +inf for padding queries
  // everything fetched so far has landed before the first tile load is issued: the compiler's wait
  // counters then never tie a query fragment to the (deliberately long-lived) tile prefetches
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)

  // this thread's chunks of a tile: c = j*512 + tid -> LDS byte offset of its 8 int8 values
  // (kept in registers up to D = 768; recomputed per use beyond that, where registers are short)
  constexpr bool WOFF_REGS = (D <= 768);
  auto woff_of = [&](int j) {
    const uint32_t c = j * Q8_THREADS + tid;
    return (c / CPR) * L::RS + (c % CPR) * EPC;
  };
  uint32_t woff[WOFF_REGS ? CPT : 1];
  if constexpr (WOFF_REGS) {
#pragma unroll
    for (int j = 0; j < CPT; ++j) woff[j] = woff_of(j);
  }
  // A fragment: of k-step ks at + 32 ks; M16: of k step s, row block rb at + 64 s + 16 rb RS (a 16-lane group reads 16 rows: RS = D + 16
  // puts them in 16 distinct bank groups)
  const uint32_t aoff = M16 ? (lane & 15) * L::RS + 16 * rq : row * L::RS + 16 * h;

  const uint32_t t0 = p.t_begin + blockIdx.x, stride = gridDim.x;
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
    float mul[FMT == 1 ? CPT : 1];  // fp8: the multiplier of each chunk's row (row scale x tile scale)
    float2 meta;   // (fp16 scale | fp16 R_t in one word, 1/scale)
    uint32_t thr;
    uint32_t hw;
  };
  auto clamp_tile = [&](uint32_t t) { return t < p.n_tiles ? t : 0u; };  // past the end: tile 0 (an L2 hit)
  // (small items first: if the register allocator decides to move one of these long-lived values
  // while it is still in flight, the wait it needs then is for the OLDEST entries of the newest group,
  // i.e. no more than the wait for the group about to be consumed anyway)
  // `refresh`: also re-read the lane's published threshold and the owner's histogram word.  With four fetch groups
  // (narrow rows) only one group in four does: at D = 384 the three small loads were half of a tile's vector-memory
  // instructions, and a CU issues one wave-instruction per ~45 cycles whatever its size (48 per tile = the tile period).
  auto fetch = [&](Fetch& f, uint32_t tile, auto refresh) {
    f.meta = *(const float2*)(p.tmeta + (size_t)tile * MSTRIDE);
    if constexpr (FMT == 1) {
#pragma unroll
      for (int j = 0; j < CPT; ++j)
        f.mul[j] = p.tmeta[(size_t)tile * MSTRIDE + 2 + (j * Q8_THREADS + tid) / CPR];
    }
    if constexpr (decltype(refresh)::value) {
      f.thr = __hip_atomic_load(&p.thr[qidx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      f.hw = __hip_atomic_load(hword, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __builtin_amdgcn_sched_barrier(0);
    const uint4* src = p.corpus + (size_t)tile * TCH + tid;
#pragma unroll
    for (int j = 0; j < CPT; ++j) f.c[j] = src[j * Q8_THREADS];
  };
  uint32_t opaque_zero = 0;
  asm volatile("" : "+v"(opaque_zero));  // a zero the compiler cannot fold (FMT 2, see convert_chunk)
  // chunk j of a fetch group -> int8 -> its place in the LDS tile `dst`
  auto convert_chunk = [&](const Fetch& f, int j, half_t s, char* dst) {
    char* at = dst + (WOFF_REGS ? woff[WOFF_REGS ? j : 0] : woff_of(j));
    if constexpr (FMT == 2) {
      // the shadow image already holds what the conversion would produce.  (The bytes pass through one
      // VALU op on purpose: stored to LDS straight from the load's destination registers, the register
      // allocator parks each load in a scratch tuple and copies it home right after issuing it — which
      // means waiting for it on the spot.)
      const uint4 v = f.c[j];
      *(uint4*)at = make_uint4(v.x ^ opaque_zero, v.y ^ opaque_zero, v.z ^ opaque_zero, v.w ^ opaque_zero);
    } else if constexpr (FMT == 1) {
      if (ABL & 16) {            // (ablation: the raw bytes, no conversion arithmetic)
        const uint4 v = f.c[j];
        *(uint4*)at = make_uint4(v.x ^ opaque_zero, v.y ^ opaque_zero, v.z ^ opaque_zero, v.w ^ opaque_zero);
      } else {
        const uint4 v = rarc_quant8_chunk_f8(f.c[j], (half_t)f.mul[j]);
        if (ABL & 32768) asm volatile("" ::"v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w));   // (ablation: converted, not written to LDS)
        else *(uint4*)at = v;
      }
    } else {
      uint2 o;
      if (ABL & 16) { o.x = f.c[j].x ^ f.c[j].z; o.y = f.c[j].y ^ f.c[j].w; }
      else o = rarc_quant8_chunk(f.c[j], s);
      *(uint2*)at = o;
    }
  };
  auto convert_tile = [&](const Fetch& f, int buf) {
    const half_t s = q8_tile_scale(f.meta.x);
    char* dst = smem + buf * L::TILE;
#pragma unroll
    for (int j = 0; j < CPT; ++j) convert_chunk(f, j, s, dst);
  };

  // 32 rows x 32 queries per wave: D/32 chained int8 MFMAs, A fragments read a few steps ahead —
  // and, in the shadow of those MFMAs (the matrix pipe takes 32 cycles per instruction, the wave is
  // free in between), the conversion of the NEXT tile's chunks (fetched two iterations ago) into the
  // other LDS buffer.  Unconditional: past the end of the shard the chunks are the tile-0 dummy.
  // (Two accumulators, or running waves w / w+4 of a SIMD in opposite phase order, changed cycles per
  // tile by a few per cent and the clock the other way: the kernel runs at its power limit.)
  auto mfma_convert = [&](int buf, const Fetch& nx) __attribute__((always_inline)) -> i32x16 {
    const char* a_base = smem + buf * L::TILE + aoff;
    char* dst = smem + (buf ^ 1) * L::TILE;
    const half_t s = q8_tile_scale(nx.meta.x);
    if constexpr (M16) {
      i32x4 c00 = {0, 0, 0, 0}, c01 = {0, 0, 0, 0}, c10 = {0, 0, 0, 0}, c11 = {0, 0, 0, 0};  // [row block][query block]
      // (D = 1024 fp8 with a deeper prefetch and/or one fetch group instead of two — 230-256 VGPRs, no spill — measured the same
      //  15.7-15.9 ms per 50M rows as this: LDS prefetch depth is not what that kernel waits for)
      constexpr int PF = (D <= 768) ? 2 : Q8_BIG_PF;     // k steps of 64 read ahead (two fragments each)
      constexpr int CSTEP = KS2 / CPT;           // one chunk converted every CSTEP steps (KS2 = 4·CPT for fp8)
      i32x4 a0[PF], a1[PF];
      if (!(ABL & 4) && !((ABL & 65536) && wave >= 4)) {
#pragma unroll
        for (int i = 0; i < PF; ++i) {
          a0[i] = *(const i32x4*)(a_base + 64 * i);
          a1[i] = *(const i32x4*)(a_base + 16 * L::RS + 64 * i);
        }
      }
#pragma unroll
      for (int ks = 0; ks < KS2; ++ks) {
        if (!(ABL & 4) && !((ABL & 65536) && wave >= 4)) {
          c00 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a0[ks % PF], qf[2 * ks], c00, 0, 0, 0);
          c01 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a0[ks % PF], qf[2 * ks + 1], c01, 0, 0, 0);
          c10 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a1[ks % PF], qf[2 * ks], c10, 0, 0, 0);
          c11 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a1[ks % PF], qf[2 * ks + 1], c11, 0, 0, 0);
          if (ks + PF < KS2 && !(ABL & 131072)) {   // (131072: the first PF k steps' fragments feed every MFMA — no further LDS reads)
            a0[ks % PF] = *(const i32x4*)(a_base + 64 * (ks + PF));
            a1[ks % PF] = *(const i32x4*)(a_base + 16 * L::RS + 64 * (ks + PF));
          }
        }
        if (ks % CSTEP == CSTEP / 2) convert_chunk(nx, ks / CSTEP, s, dst);
      }
      // (narrow rows: the chain is short and the pruning reads the scores right behind a branch — a window with free
      //  instructions in it once the walk of tests/codeobj.py follows branches; the wide instantiations have the next tile's
      //  conversion between the last MFMA and the first read and are left as measured)
      if constexpr (D <= 512) RARC_MFMA_SETTLE(c11);
      // as the MFMAs leave them: elements 0-7 = the lane's scores of query block 0 (rows 16 (r >> 2) + 4 rq + (r & 3)), 8-15 = block 1
      return (i32x16){c00[0], c00[1], c00[2], c00[3], c10[0], c10[1], c10[2], c10[3],
                      c01[0], c01[1], c01[2], c01[3], c11[0], c11[1], c11[2], c11[3]};
    } else {
      i32x16 c0 = {0};
      constexpr int PF = (D <= 768) ? 4 : 2;
      constexpr int CSTEP = KS / CPT;  // one chunk converted every CSTEP MFMAs (KS = 4·CPT for fp16)
      i32x4 a[PF];
      if (!(ABL & 4)) {
#pragma unroll
        for (int i = 0; i < PF; ++i) a[i] = *(const i32x4*)(a_base + 32 * i);
      }
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        if (!(ABL & 4)) {
          c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[ks % PF], qf[ks], c0, 0, 0, 0);
          if (ks + PF < KS) a[ks % PF] = *(const i32x4*)(a_base + 32 * (ks + PF));
        }
        if (ks % CSTEP == CSTEP / 2) convert_chunk(nx, ks / CSTEP, s, dst);
      }
      if constexpr (D <= 512) RARC_MFMA_SETTLE(c0);
      return c0;
    }
  };
  // regroup (at the start of the pruning, not behind the last MFMA: group B prunes a tile one iteration after its MFMAs):
  // swapping block 1 of lanes 0-31 with block 0 of lanes 32-63 leaves lanes 0-31 with sixteen scores of q_lo (their own block 0,
  // and in elements 8-15 the block 0 of the lane 32 above: row quad rq + 2) and lanes 32-63 with sixteen of q_lo + 16 (in
  // elements 0-7 the block 1 of the lane 32 below: row quad rq - 2, then their own): element 8 g + r = row
  // 16 (r >> 2) + 4 (rql + 2 g) + (r & 3) on every lane.
  // (inline asm consumers get no MFMA-result -> VALU-read wait states from the compiler's hazard recognizer — a first version
  //  that swapped straight behind the last MFMA read its registers stale: a handful of wrong candidates.  Here the pruning's
  //  fast path, compiler-scheduled VALU with its own hazard handling, has read every accumulator register before this runs.)
  auto regroup = [&](i32x16& a) __attribute__((always_inline)) {
    asm volatile("v_nop\n\tv_nop\n\t"
                 "v_permlane32_swap_b32 %0, %8\n\tv_permlane32_swap_b32 %1, %9\n\tv_permlane32_swap_b32 %2, %10\n\t"
                 "v_permlane32_swap_b32 %3, %11\n\tv_permlane32_swap_b32 %4, %12\n\tv_permlane32_swap_b32 %5, %13\n\t"
                 "v_permlane32_swap_b32 %6, %14\n\tv_permlane32_swap_b32 %7, %15"
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]),
                   "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]), "+v"(a[13]), "+v"(a[14]), "+v"(a[15]));
  };

  // one survivor straight to global memory (flush, and the overflow path of the staging buffer)
  // `hot`: the survivor also goes into the query's global histogram.  The histogram only serves to locate the
  // k-th best approximate score A (the owner publishes thr = A - 2·eps8; the finalize starts from the bin of A),
  // and A only rises: a survivor below thr + eps8 = A - eps8 can never take part in that, while such survivors
  // are ~85 % of all (the margin region grows exponentially towards lower scores) — and every histogram update
  // is a device-scope atomic.  Every row with a >= A(final) is counted whatever thresholds were current when it
  // was flushed, because thr(at flush) + eps8 <= A(final) - eps8.
  auto emit = [&](uint32_t q, uint64_t key, bool hot) {
    const uint32_t slot = atomicAdd(&s_cnt[q], 1u);
    if (slot < p.seg) p.cand[((size_t)q * RARC_MAX_WG + blockIdx.x) * p.seg + slot] = key;
    if (hot) atomicAdd(&p.hist[q * RARC_NB + rarc_bin_of(rarc_candscore(key), s_binlo[q], s_binscale[q])], 1u);
  };
  // Survivors go to an LDS staging area (no vector-memory traffic), one region per wave so that
  // slots are handed out with lane arithmetic (ballot + mbcnt) instead of an LDS atomic round trip;
  // a full region sends the survivor straight to global memory.
  uint32_t wcount = 0;  // wave-uniform: entries staged by this wave since the last flush
  uint64_t* my_skey = s_skey + wave * Q8_WSTAGE;
  uint8_t* my_sq = s_sq + wave * Q8_WSTAGE;
  auto stage_q = [&](bool want, float a, uint32_t doc, uint32_t qq) __attribute__((always_inline)) {  // called by the whole wave (want: this lane has one)
    const unsigned long long b = __builtin_amdgcn_ballot_w64(want);
    if (b == 0) return;
    const uint32_t pos = wcount + __builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0u));
    if (want) {
      const uint64_t key = rarc_candkey(a, doc);
      if (pos < (uint32_t)Q8_WSTAGE) {
        my_skey[pos] = key;
        my_sq[pos] = (uint8_t)qq;
      } else {
        emit(qq, key, true);
      }
    }
    wcount += (uint32_t)__builtin_popcountll(b);
  };
  auto stage_n = [&](bool want, float a, uint32_t doc) { stage_q(want, a, doc, qidx); };
  char* my_tb = smem + L::TB + wave * (Q8_TB_SLOTS * Q8_TB_STRIDE);
  // Passing lanes park their 16 scores (+ threshold, scale, query, first row) in these slots; the wave walks the
  // parked scores one per lane once four slots (64 scores: a full wave) have gathered, before a flush, and at
  // the end.  LDS returns a wave's operations in order, so the reads see the writes.
  uint32_t tb_n = 0;  // wave-uniform: slots in use
  auto drain = [&]() __attribute__((always_inline)) {
    if (tb_n == 0) return;
    __builtin_amdgcn_wave_barrier();
    // (the lane index goes through an opaque copy: otherwise the slot address below is hoisted out of the scan loop
    //  as a loop invariant, finds no free register there, and is SPILLED — and its reload here is a vector-memory
    //  load whose wait, vmcnt(0), drains the whole two-tile prefetch queue on every drain)
    int dl = lane;
    asm volatile("" : "+v"(dl));
    for (int base = 0; base < (int)tb_n * 16; base += 64) {
      const int item = base + dl;  // (slot, position) pair; slots past tb_n hold stale data and are masked out
      const bool valid = item < (int)tb_n * 16;
      const char* sp = my_tb + (item >> 4) * Q8_TB_STRIDE;
      const int r = item & 15;
      const int sv = *(const int*)(sp + 4 * r);
      const i32x4 mt = *(const i32x4*)(sp + 64);
      const float a = (float)sv * __uint_as_float((uint32_t)mt[1]);
      const uint32_t doc = (uint32_t)mt[3] + row_of((uint32_t)r);
      stage_q(valid && a >= __uint_as_float((uint32_t)mt[0]) && doc < p.n_rows, a, doc, (uint32_t)mt[2]);
    }
    __builtin_amdgcn_wave_barrier();
    tb_n = 0;
  };
  // lane holds 16 integer scores of its query: element r is row 8*(r>>2) + 4*h + (r&3) of the tile (M16: 16*((r>>2)&1) + 8*(r>>3) + 4*rql + (r&3)).
  // Fast path (every tile): max of (score << 4 | r) — the best score and where it sits — and one
  // compare.  When some lane's best clears its threshold, the wave counts per lane how many of the 16
  // scores clear a (conservative) integer threshold: when no lane has more than one — by far the usual
  // case — each passing lane's best is its only survivor and is staged directly; otherwise all 16
  // positions are walked.
  // `thr_in` already carries the tile's bonus: every row of tile t is within eps8 − hq·(R − R_t) of its approximate
  // score, so inside the tile the query's threshold may sit hq·(R − R_t) higher (an outlier tile anywhere in the
  // shard sets R; the typical tile is 1.5 - 2x better)
  auto prune = [&](const i32x16& acc_in, uint32_t tile, float tinv, float tmw) __attribute__((always_inline)) {
    i32x16 acc = acc_in;
    const float thr_g = thr;   // the query's threshold as published (flush compares against it)
    const float tsc = (float)q8_tile_scale(tmw);
    const float thr = __builtin_fmaf(my_hq, fmaxf(r_max - q8_tile_rt(tmw), 0.f), thr_g);
    // fast path: the lane's best score only (v_max3: 8 instructions); WHERE it sits is worked out in the slow
    // path, together with the count of scores above the integer threshold.  (Carrying the position along as
    // (score << 4 | r) cost 16 more instructions on every tile for something 70 % of the tiles never use.)
    int m, pm = 0;
    if (ABL & 2048) {  // (A/B: the packed form)
      pm = acc[0] << 4;
#pragma unroll
      for (int r = 1; r < 16; ++r) {
        const int v = (acc[r] << 4) | r;  // |score| < 2^24: no overflow
        pm = v > pm ? v : pm;
      }
      m = pm >> 4;
    } else if constexpr (M16) {
      // the scores are still as the MFMAs left them (8 of q_lo, 8 of q_lo + 16 on every lane): the best score of the lane's
      // query is the larger of its own half's maximum and its partner's (lane ^ 32) — ONE v_permlane32_swap of the two
      // maxima instead of the eight that regroup the scores, which only the tiles with a passing lane go on to need
      int mx = acc[0], my = acc[8];
#pragma unroll
      for (int r = 1; r < 8; ++r) { mx = acc[r] > mx ? acc[r] : mx; my = acc[8 + r] > my ? acc[8 + r] : my; }
      asm volatile("v_nop\n\tv_nop\n\tv_permlane32_swap_b32 %0, %1" : "+v"(mx), "+v"(my));
      m = mx > my ? mx : my;
    } else {
      m = acc[0];
#pragma unroll
      for (int r = 1; r < 16; ++r) m = acc[r] > m ? acc[r] : m;
    }
    const float sc = my_qinv * tinv;
    const bool pass = (float)m * sc >= thr;
    if (ABL & 32) {
      if (__builtin_amdgcn_ballot_w64((float)m * sc >= 1e30f) != 0) p.cnt2[1] = 1;
      return;
    }
    const unsigned long long pmask = __builtin_amdgcn_ballot_w64(pass);
    if (pmask != 0) {
      if constexpr (M16) regroup(acc);
      const uint32_t row0 = tile * 32 + 4 * (M16 ? rql : h);
      const int np = __builtin_popcountll(pmask);
      if (np <= Q8_TB_SLOTS && !(ABL & 4096)) {
        // Usually one to three of the 64 lanes pass.  Instead of every lane walking its own 16 scores, the
        // passing lanes park theirs in LDS and the wave walks parked scores one per lane (drain): a quarter of
        // the instructions of the per-lane walk at one slot per walk, less with four.  Same acceptance test
        // ((float)s * sc >= thr, with the threshold of the moment the lane passed), hence the same candidates.
        if (tb_n + (uint32_t)np > (uint32_t)Q8_TB_SLOTS) drain();
        if (pass) {
          const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(pmask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)pmask, 0u));
          char* dst = my_tb + (tb_n + rank) * Q8_TB_STRIDE;
          *(i32x4*)(dst) = (i32x4){acc[0], acc[1], acc[2], acc[3]};
          *(i32x4*)(dst + 16) = (i32x4){acc[4], acc[5], acc[6], acc[7]};
          *(i32x4*)(dst + 32) = (i32x4){acc[8], acc[9], acc[10], acc[11]};
          *(i32x4*)(dst + 48) = (i32x4){acc[12], acc[13], acc[14], acc[15]};
          *(i32x4*)(dst + 64) = (i32x4){(int)__float_as_uint(thr), (int)__float_as_uint(sc), (int)qidx, (int)row0};
        }
        tb_n += (uint32_t)np;
        if ((ABL & 8192) || tb_n >= 4) drain();
      } else {
      // every score s with (float)s*sc >= thr satisfies s >= ti (one unit + 1e-6 relative of slack)
      const float my_sq8 = 1.0f / my_qinv;  // the query's int8 scale (recomputed here: this path is rare, registers are not)
      const float tq = fmaxf(thr * (my_sq8 * tsc), -2.0e9f);  // thr / sc up to rounding; -inf (no threshold yet) clamped
      const int ti = pass ? (int)__builtin_floorf(tq - 1.0f - __builtin_fabsf(tq) * 2e-6f) : 0x7fffffff;
      int c = 0;
      uint32_t rs = (uint32_t)pm & 15u;
      if (ABL & 2048) {
#pragma unroll
        for (int r = 0; r < 16; ++r) c += (acc[r] >= ti) ? 1 : 0;
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          c += (acc[r] >= ti) ? 1 : 0;
          rs = acc[r] == m ? (uint32_t)r : rs;  // (ties: the highest position, as the packed maximum picked)
        }
      }
      if (__builtin_amdgcn_ballot_w64(c > 1) == 0) {
        const uint32_t doc = row0 + row_of(rs);
        stage_n(pass && doc < p.n_rows, (float)m * sc, doc);
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float a = (float)acc[r] * sc;
          const uint32_t doc = row0 + row_of((uint32_t)r);
          stage_n(a >= thr && doc < p.n_rows, a, doc);
        }
      }
      }
    }
  };
  // all threads; staged entries -> private candidate segments + global histogram.  Each wave drains
  // its own region (it knows its count; nothing to exchange), so no barrier is needed around it.
  auto flush = [&]() __attribute__((always_inline)) {
    drain();
    const uint32_t n = wcount < (uint32_t)Q8_WSTAGE ? wcount : (uint32_t)Q8_WSTAGE;
    for (uint32_t e0 = 0; e0 < n; e0 += 64) {  // wave-uniform trips: the threshold of an entry's query comes by shuffle
      const uint32_t e = e0 + lane;
      const bool v = e < n;
      const uint32_t qq = v ? (uint32_t)my_sq[e] : qidx;
      const uint64_t key = v ? my_skey[e] : 0ull;
      // query 32·wave + j sits on lanes j and j + 32; M16: on lanes (j & 15) + 32 (j >> 4) and 16 above
      const float tcur = __shfl(thr, (int)(M16 ? (qq & 15u) + 2u * (qq & 16u) : (qq & 31u)), 64);
      const bool hot = (ABL & 16384) || rarc_candscore(key) >= tcur + p.hot_margin * s_eps8[qq];
      // (a survivor staged under an older, lower threshold that no longer clears the current one is dropped:
      //  the final threshold is at least the current one)
      if (v && rarc_candscore(key) >= tcur) emit(qq, key, hot);
    }
    wcount = 0;
  };

  // fetch groups in flight: two up to D = 896; one beyond (a group is 36 registers at D = 1024 and a
  // spill would put scratch traffic into the very queue the counted waits rely on); FOUR for narrow rows, where two
  // tiles are too few bytes in flight to cover the HBM latency (48 KB per CU at D = 384 against ~45 KB needed:
  // cycles per tile were 1170 + 2.4 D, the constant being exposed latency — DESIGN.md, narrow rows)
  constexpr int NG = D <= Q8_DEEP_D ? 4 : (FMT != 0 ? (D <= 768 ? 2 : Q8_BIG_NG) : (D <= 896 ? 2 : 1));
  // ---- prologue: tile t0 straight into LDS buffer 0; tiles t0+stride, t0+2·stride in flight ----
  // (the launch guarantees gridDim.x <= n_tiles, so tile t0 exists)
  Fetch f[NG];
  fetch(f[0], t0, Q8Flag<true>{});
  convert_tile(f[0], 0);
  float2 mcur[2];   // (scale | R_t word, 1/scale) of the tile sitting in LDS buffer 0 / 1
  mcur[0] = f[0].meta;
  // (scheduling fences: the groups must be ISSUED in this order, or the counted waits the compiler
  // derives for the first loop iteration assume the wrong group is the newest)
  __builtin_amdgcn_sched_barrier(0);
  // ping-pong: waves w and w+4 share a SIMD.  Group A (waves 0-3) runs a tile's MFMAs first and prunes it
  // afterwards; group B (waves 4-7) prunes the PREVIOUS tile first and runs the MFMAs second — so one
  // wave's VALU / VMEM phase (prune, survivors, prefetch issue) overlaps its partner's matrix phase
  // instead of both leaving the matrix pipe idle at the same time.
  constexpr bool PP = (D <= 768) && !(ABL & 256);  // (where the second accumulator still fits in registers)
  const bool grp_b = PP && wave >= Q8_WAVES / 2;
  // EB — without the ping-pong (D > 768: the fp8 rows of config 5) the iteration's barrier sits right BEHIND the matrix
  // phase instead of at the end of the iteration.  What the barrier orders is the tile buffers: every wave has finished
  // reading buffer PAR (its MFMAs) and writing buffer PAR^1 (its share of the next tile's conversion) — both are over when
  // mfma_convert returns; pruning, survivor staging and the refill of the fetch group touch per-wave LDS and registers only.
  // With the barrier at the end (round 3), the s_memtime timeline of 100M x 1024 fp8 rows showed (profiles/r04_f8_timeline.txt)
  // waves 0-3 of a workgroup done with MFMA + prune + refill at ~2700 of a 4350-cycle iteration and then parked at the barrier
  // until ~4100, waiting for their SIMD partners (waves 4-7 lose the arbitration for the matrix pipe, finish their MFMAs
  // at ~2850 and prune afterwards) — a third of every iteration with the matrix pipe idle.  Now the early waves wait only
  // for the partners' MFMAs, and a wave's pruning overlaps the other waves' next matrix phase.
  constexpr bool EB = Q8_EB && !PP && !(ABL & 512);
  if constexpr (NG == 4) {
    fetch(f[1], clamp_tile(t0 + stride), Q8Flag<true>{});
    __builtin_amdgcn_sched_barrier(0);
    fetch(f[2], clamp_tile(t0 + 2 * stride), Q8Flag<true>{});
    __builtin_amdgcn_sched_barrier(0);
    fetch(f[3], clamp_tile(t0 + 3 * stride), Q8Flag<true>{});
    __builtin_amdgcn_sched_barrier(0);
    if (!grp_b) fetch(f[0], clamp_tile(t0 + 4 * stride), Q8Flag<true>{});  // (group B: in its first iteration, as below)
  } else if constexpr (NG == 2) {
    fetch(f[1], clamp_tile(t0 + stride), Q8Flag<true>{});
    __builtin_amdgcn_sched_barrier(0);
    // (group B issues this one in its first iteration, by the same formula as in every later one)
    if (!grp_b) fetch(f[0], clamp_tile(t0 + 2 * stride), Q8Flag<true>{});
  } else {
    fetch(f[0], clamp_tile(t0 + stride), Q8Flag<true>{});
  }
  __builtin_amdgcn_sched_barrier(0);
  // owner lane: last threshold it published (starts from what the seed pass / the tightening pass left there)
  float last_pub = has_own ? __uint_as_float(p.thr[own_q]) : -INFINITY;
  // Drain once before the loop.  Otherwise the loop header sees, from this path only, prologue loads in
  // flight, and the wait the compiler places there for them (an absolute "at most N outstanding") is
  // executed on every trip and empties the prefetch queue each time.  Costs one memory latency per launch.
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
  q8_lds_barrier();

  i32x16 acc_b = {0};      // group B: scores of the tile whose pruning is still to come
  bool live_prev = false;  // group B: that tile exists
  uint32_t it = 0;
  // one iteration = one tile; PAR (its parity) names the LDS buffer read and the fetch group consumed
#define Q8_STAMP(slot)                                                                                        \
  if ((ABL & 1024) && blockIdx.x == 0 && it >= 1000 && it < 1016 && lane == 0)                                 \
    p.dbg[8 + ((it - 1000) * Q8_WAVES + wave) * 8 + (slot)] = __builtin_amdgcn_s_memtime();
// G = the fetch group this iteration converts (tile cur + stride), GP = the one the previous iteration converted
// RF(g): does fetch group g carry the threshold / histogram refresh (all of them unless NG == 4: then group 1 only)
#define Q8_RF(g) (NG != 4 || (g) == 1)
#define Q8_ITER(PAR, GB, G, GP)                                                                               \
  {                                                                                                           \
    Q8_STAMP(0)                                                                                               \
    if (!(GB)) { /* ---- group A: MFMA(cur) + convert(next), then prune(cur), then refill ---- */            \
      const float tinv = mcur[PAR].y, tsc = mcur[PAR].x;                                                      \
      i32x16 acc;                                                                                             \
      acc = mfma_convert(PAR, f[G]);                                                                  \
      mcur[(PAR) ^ 1] = f[G].meta;                                                                    \
      Q8_STAMP(1)                                                                                             \
      if (!(ABL & 8) && Q8_RF(G)) {                                                                           \
        thr = fmaxf(thr, __uint_as_float(f[G].thr));                                                  \
        if (lane < 32) s_hland[32 * wave + lane] = f[G].hw;                                           \
      }                                                                                                       \
      if constexpr (EB) { /* the iteration's ONE barrier, here: see EB above */                               \
        Q8_STAMP(3)                                                                                           \
        q8_lds_barrier();                                                                                     \
        Q8_STAMP(4)                                                                                           \
      }                                                                                                       \
      if (!(ABL & 1) && live) prune(acc, cur, tinv, tsc);                                                     \
      else if (q8_all16(acc) == 0x7fffffff) p.cnt2[0] = 1; /* keep ALL the MFMAs alive (M16: four accumulators) */ \
      Q8_STAMP(2)                                                                                             \
      /* refill that group with tile cur+3·stride */                                                          \
      if (!(ABL & 2)) fetch(f[G], clamp_tile(cur + (NG + 1) * stride), Q8Flag<Q8_RF(G)>{});           \
    } else { /* ---- group B: prune(previous tile), refill the group consumed last iteration, then MFMA ---- */ \
      if (!(ABL & 1) && live_prev) prune(acc_b, cur - stride, mcur[(PAR) ^ 1].y, mcur[(PAR) ^ 1].x);          \
      live_prev = live;                                                                                       \
      Q8_STAMP(1)                                                                                             \
      if (!(ABL & 2)) fetch(f[GP], clamp_tile(cur + NG * stride), Q8Flag<Q8_RF(GP)>{});                       \
      __builtin_amdgcn_sched_barrier(0); /* the chunk loads must be ISSUED before the matrix phase */          \
      Q8_STAMP(2)                                                                                             \
      acc_b = mfma_convert(PAR, f[G]);                                                                        \
      mcur[(PAR) ^ 1] = f[G].meta;                                                                            \
      if (!(ABL & 8) && Q8_RF(G)) {                                                                           \
        thr = fmaxf(thr, __uint_as_float(f[G].thr));                                                          \
        if (lane < 32) s_hland[32 * wave + lane] = f[G].hw;                                                   \
      }                                                                                                       \
      if ((ABL & 1) && q8_all16(acc_b) == 0x7fffffff) p.cnt2[0] = 1;                                          \
    }                                                                                                         \
    if constexpr (!EB) {                                                                                      \
      Q8_STAMP(3)                                                                                             \
      q8_lds_barrier();                                                                                       \
      Q8_STAMP(4)                                                                                             \
    }                                                                                                         \
    ++it;                                                                                                     \
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
    if (!(ABL & 64) && (wcount | tb_n) != 0 &&                                                                 \
        (it <= 16 || (it & (it - 1)) == 0 || (it & 127) == 0 || wcount > (uint32_t)Q8_WSTAGE / 2))            \
      flush();                                                                                                \
  }

  // Always whole pairs of iterations, no exit from the middle of the loop body (a mid-body break makes
  // the compiler's wait-counter merge treat the wrong fetch group as the newest one and drain the
  // queue every iteration).  When a workgroup's tile count is odd the last half-iteration runs on a
  // tile index past the end: its scores are ignored (`live`), its fetch is the usual tile-0 dummy.
  // (two copies of the loop, one per wave group: inside ONE loop the compiler would merge the wait counters
  // of the two instruction orders and fall back to draining the queue)
#define Q8_STEP(OFF, PAR, GB, G, GP)                                                  \
    {                                                                                 \
      const uint32_t cur = base + (OFF) * stride;                                     \
      const bool live = (OFF) == 0 || cur < p.n_tiles;                                \
      Q8_ITER(PAR, GB, G, GP)                                                         \
    }
#define Q8_LOOP(GB)                                                                   \
  if constexpr (NG == 4) {                                                            \
    for (uint32_t base = t0; base < p.n_tiles; base += 4 * stride) {                  \
      Q8_STEP(0, 0, GB, 1, 0) Q8_STEP(1, 1, GB, 2, 1) Q8_STEP(2, 0, GB, 3, 2) Q8_STEP(3, 1, GB, 0, 3) \
    }                                                                                 \
  } else {                                                                            \
    for (uint32_t base = t0; base < p.n_tiles; base += 2 * stride) {                  \
      Q8_STEP(0, 0, GB, NG == 2 ? 1 : 0, 0) Q8_STEP(1, 1, GB, 0, NG == 2 ? 1 : 0)     \
    }                                                                                 \
  }
  if (grp_b) {
    Q8_LOOP(true)
  } else {
    Q8_LOOP(false)
  }
#undef Q8_STEP
#undef Q8_LOOP
#undef Q8_ITER
#undef Q8_RF
  // group B still owes the pruning of its last tile (the second half of the last pair: LDS buffer 1)
  if (grp_b && !(ABL & 1) && live_prev)
    prune(acc_b, t0 + (((p.n_tiles - 1 - t0) / stride) | (NG == 4 ? 3u : 1u)) * stride, mcur[1].y, mcur[1].x);
  flush();
  __syncthreads();
  if (tid < RARC_MAX_QUERIES) p.cnt2[(size_t)blockIdx.x * RARC_MAX_QUERIES + tid] = s_cnt[tid];
  if (p.dbg && blockIdx.x == 0 && tid == 0) {
    p.dbg[0] = __builtin_amdgcn_s_memtime() - dbg_c0;
    p.dbg[1] = __builtin_amdgcn_s_memrealtime() - dbg_r0;
  }
}

// ---- host side -----------------------------------------------------------------------------------
bool rarc_prof_next(hipEvent_t* start, hipEvent_t* stop);  // rarc_api.hip
int rarc_seed_launch(const void* corpus, const float* rowscale, int fmt, int64_t n_rows, int d_pad,
                     const uint16_t* q16, int nq, int kprime, float bin_lo, float bin_hi, const float* sub_a,
                     const float* sub_b, const RarcWs& ws, hipStream_t s, int64_t rows_covered, const float* floor);  // scan_f16.hip
int rarc_scan_f16_stage_launch(const uint16_t* corpus, int64_t n_rows, uint32_t stage_tiles, int d_pad, const uint16_t* q16,
                               const float* eps16, int nq, int kprime, const RarcWs& ws, int cap, int grid, hipStream_t s);  // scan_f16.hip

template <int D, int FMT, int ABL = 0>
static int launch_scan_q8(const ScanQ8Params& p, int grid, hipStream_t s) {
  constexpr size_t lds = ScanQ8Lds<D, (FMT != 0) || Q8_M16_FP16_ROWS>::TOTAL;
  static RarcPerDevice attr_done;
  if (size_t& done = attr_done.cur(); !done) {
    RARC_HIP_CHECK(hipFuncSetAttribute((const void*)rarc_scan_q8_kernel<D, FMT, ABL>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    done = 1;
  }
#ifdef RARC_Q8_ABLATIONS
  // (measurement builds only: RARC_Q8_ABL selects an ablated instantiation of the two headline shapes — results are wrong)
  if constexpr (ABL == 0 && ((D == 1024 && FMT == 1) || (D == 768 && FMT == 0))) {
    static const int abl = getenv("RARC_Q8_ABL") ? atoi(getenv("RARC_Q8_ABL")) : 0;
    if (abl == 1) return launch_scan_q8<D, FMT, 1>(p, grid, s);
    if (abl == 4) return launch_scan_q8<D, FMT, 4>(p, grid, s);
    if (abl == 5) return launch_scan_q8<D, FMT, 5>(p, grid, s);
    if (abl == 9) return launch_scan_q8<D, FMT, 9>(p, grid, s);
    if constexpr (FMT == 1) {
      if (abl == 1024 || abl == 1025) {   // s_memtime timeline of workgroup 0, iterations 1000-1015 -> the file RARC_Q8_TIMELINE names
        static unsigned long long* d_dbg = nullptr;
        if (!d_dbg) RARC_HIP_CHECK(hipMalloc((void**)&d_dbg, 65536));
        RARC_HIP_CHECK(hipMemsetAsync(d_dbg, 0, 65536, s));
        ScanQ8Params pd = p;
        pd.dbg = d_dbg;
        hipFuncSetAttribute((const void*)rarc_scan_q8_kernel<D, FMT, 1024>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipFuncSetAttribute((const void*)rarc_scan_q8_kernel<D, FMT, 1025>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (abl == 1024) hipLaunchKernelGGL((rarc_scan_q8_kernel<D, FMT, 1024>), dim3(grid), dim3(Q8_THREADS), lds, s, pd);
        else hipLaunchKernelGGL((rarc_scan_q8_kernel<D, FMT, 1025>), dim3(grid), dim3(Q8_THREADS), lds, s, pd);
        RARC_HIP_CHECK(hipStreamSynchronize(s));
        if (const char* path = getenv("RARC_Q8_TIMELINE"); path && p.n_tiles - p.t_begin > 1100u * (uint32_t)grid) {
          static unsigned long long h[8192];
          RARC_HIP_CHECK(hipMemcpy(h, d_dbg, 65536, hipMemcpyDeviceToHost));
          if (FILE* f = fopen(path, "a")) {
            fprintf(f, "launch tiles [%u, %u) abl %d: per iteration and wave, shader cycles relative to wave 0's start of the iteration: start | after mfma+convert | after prune | after fetch issue | after barrier\n", p.t_begin, p.n_tiles, abl);
            for (int it = 2; it < 10; ++it)
              for (int w = 0; w < Q8_WAVES; ++w) {
                const unsigned long long* r = &h[8 + (it * Q8_WAVES + w) * 8];
                const unsigned long long b0 = h[8 + (it * Q8_WAVES) * 8];
                fprintf(f, "it %2d w%d: %6lld %6lld %6lld %6lld %6lld   (next iteration starts at %lld)\n", it, w, (long long)(r[0] - b0), (long long)(r[1] - b0),
                        (long long)(r[2] - b0), (long long)(r[3] - b0), (long long)(r[4] - b0), (long long)(h[8 + ((it + 1) * Q8_WAVES + w) * 8] - b0));
              }
            fclose(f);
          }
        }
        return RARC_OK;
      }
      // round 4: where the fp8 scan's time goes (profiles/r04_f8_attribution.txt)
      if (abl == 512) return launch_scan_q8<D, FMT, 512>(p, grid, s);        // round 3's barrier placement (A/B of EB; results are right)
      if (abl == 17) return launch_scan_q8<D, FMT, 17>(p, grid, s);          // no conversion, no pruning (MFMAs on raw bytes)
      if (abl == 21) return launch_scan_q8<D, FMT, 21>(p, grid, s);          // fetch + LDS write only
      if (abl == 32769) return launch_scan_q8<D, FMT, 32769>(p, grid, s);    // converted but not written to LDS, MFMAs on stale LDS, no pruning
      if (abl == 32773) return launch_scan_q8<D, FMT, 32773>(p, grid, s);    // fetch + conversion only
      // round 5: do the streaming phase and the matrix phase overlap?  half the MFMAs (waves 4-7 issue none), with and without pruning
      if (abl == 131073) return launch_scan_q8<D, FMT, 131073>(p, grid, s);   // no pruning, all MFMAs, 1/8 of the LDS fragment reads: what do the reads cost?
      if (abl == 65536) return launch_scan_q8<D, FMT, 65536>(p, grid, s);
      if (abl == 65537) return launch_scan_q8<D, FMT, 65537>(p, grid, s);
    }
  }
#endif
  hipEvent_t e0, e1;
  const bool prof = rarc_prof_next(&e0, &e1);
  if (prof) RARC_HIP_CHECK(hipEventRecord(e0, s));
  hipLaunchKernelGGL((rarc_scan_q8_kernel<D, FMT, ABL>), dim3(grid), dim3(Q8_THREADS), lds, s, p);
  RARC_HIP_CHECK(hipGetLastError());
  if (prof) RARC_HIP_CHECK(hipEventRecord(e1, s));
  return RARC_OK;
}

// Host entry used by rarc_api.hip.  *grid_out = workgroups launched (owners of candidate segments).
// fmt 0: fp16 rows; fmt 1: fp8 (e4m3fn) rows with per-row scales `rowscale`.
// fmt 2: `shadow8` is the int8 image of the fp16 rows `corpus` (which the seed pass still reads).
template <int FMTV>
static int dispatch_scan_q8(const ScanQ8Params& p, int d_pad, int grid, hipStream_t s) {
  switch (d_pad) {
    case 256: return launch_scan_q8<256, FMTV>(p, grid, s);
    case 512: return launch_scan_q8<512, FMTV>(p, grid, s);
    case 768: return launch_scan_q8<768, FMTV>(p, grid, s);
    case 1024: return launch_scan_q8<1024, FMTV>(p, grid, s);
    default: break;
  }
  if (FMTV == 0) {
    switch (d_pad) {
      case 128: return launch_scan_q8<128, 0>(p, grid, s);
      case 384: return launch_scan_q8<384, 0>(p, grid, s);
      case 640: return launch_scan_q8<640, 0>(p, grid, s);
      case 896: return launch_scan_q8<896, 0>(p, grid, s);
      default: break;
    }
  }
  rarc_set_error("rarc_scan_q8 (row format %d): padded dim %d unsupported (multiple of %d, <= 1024)", FMTV, d_pad,
                 FMTV == 0 ? 128 : 256);
  return RARC_E_UNSUPPORTED;
}

int rarc_scan_q8_launch(const void* corpus, const float* rowscale, int fmt, int64_t n_rows, int d_pad,
                        const float* qmeta, const uint16_t* q16, const int8_t* q8, const float* qinv,
                        const float* eps16, const float* eps8, int nq, int kprime, float bin_lo, float bin_hi,
                        const RarcWs& ws, int cap, int* grid_out, hipStream_t s, const int8_t* shadow8,
                        int (*tighten)(void* ctx, int n_wg, int mode), void* tighten_ctx, const float* hq, const float* floor,
                        int* hybrid_out) {
  ScanQ8Params p;
  p.hq = hq;
  p.corpus = fmt == 2 ? (const uint4*)shadow8 : (const uint4*)corpus;
  p.tmeta = qmeta + RARC_QMETA_HDR;
  p.q8 = q8;
  p.qinv = qinv;
  p.eps8 = eps8;
  p.n_rows = (uint32_t)n_rows;
  p.n_tiles = (uint32_t)((n_rows + 31) / 32);
  p.t_begin = 0;
  p.resume = 0;
  p.hot_margin = 1.0f;
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

  int dev = 0, cus = 256;
  RARC_HIP_CHECK(hipGetDevice(&dev));
  RARC_HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  int grid = cus < RARC_MAX_WG ? cus : RARC_MAX_WG;
  // RARC_SCAN_Q8_WGS=n: fewer persistent workgroups than CUs — the CUs left free are where a neighbouring search context's
  // finalize (153 KB of LDS: it cannot share a CU with a scan workgroup) runs UNDER this scan (engine.py)
  if (const char* e = getenv("RARC_SCAN_Q8_WGS")) { const int g = atoi(e); if (g >= 8 && g < grid) grid = g; }
  if ((uint32_t)grid > p.n_tiles) grid = (int)p.n_tiles;
  *grid_out = grid;
  auto launch = [&](const ScanQ8Params& pp) {
    return fmt == 2 ? dispatch_scan_q8<2>(pp, d_pad, grid, s)
                    : fmt == 1 ? dispatch_scan_q8<1>(pp, d_pad, grid, s) : dispatch_scan_q8<0>(pp, d_pad, grid, s);
  };
  // Split scan.  While it follows the k-th best APPROXIMATE score the threshold sits two error bounds below it
  // (one because that score only bounds the true k-th best score L from below by eps8, one because a row as good
  // as L may score eps8 lower).  At a few points of the scan — after 1/512, 1/64 and 1/8 of the shard, as far as
  // each is large enough to be worth a launch — the caller's `tighten` pass rescores the best candidates so far
  // exactly: their k-th best canonical score L1 needs no first margin, and the next stretch runs under
  // max(thr, L1 - eps8) from its first tile.  The k-th best score of the first n rows grows like the (1 - k/n)
  // quantile, so each stretch is 8x the previous one at a threshold roughly a third of a sigma higher; the lag of
  // 2 eps8 behind that quantile is where most candidates came from (100M x 1024 fp8 rows on N(0,1) data: 97 K
  // candidates per query with one pass at 1/8, mostly from the stretch [1M, 12.5M) rows).
  // (RARC_SCAN_SPLIT=0: one launch; =1: the single pass at 1/8 of round 1.)
  static const int split_mode = getenv("RARC_SCAN_SPLIT") ? atoi(getenv("RARC_SCAN_SPLIT")) : 3;
  const uint32_t pair = 2u * (uint32_t)(grid > 0 ? grid : 1);
  uint32_t cuts[3];
  int n_cuts = 0;
  if (tighten && split_mode > 0) {
    uint32_t div[3] = {512u, 64u, 8u};
    if (const char* e = getenv("RARC_SPLIT_DIVS")) {          // (experiments: "64,8" / "0,0,16" ...; 0 = no cut)
      unsigned a = 0, b = 0, c = 0;
      const int got = sscanf(e, "%u,%u,%u", &a, &b, &c);
      if (got >= 1) { div[0] = a; div[1] = got >= 2 ? b : 0; div[2] = got >= 3 ? c : 0; }
    }
    for (int i = (split_mode >= 3 ? 0 : (split_mode == 2 ? 1 : 2)); i < 3; ++i) {
      if (div[i] == 0) continue;
      const uint32_t t = (p.n_tiles / div[i]) / pair * pair;
      // (a stretch shorter than 8 pairs of tile rounds per workgroup costs more in launches than it saves; the
      //  1/8 cut keeps its measured limit of 16: below ~2M rows the second launch does not pay)
      static const int min_env = getenv("RARC_SPLIT_MIN") ? atoi(getenv("RARC_SPLIT_MIN")) : 0;  // (experiments)
      const uint32_t min_pairs = min_env > 0 ? (uint32_t)min_env : (div[i] == 8u ? 16u : 8u);
      if (t >= min_pairs * pair && (n_cuts == 0 || t > cuts[n_cuts - 1])) cuts[n_cuts++] = t;
    }
  }
  // seed pass (fp16 MFMA on a strided sample of the whole shard): t = k'-th best sample score, accurate to eps16,
  // so t − eps16 bounds the k-th best canonical score from below; rows whose int8 score is under t − eps16 − eps8
  // are out.  The sample is sized for the rows that run under it: the first launch only, when the scan is split.
  int rc = rarc_seed_launch(corpus, rowscale, fmt == 2 ? 0 : fmt, n_rows, d_pad, q16, nq, kprime, bin_lo, bin_hi, eps16,
                            eps8, ws, s, n_cuts ? (int64_t)cuts[0] * 32 : 0, floor);
  if (rc) return rc;
  if (p.n_tiles == 0) return RARC_OK;  // (the seed pass above still initialised thresholds, histograms and flags)
  RARC_REQUIRE(rarc_gate_scan(s) == 0, RARC_E_HIP, "rarc_scan_q8: hipStreamWaitEvent on the gate event failed");
  uint32_t begin = 0;
  // Hybrid search of a SMALL shard (fp16 rows, no cascade cut: below ~2M rows).  There the int8 scan's weakness is its
  // start: the 4096-row sample puts the threshold two int8 error bounds under a k-th best score that is itself far below
  // the final one (11 K candidates per query at 1M rows, a 0.1 ms finalize), while the fp16 MFMA scan — tight margin, a few
  // hundred candidates — is held at 0.36 ms per 1M rows by the matrix pipe's power draw.  So the FIRST EIGHTH of the
  // shard runs through the fp16 kernel (scan_f16.hip, rigorous 2·eps16 margin: no certificate needed), one exact pass
  // turns its candidates into L1 = the k-th best canonical score of 1/8 of the rows, and the int8 kernel streams the other
  // seven eighths under L1 - eps8 from its first tile, continuing the same candidate segments.
  static const int hybrid_env = getenv("RARC_HYBRID") ? atoi(getenv("RARC_HYBRID")) : 1;
  static const int hybrid_div = getenv("RARC_HYBRID_DIV") ? atoi(getenv("RARC_HYBRID_DIV")) : 8;
  if (hybrid_out) *hybrid_out = 0;
  if (hybrid_out && hybrid_env && tighten && fmt == 0 && d_pad <= 768 && n_cuts == 0 && hybrid_div >= 2) {
    const uint32_t stage = (p.n_tiles / (uint32_t)hybrid_div) / pair * pair;   // whole pairs of tile rounds, as the cuts
    if (stage >= 2u * pair && stage < p.n_tiles) {
      if ((rc = rarc_scan_f16_stage_launch((const uint16_t*)corpus, n_rows, stage, d_pad, q16, eps16, nq, kprime, ws, cap, grid,
                                           s)) != RARC_OK)
        return rc;
      if ((rc = tighten(tighten_ctx, grid, 2)) != RARC_OK) return rc;
      ScanQ8Params pi = p;
      pi.t_begin = stage;
      pi.resume = 1u;
      pi.hot_margin = 0.f;
      *hybrid_out = 1;
      return launch(pi);
    }
  }
  for (int i = 0; i <= n_cuts; ++i) {
    ScanQ8Params pi = p;
    pi.t_begin = begin;
    pi.n_tiles = i < n_cuts ? cuts[i] : p.n_tiles;
    pi.resume = i > 0 ? 1u : 0u;
    // (after a pass the tightened threshold is no longer "k-th approximate score - 2 eps8": count everything)
    pi.hot_margin = i > 0 ? 0.f : 1.0f;
    if ((rc = launch(pi)) != RARC_OK) return rc;
    if (i < n_cuts && (rc = tighten(tighten_ctx, grid, 1)) != RARC_OK) return rc;
    begin = pi.n_tiles;
  }
  return RARC_OK;
}
