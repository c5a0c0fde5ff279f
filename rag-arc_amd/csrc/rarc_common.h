// rarc_common.h — shared device/host helpers for librarc_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/rarc.h"

// ---- measurement-build switches --------------------------------------------------------------------------------------------
// A few -D switches build kernels that return WRONG results on purpose: ablations, which exist to time a loop with one of
// its parts removed (SCAN_HALFREAD: every second A fragment of the fp16 scan not read; RARC_Q8_ABLATIONS: the int8 scan with
// pruning / loads / MFMAs removed, selected at run time by RARC_Q8_ABL; G256_SKIP_A1: a third of a GEMM's fragment reads).
// None of them may reach a product library: they compile only together with -DRARC_EXPERIMENT, and a library built that
// way announces itself — rarc_version() returns RARC_VERSION + 100000, which binding.py and tests/test_cabi.py refuse.
#if (defined(SCAN_HALFREAD) || defined(RARC_Q8_ABLATIONS) || defined(G256_SKIP_A1)) && !defined(RARC_EXPERIMENT)
#error "SCAN_HALFREAD / RARC_Q8_ABLATIONS / G256_SKIP_A1 build kernels with wrong results: measurement builds only, add -DRARC_EXPERIMENT"
#endif
#ifdef RARC_EXPERIMENT
#define RARC_BUILD_VERSION (RARC_VERSION + 100000)
#else
#define RARC_BUILD_VERSION RARC_VERSION
#endif

typedef _Float16 half_t;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define RARC_GPTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define RARC_LPTR(p) ((__attribute__((address_space(3))) void*)(p))

// ---- workspace layout (bytes from a 256-byte aligned base) -------------------------------
// thr      : float   [256]         running per-query pruning threshold (monotone increasing)
// binlo    : float   [256]         per-query histogram window: bin(s) = floor((s - lo) * scale)
// binscale : float   [256]
// bininv   : float   [256]         1 / scale (bin width)
// cnt      : uint32  [256]         (repair scratch counter lives in cnt[0])
// flags    : uint32  [64]
// hist     : uint32  [256][NB]     per-query score histogram of appended candidates
// cnt2     : uint32  [256 wg][256] candidates appended per (workgroup, query)
// seed     : float   [256][65536]  scores of the strided seed sample (initial thresholds)
// cand     : uint64  [256][cap]    query q, workgroup w owns slots [w*seg, (w+1)*seg), seg = cap/256;
//                                  key = ordkey(score)<<32 | ~local_row
constexpr int RARC_NB = 256;          // histogram bins per query
constexpr int RARC_MAX_WG = 256;      // scan workgroups (one per CU); owners of candidate segments
constexpr int RARC_SEED_TILES = 128;  // strided sample tiles scored before the fp16 scan (4096 rows)
constexpr int RARC_SEED_MAX_TILES = 2048;  // ... before the int8 scan: n_tiles/128 clamped to [128, 2048]
constexpr size_t RARC_WS_THR = 0;
constexpr size_t RARC_WS_BINLO = 1024;
constexpr size_t RARC_WS_BINSCALE = 2048;
constexpr size_t RARC_WS_BININV = 3072;
constexpr size_t RARC_WS_CNT = 4096;
constexpr size_t RARC_WS_FLAGS = 5120;
static_assert(RARC_WS_FLAGS + 4 == RARC_WS_ANYFLAG_OFFSET, "include/rarc.h out of sync");
constexpr size_t RARC_WS_HIST = 8192;
constexpr size_t RARC_WS_CNT2 = RARC_WS_HIST + (size_t)RARC_MAX_QUERIES * RARC_NB * 4;
constexpr size_t RARC_WS_SEED = RARC_WS_CNT2 + (size_t)RARC_MAX_WG * RARC_MAX_QUERIES * 4;
constexpr size_t RARC_WS_CAND = RARC_WS_SEED + (size_t)RARC_MAX_QUERIES * RARC_SEED_MAX_TILES * 32 * 4;

struct RarcWs {
  float* thr;
  float* binlo;
  float* binscale;
  float* bininv;
  uint32_t* cnt;
  uint32_t* flags;
  uint32_t* hist;
  uint32_t* cnt2;
  float* seed;
  uint64_t* cand;
};
static inline RarcWs rarc_ws_carve(void* base) {
  char* b = (char*)base;
  RarcWs w;
  w.thr = (float*)(b + RARC_WS_THR);
  w.binlo = (float*)(b + RARC_WS_BINLO);
  w.binscale = (float*)(b + RARC_WS_BINSCALE);
  w.bininv = (float*)(b + RARC_WS_BININV);
  w.cnt = (uint32_t*)(b + RARC_WS_CNT);
  w.flags = (uint32_t*)(b + RARC_WS_FLAGS);
  w.hist = (uint32_t*)(b + RARC_WS_HIST);
  w.cnt2 = (uint32_t*)(b + RARC_WS_CNT2);
  w.seed = (float*)(b + RARC_WS_SEED);
  w.cand = (uint64_t*)(b + RARC_WS_CAND);
  return w;
}


// ---- query block (written by rarc_prep_queries, read by search / repair) ------------------------
// q32 f32 [256][d_pad] | q16 f16 [256][d_pad] | q8 i8 [256][d_pad] | eps16 f32 [256] | eps8 f32 [256]
// | qinv f32 [256] | hq f32 [256] (||q8/s_q|| rounded down: eps8 shrinks by hq·(R − R_t) inside tile t)
// | floor f32 [256] (a known lower bound of the query's k-th best canonical score, -inf when there is none:
// rarc_prep_queries resets it, rarc_qblock_set_floor fills it before a search is re-run).
// d_pad is a multiple of 128, so every part starts 256-byte aligned.
struct RarcQb {
  float* q32;
  uint16_t* q16;
  int8_t* q8;
  float* eps16;
  float* eps8;
  float* qinv;
  float* hq;
  float* floor;
};
static inline size_t rarc_qb_bytes(int d_pad) { return (size_t)RARC_MAX_QUERIES * (size_t)d_pad * 7 + 5 * 1024; }
static inline RarcQb rarc_qb_carve(const void* base, int d_pad) {
  char* b = (char*)base;
  const size_t n = (size_t)RARC_MAX_QUERIES * (size_t)d_pad;
  RarcQb q;
  q.q32 = (float*)b;
  q.q16 = (uint16_t*)(b + n * 4);
  q.q8 = (int8_t*)(b + n * 6);
  q.eps16 = (float*)(b + n * 7);
  q.eps8 = (float*)(b + n * 7 + 1024);
  q.qinv = (float*)(b + n * 7 + 2048);
  q.hq = (float*)(b + n * 7 + 3072);
  q.floor = (float*)(b + n * 7 + 4096);
  return q;
}

// Threshold implied by a histogram: highest bin b whose suffix count reaches k'.  Every row whose
// score is below lo + (b-1)*width is provably in a bin < b (one bin of slack absorbs the fp32
// rounding of the bin computation), and at least k' appended rows sit in bins >= b, so dropping
// rows below the returned value never drops a top-k' row.  Returns -inf when no bin qualifies.
__host__ __device__ static inline float rarc_bin_threshold(int b, float lo, float width) {
  return (b >= 2) ? lo + (float)(b - 1) * width : -__builtin_inff();
}
__host__ __device__ static inline int rarc_bin_of(float s, float lo, float scale) {
  const float x = (s - lo) * scale;
  int b = (x >= (float)(RARC_NB - 1)) ? RARC_NB - 1 : (x > 0.f ? (int)x : 0);
  return b;
}

// Wave-level search used by the seed-threshold and finalize kernels: over `nbins` counters in LDS
// (nbins multiple of 64), find the highest bin b with sum(cnt[b..nbins)) >= need.  Called by ONE
// full wave (64 lanes); returns b (or -1) and the count strictly above b through *above.
#ifdef __HIPCC__
// lane id recomputed on the spot (volatile: never hoisted out of a rarely executed block, so it
// does not occupy a VGPR across a register-starved hot loop)
__device__ __forceinline__ int rarc_fresh_lane() {
  int l;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
  return l;
}
// LDS accesses the compiler must not see: with LDS-DMA (global_load_lds) in flight hipcc puts
// s_waitcnt vmcnt(0) in front of every DS access it knows about (possible alias with the DMA
// destination), which drains the whole HBM pipeline.  These touch disjoint bookkeeping words.
__device__ __forceinline__ uint32_t rarc_lds_add_rtn(uint32_t byte_addr, uint32_t v) {
  uint32_t old;
  asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(old) : "v"(byte_addr), "v"(v) : "memory");
  return old;
}
__device__ __forceinline__ float rarc_lds_read_f32(uint32_t byte_addr) {
  float x;
  asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(x) : "v"(byte_addr) : "memory");
  return x;
}
__device__ __forceinline__ void rarc_lds_read_u32x4(uint32_t byte_addr, uint32_t& a, uint32_t& b, uint32_t& c,
                                                    uint32_t& d) {
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  u32x4 v;
  asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(byte_addr) : "memory");
  a = v[0]; b = v[1]; c = v[2]; d = v[3];
}
// 256-bin variant of rarc_wave_find_from_top on register-held counts (lane holds bins 4*lane..+3)
__device__ static inline int rarc_wave_find_from_top_256(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                         uint32_t need) {
  const int lane = rarc_fresh_lane();
  const uint32_t mine = c0 + c1 + c2 + c3;
  uint32_t suf = mine;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    // (the source lane is derived from the fresh lane id: __shfl_down's own six lane addresses are loop invariants of the
    //  caller's scan loop, get hoisted out of it and cost six VGPRs there — or a spill — for a block that runs on wave 0 only)
    const uint32_t o = (uint32_t)__builtin_amdgcn_ds_bpermute(((lane + d) & 63) << 2, (int)suf);
    if (lane + d < 64) suf += o;
  }
  const unsigned long long ge = __builtin_amdgcn_ballot_w64(suf >= need);
  if (!ge) return -1;
  const int hl = 63 - __builtin_clzll(ge);
  const uint32_t above = __shfl(suf - mine, hl, 64);
  const uint32_t h3 = __shfl(c3, hl, 64), h2 = __shfl(c2, hl, 64), h1 = __shfl(c1, hl, 64);
  int b = 4 * hl;
  if (above + h3 >= need) b += 3;
  else if (above + h3 + h2 >= need) b += 2;
  else if (above + h3 + h2 + h1 >= need) b += 1;
  return b;
}
__device__ static inline int rarc_wave_find_from_top(const uint32_t* cnt, int nbins, uint32_t need,
                                                     uint32_t* above_out) {
  const int lane = rarc_fresh_lane();
  const int per = nbins / 64;
  const uint32_t* mine = cnt + lane * per;
  uint32_t tot = 0;
  for (int i = 0; i < per; ++i) tot += mine[i];
  uint32_t suf = tot;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t o = __shfl_down(suf, d, 64);
    if (lane + d < 64) suf += o;
  }
  const unsigned long long ge = __builtin_amdgcn_ballot_w64(suf >= need);
  if (!ge) return -1;
  const int hl = 63 - __builtin_clzll(ge);
  uint32_t above = __shfl(suf - tot, hl, 64);
  int b = -1;
  if (lane == hl) {
    for (int i = per - 1; i >= 0; --i) {
      const uint32_t c = mine[i];
      if (above + c >= need) { b = lane * per + i; break; }
      above += c;
    }
  }
  b = __shfl(b, hl, 64);
  *above_out = __shfl(above, hl, 64);
  return b;
}
#endif

// ---- order-preserving float <-> uint32 map ------------------------------------------------
__host__ __device__ static inline uint32_t rarc_ordkey(float f) {
  union { float f; uint32_t u; } x; x.f = f;
  return (x.u & 0x80000000u) ? ~x.u : (x.u | 0x80000000u);
}
__host__ __device__ static inline float rarc_unordkey(uint32_t k) {
  union { float f; uint32_t u; } x;
  x.u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
  return x.f;
}
// candidate key: larger == better (score desc, then row asc)
__host__ __device__ static inline uint64_t rarc_candkey(float score, uint32_t row) {
  return ((uint64_t)rarc_ordkey(score) << 32) | (uint64_t)(~row);
}
__host__ __device__ static inline uint32_t rarc_candrow(uint64_t key) { return ~(uint32_t)key; }
__host__ __device__ static inline float rarc_candscore(uint64_t key) {
  return rarc_unordkey((uint32_t)(key >> 32));
}

// ---- canonical fp32 inner product -----------------------------------------------------------
// score = tree(acc[0..7]) with acc[j] = fma chain over elements 8m+j, m ascending.
// The same order is used by oracle/rarc_oracle.c (canon_dot_*), so results are bit-identical.
__host__ __device__ static inline float rarc_canon_tree(const float a[8]) {
  return ((a[0] + a[4]) + (a[2] + a[6])) + ((a[1] + a[5]) + (a[3] + a[7]));
}


// ---- int8 prefilter: shared quantisation of an fp16 chunk ------------------------------------
// The q8 scan scores int8 images of the fp16 rows.  One 16-byte chunk (8 fp16 values of one row)
// becomes 8 int8 values d8 = RNE(x * s) with the tile's scale s (an fp16 number, x*s <= 127 by
// construction): fp16 fma(x, s, 1536) lands in [1409, 1663] where the fp16 ulp is 1, so the low byte
// of the result's bit pattern is the two's-complement int8 (1536 = 6*256 leaves the low byte alone).
// rarc_quant_meta_f16 (which derives the error bound) and the scan kernel both call this function,
// so they agree on every byte.
// Round 4 — MFMA result settle.  hipcc's hazard recognizer counts EVERY instruction between an MFMA and the first VALU read of
// its result as one wait state, an s_waitcnt too — and gfx950 retires an s_waitcnt whose counters are already satisfied without
// spending an issue cycle.  With one to three of them inside a window the compiler sized exactly (12 states for
// v_mfma_f32_32x32x16_f16: measured, tools/lab/mfma_wait_probe.hip), the read comes one to three cycles early and sees the old
// register: random wrong rows, at a rate that follows how quickly the LDS happened to answer (rarc_e32_attention_split_kernel,
// 1.5 % of forwards; a build with the accumulators in AGPRs: every forward).  Put this behind the LAST MFMA of a chain whose
// result VALU code consumes soon: five real wait states on top of whatever the compiler inserts (inline asm is not counted).
// tests/test_codeobj.py walks every kernel's listing and fails on a window that is short once s_waitcnt counts as zero.
// The pad is `s_nop 4` = five wait states: it covers a window with up to five free instructions in it (the worst found: three).
// Where the consumer is a register copy the compiler materialises after scheduling (accumulators leaving their AGPRs at a
// loop exit), put the pad INSIDE the loop, behind the last MFMA of the body: behind the loop the copies slip in front of it.
// (Not tied to the accumulator: with an in/out operand hipcc may copy the MFMA result into the operand's register first — a
//  read in front of the pad.  The two scheduling barriers keep the pad directly behind the MFMA and everything else behind it.)
#define RARC_MFMA_SETTLE(acc)                  \
  do {                                         \
    __builtin_amdgcn_sched_barrier(0);         \
    asm volatile("s_nop 4");                   \
    __builtin_amdgcn_sched_barrier(0);         \
  } while (0)

// What rarc_search_batch (rarc_api.hip, ABI 600) hands to the launchers underneath without widening every signature on the way:
// set for the duration of ONE call on the calling thread, cleared before it returns.
struct RarcLaunchExtras {
  uint32_t* status_zero = nullptr;  // device, RARC_MAX_QUERIES + 1 words: zeroed by the query-prep kernel (no fill launch in front)
  uint32_t* flag_host = nullptr;    // pinned host word: zeroed by the query-prep kernel, set non-zero by the finalize kernel of a
                                    // flagged query (no copy launch behind the search; visible to the host once the stream's
                                    // next event completes: a kernel's end releases its stores to system scope)
  hipEvent_t gate = nullptr;        // the first SCAN launch waits for this event (the query prep and the seed pass in front
                                    // of it do not): a neighbouring search context's finalize, see engine.py _PipelinedPair
};
RarcLaunchExtras& rarc_launch_extras();   // thread-local (rarc_api.hip)
// the scan launchers call this right before their first scan kernel
inline int rarc_gate_scan(hipStream_t s) {
  RarcLaunchExtras& x = rarc_launch_extras();
  if (x.gate) {
    hipEvent_t g = x.gate;
    x.gate = nullptr;
    if (hipStreamWaitEvent(s, g, 0) != hipSuccess) return -2;
  }
  return 0;
}

typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
#ifdef __HIPCC__
__device__ __forceinline__ uint2 rarc_quant8_chunk(const uint4 v, const half_t s) {
  const half2_t s2 = {s, s}, c2 = {(half_t)1536.f, (half_t)1536.f};
  const half2_t y0 = __builtin_elementwise_fma(__builtin_bit_cast(half2_t, v.x), s2, c2);
  const half2_t y1 = __builtin_elementwise_fma(__builtin_bit_cast(half2_t, v.y), s2, c2);
  const half2_t y2 = __builtin_elementwise_fma(__builtin_bit_cast(half2_t, v.z), s2, c2);
  const half2_t y3 = __builtin_elementwise_fma(__builtin_bit_cast(half2_t, v.w), s2, c2);
  uint2 o;
  o.x = __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, y1), __builtin_bit_cast(uint32_t, y0), 0x06040200u);
  o.y = __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, y3), __builtin_bit_cast(uint32_t, y2), 0x06040200u);
  return o;
}
#endif
// ---- fp8 (OCP e4m3fn) storage -------------------------------------------------------------------
// value[m] = rowscale * decode(byte[m]); same integer algorithm as oracle/rarc_oracle.c (f8_encode /
// f8_decode), so ingest is bit-identical to the oracle's.
__host__ __device__ static inline uint8_t rarc_f8_encode(float x) {
  union { float f; uint32_t u; } c; c.f = x;
  uint32_t u = c.u;
  const uint8_t sign = (uint8_t)((u >> 24) & 0x80);
  u &= 0x7fffffffu;
  c.u = u;
  const float a = c.f;
  if (!(a == a)) return sign;
  if (a >= 448.0f) return sign | 0x7e;
  if (a < 0.015625f) return sign | (uint8_t)(int)__builtin_rintf(a * 512.0f);
  uint32_t r = u + 0x0007ffffu + ((u >> 20) & 1u);
  r >>= 20;
  uint32_t code = r - ((127u - 7u) << 3);
  if (code > 0x7eu) code = 0x7eu;
  return sign | (uint8_t)code;
}
#ifdef __HIPCC__
typedef float float2_t __attribute__((ext_vector_type(2)));
// 4 fp8 in a dword -> 4 fp32 (exact; v_cvt_pk_f32_fp8)
__device__ __forceinline__ void rarc_f8x4_to_f32(uint32_t w, float (&o)[4]) {
  const float2_t lo = __builtin_amdgcn_cvt_pk_f32_fp8(w, false), hi = __builtin_amdgcn_cvt_pk_f32_fp8(w, true);
  o[0] = lo.x; o[1] = lo.y; o[2] = hi.x; o[3] = hi.y;
}
// One 16-byte chunk of an fp8 row (16 values) -> 16 int8 = RNE(value * mul): fp8 -> fp16 is exact
// (v_cvt_scalef32_pk_f16_fp8 with scale 1), then the same fma(x, mul, 1536) low-byte trick as
// rarc_quant8_chunk.  `mul` is the row's fp16 multiplier (row scale x tile scale, rounded down).
__device__ __forceinline__ uint4 rarc_quant8_chunk_f8(const uint4 v, const half_t mul) {
  const half2_t m2 = {mul, mul}, c2 = {(half_t)1536.f, (half_t)1536.f};
  const uint32_t w[4] = {v.x, v.y, v.z, v.w};
  uint32_t o[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const half2_t lo = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(w[i], 1.0f, false);
    const half2_t hi = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(w[i], 1.0f, true);
    const half2_t ylo = __builtin_elementwise_fma(lo, m2, c2), yhi = __builtin_elementwise_fma(hi, m2, c2);
    o[i] = __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, yhi), __builtin_bit_cast(uint32_t, ylo), 0x06040200u);
  }
  return make_uint4(o[0], o[1], o[2], o[3]);
}
#endif
constexpr int RARC_DIM_ALIGN_F8 = 256;  // fp8 rows: 16-byte chunks of 16 values, 512 threads per 32-row tile
// fp8 quantisation metadata: [0] R, [1..3] as below, then 34 floats per 32-row tile:
//   [4 + 34t] packed (s_t, R_t) word, [5 + 34t] 1/s_t, [6 + 34t + r] fp16-representable multiplier of row r of the tile
constexpr int RARC_QMETA_F8_STRIDE = 34;

// quantisation metadata (float array owned by the caller, see include/rarc.h):
//   [0] max over rows of ||d - d8/s||_2 (as float bits, raised by atomicMax)   [1..3] reserved
//   [4 + 2t] : one 32-bit word: low half = the fp16 bits of the scale s_t of 32-row tile t, high half = the fp16 bits
//              of R_t, the largest residual norm ||d - d8/s_t|| among the tile's rows, rounded UP (R = max R_t)
//   [5 + 2t] : 1/s_t (float)
constexpr int RARC_QMETA_HDR = 4;
constexpr int RARC_QMETA_STRIDE = 2;
static inline __host__ __device__ float rarc_tmeta_pack(uint16_t s_half_bits, uint16_t rt_half_bits) {
  const uint32_t w = ((uint32_t)rt_half_bits << 16) | (uint32_t)s_half_bits;
  return __builtin_bit_cast(float, w);
}

// ---- error plumbing (host) ------------------------------------------------------------------
void rarc_set_error(const char* fmt, ...);
#define RARC_HIP_CHECK(expr)                                                              \
  do {                                                                                    \
    hipError_t _e = (expr);                                                               \
    if (_e != hipSuccess) {                                                               \
      rarc_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__,     \
                     __LINE__);                                                           \
      return RARC_E_HIP;                                                                  \
    }                                                                                     \
  } while (0)
// hipFuncSetAttribute applies to the CURRENT device: a "done" flag is kept per (call site, device), so a process
// that drives several GPUs sets the attribute once on each (racing threads at worst set it twice: harmless).
struct RarcPerDevice {
  size_t v[64] = {};
  size_t& cur() {
    int d = 0;
    (void)hipGetDevice(&d);
    return v[(d < 0 ? 0 : d) & 63];
  }
};

#define RARC_REQUIRE(cond, code, ...)  \
  do {                                 \
    if (!(cond)) {                     \
      rarc_set_error(__VA_ARGS__);     \
      return (code);                   \
    }                                  \
  } while (0)

// ---- optional roctx ranges around the C-ABI entry points (rocprofv3 --marker-trace) -------------------------------
// RARC_ROCTX=1 resolves roctxRangePushA / roctxRangePop from the profiler SDK's library at first use (dlopen: no link
// dependency, nothing happens without the variable).  One range per entry point, named after it.
void rarc_roctx_push(const char* name);
void rarc_roctx_pop();
struct RarcRange {
  explicit RarcRange(const char* name) { rarc_roctx_push(name); }
  ~RarcRange() { rarc_roctx_pop(); }
  RarcRange(const RarcRange&) = delete;
  RarcRange& operator=(const RarcRange&) = delete;
};
#define RARC_RANGE() RarcRange rarc_range_guard_(__func__)
