// rarc_common.h — shared device/host helpers for librarc_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/rarc.h"

typedef _Float16 half_t;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define RARC_GPTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define RARC_LPTR(p) ((__attribute__((address_space(3))) void*)(p))

// ---- workspace layout (bytes from a 256-byte aligned base) -------------------------------
// thr   : float   [256]        running per-query pruning threshold (monotone increasing)
// cnt   : uint32  [256]        candidates appended per query
// flags : uint32  [64]         [0] = any overflow
// hist  : uint32  [256][NB]    per-query score histogram of appended candidates
// cand  : uint64  [256][cap]   appended candidates, key = ordkey(score)<<32 | ~local_row
constexpr int RARC_NB = 256;  // histogram bins per query
constexpr size_t RARC_WS_THR = 0;
constexpr size_t RARC_WS_CNT = 1024;
constexpr size_t RARC_WS_FLAGS = 2048;
constexpr size_t RARC_WS_HIST = 4096;
constexpr size_t RARC_WS_CAND = RARC_WS_HIST + (size_t)RARC_MAX_QUERIES * RARC_NB * 4;

struct RarcWs {
  float* thr;
  uint32_t* cnt;
  uint32_t* flags;
  uint32_t* hist;
  uint64_t* cand;
};
static inline RarcWs rarc_ws_carve(void* base) {
  char* b = (char*)base;
  RarcWs w;
  w.thr = (float*)(b + RARC_WS_THR);
  w.cnt = (uint32_t*)(b + RARC_WS_CNT);
  w.flags = (uint32_t*)(b + RARC_WS_FLAGS);
  w.hist = (uint32_t*)(b + RARC_WS_HIST);
  w.cand = (uint64_t*)(b + RARC_WS_CAND);
  return w;
}

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
#define RARC_REQUIRE(cond, code, ...)  \
  do {                                 \
    if (!(cond)) {                     \
      rarc_set_error(__VA_ARGS__);     \
      return (code);                   \
    }                                  \
  } while (0)
