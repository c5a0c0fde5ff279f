// tools/lab/mfma_probe.hip — micro-probes for the scan's MFMA loop (development tool).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// MODE 0: 48 dependent MFMAs per "tile" on ONE accumulator, operands in registers only
// MODE 1: two accumulators alternating
// MODE 2: like 0 plus one ds_read_b128 per MFMA into a ring (no waits on it: timing only)
// MODE 3: like 2 with counted lgkmcnt(7) waits (the product pattern), LDS filled with zeros
// MODE 4: MODE 3 with 2 accumulators
template <int MODE, int BAR>
__global__ __launch_bounds__(512, 2) void probe(const _Float16* q, float* out, int tiles, int zero) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  half8 qf[48];
#pragma unroll
  for (int i = 0; i < 48; ++i) qf[i] = *(const half8*)(q + (size_t)(lane + 64 * i) * 8);
  half8 rg[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) rg[i] = qf[i];
  for (int i = threadIdx.x; i < 48 * 1024 / 2; i += blockDim.x) ((_Float16*)smem)[i] = zero ? (_Float16)0.f : q[(i * 7 + 13) % (64 * 48 * 8)];
  __syncthreads();
  const int row = lane & 31, h = lane >> 5, sw = (row >> 1) & 7;
  int ad[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) ad[kk] = row * 128 + (((2 * kk + h) ^ sw) << 4);
  const int addr = ad[0];
  f32x16 a0 = {0}, a1 = {0};
  for (int t = 0; t < tiles; ++t) {
    if (MODE >= 2) {
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(rg[i]) : "v"(ad[i & 3]), "n"(0));
    }
#pragma unroll
    for (int s = 0; s < 48; ++s) {
      if (MODE == 0) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(a0) : "v"(rg[s & 7]), "v"(qf[s]));
      if (MODE == 1) {
        if (s & 1) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(a1) : "v"(rg[s & 7]), "v"(qf[s]));
        else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(a0) : "v"(rg[s & 7]), "v"(qf[s]));
      }
      if (MODE == 2) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0\n\tds_read_b128 %1, %3 offset:4096" : "+v"(a0), "+v"(rg[s & 7]) : "v"(qf[s]), "v"(ad[s & 3]));
      if (MODE == 3) asm volatile("s_waitcnt lgkmcnt(7)\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0\n\tds_read_b128 %1, %3 offset:4096" : "+v"(a0), "+v"(rg[s & 7]) : "v"(qf[s]), "v"(ad[s & 3]));
      if (MODE == 4) {
        if (s & 1) asm volatile("s_waitcnt lgkmcnt(7)\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0\n\tds_read_b128 %1, %3 offset:4096" : "+v"(a1), "+v"(rg[s & 7]) : "v"(qf[s]), "v"(ad[s & 3]));
        else asm volatile("s_waitcnt lgkmcnt(7)\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0\n\tds_read_b128 %1, %3 offset:4096" : "+v"(a0), "+v"(rg[s & 7]) : "v"(qf[s]), "v"(ad[s & 3]));
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (BAR == 1) __builtin_amdgcn_s_barrier();
    if (BAR == 2) { asm volatile("s_nop 15\n\ts_nop 3" : "+v"(a0)); float m = a0[0]; 
      _Pragma("unroll") for (int i = 1; i < 16; ++i) m = fmaxf(m, a0[i]);
      if (__builtin_amdgcn_ballot_w64(m > 1e30f)) out[threadIdx.x] = m; __builtin_amdgcn_s_barrier(); }
  }
  asm volatile("s_nop 15\n\ts_nop 3" : "+v"(a0), "+v"(a1));
  float r = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) r += a0[i] + a1[i];
  out[blockIdx.x * 512 + threadIdx.x] = r;
}

template <int MODE, int BAR>
static void run(const _Float16* q, float* out, int tiles, int waves_per_simd, int zero) {
  hipFuncSetAttribute((const void*)probe<MODE, BAR>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e30f;
  for (int it = 0; it < 4; ++it) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((probe<MODE, BAR>), dim3(256), dim3(waves_per_simd * 256), 150 * 1024, 0, q, out, tiles, zero);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
  }
  const double n_mfma_per_simd = (double)tiles * 48 * waves_per_simd;
  const double tf = 256.0 * 4 * n_mfma_per_simd * 32768.0 / (best * 1e-3) / 1e12;
  printf("zero=%d MODE %d BAR %d waves/SIMD %d: %8.1f us  -> %.0f TF/s, %.1f ns per MFMA per SIMD\n", zero, MODE, BAR, waves_per_simd, best * 1e3, tf, best * 1e6 / n_mfma_per_simd);
}

int main() {
  _Float16* q; float* out;
  hipMalloc(&q, 64 * 48 * 16); hipMalloc(&out, 256 * 512 * 4);
  { _Float16* h = (_Float16*)malloc(64 * 48 * 16); uint32_t st = 99; for (int i = 0; i < 64 * 48 * 8; ++i) { float a = 0; for (int j = 0; j < 4; ++j) { st = st * 1664525u + 1013904223u; a += (float)(st >> 8) / 16777216.f - 0.5f; } h[i] = (_Float16)(a * 0.0625f); } hipMemcpy(q, h, 64 * 48 * 16, hipMemcpyHostToDevice); }
  const int tiles = 122;
  for (int z = 1; z >= 0; --z) { run<0, 0>(q, out, tiles, 2, z); run<3, 0>(q, out, tiles, 2, z); run<3, 2>(q, out, tiles, 2, z); run<3, 2>(q, out, tiles, 1, z); }
  return 0;
}
