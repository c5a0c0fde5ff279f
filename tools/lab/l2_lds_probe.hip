// tools/lab/l2_lds_probe.hip — how fast can a CU move L2-resident operand tiles into LDS? (development tool)
// The 256 x 256 x 64 GEMM tile loop issues 64 LDS-DMA instructions (1 KiB each) per k tile per CU and its interior k tile takes
// ~2740 cycles = ~43 cycles per instruction (profiles/r03_gemm_seamless.txt): is that the vector-memory path's rate for ANY 1-KiB
// wave load, or the LDS-DMA form's?   MODE 0: global_load_lds_dwordx4;  MODE 1: global_load_dwordx4 -> VGPR -> ds_write_b128;
// MODE 2: global_load_dwordx4 -> VGPR only.  Every workgroup (8 waves) moves 64 KiB per iteration out of a region small enough to
// stay in L2 / MALL.  MODE 3: LDS-DMA with the GEMM's operand pattern — an instruction covers 8 rows x 128 B of a row-major [rows][stride]
// matrix (all workgroups at the same 128-B column, as the tile loops are), stride in bytes from argv[2].  build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/lab/l2_lds_probe.hip -o tools/l2_lds_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define GPTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define LPTR(p) ((__attribute__((address_space(3))) void*)(p))

template <int MODE>
__global__ __launch_bounds__(512) void probe(const char* src, size_t region, int iters, unsigned long long* cyc, uint32_t* sink, int stride) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t blocks = region / 65536;
  unsigned long long t0 = 0;
  u32x4 keep = {0, 0, 0, 0};
  for (int it = -2; it < iters; ++it) {
    if (it == 0) { __syncthreads(); t0 = __builtin_amdgcn_s_memtime(); }
    const size_t blk = ((size_t)blockIdx.x * 7 + (size_t)(it + 2) * 13) % blocks;
    const char* p = src + blk * 65536 + wave * 8192 + lane * 16;
    if (MODE == 3) {
      // 512 rows of one 128-B column per iteration and workgroup: wave w rows [64 w, 64 w + 64), instruction j rows 8 j .. 8 j + 7
      const size_t rows = region / (size_t)stride, cols = (size_t)stride / 128;
      const size_t r0 = (((size_t)blockIdx.x * 7) % (rows / 512 ? rows / 512 : 1)) * (rows >= 512 ? 512 : 0) + wave * 64 + (lane >> 3);
      const char* ps = src + r0 * (size_t)stride + ((size_t)(it + 2) % cols) * 128 + (lane & 7) * 16;
#pragma unroll
      for (int j = 0; j < 8; ++j)
        __builtin_amdgcn_global_load_lds(GPTR(ps + (size_t)j * 8 * stride), LPTR(smem + wave * 8192 + j * 1024), 16, 0, 0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else if (MODE == 0) {
#pragma unroll
      for (int j = 0; j < 8; ++j)
        __builtin_amdgcn_global_load_lds(GPTR(p + j * 1024), LPTR(smem + wave * 8192 + j * 1024), 16, 0, 0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      u32x4 r[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r[j]) : "v"(p + j * 1024) : "memory");
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (j == 0) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
        if (j == 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        if (j == 2) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        if (j == 3) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        if (j == 4) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        if (j == 5) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        if (j == 6) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        if (j == 7) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (MODE == 1)
          asm volatile("ds_write_b128 %0, %1" ::"v"((uint32_t)(wave * 8192 + j * 1024 + lane * 16)), "v"(r[j]) : "memory");
        else
          asm volatile("v_xor_b32 %0, %0, %1" : "+v"(keep[0]) : "v"(r[j][0]));
      }
      if (MODE == 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  }
  __syncthreads();
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
  if (MODE != 2) keep[0] ^= ((const uint32_t*)smem)[threadIdx.x];
  if (keep[0] == 0x12345678u) sink[0] = keep[0];
}

template <int MODE>
static void run(const char* name, const char* src, size_t region, int iters, unsigned long long* d_cyc, uint32_t* sink, int stride = 2048) {
  hipFuncSetAttribute((const void*)probe<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e30f; double cycs = 0;
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(512), 65536, 0, src, region, iters, d_cyc, sink, stride);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) {
      best = ms;
      unsigned long long h[256]; hipMemcpy(h, d_cyc, sizeof(h), hipMemcpyDeviceToHost);
      cycs = 0; for (int i = 0; i < 256; ++i) cycs += (double)h[i]; cycs /= 256;
    }
  }
  const double bytes = 256.0 * iters * 65536;
  printf("%-44s region %6.1f MiB: %8.1f us  %6.2f TB/s  %6.1f cycles per 1-KiB wave instruction (per CU)  %5.1f B/clk/CU\n", name,
         region / 1048576.0, best * 1e3, bytes / best * 1e-9, cycs / (iters * 64.0), 65536.0 * iters / cycs);
  fflush(stdout);
}

int main(int argc, char** argv) {
  const int only = argc > 1 ? atoi(argv[1]) : -1;
  const size_t maxr = (size_t)256 << 20;
  char* src; hipMalloc(&src, maxr); hipMemset(src, 1, maxr);
  unsigned long long* d_cyc; hipMalloc(&d_cyc, 256 * 8);
  uint32_t* sink; hipMalloc(&sink, 4);
  for (size_t region : {(size_t)1 << 20, (size_t)16 << 20, (size_t)128 << 20, (size_t)2 << 20, (size_t)8 << 20}) {
    if (only < 0 || only == 0) run<0>("global_load_lds_dwordx4 (LDS-DMA)", src, region, 400, d_cyc, sink);
    if (only < 0 || only == 1) run<1>("global_load_dwordx4 -> VGPR -> ds_write_b128", src, region, 400, d_cyc, sink);
    if (only < 0 || only == 2) run<2>("global_load_dwordx4 -> VGPR", src, region, 400, d_cyc, sink);
    if (only == 3 && region != ((size_t)1 << 20)) for (int stride : {2048, 4096, 8192, 2048 + 128, 8192 + 128}) {
      char nm[64]; snprintf(nm, sizeof nm, "LDS-DMA, 8 rows x 128 B, row stride %d", stride);
      run<3>(nm, src, region, 400, d_cyc, sink, stride);
    }
  }
  return 0;
}
