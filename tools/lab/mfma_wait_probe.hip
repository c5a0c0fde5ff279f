// How many wait states does gfx950 need between an MFMA (last of a dependent chain) and a VALU read of its result, for each
// MFMA opcode this library uses?  hipcc's hazard recognizer counts EVERY instruction in between as one wait state, also an
// s_waitcnt — which the hardware retires without spending an issue cycle when its counters are already satisfied.  Everything
// sits in ONE asm block with fixed registers, so the compiler adds nothing: prefill the accumulator, run the chain, K
// `s_nop 0`, copy the last element.  Reported: how many reads saw the result too early.  (tools/mfma_wait_probe, built with
// hipcc --offload-arch=gfx950 -O2 -w; round 4: the cause of random wrong rows in rarc_e32_attention_split_kernel.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define STR2(x) #x
#define STR(x) STR2(x)
// OP: opcode string; NA / NB: VGPRs per A / B operand; ND: accumulator registers; FILL: operand pattern (fp16 1.0 pairs, int8 1s, fp32 1.0)
#define PROBE_KERNEL(NAME, OP, AREGS, BREGS, DLAST, DREGS, FILLA, PRE)                                                           \
  template <int K, int CHAIN>                                                                                                    \
  __global__ void NAME(float* out, int iters) {                                                                                  \
    float bad = 0.f;                                                                                                             \
    unsigned first = 0;                                                                                                          \
    for (int it = 0; it < iters; ++it) {                                                                                         \
      unsigned v;                                                                                                                \
      asm volatile(                                                                                                              \
          "v_mov_b32 v120, " FILLA "\n\tv_mov_b32 v121, " FILLA "\n\tv_mov_b32 v122, " FILLA "\n\tv_mov_b32 v123, " FILLA "\n\t" \
          "v_mov_b32 v124, " FILLA "\n\tv_mov_b32 v125, " FILLA "\n\tv_mov_b32 v126, " FILLA "\n\tv_mov_b32 v127, " FILLA "\n\t" \
          "v_mov_b32 " DLAST ", " PRE "\n\t"                                                                                    \
          "s_nop 15\n\ts_nop 15\n\t" OP " " DREGS ", " AREGS ", " BREGS ", 0\n\t"                                                \
          ".rept " STR(%c2) "\n\t" OP " " DREGS ", " AREGS ", " BREGS ", " DREGS "\n\t.endr\n\t"                                 \
          ".rept " STR(%c1) "\n\ts_nop 0\n\t.endr\n\t"                                                                           \
          "v_mov_b32 %0, " DLAST "\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\t"                                          \
          "v_mov_b32 v99, " DLAST                                                                                                \
          : "=v"(v) : "n"(K), "n"(CHAIN - 1)                                                                                     \
          : "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112",       \
            "v113", "v114", "v115", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127");                            \
      unsigned late;                                                                                                             \
      asm volatile("v_mov_b32 %0, v99" : "=v"(late));   /* the settled value, read 64 states later */                            \
      if (v != late) bad += 1.f;                                                                                                 \
      first = late;                                                                                                              \
    }                                                                                                                            \
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = bad;                                                                    \
    if (first == 0x12345u) out[0] = 1.f;                                                                                         \
  }
PROBE_KERNEL(p_f16_32, "v_mfma_f32_32x32x16_f16", "v[120:123]", "v[124:127]", "v115", "v[100:115]", "0x3c003c00", "-1.0")
PROBE_KERNEL(p_f16_16, "v_mfma_f32_16x16x32_f16", "v[120:123]", "v[124:127]", "v103", "v[100:103]", "0x3c003c00", "-1.0")
PROBE_KERNEL(p_i8_16, "v_mfma_i32_16x16x64_i8", "v[120:123]", "v[124:127]", "v103", "v[100:103]", "0x01010101", "-1")
PROBE_KERNEL(p_i8_32, "v_mfma_i32_32x32x32_i8", "v[120:123]", "v[124:127]", "v115", "v[100:115]", "0x01010101", "-1")
PROBE_KERNEL(p_f32_32, "v_mfma_f32_32x32x2_f32", "v120", "v124", "v115", "v[100:115]", "1.0", "-1.0")

template <typename F>
static double launch(F kern, float* d, int blocks) {
  const int iters = 1000, threads = 256;
  (void)hipMemset(d, 0, (size_t)blocks * threads * 4);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, d, iters);
  (void)hipDeviceSynchronize();
  std::vector<float> h((size_t)blocks * threads);
  (void)hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
  double bad = 0;
  for (float x : h) bad += x;
  return 100.0 * bad / ((double)h.size() * iters);
}

// MFMA -> MFMA that takes the first one's result as SrcC and writes a DIFFERENT destination (the fp8 scan's 2 x 2 blocks, the
// 256 x 128 GEMM's epilogue order): is that interlocked by the hardware?
#define PROBE_SRCC(NAME, OP, AREGS, BREGS, D1, D2, D2LAST, FILLA)                                                                \
  template <int K>                                                                                                              \
  __global__ void NAME(float* out, int iters) {                                                                                 \
    float bad = 0.f;                                                                                                            \
    for (int it = 0; it < iters; ++it) {                                                                                        \
      unsigned early, late;                                                                                                     \
      asm volatile(                                                                                                             \
          "v_mov_b32 v120, " FILLA "\n\tv_mov_b32 v121, " FILLA "\n\tv_mov_b32 v122, " FILLA "\n\tv_mov_b32 v123, " FILLA "\n\t" \
          "v_mov_b32 v124, " FILLA "\n\tv_mov_b32 v125, " FILLA "\n\tv_mov_b32 v126, " FILLA "\n\tv_mov_b32 v127, " FILLA "\n\t" \
          "s_nop 15\n\ts_nop 15\n\t" OP " " D1 ", " AREGS ", " BREGS ", 0\n\t"                                                  \
          ".rept " STR(%c2) "\n\ts_nop 0\n\t.endr\n\t"                                                                          \
          OP " " D2 ", " AREGS ", " BREGS ", " D1 "\n\t"                                                                        \
          "s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\t"                                                                    \
          "v_mov_b32 %0, " D2LAST "\n\t"                                                                                        \
          /* reference: the same with 64 wait states in between */                                                              \
          OP " " D1 ", " AREGS ", " BREGS ", 0\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\t"                              \
          OP " " D2 ", " AREGS ", " BREGS ", " D1 "\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\t"                         \
          "v_mov_b32 %1, " D2LAST                                                                                               \
          : "=v"(early), "=v"(late) : "n"(K)                                                                                    \
          : "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113",     \
            "v114", "v115", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v80", "v81", "v82", "v83", "v84",  \
            "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95");                                       \
      if (early != late) bad += 1.f;                                                                                            \
    }                                                                                                                           \
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = bad;                                                                   \
  }
PROBE_SRCC(c_i8_16, "v_mfma_i32_16x16x64_i8", "v[120:123]", "v[124:127]", "v[100:103]", "v[80:83]", "v83", "0x01010101")
PROBE_SRCC(c_f16_16, "v_mfma_f32_16x16x32_f16", "v[120:123]", "v[124:127]", "v[100:103]", "v[80:83]", "v83", "0x3c003c00")
PROBE_SRCC(c_f16_32, "v_mfma_f32_32x32x16_f16", "v[120:123]", "v[124:127]", "v[100:115]", "v[80:95]", "v95", "0x3c003c00")
#define SWEEPC(NAME, LABEL)                                                                                                    \
  for (int blocks : {256, 2048}) {                                                                                             \
    printf("%-26s result -> SrcC of another MFMA, %d wave(s)/SIMD, %% wrong at K wait states:", LABEL, blocks == 256 ? 1 : 8);  \
    printf(" K=0 %.2f", launch(NAME<0>, d, blocks)); printf(" | 1 %.2f", launch(NAME<1>, d, blocks));                           \
    printf(" | 2 %.2f", launch(NAME<2>, d, blocks)); printf(" | 3 %.2f", launch(NAME<3>, d, blocks));                           \
    printf(" | 4 %.2f", launch(NAME<4>, d, blocks)); printf(" | 6 %.2f", launch(NAME<6>, d, blocks));                           \
    printf(" | 8 %.2f", launch(NAME<8>, d, blocks)); printf(" | 12 %.2f\n", launch(NAME<12>, d, blocks));                       \
  }
#define SWEEP(NAME, LABEL)                                                                                                      \
  for (int blocks : {256, 2048}) {                                                                                              \
    printf("%-26s chain 3, %d wave(s)/SIMD, %% of reads too early at K wait states:", LABEL, blocks == 256 ? 1 : 8);             \
    printf(" K=2 %.2f", launch(NAME<2, 3>, d, blocks));   printf(" | 4 %.2f", launch(NAME<4, 3>, d, blocks));                    \
    printf(" | 5 %.2f", launch(NAME<5, 3>, d, blocks));   printf(" | 6 %.2f", launch(NAME<6, 3>, d, blocks));                    \
    printf(" | 7 %.2f", launch(NAME<7, 3>, d, blocks));   printf(" | 8 %.2f", launch(NAME<8, 3>, d, blocks));                    \
    printf(" | 10 %.2f", launch(NAME<10, 3>, d, blocks)); printf(" | 11 %.2f", launch(NAME<11, 3>, d, blocks));                  \
    printf(" | 12 %.2f", launch(NAME<12, 3>, d, blocks)); printf(" | 13 %.2f", launch(NAME<13, 3>, d, blocks));                  \
    printf(" | 16 %.2f", launch(NAME<16, 3>, d, blocks)); printf(" | 18 %.2f", launch(NAME<18, 3>, d, blocks));                  \
    printf(" | 19 %.2f", launch(NAME<19, 3>, d, blocks)); printf(" | 20 %.2f", launch(NAME<20, 3>, d, blocks));                  \
    printf(" | 22 %.2f\n", launch(NAME<22, 3>, d, blocks));                                                                      \
  }
int main() {
  float* d;
  (void)hipMalloc(&d, 1024 * 1024 * 4);
  SWEEP(p_f16_32, "v_mfma_f32_32x32x16_f16")
  SWEEP(p_f16_16, "v_mfma_f32_16x16x32_f16")
  SWEEP(p_i8_16, "v_mfma_i32_16x16x64_i8")
  SWEEP(p_i8_32, "v_mfma_i32_32x32x32_i8")
  SWEEP(p_f32_32, "v_mfma_f32_32x32x2_f32")
  SWEEPC(c_i8_16, "v_mfma_i32_16x16x64_i8")
  SWEEPC(c_f16_16, "v_mfma_f32_16x16x32_f16")
  SWEEPC(c_f16_32, "v_mfma_f32_32x32x16_f16")
  return 0;
}
