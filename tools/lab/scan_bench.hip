// tools/lab/scan_bench.hip — ablation timing of the scan kernel (development tool, not product).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/lab/scan_bench.hip -o tools/scan_bench
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "../rag-arc_amd/csrc/scan_f16.hip"
#include "../rag-arc_amd/csrc/prep.hip"
void rarc_set_error(const char* fmt, ...) { (void)fmt; }
void rarc_roctx_push(const char*) {}
void rarc_roctx_pop() {}
bool rarc_prof_next(hipEvent_t*, hipEvent_t*) { return false; }

template <int ABL>
static float run(const ScanParams& p, int grid, int iters, uint32_t nq, int kprime, RarcWs ws, uint32_t seed_tiles) {
  constexpr int D = 768;
  constexpr size_t lds = ScanLds<D>::TOTAL + ((ABL & 64) ? 8192 : 0);
  hipFuncSetAttribute((const void*)rarc_scan_f16_kernel<D, ABL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e30f;
  for (int it = 0; it < iters; ++it) {
    hipLaunchKernelGGL((rarc_seed_kernel<D, 0>), dim3(8, seed_tiles), dim3(256), 0, 0, (const void*)p.corpus, (const float*)nullptr, p.q16, p.n_rows, p.n_tiles, seed_tiles, ws.seed);
    hipLaunchKernelGGL(rarc_seed_thr_kernel, dim3(256), dim3(1024), 0, 0, ws.seed, seed_tiles * 32, (uint32_t)kprime, nq, -1.f, 1.f, (const float*)nullptr, (const float*)nullptr, (uint32_t*)ws.thr, ws.binlo, ws.binscale, ws.bininv, ws.flags, ws.hist, (const float*)nullptr, (const float*)nullptr);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((rarc_scan_f16_kernel<D, ABL>), dim3(grid), dim3(512), lds, 0, p);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  return best * 1000.f;
}

int main(int argc, char** argv) {
  const int64_t N = argc > 1 ? atoll(argv[1]) : 1000000;
  const int D = 768, NQ = 256, KP = 128, CAP = 16384;
  half_t *corpus, *q16; void* wsb;
  hipMalloc(&corpus, (size_t)(N + 32) * D * 2); hipMalloc(&q16, 256 * D * 2);
  size_t wsbytes = RARC_WS_CAND + (size_t)256 * CAP * 8; hipMalloc(&wsb, wsbytes);
  rarc_synth_rows_f16((uint16_t*)corpus, D, D, 0, N, 1234, 0);
  { float* qf; hipMalloc(&qf, 256 * D * 4); rarc_synth_rows_f32(qf, D, D, 0, 256, 4321, 0);
    void* qblock; hipMalloc(&qblock, rarc_qb_bytes(D));
    rarc_prep_queries(qf, D, 256, D, D, 1, 1.001f, nullptr, qblock, 0); hipDeviceSynchronize();
    hipMemcpy(q16, rarc_qb_carve(qblock, D).q16, 256 * D * 2, hipMemcpyDeviceToDevice); }
  RarcWs ws = rarc_ws_carve(wsb);
  ScanParams p; p.corpus = corpus; p.q16 = q16; p.n_rows = (uint32_t)N; p.n_tiles = (uint32_t)((N + 31) / 32);
  p.thr = (uint32_t*)ws.thr; p.hist = ws.hist; p.cnt2 = ws.cnt2; p.cand = ws.cand; p.seg = CAP / 256; p.kprime = KP; p.nq = NQ;
  p.binlo = ws.binlo; p.binscale = ws.binscale; p.bininv = ws.bininv;
  unsigned long long* dbg; hipMalloc(&dbg, 64 * 8 * 8 * 8); hipMemset(dbg, 0, 64 * 8 * 8 * 8); p.dbg = dbg;
  int grid = 256; uint32_t st = p.n_tiles < 128 ? p.n_tiles : 128;
  const double gb = (double)N * D * 2 / 1e9;
#define RUN(A) { float us = run<A>(p, grid, 8, NQ, KP, ws, st); printf("ABL=%2d  %8.1f us  %6.2f TB/s\n", A, us, gb / us * 1e-3); }
  RUN(0) RUN(0) RUN(1) RUN(2) RUN(3)
  run<64>(p, grid, 2, NQ, KP, ws, st);
  { std::vector<unsigned long long> h(64 * 8 * 8); hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost);
    printf("timeline of workgroup 0 (s_memtime ticks, 100 MHz => x10 ns), relative to wave 0 stamp 0 of each iteration\n");
    printf("iter wave:  barrier->stamp1 ->mfma_start ->mfma_end ->pruned(A) ->events_done ->dma_landed | next barrier\n");
    for (int it = 56; it < 60; ++it) for (int w = 0; w < 8; ++w) { const unsigned long long* r = &h[(it * 8 + w) * 8]; unsigned long long b0 = h[(it * 8) * 8];
      unsigned long long nb = h[((it + 1) * 8 + w) * 8];
      printf("%3d %d: %6lld %6lld %6lld %6lld %6lld %6lld %6lld | %6lld\n", it, w, (long long)(r[0] - b0), (long long)(r[1] - b0), (long long)(r[2] - b0), (long long)(r[3] - b0), (long long)(r[4] - b0), (long long)(r[5] - b0), (long long)(r[6] - b0), (long long)(nb - b0)); } }
  run<0>(p, grid, 1, NQ, KP, ws, st);
  std::vector<uint32_t> cnt(256 * 256); hipMemcpy(cnt.data(), ws.cnt2, 256 * 256 * 4, hipMemcpyDeviceToHost);
  uint64_t tot = 0; uint32_t mx = 0; for (auto c : cnt) { tot += c; mx = c > mx ? c : mx; }
  printf("mean candidates/query: %.0f   max per (wg,query) segment: %u\n", tot / 256.0, mx);
  return 0;
}
