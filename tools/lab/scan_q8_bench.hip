// tools/lab/scan_q8_bench.hip — ablation timing of the int8-prefilter scan kernel (development tool).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/lab/scan_q8_bench.hip -o tools/scan_q8_bench
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "../rag-arc_amd/csrc/scan_f16.hip"
#include "../rag-arc_amd/csrc/scan_q8.hip"
#include "../rag-arc_amd/csrc/quant.hip"
#include "../rag-arc_amd/csrc/prep.hip"
#include "../rag-arc_amd/csrc/finalize.hip"
void rarc_set_error(const char* fmt, ...) { (void)fmt; }
void rarc_roctx_push(const char*) {}
void rarc_roctx_pop() {}
bool rarc_prof_next(hipEvent_t*, hipEvent_t*) { return false; }

#ifndef BD
#define BD 768
#endif
template <int ABL>
static float run(const ScanQ8Params& p, int grid, int iters, const uint16_t* corpus, int64_t N, const RarcQb& qb,
                 int kprime, RarcWs ws) {
  constexpr int D = BD;
  constexpr size_t lds = ScanQ8Lds<D>::TOTAL;
  hipFuncSetAttribute((const void*)rarc_scan_q8_kernel<D, 0, ABL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e30f;
  for (int it = 0; it < iters; ++it) {
    rarc_seed_launch(corpus, nullptr, 0, N, D, qb.q16, 256, kprime, -1.f, 1.f, qb.eps16, qb.eps8, ws, 0, 0, nullptr);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((rarc_scan_q8_kernel<D, 0, ABL>), dim3(grid), dim3(Q8_THREADS), lds, 0, p);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  { unsigned long long h[2]; hipMemcpy(h, p.dbg, 16, hipMemcpyDeviceToHost); printf("  [wg0: %.0f kcycles, clock %.0f MHz] ", h[0] / 1e3, h[1] ? 100.0 * h[0] / h[1] : 0.0); }
  return best * 1000.f;
}

int main(int argc, char** argv) {
  const int64_t N = argc > 1 ? atoll(argv[1]) : 1000000;
  const int K = argc > 2 ? atoi(argv[2]) : 10, KP = K;  // (the engine passes k' = k on this path)
  const int D = BD, NQ = 256, CAP = 131072;
  uint16_t* corpus; void* wsb; void* qblock; float* qmeta;
  hipMalloc(&corpus, (size_t)(N + 32) * D * 2); hipMemset(corpus, 0, (size_t)(N + 32) * D * 2);
  size_t wsbytes = RARC_WS_CAND + (size_t)256 * CAP * 8; hipMalloc(&wsb, wsbytes);
  hipMalloc(&qblock, rarc_qb_bytes(D));
  size_t nm = rarc_quant_meta_floats(N); hipMalloc(&qmeta, nm * 4); hipMemset(qmeta, 0, nm * 4);
  rarc_synth_rows_f16(corpus, D, D, 0, N, 1234, 0);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0, 0);
  rarc_quant_meta_f16(corpus, N, D, 0, qmeta, 0);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  { float ms; hipEventElapsedTime(&ms, e0, e1); printf("quant_meta: %.1f us (%.2f TB/s)\n", ms * 1000, (double)N * D * 2 / ms * 1e-9); }
  { float* qf; hipMalloc(&qf, 256 * D * 4); rarc_synth_rows_f32(qf, D, D, 0, 256, 4321, 0);
    rarc_prep_queries(qf, D, 256, D, D, 1, 1.001f, qmeta, qblock, 0); hipDeviceSynchronize(); }
  RarcQb qb = rarc_qb_carve(qblock, D);
  { float h[4]; hipMemcpy(h, qmeta, 16, hipMemcpyDeviceToHost); float e8[4], e16[4]; hipMemcpy(e8, qb.eps8, 16, hipMemcpyDeviceToHost); hipMemcpy(e16, qb.eps16, 16, hipMemcpyDeviceToHost);
    printf("R = %.5f  eps8[0..1] = %.5f %.5f  eps16[0] = %.6f\n", h[0], e8[0], e8[1], e16[0]); }
  RarcWs ws = rarc_ws_carve(wsb);
  ScanQ8Params p; p.corpus = (const uint4*)corpus; p.tmeta = qmeta + RARC_QMETA_HDR; p.q8 = qb.q8; p.qinv = qb.qinv; p.eps8 = qb.eps8; p.hq = qb.hq;
  p.n_rows = (uint32_t)N; p.n_tiles = (uint32_t)((N + 31) / 32); p.t_begin = 0; p.resume = 0; p.hot_margin = 1.0f;
  p.thr = (uint32_t*)ws.thr; p.hist = ws.hist; p.cnt2 = ws.cnt2; p.cand = ws.cand; p.seg = CAP / 256; p.kprime = KP; p.nq = NQ;
  p.binlo = ws.binlo; p.binscale = ws.binscale; p.bininv = ws.bininv;
  { unsigned long long* d; hipMalloc(&d, 65536); hipMemset(d, 0, 65536); p.dbg = d; }
  int grid = 256;
  const double gb = (double)N * D * 2 / 1e9;
#define RUN(A) { float us = run<A>(p, grid, 6, corpus, N, qb, KP, ws); printf("ABL=%2d  %8.1f us  %6.2f TB/s\n", A, us, gb / us * 1e3); }
  RUN(0) RUN(0) RUN(256) RUN(1) RUN(257)
  for (int v = 0; v < 1; ++v) { if (v == 0) run<1024>(p, grid, 1, corpus, N, qb, KP, ws); else run<1025>(p, grid, 1, corpus, N, qb, KP, ws);
    std::vector<unsigned long long> h(8192); hipMemcpy(h.data(), p.dbg, 65536, hipMemcpyDeviceToHost);
    printf("\ntimeline wg0 (%s), shader cycles relative to wave0 stamp0 of the iteration:  start  mfma+conv  pruned  fetched  barrier_out | next_start\n", v ? "no prune" : "full");
    for (int it = 4; it < 8; ++it) for (int w = 0; w < 8; ++w) { const unsigned long long* r = &h[8 + (it * 8 + w) * 8]; unsigned long long b0 = h[8 + (it * 8) * 8]; unsigned long long nb = h[8 + ((it + 1) * 8 + w) * 8];
      printf("%2d w%d: %6lld %6lld %6lld %6lld %6lld | %6lld\n", it, w, (long long)(r[0] - b0), (long long)(r[1] - b0), (long long)(r[2] - b0), (long long)(r[3] - b0), (long long)(r[4] - b0), (long long)(nb - b0)); } }
  run<0>(p, grid, 1, corpus, N, qb, KP, ws);
  { // finalize timing
    int64_t* oi; float* os; uint32_t* st; hipMalloc(&oi, 256 * 1024 * 8); hipMalloc(&os, 256 * 1024 * 4); hipMalloc(&st, 257 * 4); hipMemset(st, 0, 257 * 4);
    unsigned long long* fd; hipMalloc(&fd, 16384); hipMemset(fd, 0, 16384); g_fin8_dbg = fd;
    hipEvent_t f0, f1; hipEventCreate(&f0); hipEventCreate(&f1);
    for (int rep = 0; rep < 6; ++rep) {
      if (rep >= 3) { run<0>(p, grid, 1, corpus, N, qb, KP, ws); printf("(after a scan) "); }
      hipEventRecord(f0, 0);
      rarc_finalize_q8_launch(corpus, nullptr, 0, D, qb.q32, qb.eps8, 256, K, 0, ws, CAP, grid, oi, os, st, 0, false, qmeta, qb.hq);
      hipEventRecord(f1, 0); hipEventSynchronize(f1); float ms; hipEventElapsedTime(&ms, f0, f1);
      unsigned long long h[16 + 1024]; hipMemcpy(h, fd, sizeof(h), hipMemcpyDeviceToHost);
      { double s1 = 0, s2 = 0; unsigned long long m1 = 0, m2 = 0, tmax = 0; int qmax = 0;
        for (int q = 0; q < 256; ++q) { s1 += h[16 + 4 * q]; s2 += h[17 + 4 * q]; if (h[16 + 4 * q] > m1) m1 = h[16 + 4 * q]; if (h[17 + 4 * q] > m2) m2 = h[17 + 4 * q];
          if (h[18 + 4 * q] > tmax) { tmax = h[18 + 4 * q]; qmax = q; } }
        { int qb2 = 0; for (int q = 0; q < 256; ++q) if (h[16 + 4 * q] > h[16 + 4 * qb2]) qb2 = q;
          unsigned long long v = h[19 + 4 * qb2]; unsigned int a = (unsigned int)(v >> 32), b = (unsigned int)v; float t1f, Lf; memcpy(&t1f, &a, 4); memcpy(&Lf, &b, 4);
          float th[256]; hipMemcpy(th, ws.thr, 1024, hipMemcpyDeviceToHost); float lo[256]; hipMemcpy(lo, ws.binlo, 1024, hipMemcpyDeviceToHost); float bi[256]; hipMemcpy(bi, ws.bininv, 1024, hipMemcpyDeviceToHost);
          printf("  worst query %d: |G1| %llu  T1 %.5f  L %.5f  thr %.5f  binlo %.5f  binwidth %.6f\n", qb2, h[16 + 4 * qb2], t1f, Lf, th[qb2], lo[qb2], bi[qb2]); }
        printf("  |G1| mean %.0f max %llu   |G1+G2| mean %.0f max %llu   last block q=%d ends %.1f us after block0 start (its |G1+G2| = %llu)\n", s1 / 256, m1, s2 / 256, m2, qmax, (tmax - h[0]) / 100.0, h[17 + 4 * qmax]); }
      printf("finalize: %.1f us; block0 phases (us): init %.1f collect1 %.1f rescore1 %.1f rankL %.1f collect2 %.1f rescore2 %.1f final %.1f; |G1|=%llu |G1+G2|=%llu\n", ms * 1000,
             (h[1]-h[0])/100.0, (h[2]-h[1])/100.0, (h[3]-h[2])/100.0, (h[4]-h[3])/100.0, (h[5]-h[4])/100.0, (h[6]-h[5])/100.0, (h[7]-h[6])/100.0, h[8], h[9]);
    }
  }
  std::vector<uint32_t> cnt(256 * 256); hipMemcpy(cnt.data(), ws.cnt2, 256 * 256 * 4, hipMemcpyDeviceToHost);
  uint64_t tot = 0; uint32_t mx = 0; for (auto c : cnt) { tot += c; mx = c > mx ? c : mx; }
  printf("mean candidates/query: %.0f   max per (wg,query) segment: %u (seg %u)\n", tot / 256.0, mx, p.seg);
  return 0;
}
