// tools/lab/small_bench.hip — steady-state timing of the small pipeline kernels (development tool).
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include "../include/rarc.h"
__global__ void empty_kernel(int* p) { if (p && threadIdx.x == 9999) *p = 1; }
int main() {
  const int64_t N = 1000000; const int D = 768, NQ = 256, K = 100, KP = 128, CAP = 16384;
  uint16_t* corpus; float *qin, *osc; int64_t* oid; uint32_t* st; void *ws, *qblock;
  hipMalloc(&corpus, (size_t)(N + 32) * D * 2); hipMalloc(&qblock, rarc_query_block_bytes(D));
  hipMalloc(&qin, 256 * D * 4); hipMalloc(&osc, 256 * K * 4); hipMalloc(&oid, 256 * K * 8); hipMalloc(&st, 1028); hipMemset(st, 0, 1028);
  size_t wsb = rarc_search_workspace_bytes(CAP); hipMalloc(&ws, wsb);
  rarc_synth_rows_f16(corpus, D, D, 0, N, 1234, 0); rarc_synth_rows_f32(qin, D, D, 0, NQ, 4321, 0);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto timeit = [&](const char* name, auto fn, int reps) {
    fn(); hipDeviceSynchronize();
    hipEventRecord(e0, 0); for (int i = 0; i < reps; ++i) fn(); hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); printf("%-28s %8.1f us per call (%d back-to-back)\n", name, ms * 1e3 / reps, reps);
  };
  timeit("empty 256x256", [&] { hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(256), 0, 0, (int*)nullptr); }, 50);
  timeit("prep_queries", [&] { rarc_prep_queries(qin, D, NQ, D, D, 1, 1.001f, nullptr, qblock, 0); }, 50);
  timeit("search (all kernels)", [&] { rarc_search_f16(corpus, N, D, nullptr, qblock, NQ, K, KP, 0, -1.f, 1.f, oid, osc, st, ws, wsb, CAP, 0); }, 20);
  timeit("prep+search", [&] { rarc_prep_queries(qin, D, NQ, D, D, 1, 1.001f, nullptr, qblock, 0); rarc_search_f16(corpus, N, D, nullptr, qblock, NQ, K, KP, 0, -1.f, 1.f, oid, osc, st, ws, wsb, CAP, 0); }, 20);
  // tiny corpus: the scan is negligible, what remains is seed + seed_thr + finalize + launch gaps
  timeit("search on 4096 rows", [&] { rarc_search_f16(corpus, 4096, D, nullptr, qblock, NQ, K, KP, 0, -1.f, 1.f, oid, osc, st, ws, wsb, CAP, 0); }, 50);
  timeit("search on 65536 rows", [&] { rarc_search_f16(corpus, 65536, D, nullptr, qblock, NQ, K, KP, 0, -1.f, 1.f, oid, osc, st, ws, wsb, CAP, 0); }, 50);
  std::vector<uint32_t> h(256); hipMemcpy(h.data(), st, 1024, hipMemcpyDeviceToHost); int bad = 0; for (auto v : h) bad += v != 0; printf("flagged queries: %d\n", bad);
  return 0;
}
