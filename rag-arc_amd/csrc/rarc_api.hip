// rarc_api.hip — C-ABI entry points that chain the kernels (see include/rarc.h).
#include <stdarg.h>
#include <stdio.h>
#include "rarc_common.h"

int rarc_scan_f16_launch(const uint16_t* corpus, int64_t n_rows, int d_pad, const uint16_t* q16, int nq,
                         int kprime, float bin_lo, float bin_hi, const RarcWs& ws, int cap, int* grid_out,
                         hipStream_t s, const float* floor, const float* floor_eps);
int rarc_finalize_launch(const uint16_t* corpus, int d_pad, const float* q32, const float* eps, int nq,
                         int k, int kprime, int64_t id_base, const RarcWs& ws, int cap, int n_wg,
                         int64_t* out_ids, float* out_scores, uint32_t* status, hipStream_t s);
int rarc_scan_q8_launch(const void* corpus, const float* rowscale, int fmt, int64_t n_rows, int d_pad,
                        const float* qmeta, const uint16_t* q16, const int8_t* q8, const float* qinv,
                        const float* eps16, const float* eps8, int nq, int kprime, float bin_lo, float bin_hi,
                        const RarcWs& ws, int cap, int* grid_out, hipStream_t s, const int8_t* shadow8,
                        int (*tighten)(void* ctx, int n_wg, int mode), void* tighten_ctx, const float* hq, const float* floor,
                        int* hybrid_out);
int rarc_finalize_q8_launch(const void* corpus, const float* rowscale, int fmt, int d_pad, const float* q32,
                            const float* eps8, int nq, int k, int64_t id_base, const RarcWs& ws, int cap, int n_wg,
                            int64_t* out_ids, float* out_scores, uint32_t* status, hipStream_t s, int tighten,
                            const float* qmeta, const float* hq, const float* eps16);

// int8-prefilter search = seed, scan (split in two launches around an exact mid-scan pass on large shards),
// canonical finalize; shared by the fp16, fp8 and shadow-image entry points
struct Q8Search {
  const void* rows;       // what the finalize rescans: fp16 rows, or fp8 bytes
  const float* rowscale;  // fp8 only
  int fin_fmt;            // row format of `rows` for the finalize (0 fp16, 1 fp8, 2 fp32)
  int d_pad, nq, k, cap;
  int64_t id_base;
  const float *q32, *eps8;
  const float *qmeta, *hq;  // per-tile error bounds for the finalize (finalize.hip: reaches())
  const float* eps16;       // the fp16 scorer's bound (hybrid search: the first stage's candidates carry fp16 scores)
  const RarcWs* ws;
  int64_t* out_ids;
  float* out_scores;
  uint32_t* status;
  hipStream_t s;
};
static int q8_tighten(void* ctx, int n_wg, int mode) {   // mode 1: raise thr; 2: set it (after the fp16 stage of a hybrid search)
  const Q8Search& a = *(const Q8Search*)ctx;
  return rarc_finalize_q8_launch(a.rows, a.rowscale, a.fin_fmt, a.d_pad, a.q32, a.eps8, a.nq, a.k, a.id_base, *a.ws, a.cap,
                                 n_wg, a.out_ids, a.out_scores, a.status, a.s, mode, a.qmeta, a.hq, a.eps16);
}
int rarc_repair_launch(const void* corpus, const float* rowscale, int fmt, int64_t n_rows, int d_pad,
                       const float* qv, int k, int64_t id_base, int64_t* ids, float* scores, uint32_t* found,
                       const RarcWs& ws, int cap, hipStream_t s);
int rarc_verify_launch(const void* corpus, const float* rowscale, int fmt, int64_t n_rows, int d_pad, const float* q32,
                       int q_first, int nq, int k, int64_t id_base, const int64_t* ids, const float* scores,
                       uint32_t* counts, hipStream_t s);
int rarc_merge_launch(const int64_t* ids, const float* scores, int G, int nq, int k, int64_t* out_ids,
                      float* out_scores, hipStream_t s, bool packed);
int rarc_pack_launch(const int64_t* ids, const float* scores, int n, uint32_t* out, hipStream_t s);

static thread_local char g_err[512] = "";

// ---- scan timing hooks -------------------------------------------------------------------------
#include <atomic>
#include <mutex>
#include <vector>
static std::mutex g_prof_mu;               // searches may come from pool threads (core/retrieval/base.py:92-96)
static std::vector<hipEvent_t> g_prof_ev;  // pairs: [2i] start, [2i+1] stop
static int g_prof_n = 0;
static std::atomic<bool> g_prof_on{false};
bool rarc_prof_next(hipEvent_t* start, hipEvent_t* stop) {
  if (!g_prof_on.load(std::memory_order_acquire)) return false;  // the common case takes no lock
  std::lock_guard<std::mutex> lk(g_prof_mu);
  if (!g_prof_on || (size_t)(2 * g_prof_n + 1) >= g_prof_ev.size()) return false;
  *start = g_prof_ev[2 * g_prof_n];
  *stop = g_prof_ev[2 * g_prof_n + 1];
  ++g_prof_n;
  return true;
}
extern "C" int rarc_profile_begin(int max_launches) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  RARC_REQUIRE(max_launches > 0 && max_launches <= 65536 && !g_prof_on, RARC_E_INVALID, "rarc_profile_begin: bad state");
  g_prof_ev.resize((size_t)2 * max_launches);
  for (auto& e : g_prof_ev) RARC_HIP_CHECK(hipEventCreate(&e));
  g_prof_n = 0;
  g_prof_on = true;
  return RARC_OK;
}
extern "C" int rarc_profile_end(double* total_scan_ms, int* n_launches) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  RARC_REQUIRE(g_prof_on && total_scan_ms && n_launches, RARC_E_INVALID, "rarc_profile_end: not profiling");
  double tot = 0;
  for (int i = 0; i < g_prof_n; ++i) {
    RARC_HIP_CHECK(hipEventSynchronize(g_prof_ev[2 * i + 1]));
    float ms = 0;
    RARC_HIP_CHECK(hipEventElapsedTime(&ms, g_prof_ev[2 * i], g_prof_ev[2 * i + 1]));
    tot += ms;
  }
  *total_scan_ms = tot;
  *n_launches = g_prof_n;
  for (auto& e : g_prof_ev) (void)hipEventDestroy(e);
  g_prof_ev.clear();
  g_prof_on = false;
  return RARC_OK;
}

void rarc_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// ---- roctx (see rarc_common.h) ----
#include <dlfcn.h>
static int (*g_roctx_push)(const char*) = nullptr;
static int (*g_roctx_pop)() = nullptr;
static bool roctx_ready() {
  static const bool ok = [] {
    const char* e = getenv("RARC_ROCTX");
    if (!e || atoi(e) == 0) return false;
    void* h = dlopen("librocprofiler-sdk-roctx.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return false;
    g_roctx_push = (int (*)(const char*))dlsym(h, "roctxRangePushA");
    g_roctx_pop = (int (*)())dlsym(h, "roctxRangePop");
    return g_roctx_push && g_roctx_pop;
  }();
  return ok;
}
void rarc_roctx_push(const char* name) { if (roctx_ready()) g_roctx_push(name); }
void rarc_roctx_pop() { if (roctx_ready()) g_roctx_pop(); }

RarcLaunchExtras& rarc_launch_extras() {
  static thread_local RarcLaunchExtras x;
  return x;
}

extern "C" int rarc_version(void) { return RARC_BUILD_VERSION; }
extern "C" const char* rarc_last_error(void) { return g_err; }
extern "C" int rarc_padded_dim(int d) {
  return d <= 0 ? 0 : ((d + RARC_DIM_ALIGN - 1) / RARC_DIM_ALIGN) * RARC_DIM_ALIGN;
}

extern "C" size_t rarc_search_workspace_bytes(int cand_cap) {
  if (cand_cap < 4096) cand_cap = 4096;
  return RARC_WS_CAND + (size_t)RARC_MAX_QUERIES * (size_t)cand_cap * 8;
}

static int check_ws(void* ws, size_t bytes, int cap, const char* who) {
  RARC_REQUIRE(ws != nullptr && ((uintptr_t)ws % 256) == 0, RARC_E_WORKSPACE,
               "%s: workspace must be a 256-byte aligned device pointer", who);
  RARC_REQUIRE(cap >= 4096 && cap % RARC_MAX_WG == 0 && bytes >= rarc_search_workspace_bytes(cap), RARC_E_WORKSPACE,
               "%s: workspace of %zu bytes too small for cand_cap=%d (need %zu)", who, bytes, cap,
               rarc_search_workspace_bytes(cap));
  return RARC_OK;
}

extern "C" int rarc_search_f16(const uint16_t* d_corpus_f16, int64_t n_rows, int d_pad, const float* d_qmeta,
                               const void* d_qblock, int nq, int k, int kprime, int64_t id_base, float bin_lo,
                               float bin_hi, int64_t* d_out_ids, float* d_out_scores, uint32_t* d_status,
                               void* d_workspace, size_t workspace_bytes, int cand_cap, void* stream) {
  RARC_RANGE();
  RARC_REQUIRE(d_qblock && d_out_ids && d_out_scores && d_status, RARC_E_INVALID, "rarc_search_f16: null pointer");
  RARC_REQUIRE(n_rows >= 0 && n_rows < (int64_t)0xffffffe0ll, RARC_E_INVALID,
               "rarc_search_f16: n_rows=%lld outside [0, 2^32-32)", (long long)n_rows);
  RARC_REQUIRE(d_corpus_f16 || n_rows == 0, RARC_E_INVALID, "rarc_search_f16: null corpus");
  RARC_REQUIRE(d_pad > 0 && d_pad % RARC_DIM_ALIGN == 0, RARC_E_INVALID, "rarc_search_f16: d_pad=%d", d_pad);
  RARC_REQUIRE(nq >= 0 && nq <= RARC_MAX_QUERIES, RARC_E_INVALID, "rarc_search_f16: nq=%d outside [0,%d]", nq,
               RARC_MAX_QUERIES);
  RARC_REQUIRE(k >= 1 && k <= kprime && kprime <= RARC_MAX_K, RARC_E_INVALID,
               "rarc_search_f16: need 1 <= k (%d) <= kprime (%d) <= %d", k, kprime, RARC_MAX_K);
  RARC_REQUIRE(bin_hi > bin_lo, RARC_E_INVALID, "rarc_search_f16: empty histogram range");
  int rc = check_ws(d_workspace, workspace_bytes, cand_cap, "rarc_search_f16");
  if (rc) return rc;
  if (nq == 0) return RARC_OK;
  const RarcWs ws = rarc_ws_carve(d_workspace);
  const RarcQb qb = rarc_qb_carve(d_qblock, d_pad);
  hipStream_t s = (hipStream_t)stream;
  int n_wg = 0, hybrid = 0;
  if (d_qmeta) {  // int8 prefilter scan + two-step canonical finalize (always exact unless a buffer overflows)
    Q8Search a{d_corpus_f16, nullptr, 0, d_pad, nq, k, cand_cap, id_base, qb.q32, qb.eps8, d_qmeta, qb.hq, qb.eps16, &ws, d_out_ids, d_out_scores,
               d_status, s};
    rc = rarc_scan_q8_launch(d_corpus_f16, nullptr, 0, n_rows, d_pad, d_qmeta, qb.q16, qb.q8, qb.qinv, qb.eps16,
                             qb.eps8, nq, kprime, bin_lo, bin_hi, ws, cand_cap, &n_wg, s, nullptr, q8_tighten, &a, qb.hq, qb.floor, &hybrid);
    if (rc) return rc;
    return rarc_finalize_q8_launch(d_corpus_f16, nullptr, 0, d_pad, qb.q32, qb.eps8, nq, k, id_base, ws, cand_cap,
                                   n_wg, d_out_ids, d_out_scores, d_status, s, 0, d_qmeta, qb.hq, hybrid ? qb.eps16 : nullptr);
  }
  // fp16 MFMA scan + k' selection + exactness certificate (kept for comparison)
  rc = rarc_scan_f16_launch(d_corpus_f16, n_rows, d_pad, qb.q16, nq, kprime, bin_lo, bin_hi, ws, cand_cap, &n_wg, s,
                            qb.floor, qb.eps16);
  if (rc) return rc;
  return rarc_finalize_launch(d_corpus_f16, d_pad, qb.q32, qb.eps16, nq, k, kprime, id_base, ws, cand_cap, n_wg,
                              d_out_ids, d_out_scores, d_status, s);
}

extern "C" int rarc_repair_f16(const uint16_t* d_corpus_f16, int64_t n_rows, int d_pad, const void* d_qblock,
                               int q, int k, int64_t id_base, int64_t* d_out_ids, float* d_out_scores,
                               uint32_t* d_found, void* d_workspace, size_t workspace_bytes, void* stream) {
  RARC_RANGE();
  RARC_REQUIRE(d_corpus_f16 && d_qblock && d_out_ids && d_out_scores && d_found, RARC_E_INVALID,
               "rarc_repair_f16: null pointer");
  RARC_REQUIRE(q >= 0 && q < RARC_MAX_QUERIES && k >= 1 && k <= RARC_MAX_K && d_pad > 0 &&
                   d_pad % RARC_DIM_ALIGN == 0 && n_rows >= 0 && n_rows < (int64_t)0xffffffe0ll,
               RARC_E_INVALID, "rarc_repair_f16: bad arguments (q=%d k=%d)", q, k);
  int rc = check_ws(d_workspace, workspace_bytes, 4096, "rarc_repair_f16");
  if (rc) return rc;
  const int cap = (int)((workspace_bytes - RARC_WS_CAND) / 8);
  const RarcWs ws = rarc_ws_carve(d_workspace);
  const RarcQb qb = rarc_qb_carve(d_qblock, d_pad);
  return rarc_repair_launch(d_corpus_f16, nullptr, 0, n_rows, d_pad, qb.q32 + (size_t)q * d_pad, k, id_base,
                            d_out_ids + (size_t)q * k, d_out_scores + (size_t)q * k, d_found, ws, cap,
                            (hipStream_t)stream);
}

// ---- fp16 rows + int8 shadow image: the prefilter scan reads the image (half the bytes, no conversion) ----
extern "C" int rarc_search_f16_shadow(const uint16_t* d_corpus_f16, const int8_t* d_shadow8, int64_t n_rows,
                                      int d_pad, const float* d_qmeta, const void* d_qblock, int nq, int k,
                                      int kprime, int64_t id_base, float bin_lo, float bin_hi, int64_t* d_out_ids,
                                      float* d_out_scores, uint32_t* d_status, void* d_workspace,
                                      size_t workspace_bytes, int cand_cap, void* stream) {
  RARC_RANGE();
  RARC_REQUIRE(d_qblock && d_qmeta && d_out_ids && d_out_scores && d_status, RARC_E_INVALID,
               "rarc_search_f16_shadow: null pointer");
  RARC_REQUIRE(n_rows >= 0 && n_rows < (int64_t)0xffffffe0ll, RARC_E_INVALID,
               "rarc_search_f16_shadow: n_rows=%lld outside [0, 2^32-32)", (long long)n_rows);
  RARC_REQUIRE((d_corpus_f16 && d_shadow8) || n_rows == 0, RARC_E_INVALID, "rarc_search_f16_shadow: null corpus");
  RARC_REQUIRE(d_pad > 0 && d_pad % 256 == 0, RARC_E_INVALID, "rarc_search_f16_shadow: d_pad=%d (multiple of 256)", d_pad);
  RARC_REQUIRE(nq >= 0 && nq <= RARC_MAX_QUERIES, RARC_E_INVALID, "rarc_search_f16_shadow: nq=%d outside [0,%d]", nq,
               RARC_MAX_QUERIES);
  RARC_REQUIRE(k >= 1 && k <= kprime && kprime <= RARC_MAX_K, RARC_E_INVALID,
               "rarc_search_f16_shadow: need 1 <= k (%d) <= kprime (%d) <= %d", k, kprime, RARC_MAX_K);
  RARC_REQUIRE(bin_hi > bin_lo, RARC_E_INVALID, "rarc_search_f16_shadow: empty histogram range");
  int rc = check_ws(d_workspace, workspace_bytes, cand_cap, "rarc_search_f16_shadow");
  if (rc) return rc;
  if (nq == 0) return RARC_OK;
  const RarcWs ws = rarc_ws_carve(d_workspace);
  const RarcQb qb = rarc_qb_carve(d_qblock, d_pad);
  hipStream_t s = (hipStream_t)stream;
  int n_wg = 0, hybrid = 0;
  Q8Search a{d_corpus_f16, nullptr, 0, d_pad, nq, k, cand_cap, id_base, qb.q32, qb.eps8, d_qmeta, qb.hq, qb.eps16, &ws, d_out_ids, d_out_scores,
             d_status, s};
  rc = rarc_scan_q8_launch(d_corpus_f16, nullptr, 2, n_rows, d_pad, d_qmeta, qb.q16, qb.q8, qb.qinv, qb.eps16, qb.eps8,
                           nq, kprime, bin_lo, bin_hi, ws, cand_cap, &n_wg, s, d_shadow8, q8_tighten, &a, qb.hq, qb.floor, nullptr);
  if (rc) return rc;
  return rarc_finalize_q8_launch(d_corpus_f16, nullptr, 0, d_pad, qb.q32, qb.eps8, nq, k, id_base, ws, cand_cap,
                                 n_wg, d_out_ids, d_out_scores, d_status, s, 0, d_qmeta, qb.hq, hybrid ? qb.eps16 : nullptr);
}

// ---- fp32 rows (the reference's own storage) + their fp16 image: the scan reads the image, every returned score is
// the canonical fp32 dot with the fp32 row; the image's rounding is part of the error bounds (qmeta[1], prep.hip)
extern "C" int rarc_search_f32(const float* d_corpus_f32, const uint16_t* d_image_f16, int64_t n_rows, int d_pad,
                               const float* d_qmeta, const void* d_qblock, int nq, int k, int kprime, int64_t id_base,
                               float bin_lo, float bin_hi, int64_t* d_out_ids, float* d_out_scores,
                               uint32_t* d_status, void* d_workspace, size_t workspace_bytes, int cand_cap,
                               void* stream) {
  RARC_RANGE();
  RARC_REQUIRE(d_qblock && d_qmeta && d_out_ids && d_out_scores && d_status, RARC_E_INVALID,
               "rarc_search_f32: null pointer");
  RARC_REQUIRE(n_rows >= 0 && n_rows < (int64_t)0xffffffe0ll, RARC_E_INVALID,
               "rarc_search_f32: n_rows=%lld outside [0, 2^32-32)", (long long)n_rows);
  RARC_REQUIRE((d_corpus_f32 && d_image_f16) || n_rows == 0, RARC_E_INVALID, "rarc_search_f32: null corpus");
  RARC_REQUIRE(d_pad > 0 && d_pad % RARC_DIM_ALIGN == 0, RARC_E_INVALID, "rarc_search_f32: d_pad=%d", d_pad);
  RARC_REQUIRE(nq >= 0 && nq <= RARC_MAX_QUERIES, RARC_E_INVALID, "rarc_search_f32: nq=%d outside [0,%d]", nq,
               RARC_MAX_QUERIES);
  RARC_REQUIRE(k >= 1 && k <= kprime && kprime <= RARC_MAX_K, RARC_E_INVALID,
               "rarc_search_f32: need 1 <= k (%d) <= kprime (%d) <= %d", k, kprime, RARC_MAX_K);
  RARC_REQUIRE(bin_hi > bin_lo, RARC_E_INVALID, "rarc_search_f32: empty histogram range");
  int rc = check_ws(d_workspace, workspace_bytes, cand_cap, "rarc_search_f32");
  if (rc) return rc;
  if (nq == 0) return RARC_OK;
  const RarcWs ws = rarc_ws_carve(d_workspace);
  const RarcQb qb = rarc_qb_carve(d_qblock, d_pad);
  hipStream_t s = (hipStream_t)stream;
  int n_wg = 0, hybrid = 0;
  Q8Search a{d_corpus_f32, nullptr, 2, d_pad, nq, k, cand_cap, id_base, qb.q32, qb.eps8, d_qmeta, qb.hq, qb.eps16, &ws, d_out_ids, d_out_scores,
             d_status, s};
  rc = rarc_scan_q8_launch(d_image_f16, nullptr, 0, n_rows, d_pad, d_qmeta, qb.q16, qb.q8, qb.qinv, qb.eps16, qb.eps8,
                           nq, kprime, bin_lo, bin_hi, ws, cand_cap, &n_wg, s, nullptr, q8_tighten, &a, qb.hq, qb.floor, nullptr);
  if (rc) return rc;
  return rarc_finalize_q8_launch(d_corpus_f32, nullptr, 2, d_pad, qb.q32, qb.eps8, nq, k, id_base, ws, cand_cap, n_wg,
                                 d_out_ids, d_out_scores, d_status, s, 0, d_qmeta, qb.hq, hybrid ? qb.eps16 : nullptr);
}

extern "C" int rarc_repair_f32(const float* d_corpus_f32, int64_t n_rows, int d_pad, const void* d_qblock, int q, int k,
                               int64_t id_base, int64_t* d_out_ids, float* d_out_scores, uint32_t* d_found,
                               void* d_workspace, size_t workspace_bytes, void* stream) {
  RARC_RANGE();
  RARC_REQUIRE(d_corpus_f32 && d_qblock && d_out_ids && d_out_scores && d_found, RARC_E_INVALID,
               "rarc_repair_f32: null pointer");
  RARC_REQUIRE(q >= 0 && q < RARC_MAX_QUERIES && k >= 1 && k <= RARC_MAX_K && d_pad > 0 &&
                   d_pad % RARC_DIM_ALIGN == 0 && n_rows >= 0 && n_rows < (int64_t)0xffffffe0ll,
               RARC_E_INVALID, "rarc_repair_f32: bad arguments (q=%d k=%d)", q, k);
  int rc = check_ws(d_workspace, workspace_bytes, 4096, "rarc_repair_f32");
  if (rc) return rc;
  const int cap = (int)((workspace_bytes - RARC_WS_CAND) / 8);
  const RarcWs ws = rarc_ws_carve(d_workspace);
  const RarcQb qb = rarc_qb_carve(d_qblock, d_pad);
  return rarc_repair_launch(d_corpus_f32, nullptr, 2, n_rows, d_pad, qb.q32 + (size_t)q * d_pad, k, id_base,
                            d_out_ids + (size_t)q * k, d_out_scores + (size_t)q * k, d_found, ws, cap,
                            (hipStream_t)stream);
}

// ---- fp8 (e4m3fn + per-row scale) corpus: BASELINE config 5's storage --------------------------------
extern "C" int rarc_search_f8(const uint8_t* d_corpus_f8, const float* d_row_scale, int64_t n_rows, int d_pad,
                              const float* d_qmeta, const void* d_qblock, int nq, int k, int kprime,
                              int64_t id_base, float bin_lo, float bin_hi, int64_t* d_out_ids,
                              float* d_out_scores, uint32_t* d_status, void* d_workspace, size_t workspace_bytes,
                              int cand_cap, void* stream) {
  RARC_RANGE();
  RARC_REQUIRE(d_qblock && d_qmeta && d_out_ids && d_out_scores && d_status, RARC_E_INVALID,
               "rarc_search_f8: null pointer");
  RARC_REQUIRE(n_rows >= 0 && n_rows < (int64_t)0xffffffe0ll, RARC_E_INVALID,
               "rarc_search_f8: n_rows=%lld outside [0, 2^32-32)", (long long)n_rows);
  RARC_REQUIRE((d_corpus_f8 && d_row_scale) || n_rows == 0, RARC_E_INVALID, "rarc_search_f8: null corpus");
  RARC_REQUIRE(d_pad > 0 && d_pad % RARC_DIM_ALIGN_F8 == 0, RARC_E_INVALID, "rarc_search_f8: d_pad=%d", d_pad);
  RARC_REQUIRE(nq >= 0 && nq <= RARC_MAX_QUERIES, RARC_E_INVALID, "rarc_search_f8: nq=%d outside [0,%d]", nq,
               RARC_MAX_QUERIES);
  RARC_REQUIRE(k >= 1 && k <= kprime && kprime <= RARC_MAX_K, RARC_E_INVALID,
               "rarc_search_f8: need 1 <= k (%d) <= kprime (%d) <= %d", k, kprime, RARC_MAX_K);
  RARC_REQUIRE(bin_hi > bin_lo, RARC_E_INVALID, "rarc_search_f8: empty histogram range");
  int rc = check_ws(d_workspace, workspace_bytes, cand_cap, "rarc_search_f8");
  if (rc) return rc;
  if (nq == 0) return RARC_OK;
  const RarcWs ws = rarc_ws_carve(d_workspace);
  const RarcQb qb = rarc_qb_carve(d_qblock, d_pad);
  hipStream_t s = (hipStream_t)stream;
  int n_wg = 0, hybrid = 0;
  Q8Search a{d_corpus_f8, d_row_scale, 1, d_pad, nq, k, cand_cap, id_base, qb.q32, qb.eps8, d_qmeta, qb.hq, qb.eps16, &ws, d_out_ids, d_out_scores,
             d_status, s};
  rc = rarc_scan_q8_launch(d_corpus_f8, d_row_scale, 1, n_rows, d_pad, d_qmeta, qb.q16, qb.q8, qb.qinv, qb.eps16,
                           qb.eps8, nq, kprime, bin_lo, bin_hi, ws, cand_cap, &n_wg, s, nullptr, q8_tighten, &a, qb.hq, qb.floor, nullptr);
  if (rc) return rc;
  return rarc_finalize_q8_launch(d_corpus_f8, d_row_scale, 1, d_pad, qb.q32, qb.eps8, nq, k, id_base, ws, cand_cap,
                                 n_wg, d_out_ids, d_out_scores, d_status, s, 0, d_qmeta, qb.hq, hybrid ? qb.eps16 : nullptr);
}

extern "C" int rarc_repair_f8(const uint8_t* d_corpus_f8, const float* d_row_scale, int64_t n_rows, int d_pad,
                              const void* d_qblock, int q, int k, int64_t id_base, int64_t* d_out_ids,
                              float* d_out_scores, uint32_t* d_found, void* d_workspace, size_t workspace_bytes,
                              void* stream) {
  RARC_RANGE();
  RARC_REQUIRE(d_corpus_f8 && d_row_scale && d_qblock && d_out_ids && d_out_scores && d_found, RARC_E_INVALID,
               "rarc_repair_f8: null pointer");
  RARC_REQUIRE(q >= 0 && q < RARC_MAX_QUERIES && k >= 1 && k <= RARC_MAX_K && d_pad > 0 &&
                   d_pad % RARC_DIM_ALIGN_F8 == 0 && n_rows >= 0 && n_rows < (int64_t)0xffffffe0ll,
               RARC_E_INVALID, "rarc_repair_f8: bad arguments (q=%d k=%d)", q, k);
  int rc = check_ws(d_workspace, workspace_bytes, 4096, "rarc_repair_f8");
  if (rc) return rc;
  const int cap = (int)((workspace_bytes - RARC_WS_CAND) / 8);
  const RarcWs ws = rarc_ws_carve(d_workspace);
  const RarcQb qb = rarc_qb_carve(d_qblock, d_pad);
  return rarc_repair_launch(d_corpus_f8, d_row_scale, 1, n_rows, d_pad, qb.q32 + (size_t)q * d_pad, k, id_base,
                            d_out_ids + (size_t)q * k, d_out_scores + (size_t)q * k, d_found, ws, cap,
                            (hipStream_t)stream);
}

// exact batched verification of up to 8 answers against a canonical scan of the shard (finalize.hip)
extern "C" int rarc_verify_batch(const void* d_rows, const float* d_row_scale, int row_format, int64_t n_rows, int d_pad,
                                 const void* d_qblock, int q_first, int nq, int k, int64_t id_base, const int64_t* d_ids,
                                 const float* d_scores, uint32_t* d_counts, void* stream) {
  RARC_RANGE();
  RARC_REQUIRE(d_rows && d_qblock && d_ids && d_scores && d_counts, RARC_E_INVALID, "rarc_verify_batch: null pointer");
  RARC_REQUIRE(row_format >= 0 && row_format <= 2 && (row_format != 1 || d_row_scale), RARC_E_INVALID,
               "rarc_verify_batch: row_format %d (0 fp16, 1 fp8 + scales, 2 fp32)", row_format);
  RARC_REQUIRE(nq >= 1 && nq <= 8 && q_first >= 0 && q_first + nq <= RARC_MAX_QUERIES && k >= 1 && k <= RARC_MAX_K &&
                   d_pad > 0 && d_pad % RARC_DIM_ALIGN == 0 && d_pad <= 1024 && n_rows >= 0 && n_rows < (int64_t)0xffffffe0ll,
               RARC_E_INVALID, "rarc_verify_batch: bad arguments (q_first=%d nq=%d k=%d d_pad=%d)", q_first, nq, k, d_pad);
  const RarcQb qb = rarc_qb_carve(d_qblock, d_pad);
  return rarc_verify_launch(d_rows, d_row_scale, row_format, n_rows, d_pad, qb.q32, q_first, nq, k, id_base, d_ids,
                            d_scores, d_counts, (hipStream_t)stream);
}

extern "C" int rarc_topk_merge(const int64_t* d_ids, const float* d_scores, int n_lists, int nq, int k,
                               int64_t* d_out_ids, float* d_out_scores, void* stream) {
  RARC_RANGE();
  RARC_REQUIRE(d_ids && d_scores && d_out_ids && d_out_scores, RARC_E_INVALID, "rarc_topk_merge: null pointer");
  RARC_REQUIRE(n_lists >= 1 && nq >= 0 && k >= 1, RARC_E_INVALID, "rarc_topk_merge: bad sizes");
  if (nq == 0) return RARC_OK;
  return rarc_merge_launch(d_ids, d_scores, n_lists, nq, k, d_out_ids, d_out_scores, (hipStream_t)stream, false);
}

extern "C" int rarc_pack_results(const int64_t* d_ids, const float* d_scores, int nq, int k, int32_t* d_packed, void* stream) {
  RARC_REQUIRE(d_ids && d_scores && d_packed && nq >= 0 && k >= 1, RARC_E_INVALID, "rarc_pack_results: bad arguments");
  if (nq == 0) return RARC_OK;
  return rarc_pack_launch(d_ids, d_scores, nq * k, (uint32_t*)d_packed, (hipStream_t)stream);
}

extern "C" int rarc_topk_merge_packed(const int32_t* d_packed, int n_lists, int nq, int k, int64_t* d_out_ids,
                                      float* d_out_scores, void* stream) {
  RARC_RANGE();
  RARC_REQUIRE(d_packed && d_out_ids && d_out_scores, RARC_E_INVALID, "rarc_topk_merge_packed: null pointer");
  RARC_REQUIRE(n_lists >= 1 && nq >= 0 && k >= 1, RARC_E_INVALID, "rarc_topk_merge_packed: bad sizes");
  if (nq == 0) return RARC_OK;
  return rarc_merge_launch((const int64_t*)d_packed, nullptr, n_lists, nq, k, d_out_ids, d_out_scores, (hipStream_t)stream, true);
}


// ---- one batch, one call (ABI 600): query prep + search, status words zeroed by the prep kernel, the any-flag word written
//      to pinned host memory by the finalize kernel, the scan optionally gated on a neighbouring context's event ----
extern "C" int rarc_search_batch(const RarcSearchBatch* b, void* stream) {
  RARC_RANGE();
  RARC_REQUIRE(b != nullptr, RARC_E_INVALID, "rarc_search_batch: null descriptor");
  RARC_REQUIRE(b->row_format >= 0 && b->row_format <= 3, RARC_E_INVALID, "rarc_search_batch: row_format %d", b->row_format);
  RARC_REQUIRE(b->d_status != nullptr, RARC_E_INVALID, "rarc_search_batch: null status");
  RarcLaunchExtras& x = rarc_launch_extras();
  struct Clear {   // whatever path returns: the extras are this call's only
    RarcLaunchExtras& x;
    ~Clear() { x = RarcLaunchExtras(); }
  } clear{x};
  x.status_zero = b->d_status;
  x.flag_host = b->flag_host;
  x.gate = (hipEvent_t)b->gate_event;
  int rc = rarc_prep_queries(b->d_queries, b->ld_queries, b->nq, b->d, b->d_pad, b->normalize, b->corpus_max_norm, b->d_qmeta,
                             b->d_qblock, stream);
  if (rc) return rc;
  switch (b->row_format) {
    case 0:
      return rarc_search_f16((const uint16_t*)b->d_rows, b->n_rows, b->d_pad, b->d_qmeta, b->d_qblock, b->nq, b->k, b->kprime,
                             b->id_base, b->bin_lo, b->bin_hi, b->d_out_ids, b->d_out_scores, b->d_status, b->d_workspace,
                             b->workspace_bytes, b->cand_cap, stream);
    case 1:
      return rarc_search_f8((const uint8_t*)b->d_rows, (const float*)b->d_aux, b->n_rows, b->d_pad, b->d_qmeta, b->d_qblock, b->nq,
                            b->k, b->kprime, b->id_base, b->bin_lo, b->bin_hi, b->d_out_ids, b->d_out_scores, b->d_status,
                            b->d_workspace, b->workspace_bytes, b->cand_cap, stream);
    case 2:
      return rarc_search_f32((const float*)b->d_rows, (const uint16_t*)b->d_aux, b->n_rows, b->d_pad, b->d_qmeta, b->d_qblock, b->nq,
                             b->k, b->kprime, b->id_base, b->bin_lo, b->bin_hi, b->d_out_ids, b->d_out_scores, b->d_status,
                             b->d_workspace, b->workspace_bytes, b->cand_cap, stream);
    default:
      return rarc_search_f16_shadow((const uint16_t*)b->d_rows, (const int8_t*)b->d_aux, b->n_rows, b->d_pad, b->d_qmeta, b->d_qblock,
                                    b->nq, b->k, b->kprime, b->id_base, b->bin_lo, b->bin_hi, b->d_out_ids, b->d_out_scores,
                                    b->d_status, b->d_workspace, b->workspace_bytes, b->cand_cap, stream);
  }
}
