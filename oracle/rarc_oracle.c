/*
 * rarc_oracle.c — CPU ORACLE (test infrastructure, never shipped, never on the product path).
 *
 * A restatement of the arithmetic of RAG-ARC's dense-retrieval hot path, used only by tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg as the checker / timed CPU baseline.
 *
 * PARITY STATUS: PINNED (round 2) to numbers the reference itself produced.  The reference reaches this arithmetic
 * through faiss, which is NOT vendored in /root/reference (requirements.txt is empty, no version pinned) and is not
 * installed here, so faiss's own outputs cannot be recorded; what IS importable from the reference and computes the same
 * quantity is its float64 cosine (core/file_management/chunker/spliter.py:326-332, `cosine_similarity`).
 * tests/golden/make_golden.py ran it on 64 queries x 4096 rows of fp16-representable vectors at d = 384 and 768
 * (tests/golden/cosine_pin_d*.npz); tests/test_oracle_golden.py::test_flat_search_pinned_to_reference_float64 holds this
 * oracle (fp16 and fp32 rows) to those numbers within 1e-5 (measured < 2e-6) on all 262 144 pairs, with the reference's
 * top-100 set / order wherever its gaps exceed the tolerance; tests/test_gpu_reference_pin.py holds the HIP index to the
 * same check.  Beyond that pin the flat search / normalise parts restate faiss's published contract — normalize_L2 =
 * x * (1/sqrt(sum x^2)) in fp32 with zero rows untouched (fvec_renorm_L2), IndexFlatIP.search = exact fp32 inner
 * products sorted by score descending, int64 labels, -1 padding — at the reference's call sites:
 *   encapsulation/database/vector_db/VectorStore_Faiss.py:150-154  (_normalize_vectors)
 *   encapsulation/database/vector_db/VectorStore_Faiss.py:170-202  (add_texts: astype(f32), normalise, add)
 *   encapsulation/database/vector_db/VectorStore_Faiss.py:258-272  (query: astype(f32), normalise, k=min(k,ntotal), search)
 * Two things faiss leaves implementation-defined are DEFINED here (and in DESIGN.md):
 *   - summation order of an fp32 inner product / squared norm: the "canonical" order below
 *     (8 interleaved fma chains + a fixed add tree; it is what an AVX2 loop computes);
 *   - order among equal scores: id ascending.
 * The RRF / retriever-control-flow / relevance-score parts of the oracle live in cpu_ref.py and
 * ARE pinned against the reference's own importable code (tests/golden/make_golden.py).
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -mavx2 -mfma -mf16c -fopenmp).
 */
#include <immintrin.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ---- fp16 <-> fp32, bit exact (IEEE round-to-nearest-even) ---------------------------------- */
static inline float h2f(uint16_t h) { return _cvtsh_ss(h); }
static inline uint16_t f2h(float f) { return _cvtss_sh(f, _MM_FROUND_TO_NEAREST_INT | _MM_FROUND_NO_EXC); }

/* portable versions, used by the self test to pin the intrinsics */
static float h2f_soft(uint16_t h) {
  uint32_t s = (uint32_t)(h & 0x8000) << 16, e = (h >> 10) & 0x1f, m = h & 0x3ff, u;
  if (e == 0) {
    if (m == 0) u = s;
    else {
      int sh = 0;
      while (!(m & 0x400)) { m <<= 1; ++sh; }
      u = s | ((uint32_t)(127 - 15 - sh + 1) << 23) | ((m & 0x3ff) << 13);
    }
  } else if (e == 31) u = s | 0x7f800000u | (m << 13);
  else u = s | ((e + 112) << 23) | (m << 13);
  float f;
  memcpy(&f, &u, 4);
  return f;
}

/* ---- canonical fp32 reductions ---------------------------------------------------------------
 * acc[j] (j = 0..7) = fma chain over elements 8m + j, m ascending, starting from +0;
 * result = ((acc0 + acc4) + (acc2 + acc6)) + ((acc1 + acc5) + (acc3 + acc7)).
 * d must be a multiple of 8 (rows are stored zero-padded).                                      */
static inline float tree8(__m256 acc) {
  __m128 lo = _mm256_castps256_ps128(acc), hi = _mm256_extractf128_ps(acc, 1);
  __m128 s = _mm_add_ps(lo, hi);                       /* a0+a4, a1+a5, a2+a6, a3+a7 */
  __m128 t = _mm_add_ps(s, _mm_movehl_ps(s, s));        /* (a0+a4)+(a2+a6), (a1+a5)+(a3+a7) */
  return _mm_cvtss_f32(_mm_add_ss(t, _mm_shuffle_ps(t, t, 1)));
}

float oracle_canon_dot_f16(const float* q, const uint16_t* row, int d) {
  __m256 acc = _mm256_setzero_ps();
  for (int m = 0; m < d; m += 8) {
    __m256 x = _mm256_cvtph_ps(_mm_loadu_si128((const __m128i*)(row + m)));
    acc = _mm256_fmadd_ps(_mm256_loadu_ps(q + m), x, acc);
  }
  return tree8(acc);
}

/* scalar statement of the same definition (self test: must equal the AVX2 version bit for bit) */
float oracle_canon_dot_f16_scalar(const float* q, const uint16_t* row, int d) {
  float a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int m = 0; m < d; m += 8)
    for (int j = 0; j < 8; ++j) a[j] = fmaf(q[m + j], h2f_soft(row[m + j]), a[j]);
  return ((a[0] + a[4]) + (a[2] + a[6])) + ((a[1] + a[5]) + (a[3] + a[7]));
}

static float canon_sumsq_f32(const float* x, int d) { /* d arbitrary: tail treated as zeros */
  float a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int m = 0; m < d; ++m) a[m & 7] = fmaf(x[m], x[m], a[m & 7]);
  return ((a[0] + a[4]) + (a[2] + a[6])) + ((a[1] + a[5]) + (a[3] + a[7]));
}

/* faiss.normalize_L2 (fvec_renorm_L2): in place, zero rows untouched */
void oracle_normalize_rows_f32(float* x, int64_t ld, int64_t n, int d) {
#pragma omp parallel for schedule(static)
  for (int64_t r = 0; r < n; ++r) {
    float* v = x + r * ld;
    const float nr = canon_sumsq_f32(v, d);
    if (nr > 0) {
      const float inv = (float)(1.0 / sqrtf(nr));
      for (int m = 0; m < d; ++m) v[m] *= inv;
    }
  }
}

/* add_texts side: (normalise) -> fp16 rows of length d_pad, zero padded; optional squared norms */
void oracle_ingest_f16(const float* in, int64_t ld, uint16_t* out, int d_pad, float* norm2, int64_t n, int d,
                       int normalize) {
#pragma omp parallel for schedule(static)
  for (int64_t r = 0; r < n; ++r) {
    const float* v = in + r * ld;
    uint16_t* o = out + r * (int64_t)d_pad;
    float inv = 1.0f;
    int scale = 0;
    if (normalize) {
      const float nr = canon_sumsq_f32(v, d);
      if (nr > 0) { inv = (float)(1.0 / sqrtf(nr)); scale = 1; }
    }
    float a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int m = 0; m < d_pad; ++m) {
      uint16_t h = 0;
      if (m < d) h = f2h(scale ? v[m] * inv : v[m]);
      o[m] = h;
      const float f = h2f(h);
      a[m & 7] = fmaf(f, f, a[m & 7]);
    }
    if (norm2) norm2[r] = ((a[0] + a[4]) + (a[2] + a[6])) + ((a[1] + a[5]) + (a[3] + a[7]));
  }
}

/* ---- exact flat inner-product search (IndexFlatIP.search) ------------------------------------ */
static inline uint32_t ordkey(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
static inline float unordkey(uint32_t k) {
  uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
  float f;
  memcpy(&f, &u, 4);
  return f;
}
/* larger key == better: score descending, then row ascending */
static inline uint64_t candkey(float s, uint32_t row) { return ((uint64_t)ordkey(s) << 32) | (uint64_t)(~row); }

/* min-heap of the k best keys */
static inline void heap_push(uint64_t* h, int* n, int k, uint64_t key) {
  if (*n < k) {
    int i = (*n)++;
    while (i > 0 && h[(i - 1) / 2] > key) { h[i] = h[(i - 1) / 2]; i = (i - 1) / 2; }
    h[i] = key;
  } else if (key > h[0]) {
    int i = 0;
    for (;;) {
      int c = 2 * i + 1;
      if (c >= k) break;
      if (c + 1 < k && h[c + 1] < h[c]) ++c;
      if (h[c] >= key) break;
      h[i] = h[c];
      i = c;
    }
    h[i] = key;
  }
}
static int cmp_desc_u64(const void* a, const void* b) {
  const uint64_t x = *(const uint64_t*)a, y = *(const uint64_t*)b;
  return x < y ? 1 : (x > y ? -1 : 0);
}

/*
 * corpus: [n][d_pad] fp16; q: [nq][d_pad] fp32 (already normalised if the metric needs it).
 * out_ids / out_scores: [nq][k]; entries beyond min(k, n) are -1 / -inf.
 * Returns the number of threads used.  Rows are blocked 4 at a time so a query vector load is
 * shared by four accumulator chains (each chain is still exactly the canonical order).
 */
int oracle_flat_search_f16(const uint16_t* corpus, int64_t n, int d_pad, const float* q, int nq, int k,
                           int64_t id_base, int64_t* out_ids, float* out_scores) {
  int nthreads = 1;
#ifdef _OPENMP
  nthreads = omp_get_max_threads();
#endif
  if (k < 1 || nq < 1) return nthreads;
  uint64_t* heaps = (uint64_t*)malloc((size_t)nthreads * nq * k * sizeof(uint64_t));
  int* hn = (int*)calloc((size_t)nthreads * nq, sizeof(int));
#pragma omp parallel
  {
    int tid = 0;
#ifdef _OPENMP
    tid = omp_get_thread_num();
#endif
    uint64_t* H = heaps + (size_t)tid * nq * k;
    int* N = hn + (size_t)tid * nq;
    float* rowf = (float*)aligned_alloc(32, (size_t)4 * d_pad * sizeof(float));
#pragma omp for schedule(dynamic, 64)
    for (int64_t r0 = 0; r0 < n; r0 += 4) {
      const int nr = (n - r0) < 4 ? (int)(n - r0) : 4;
      for (int i = 0; i < 4; ++i) {
        const uint16_t* src = corpus + (r0 + (i < nr ? i : 0)) * (int64_t)d_pad;
        for (int m = 0; m < d_pad; m += 8)
          _mm256_store_ps(rowf + i * d_pad + m, _mm256_cvtph_ps(_mm_loadu_si128((const __m128i*)(src + m))));
      }
      for (int qi = 0; qi < nq; ++qi) {
        const float* qv = q + (size_t)qi * d_pad;
        __m256 a0 = _mm256_setzero_ps(), a1 = a0, a2 = a0, a3 = a0;
        for (int m = 0; m < d_pad; m += 8) {
          const __m256 qq = _mm256_loadu_ps(qv + m);
          a0 = _mm256_fmadd_ps(qq, _mm256_load_ps(rowf + m), a0);
          a1 = _mm256_fmadd_ps(qq, _mm256_load_ps(rowf + d_pad + m), a1);
          a2 = _mm256_fmadd_ps(qq, _mm256_load_ps(rowf + 2 * d_pad + m), a2);
          a3 = _mm256_fmadd_ps(qq, _mm256_load_ps(rowf + 3 * d_pad + m), a3);
        }
        const float s[4] = {tree8(a0), tree8(a1), tree8(a2), tree8(a3)};
        for (int i = 0; i < nr; ++i) heap_push(H + (size_t)qi * k, &N[qi], k, candkey(s[i], (uint32_t)(r0 + i)));
      }
    }
    free(rowf);
  }
  uint64_t* all = (uint64_t*)malloc((size_t)nthreads * k * sizeof(uint64_t));
  for (int qi = 0; qi < nq; ++qi) {
    int c = 0;
    for (int t = 0; t < nthreads; ++t) {
      const int m = hn[(size_t)t * nq + qi];
      memcpy(all + c, heaps + ((size_t)t * nq + qi) * k, (size_t)m * sizeof(uint64_t));
      c += m;
    }
    qsort(all, (size_t)c, sizeof(uint64_t), cmp_desc_u64);
    for (int i = 0; i < k; ++i) {
      if (i < c) {
        out_ids[(size_t)qi * k + i] = id_base + (int64_t)(uint32_t)(~(uint32_t)all[i]);
        out_scores[(size_t)qi * k + i] = unordkey((uint32_t)(all[i] >> 32));
      } else {
        out_ids[(size_t)qi * k + i] = -1;
        out_scores[(size_t)qi * k + i] = -INFINITY;
      }
    }
  }
  free(all);
  free(heaps);
  free(hn);
  return nthreads;
}

/* ---- fp32 storage: the reference's own (np.float32 rows in IndexFlatIP, VectorStore_Faiss.py:170,199-202) ----
 * rows are the normalised fp32 vectors themselves (zero padded to d_pad); the canonical score is the same
 * 8-chain fma order on fp32 x fp32.                                                                          */
float oracle_canon_dot_f32(const float* q, const float* row, int d) {
  __m256 acc = _mm256_setzero_ps();
  for (int m = 0; m < d; m += 8) acc = _mm256_fmadd_ps(_mm256_loadu_ps(q + m), _mm256_loadu_ps(row + m), acc);
  return tree8(acc);
}
void oracle_ingest_f32(const float* in, int64_t ld, float* out, int d_pad, float* norm2, int64_t n, int d, int normalize) {
#pragma omp parallel for schedule(static)
  for (int64_t r = 0; r < n; ++r) {
    const float* v = in + r * ld;
    float* o = out + r * (int64_t)d_pad;
    float inv = 1.0f;
    int scale = 0;
    if (normalize) {
      const float nr = canon_sumsq_f32(v, d);
      if (nr > 0) { inv = (float)(1.0 / sqrtf(nr)); scale = 1; }
    }
    float a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int m = 0; m < d_pad; ++m) {
      const float f = m < d ? (scale ? v[m] * inv : v[m]) : 0.0f;
      o[m] = f;
      a[m & 7] = fmaf(f, f, a[m & 7]);
    }
    if (norm2) norm2[r] = ((a[0] + a[4]) + (a[2] + a[6])) + ((a[1] + a[5]) + (a[3] + a[7]));
  }
}
int oracle_flat_search_f32(const float* corpus, int64_t n, int d_pad, const float* q, int nq, int k, int64_t id_base,
                           int64_t* out_ids, float* out_scores) {
  int nthreads = 1;
#ifdef _OPENMP
  nthreads = omp_get_max_threads();
#endif
  if (k < 1 || nq < 1) return nthreads;
  uint64_t* heaps = (uint64_t*)malloc((size_t)nthreads * nq * k * sizeof(uint64_t));
  int* hn = (int*)calloc((size_t)nthreads * nq, sizeof(int));
#pragma omp parallel
  {
    int tid = 0;
#ifdef _OPENMP
    tid = omp_get_thread_num();
#endif
    uint64_t* H = heaps + (size_t)tid * nq * k;
    int* N = hn + (size_t)tid * nq;
#pragma omp for schedule(dynamic, 64)
    for (int64_t r = 0; r < n; ++r) {
      const float* row = corpus + r * (int64_t)d_pad;
      for (int qi = 0; qi < nq; ++qi)
        heap_push(H + (size_t)qi * k, &N[qi], k, candkey(oracle_canon_dot_f32(q + (size_t)qi * d_pad, row, d_pad), (uint32_t)r));
    }
  }
  uint64_t* all = (uint64_t*)malloc((size_t)nthreads * k * sizeof(uint64_t));
  for (int qi = 0; qi < nq; ++qi) {
    int c = 0;
    for (int t = 0; t < nthreads; ++t) {
      const int m = hn[(size_t)t * nq + qi];
      memcpy(all + c, heaps + ((size_t)t * nq + qi) * k, (size_t)m * sizeof(uint64_t));
      c += m;
    }
    qsort(all, (size_t)c, sizeof(uint64_t), cmp_desc_u64);
    for (int i = 0; i < k; ++i) {
      if (i < c) {
        out_ids[(size_t)qi * k + i] = id_base + (int64_t)(uint32_t)(~(uint32_t)all[i]);
        out_scores[(size_t)qi * k + i] = unordkey((uint32_t)(all[i] >> 32));
      } else {
        out_ids[(size_t)qi * k + i] = -1;
        out_scores[(size_t)qi * k + i] = -INFINITY;
      }
    }
  }
  free(all);
  free(heaps);
  free(hn);
  return nthreads;
}

/* canonical scores of selected rows (spot checks at sizes where a full search is too slow) */
void oracle_score_rows_f16(const uint16_t* corpus, int d_pad, const float* qv, const int64_t* rows, int n,
                           float* out) {
  for (int i = 0; i < n; ++i) out[i] = oracle_canon_dot_f16(qv, corpus + rows[i] * (int64_t)d_pad, d_pad);
}

/* ---- fp8 (OCP e4m3fn) storage: BASELINE config 5's corpus format -----------------------------------
 * A stored row is d_pad bytes of e4m3fn (1 sign, 4 exponent bits bias 7, 3 mantissa bits, no inf,
 * max 448, subnormals m*2^-9) plus one fp32 scale per row: value[m] = scale * decode(byte[m]).
 * Ingest (the fp8 form of add_texts, VectorStore_Faiss.py:170-202): normalise as for fp16, scale =
 * max|x| / 448 (1 for a zero row), byte = encode(x / scale), encode = round to nearest even with
 * saturation at 448.  Canonical score = scale * (canonical fp32 dot of q with the decoded bytes): the
 * decoded values are exact in fp32 (and in fp16), the one multiply by the scale rounds once.
 * The HIP kernels (prep.hip: rarc_f8_encode / rarc_f8_decode) use the same integer algorithm.       */
static inline float f8_decode(uint8_t b) {
  const uint32_t e = (b >> 3) & 15, m = b & 7;
  float v = e ? ldexpf((float)(8 + m), (int)e - 10) : ldexpf((float)m, -9);
  if (e == 15 && m == 7) v = 0.0f; /* NaN code: never produced by the encoder; decodes to 0 */
  return (b & 0x80) ? -v : v;
}
static inline uint8_t f8_encode(float x) {
  uint32_t u;
  memcpy(&u, &x, 4);
  const uint8_t sign = (uint8_t)((u >> 24) & 0x80);
  u &= 0x7fffffffu;
  float a;
  memcpy(&a, &u, 4);
  if (!(a == a)) return sign;          /* NaN -> 0 */
  if (a >= 448.0f) return sign | 0x7e; /* saturate (also +-inf) */
  if (a < 0.015625f) {                 /* below 2^-6: subnormal grid of 2^-9; 8 -> 0x08 = 2^-6 */
    return sign | (uint8_t)(int)rintf(a * 512.0f);
  }
  uint32_t r = u + 0x0007ffffu + ((u >> 20) & 1u); /* RNE at mantissa bit 20 */
  r >>= 20;                                        /* (exp8 << 3) | mant3 */
  uint32_t code = r - ((127u - 7u) << 3);
  if (code > 0x7eu) code = 0x7eu;
  return sign | (uint8_t)code;
}
uint8_t oracle_f8_encode(float x) { return f8_encode(x); }
float oracle_f8_decode(uint8_t b) { return f8_decode(b); }

void oracle_ingest_f8(const float* in, int64_t ld, uint8_t* out, int d_pad, float* scale_out, float* norm2,
                      int64_t n, int d, int normalize) {
#pragma omp parallel for schedule(static)
  for (int64_t r = 0; r < n; ++r) {
    const float* v = in + r * ld;
    uint8_t* o = out + r * (int64_t)d_pad;
    float inv = 1.0f;
    int sc = 0;
    if (normalize) {
      const float nr = canon_sumsq_f32(v, d);
      if (nr > 0) { inv = (float)(1.0 / sqrtf(nr)); sc = 1; }
    }
    float mx = 0.0f;
    for (int m = 0; m < d; ++m) {
      const float x = fabsf(sc ? v[m] * inv : v[m]);
      if (x > mx) mx = x;
    }
    const float scale = (mx > 0.0f && mx < INFINITY) ? mx / 448.0f : 1.0f;
    float a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int m = 0; m < d_pad; ++m) {
      uint8_t b = 0;
      if (m < d) b = f8_encode((sc ? v[m] * inv : v[m]) / scale);
      o[m] = b;
      const float f = f8_decode(b);
      a[m & 7] = fmaf(f, f, a[m & 7]);
    }
    scale_out[r] = scale;
    /* squared norm of the stored row: scale^2 * canonical sum of squares of the decoded bytes */
    if (norm2) norm2[r] = (scale * scale) * (((a[0] + a[4]) + (a[2] + a[6])) + ((a[1] + a[5]) + (a[3] + a[7])));
  }
}

float oracle_canon_dot_f8(const float* q, const uint8_t* row, float scale, int d) {
  float a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int m = 0; m < d; m += 8)
    for (int j = 0; j < 8; ++j) a[j] = fmaf(q[m + j], f8_decode(row[m + j]), a[j]);
  return scale * (((a[0] + a[4]) + (a[2] + a[6])) + ((a[1] + a[5]) + (a[3] + a[7])));
}

/* exact flat search over an fp8 corpus; same contract as oracle_flat_search_f16 */
int oracle_flat_search_f8(const uint8_t* corpus, const float* scales, int64_t n, int d_pad, const float* q, int nq,
                          int k, int64_t id_base, int64_t* out_ids, float* out_scores) {
  int nthreads = 1;
#ifdef _OPENMP
  nthreads = omp_get_max_threads();
#endif
  if (k < 1 || nq < 1) return nthreads;
  float lut[256];
  for (int b = 0; b < 256; ++b) lut[b] = f8_decode((uint8_t)b);
  uint64_t* heaps = (uint64_t*)malloc((size_t)nthreads * nq * k * sizeof(uint64_t));
  int* hn = (int*)calloc((size_t)nthreads * nq, sizeof(int));
#pragma omp parallel
  {
    int tid = 0;
#ifdef _OPENMP
    tid = omp_get_thread_num();
#endif
    uint64_t* H = heaps + (size_t)tid * nq * k;
    int* N = hn + (size_t)tid * nq;
    float* rowf = (float*)aligned_alloc(32, (size_t)d_pad * sizeof(float));
#pragma omp for schedule(dynamic, 64)
    for (int64_t r = 0; r < n; ++r) {
      const uint8_t* src = corpus + r * (int64_t)d_pad;
      for (int m = 0; m < d_pad; ++m) rowf[m] = lut[src[m]];
      for (int qi = 0; qi < nq; ++qi) {
        const float* qv = q + (size_t)qi * d_pad;
        __m256 a0 = _mm256_setzero_ps();
        for (int m = 0; m < d_pad; m += 8) a0 = _mm256_fmadd_ps(_mm256_loadu_ps(qv + m), _mm256_load_ps(rowf + m), a0);
        heap_push(H + (size_t)qi * k, &N[qi], k, candkey(scales[r] * tree8(a0), (uint32_t)r));
      }
    }
    free(rowf);
  }
  uint64_t* all = (uint64_t*)malloc((size_t)nthreads * k * sizeof(uint64_t));
  for (int qi = 0; qi < nq; ++qi) {
    int c = 0;
    for (int t = 0; t < nthreads; ++t) {
      const int m = hn[(size_t)t * nq + qi];
      memcpy(all + c, heaps + ((size_t)t * nq + qi) * k, (size_t)m * sizeof(uint64_t));
      c += m;
    }
    qsort(all, (size_t)c, sizeof(uint64_t), cmp_desc_u64);
    for (int i = 0; i < k; ++i) {
      if (i < c) {
        out_ids[(size_t)qi * k + i] = id_base + (int64_t)(uint32_t)(~(uint32_t)all[i]);
        out_scores[(size_t)qi * k + i] = unordkey((uint32_t)(all[i] >> 32));
      } else {
        out_ids[(size_t)qi * k + i] = -1;
        out_scores[(size_t)qi * k + i] = -INFINITY;
      }
    }
  }
  free(all);
  free(heaps);
  free(hn);
  return nthreads;
}

/* ---- deterministic synthetic rows (mirrors rag-arc_amd/csrc/prep.hip: rarc_synth_kernel) ------
 * v(seed,row,col) ~ N(0,1) ("D ~ N(0,1) ... or an identical counter-based generator"):
 * counter hash -> 52-bit uniform p in (0,1) -> z = Phi^-1(p) by Wichura's AS 241 (PPND16, rel. error 1e-16)
 * -> integer rint(z * 2^20).  Every step uses only exactly rounded double operations (+ - * / sqrt) in a fixed
 * order — the one transcendental, log, is the explicit series below — so the GPU generator reproduces the
 * integers bit for bit; rows are then normalised from the exact integer sum of squares.                      */
static inline uint64_t synth_mix(uint64_t z) {
  z += 0x9e3779b97f4a7c15ull;
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return z ^ (z >> 31);
}
/* ln x for normal positive x: x = m * 2^e, m in (sqrt(1/2), sqrt 2], ln m = 2 atanh(s), s = (m-1)/(m+1) */
static inline double synth_log(double x) {
  uint64_t u;
  memcpy(&u, &x, 8);
  int e = (int)((u >> 52) & 0x7ff) - 1023;
  u = (u & 0x000fffffffffffffull) | 0x3ff0000000000000ull;
  double m;
  memcpy(&m, &u, 8);
  if (m > 1.4142135623730951) { m = m * 0.5; e += 1; }
  const double s = (m - 1.0) / (m + 1.0), z = s * s;
  double t = 1.0 / 27.0;
  t = t * z + 1.0 / 25.0;
  t = t * z + 1.0 / 23.0;
  t = t * z + 1.0 / 21.0;
  t = t * z + 1.0 / 19.0;
  t = t * z + 1.0 / 17.0;
  t = t * z + 1.0 / 15.0;
  t = t * z + 1.0 / 13.0;
  t = t * z + 1.0 / 11.0;
  t = t * z + 1.0 / 9.0;
  t = t * z + 1.0 / 7.0;
  t = t * z + 1.0 / 5.0;
  t = t * z + 1.0 / 3.0;
  t = t * z + 1.0;
  return (double)e * 0.6931471805599453 + (s + s) * t;
}
static inline double synth_ppnd(double p) {
  const double q = p - 0.5;
  if (q >= -0.425 && q <= 0.425) {
    const double r = 0.180625 - q * q;
    const double num = (((((((2.5090809287301226727e+3 * r + 3.3430575583588128105e+4) * r + 6.7265770927008700853e+4) * r +
                            4.5921953931549871457e+4) * r + 1.3731693765509461125e+4) * r + 1.9715909503065514427e+3) * r +
                         1.3314166789178437745e+2) * r + 3.3871328727963666080e0);
    const double den = (((((((5.2264952788528545610e+3 * r + 2.8729085735721942674e+4) * r + 3.9307895800092710610e+4) * r +
                            2.1213794301586595867e+4) * r + 5.3941960214247511077e+3) * r + 6.8718700749205790830e+2) * r +
                         4.2313330701600911252e+1) * r + 1.0);
    return q * num / den;
  }
  double r = q < 0.0 ? p : 1.0 - p;
  r = sqrt(-synth_log(r));
  double v;
  if (r <= 5.0) {
    r = r - 1.6;
    const double num = (((((((7.74545014278341407640e-4 * r + 2.27238449892691845833e-2) * r + 2.41780725177450611770e-1) * r +
                            1.27045825245236838258e0) * r + 3.64784832476320460504e0) * r + 5.76949722146069140550e0) * r +
                         4.63033784615654529590e0) * r + 1.42343711074968357734e0);
    const double den = (((((((1.05075007164441684324e-9 * r + 5.47593808499534494600e-4) * r + 1.51986665636164571966e-2) * r +
                            1.48103976427480074590e-1) * r + 6.89767334985100004550e-1) * r + 1.67638483018380384940e0) * r +
                         2.05319162663775882187e0) * r + 1.0);
    v = num / den;
  } else {
    r = r - 5.0;
    const double num = (((((((2.01033439929228813265e-7 * r + 2.71155556874348757815e-5) * r + 1.24266094738807843860e-3) * r +
                            2.65321895265761230930e-2) * r + 2.96560571828504891230e-1) * r + 1.78482653991729133580e0) * r +
                         5.46378491116411436990e0) * r + 6.65790464350110377720e0);
    const double den = (((((((2.04426310338993978564e-15 * r + 1.42151175831644588870e-7) * r + 1.84631831751005468180e-5) * r +
                            7.86869131145613259100e-4) * r + 1.48753612908506148525e-2) * r + 1.36929880922735805310e-1) * r +
                         5.99832206555887937690e-1) * r + 1.0);
    v = num / den;
  }
  return q < 0.0 ? -v : v;
}
static inline int32_t synth_val(uint64_t seed, uint64_t row, uint32_t col) {
  const uint64_t h = synth_mix(synth_mix(seed ^ (row * 0xd1342543de82ef95ull)) + col);
  const double p = ((double)(h >> 12) + 0.5) * 2.220446049250313e-16; /* (i + 1/2) * 2^-52, exact */
  return (int32_t)rint(synth_ppnd(p) * 1048576.0);
}
/* test hooks: the pieces of the generator, so the CPU suite can pin them against scipy */
double oracle_synth_ppnd(double p) { return synth_ppnd(p); }
double oracle_synth_log(double x) { return synth_log(x); }
int32_t oracle_synth_val(uint64_t seed, uint64_t row, uint32_t col) { return synth_val(seed, row, col); }
static void synth_row(float* tmp, int d, uint64_t seed, uint64_t row) {
  uint64_t ss = 0;
  int32_t* iv = (int32_t*)tmp; /* same size as float: integers first, scaled in place afterwards */
  for (int c = 0; c < d; ++c) {
    const int64_t v = synth_val(seed, row, (uint32_t)c);
    iv[c] = (int32_t)v;
    ss += (uint64_t)(v * v);
  }
  const double scale = ss > 0 ? 1.0 / sqrt((double)ss) : 0.0;
  for (int c = 0; c < d; ++c) tmp[c] = (float)((double)iv[c] * scale);
}
void oracle_synth_rows_f16(uint16_t* out, int d_pad, int d, int64_t first_row, int64_t n, uint64_t seed) {
#pragma omp parallel
  {
    float* tmp = (float*)malloc((size_t)d * sizeof(float));
#pragma omp for schedule(static)
    for (int64_t r = 0; r < n; ++r) {
      synth_row(tmp, d, seed, (uint64_t)(first_row + r));
      for (int c = 0; c < d_pad; ++c) out[r * (int64_t)d_pad + c] = c < d ? f2h(tmp[c]) : 0;
    }
    free(tmp);
  }
}
void oracle_synth_rows_f32(float* out, int64_t ld, int d, int64_t first_row, int64_t n, uint64_t seed) {
#pragma omp parallel for schedule(static)
  for (int64_t r = 0; r < n; ++r) synth_row(out + r * ld, d, seed, (uint64_t)(first_row + r));
}

/* ---- helpers exported for the python wrapper -------------------------------------------------- */
void oracle_f32_to_f16(const float* in, uint16_t* out, int64_t n) {
  for (int64_t i = 0; i < n; ++i) out[i] = f2h(in[i]);
}
int oracle_selftest(void) {
  /* the F16C conversions must agree with the portable bit-level ones on every half value */
  for (uint32_t h = 0; h < 65536; ++h) {
    const float a = h2f((uint16_t)h), b = h2f_soft((uint16_t)h);
    if (memcmp(&a, &b, 4) != 0 && !(a != a && b != b)) return 1;
    if (a == a && f2h(a) != (uint16_t)h) return 2;
  }
  return 0;
}
int oracle_cpu_ok(void) {
  return __builtin_cpu_supports("avx2") && __builtin_cpu_supports("fma") && __builtin_cpu_supports("f16c");
}
