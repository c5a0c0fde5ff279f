/* _rarc_hostmap — the host side of a search answer: (row, score) arrays -> the python objects the reference's interface
 * returns.  CPython C-API, no GPU code; compiled by gcc into rag-arc_amd/lib/_rarc_hostmap<EXT_SUFFIX> (csrc/Makefile).
 *
 * What it replaces (reference): the result loop of FaissVectorStore.similarity_search_by_vector_with_score
 * (encapsulation/database/vector_db/VectorStore_Faiss.py:265-272: two dict look-ups + float() per hit) and the
 * content-keyed bookkeeping of RRFusion.fuse (core/utils/Fusion.py:45-76: doc_scores / doc_objects dicts).  The reference
 * answers ONE query per call, so that loop is 100 iterations; a 256 x 100 batch is 25,600 — in python 60 ms over a 1M-entry
 * docstore (more than the whole 100M-row scan).  Here the loop runs in C with the Document objects prefetched ahead of
 * their INCREF (a corpus-sized docstore is far larger than any cache: every hit is a miss).
 *
 * Semantics are exactly the python loops' (tests/test_hostmap.py holds those loops and compares):
 *   rows_to_docs / rows_to_pairs : row -1 is skipped; score -> python float of the fp32 value.
 *   rrf_tables                   : keys in first-seen order of `content` (dict semantics: any hashable), the Document kept
 *                                  for a key is the LAST one seen (Fusion.py:62-63).
 */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <stdint.h>

#define PREFETCH_AHEAD 12

static PyObject *s_content;   /* interned "content" */

static int get_buffer(PyObject *obj, Py_buffer *view, Py_ssize_t itemsize, Py_ssize_t need, const char *what) {
    if (PyObject_GetBuffer(obj, view, PyBUF_C_CONTIGUOUS) < 0) return -1;
    if (view->len < need * itemsize) {
        PyBuffer_Release(view);
        PyErr_Format(PyExc_ValueError, "%s: buffer holds %zd bytes, %zd needed", what, view->len, need * itemsize);
        return -1;
    }
    return 0;
}

/* one row of the docstore: borrowed for exact lists (fast path), new reference otherwise (any sequence: the columnar
 * docstore builds its Document here) */
static inline PyObject *seq_item(PyObject *seq, int is_list, Py_ssize_t n, int64_t row) {
    if (row < 0 || row >= n) {
        PyErr_Format(PyExc_IndexError, "row %lld is outside the docstore (%zd rows)", (long long)row, n);
        return NULL;
    }
    if (is_list) {
        PyObject *o = PyList_GET_ITEM(seq, (Py_ssize_t)row);
        Py_INCREF(o);
        return o;
    }
    return PySequence_GetItem(seq, (Py_ssize_t)row);
}

static PyObject *map_rows(PyObject *args, int with_scores) {
    PyObject *seq, *rows_o, *scores_o = NULL;
    Py_ssize_t nq, k;
    long long base = 0;
    if (with_scores) {
        if (!PyArg_ParseTuple(args, "OOOnn|L", &seq, &rows_o, &scores_o, &nq, &k, &base)) return NULL;
    } else {
        if (!PyArg_ParseTuple(args, "OOnn|L", &seq, &rows_o, &nq, &k, &base)) return NULL;
    }
    if (nq < 0 || k < 0) { PyErr_SetString(PyExc_ValueError, "negative shape"); return NULL; }
    int is_list = PyList_CheckExact(seq);
    Py_ssize_t n = is_list ? PyList_GET_SIZE(seq) : PySequence_Size(seq);
    if (n < 0) return NULL;
    Py_buffer rv, sv;
    if (get_buffer(rows_o, &rv, 8, nq * k, "rows") < 0) return NULL;
    if (with_scores && get_buffer(scores_o, &sv, 4, nq * k, "scores") < 0) { PyBuffer_Release(&rv); return NULL; }
    const int64_t *rows = (const int64_t *)rv.buf;
    const float *scores = with_scores ? (const float *)sv.buf : NULL;
    PyObject *out = PyList_New(nq);
    if (!out) goto fail;
    const Py_ssize_t total = nq * k;
    for (Py_ssize_t q = 0; q < nq; ++q) {
        PyObject *one = PyList_New(0);
        if (!one) goto fail_out;
        PyList_SET_ITEM(out, q, one);
        for (Py_ssize_t j = 0; j < k; ++j) {
            const Py_ssize_t at = q * k + j;
            if (is_list && at + PREFETCH_AHEAD < total) {
                const int64_t ahead = rows[at + PREFETCH_AHEAD] - base;
                if (ahead >= 0 && ahead < n) __builtin_prefetch(PyList_GET_ITEM(seq, (Py_ssize_t)ahead), 1, 1);
            }
            if (rows[at] == -1) continue;
            PyObject *doc = seq_item(seq, is_list, n, rows[at] - base);
            if (!doc) goto fail_out;
            PyObject *item = doc;
            if (with_scores) {
                PyObject *f = PyFloat_FromDouble((double)scores[at]);
                item = f ? PyTuple_Pack(2, doc, f) : NULL;
                Py_XDECREF(f);
                Py_DECREF(doc);
                if (!item) goto fail_out;
            }
            int rc = PyList_Append(one, item);
            Py_DECREF(item);
            if (rc < 0) goto fail_out;
        }
    }
    PyBuffer_Release(&rv);
    if (with_scores) PyBuffer_Release(&sv);
    return out;
fail_out:
    Py_DECREF(out);
fail:
    PyBuffer_Release(&rv);
    if (with_scores) PyBuffer_Release(&sv);
    return NULL;
}

static PyObject *rows_to_docs(PyObject *self, PyObject *args) { (void)self; return map_rows(args, 0); }
static PyObject *rows_to_pairs(PyObject *self, PyObject *args) { (void)self; return map_rows(args, 1); }

/* rrf_tables(batch, n_lists, max_len) -> (keys: bytes int64 [nq][n_lists][max_len], lens: bytes int32 [nq][n_lists],
 *                                          docs: list [nq] of list [n_keys] Document)
 * batch[q][l] is a list of Documents (the answer of retriever l to query q). */
static PyObject *rrf_tables(PyObject *self, PyObject *args) {
    (void)self;
    PyObject *batch;
    Py_ssize_t n_lists, max_len;
    if (!PyArg_ParseTuple(args, "O!nn", &PyList_Type, &batch, &n_lists, &max_len)) return NULL;
    if (n_lists < 0 || max_len < 0) { PyErr_SetString(PyExc_ValueError, "negative shape"); return NULL; }
    const Py_ssize_t nq = PyList_GET_SIZE(batch);
    PyObject *keys_b = PyBytes_FromStringAndSize(NULL, nq * n_lists * max_len * 8);
    PyObject *lens_b = PyBytes_FromStringAndSize(NULL, nq * n_lists * 4);
    PyObject *docs = PyList_New(nq);
    PyObject *table = NULL;
    if (!keys_b || !lens_b || !docs) goto fail;
    int64_t *keys = (int64_t *)PyBytes_AS_STRING(keys_b);
    int32_t *lens = (int32_t *)PyBytes_AS_STRING(lens_b);
    memset(keys, 0, (size_t)(nq * n_lists * max_len * 8));
    memset(lens, 0, (size_t)(nq * n_lists * 4));
    for (Py_ssize_t q = 0; q < nq; ++q) {
        PyObject *lists = PyList_GET_ITEM(batch, q);
        if (!PyList_Check(lists) || PyList_GET_SIZE(lists) > n_lists) {
            PyErr_SetString(PyExc_ValueError, "batch[q] must be a list of at most n_lists lists");
            goto fail;
        }
        PyObject *by_key = PyList_New(0);
        if (!by_key) goto fail;
        PyList_SET_ITEM(docs, q, by_key);
        table = PyDict_New();
        if (!table) goto fail;
        for (Py_ssize_t l = 0; l < PyList_GET_SIZE(lists); ++l) {
            PyObject *one = PyList_GET_ITEM(lists, l);
            if (!PyList_Check(one) || PyList_GET_SIZE(one) > max_len) {
                PyErr_SetString(PyExc_ValueError, "batch[q][l] must be a list of at most max_len documents");
                goto fail;
            }
            lens[q * n_lists + l] = (int32_t)PyList_GET_SIZE(one);
            for (Py_ssize_t p = 0; p < PyList_GET_SIZE(one); ++p) {
                PyObject *doc = PyList_GET_ITEM(one, p);
                PyObject *content = PyObject_GetAttr(doc, s_content);
                if (!content) goto fail;
                PyObject *kobj = PyDict_GetItemWithError(table, content);   /* borrowed */
                Py_ssize_t key;
                if (kobj) {
                    key = PyLong_AsSsize_t(kobj);
                    Py_INCREF(doc);
                    if (PyList_SetItem(by_key, key, doc) < 0) { Py_DECREF(content); goto fail; }   /* last one wins */
                } else {
                    if (PyErr_Occurred()) { Py_DECREF(content); goto fail; }
                    key = PyList_GET_SIZE(by_key);
                    PyObject *knew = PyLong_FromSsize_t(key);
                    int rc = knew ? PyDict_SetItem(table, content, knew) : -1;
                    Py_XDECREF(knew);
                    if (rc < 0 || PyList_Append(by_key, doc) < 0) { Py_DECREF(content); goto fail; }
                }
                Py_DECREF(content);
                keys[(q * n_lists + l) * max_len + p] = (int64_t)key;
            }
        }
        Py_CLEAR(table);
    }
    {
        PyObject *ret = PyTuple_Pack(3, keys_b, lens_b, docs);
        Py_DECREF(keys_b); Py_DECREF(lens_b); Py_DECREF(docs);
        return ret;
    }
fail:
    Py_XDECREF(table);
    Py_XDECREF(keys_b); Py_XDECREF(lens_b); Py_XDECREF(docs);
    return NULL;
}

/* pick_docs(docs_by_key, fused_keys int64 [nq][width], counts int32 [nq], width) -> list [nq] of list [counts[q]] Document */
static PyObject *pick_docs(PyObject *self, PyObject *args) {
    (void)self;
    PyObject *docs, *keys_o, *counts_o;
    Py_ssize_t width;
    if (!PyArg_ParseTuple(args, "O!OOn", &PyList_Type, &docs, &keys_o, &counts_o, &width)) return NULL;
    const Py_ssize_t nq = PyList_GET_SIZE(docs);
    Py_buffer kv, cv;
    if (get_buffer(keys_o, &kv, 8, nq * width, "fused keys") < 0) return NULL;
    if (get_buffer(counts_o, &cv, 4, nq, "counts") < 0) { PyBuffer_Release(&kv); return NULL; }
    const int64_t *keys = (const int64_t *)kv.buf;
    const int32_t *counts = (const int32_t *)cv.buf;
    PyObject *out = PyList_New(nq);
    if (!out) goto fail;
    for (Py_ssize_t q = 0; q < nq; ++q) {
        PyObject *by_key = PyList_GET_ITEM(docs, q);
        Py_ssize_t n = counts[q] < 0 ? 0 : (counts[q] > width ? width : counts[q]);
        PyObject *one = PyList_New(n);
        if (!one) { Py_DECREF(out); goto fail; }
        PyList_SET_ITEM(out, q, one);
        for (Py_ssize_t i = 0; i < n; ++i) {
            const int64_t key = keys[q * width + i];
            if (!PyList_Check(by_key) || key < 0 || key >= PyList_GET_SIZE(by_key)) {
                PyErr_Format(PyExc_IndexError, "fused key %lld of query %zd is not a key of that query", (long long)key, q);
                Py_DECREF(out);
                goto fail;
            }
            PyObject *doc = PyList_GET_ITEM(by_key, (Py_ssize_t)key);
            Py_INCREF(doc);
            PyList_SET_ITEM(one, i, doc);
        }
    }
    PyBuffer_Release(&kv); PyBuffer_Release(&cv);
    return out;
fail:
    PyBuffer_Release(&kv); PyBuffer_Release(&cv);
    return NULL;
}

static PyMethodDef methods[] = {
    {"rows_to_docs", rows_to_docs, METH_VARARGS,
     "rows_to_docs(docstore_seq, rows int64 [nq][k], nq, k, base=0) -> [[Document]] (row -1 skipped)"},
    {"rows_to_pairs", rows_to_pairs, METH_VARARGS,
     "rows_to_pairs(docstore_seq, rows int64 [nq][k], scores fp32 [nq][k], nq, k, base=0) -> [[(Document, float)]]"},
    {"rrf_tables", rrf_tables, METH_VARARGS,
     "rrf_tables(batch [nq][n_lists][docs], n_lists, max_len) -> (keys bytes, lens bytes, docs_by_key)"},
    {"pick_docs", pick_docs, METH_VARARGS,
     "pick_docs(docs_by_key, fused_keys int64 [nq][width], counts int32 [nq], width) -> [[Document]]"},
    {NULL, NULL, 0, NULL}};

static struct PyModuleDef module = {PyModuleDef_HEAD_INIT, "_rarc_hostmap",
                                    "search answer -> python objects (rows -> Documents, RRF key tables)", -1, methods,
                                    NULL, NULL, NULL, NULL};

PyMODINIT_FUNC PyInit__rarc_hostmap(void) {
    s_content = PyUnicode_InternFromString("content");
    if (!s_content) return NULL;
    return PyModule_Create(&module);
}
