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

static PyObject *s_content, *s_metadata, *s_id;   /* interned attribute names */

/* The cyclic collector counts container allocations: a 256 x 100 answer allocates tens of thousands of lists / tuples /
 * Documents in one burst, i.e. dozens of young collections, every tenth of which promotes into — and eventually triggers —
 * the old generations, whose cost is the size of the PROCESS (a million-Document docstore: 100+ ms).  Nothing allocated
 * here is cyclic garbage: the collector is held off for the duration of a bulk call (process-wide, microseconds to
 * milliseconds) and the one young collection that follows sees the survivors once. */
#define GC_HOLD() int gc_was_on_ = PyGC_Disable()
#define GC_RELEASE() do { if (gc_was_on_) PyGC_Enable(); } while (0)

/* A ColumnarDocstore (encapsulation/database/vector_db/docstore.py) seen from C: byte columns + offsets; a Document is
 * built per hit WITHOUT running the dataclass __init__ (object.__new__ + three attribute stores — the same instance
 * dict; tests/test_hostmap.py compares with Document(content=..., metadata=..., id=...)). */
typedef struct {
    PyTypeObject *doc_type;
    Py_buffer text_blob, text_off, id_blob, id_off;
    int has_ids;
    PyObject *metadatas;      /* NULL / None: every document gets a fresh {} */
    Py_ssize_t n;
} Columns;

static void columns_release(Columns *c) {
    if (c->text_blob.obj) PyBuffer_Release(&c->text_blob);
    if (c->text_off.obj) PyBuffer_Release(&c->text_off);
    if (c->id_blob.obj) PyBuffer_Release(&c->id_blob);
    if (c->id_off.obj) PyBuffer_Release(&c->id_off);
}

/* columns: (Document type, text_blob, text_off, id_blob | None, id_off | None, metadatas | None) */
static int columns_parse(PyObject *tup, Columns *c) {
    memset(c, 0, sizeof(*c));
    PyObject *type_o, *tb, *to, *ib, *io, *md;
    if (!PyArg_ParseTuple(tup, "OOOOOO", &type_o, &tb, &to, &ib, &io, &md)) return -1;
    if (!PyType_Check(type_o)) { PyErr_SetString(PyExc_TypeError, "columns[0] must be the Document class"); return -1; }
    c->doc_type = (PyTypeObject *)type_o;
    if (PyObject_GetBuffer(tb, &c->text_blob, PyBUF_C_CONTIGUOUS) < 0) goto fail;
    if (PyObject_GetBuffer(to, &c->text_off, PyBUF_C_CONTIGUOUS) < 0) goto fail;
    if (c->text_off.len < 8 || c->text_off.len % 8) { PyErr_SetString(PyExc_ValueError, "text offsets: int64 [n + 1]"); goto fail; }
    c->n = c->text_off.len / 8 - 1;
    c->has_ids = (ib != Py_None);
    if (c->has_ids) {
        if (PyObject_GetBuffer(ib, &c->id_blob, PyBUF_C_CONTIGUOUS) < 0) goto fail;
        if (PyObject_GetBuffer(io, &c->id_off, PyBUF_C_CONTIGUOUS) < 0) goto fail;
        if (c->id_off.len != c->text_off.len) { PyErr_SetString(PyExc_ValueError, "one id per text"); goto fail; }
    }
    c->metadatas = (md == Py_None) ? NULL : md;
    return 0;
fail:
    columns_release(c);
    return -1;
}

static PyObject *column_str(const Py_buffer *blob, const Py_buffer *off, Py_ssize_t row) {
    const int64_t *o = (const int64_t *)off->buf;
    if (o[row] < 0 || o[row + 1] < o[row] || o[row + 1] > blob->len) {
        PyErr_Format(PyExc_ValueError, "row %zd: offsets outside the blob", row);
        return NULL;
    }
    return PyUnicode_DecodeUTF8((const char *)blob->buf + o[row], (Py_ssize_t)(o[row + 1] - o[row]), "strict");
}

static PyObject *columns_item(const Columns *c, int64_t row) {
    if (row < 0 || row >= c->n) {
        PyErr_Format(PyExc_IndexError, "row %lld is outside the docstore (%zd rows)", (long long)row, c->n);
        return NULL;
    }
    PyObject *content = column_str(&c->text_blob, &c->text_off, (Py_ssize_t)row);
    if (!content) return NULL;
    PyObject *doc_id, *meta, *doc = NULL;
    if (c->has_ids) doc_id = column_str(&c->id_blob, &c->id_off, (Py_ssize_t)row);
    else { doc_id = content; Py_INCREF(doc_id); }
    meta = c->metadatas ? PySequence_GetItem(c->metadatas, (Py_ssize_t)row) : PyDict_New();
    if (doc_id && meta) doc = c->doc_type->tp_alloc(c->doc_type, 0);
    if (doc && (PyObject_SetAttr(doc, s_content, content) < 0 || PyObject_SetAttr(doc, s_metadata, meta) < 0 ||
                PyObject_SetAttr(doc, s_id, doc_id) < 0))
        Py_CLEAR(doc);
    Py_DECREF(content);
    Py_XDECREF(doc_id);
    Py_XDECREF(meta);
    return doc;
}

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
    Columns cols;
    /* a ColumnarDocstore's columns (see columns_parse): a 6-tuple that starts with a class — any other tuple is a sequence
     * of Documents like a list */
    int is_cols = PyTuple_CheckExact(seq) && PyTuple_GET_SIZE(seq) == 6 && PyType_Check(PyTuple_GET_ITEM(seq, 0));
    if (is_cols && columns_parse(seq, &cols) < 0) return NULL;
    Py_ssize_t n = is_list ? PyList_GET_SIZE(seq) : (is_cols ? cols.n : PySequence_Size(seq));
    if (n < 0) return NULL;
    Py_buffer rv, sv;
    if (get_buffer(rows_o, &rv, 8, nq * k, "rows") < 0) { if (is_cols) columns_release(&cols); return NULL; }
    if (with_scores && get_buffer(scores_o, &sv, 4, nq * k, "scores") < 0) {
        PyBuffer_Release(&rv);
        if (is_cols) columns_release(&cols);
        return NULL;
    }
    GC_HOLD();
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
            if (is_cols) {          /* the offset of a row, then its bytes: two misses per hit in corpus-sized columns */
                if (at + PREFETCH_AHEAD < total) {
                    const int64_t r = rows[at + PREFETCH_AHEAD] - base;
                    if (r >= 0 && r < n) {
                        __builtin_prefetch((const int64_t *)cols.text_off.buf + r, 0, 1);
                        if (cols.has_ids) __builtin_prefetch((const int64_t *)cols.id_off.buf + r, 0, 1);
                    }
                }
                if (at + PREFETCH_AHEAD / 2 < total) {
                    const int64_t r = rows[at + PREFETCH_AHEAD / 2] - base;
                    if (r >= 0 && r < n) {
                        const int64_t o = ((const int64_t *)cols.text_off.buf)[r];
                        if (o >= 0 && o < cols.text_blob.len) __builtin_prefetch((const char *)cols.text_blob.buf + o, 0, 1);
                        if (cols.has_ids) {
                            const int64_t oi = ((const int64_t *)cols.id_off.buf)[r];
                            if (oi >= 0 && oi < cols.id_blob.len) __builtin_prefetch((const char *)cols.id_blob.buf + oi, 0, 1);
                        }
                    }
                }
            }
            if (rows[at] == -1) continue;
            PyObject *doc = is_cols ? columns_item(&cols, rows[at] - base) : seq_item(seq, is_list, n, rows[at] - base);
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
    GC_RELEASE();
    PyBuffer_Release(&rv);
    if (with_scores) PyBuffer_Release(&sv);
    if (is_cols) columns_release(&cols);
    return out;
fail_out:
    Py_DECREF(out);
fail:
    GC_RELEASE();
    PyBuffer_Release(&rv);
    if (with_scores) PyBuffer_Release(&sv);
    if (is_cols) columns_release(&cols);
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
    GC_HOLD();
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
            const Py_ssize_t len = PyList_GET_SIZE(one);
            for (Py_ssize_t p = 0; p < len; ++p) {
                /* the documents of an answer are scattered over a corpus-sized heap: every one is a chain of cache misses
                 * (object -> its dict -> the values array -> the content string).  Walk the chain a few items ahead. */
                if (p + 12 < len) __builtin_prefetch(PyList_GET_ITEM(one, p + 12), 0, 1);
                if (p + 8 < len) {
                    PyObject **dp = _PyObject_GetDictPtr(PyList_GET_ITEM(one, p + 8));
                    if (dp && *dp) __builtin_prefetch(*dp, 0, 1);
                }
#if PY_VERSION_HEX < 0x030B0000   /* (the layout of a split dict below is 3.10's; later versions keep the first two steps) */
                if (p + 5 < len) {
                    PyObject **dp = _PyObject_GetDictPtr(PyList_GET_ITEM(one, p + 5));
                    if (dp && *dp && PyDict_CheckExact(*dp) && ((PyDictObject *)*dp)->ma_values)
                        __builtin_prefetch(((PyDictObject *)*dp)->ma_values, 0, 1);
                }
                if (p + 2 < len) {
                    PyObject **dp = _PyObject_GetDictPtr(PyList_GET_ITEM(one, p + 2));
                    if (dp && *dp && PyDict_CheckExact(*dp) && ((PyDictObject *)*dp)->ma_values &&
                        ((PyDictObject *)*dp)->ma_values[0])
                        __builtin_prefetch(((PyDictObject *)*dp)->ma_values[0], 0, 1);   /* first attribute set: content */
                }
#endif
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
        GC_RELEASE();
        PyObject *ret = PyTuple_Pack(3, keys_b, lens_b, docs);
        Py_DECREF(keys_b); Py_DECREF(lens_b); Py_DECREF(docs);
        return ret;
    }
fail:
    GC_RELEASE();
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
    s_metadata = PyUnicode_InternFromString("metadata");
    s_id = PyUnicode_InternFromString("id");
    if (!s_content || !s_metadata || !s_id) return NULL;
    return PyModule_Create(&module);
}
