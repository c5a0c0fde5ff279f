"""ctypes binding of librarc_hip.so (C-ABI declared in include/rarc.h).

There is no CPU fallback: if the library is missing or a call fails, a RarcError is raised.
"""
from __future__ import annotations

import ctypes
import os
import threading
from ctypes import c_double, c_float, c_int, c_int64, c_size_t, c_uint64, c_void_p

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# (RARC_LIBRARY: load another build of the same ABI, e.g. a measurement build; default = the in-tree library)
_LIB = os.environ.get("RARC_LIBRARY") or os.path.join(_PKG, "lib", "librarc_hip.so")

MAX_QUERIES = 256
MAX_K = 1024          # the register-resident scans; beyond it (and beyond 1024 padded dims): the wide path
WIDE_MAX_K = 8192
WIDE_MAX_DPAD = 4096
DIM_ALIGN = 128
Q_UNCERTAIN = 1
Q_OVERFLOW = 2
WS_ANYFLAG_OFFSET = 5124


class RarcError(RuntimeError):
    """Raised for every non-zero status returned by librarc_hip.so."""


class RarcUnsupported(RarcError, NotImplementedError):
    """A search the reference's `IndexFlatIP.search` would answer (any d, any k: VectorStore_Faiss.py:262) that this backend
    REFUSES — the storage x path matrix in INTEGRATION.md ("What the backend refuses").  A refusal is loud and happens before
    anything is launched; the message names the configuration that does answer the call.  Also a NotImplementedError, the
    type the reference's VectorStore base raises for what a store does not implement (VectorStoreBase.py:145-176)."""


class EncLayer(ctypes.Structure):
    """RarcEncLayer (include/rarc.h): device pointers of one transformer layer."""
    _fields_ = [(n, c_void_p) for n in ("qkv_w", "qkv_b", "o_w", "o_b", "ln1_g", "ln1_b", "f1_w", "f1_b", "f2_w",
                                        "f2_b", "ln2_g", "ln2_b")]


class EncModel(ctypes.Structure):
    """RarcEncModel (include/rarc.h)."""
    _fields_ = [("hidden", c_int), ("heads", c_int), ("inter", c_int), ("n_layers", c_int), ("ln_eps", c_float),
                ("word", c_void_p), ("pos", c_void_p), ("type0", c_void_p), ("emb_g", c_void_p), ("emb_b", c_void_p),
                ("layers", ctypes.POINTER(EncLayer)), ("vocab", c_int), ("max_pos", c_int),
                ("rel_bias", c_void_p), ("rel_span", c_int)]


class Enc32Layer(ctypes.Structure):
    """RarcEnc32Layer (include/rarc.h): split fp16 weight images + fp32 scales / biases / LayerNorm of one layer."""
    _fields_ = [(n, c_void_p) for n in ("qkv_w3", "qkv_rw", "qkv_b", "o_w3", "o_rw", "o_b", "ln1_g", "ln1_b", "f1_w3", "f1_rw",
                                        "f1_b", "f2_w3", "f2_rw", "f2_b", "ln2_g", "ln2_b", "f1_colmax",
                                        "qkv_wq", "o_wq", "f1_wq", "f2_wq")]


class Enc32Model(ctypes.Structure):
    """RarcEnc32Model (include/rarc.h)."""
    _fields_ = [("hidden", c_int), ("heads", c_int), ("inter", c_int), ("n_layers", c_int), ("ln_eps", c_float),
                ("word", c_void_p), ("pos", c_void_p), ("type0", c_void_p), ("emb_g", c_void_p), ("emb_b", c_void_p),
                ("layers", ctypes.POINTER(Enc32Layer)), ("vocab", c_int), ("max_pos", c_int),
                ("rel_bias", c_void_p), ("rel_span", c_int)]


class SearchBatch(ctypes.Structure):
    """RarcSearchBatch (include/rarc.h): one batch — query prep + search — in one call."""
    _fields_ = [("d_rows", c_void_p), ("d_aux", c_void_p), ("row_format", c_int), ("n_rows", c_int64), ("d_pad", c_int),
                ("d_qmeta", c_void_p),
                ("d_queries", c_void_p), ("ld_queries", c_int64), ("nq", c_int), ("d", c_int), ("normalize", c_int),
                ("corpus_max_norm", c_float), ("d_qblock", c_void_p),
                ("k", c_int), ("kprime", c_int), ("id_base", c_int64), ("bin_lo", c_float), ("bin_hi", c_float),
                ("d_out_ids", c_void_p), ("d_out_scores", c_void_p), ("d_status", c_void_p), ("flag_host", c_void_p),
                ("d_workspace", c_void_p), ("workspace_bytes", c_size_t), ("cand_cap", c_int),
                ("gate_event", c_void_p)]


class IoStats(ctypes.Structure):
    """RarcIoStats (include/rarc.h): what a shard-file transfer moved and how fast."""
    _fields_ = [("bytes", c_int64), ("seconds", c_double), ("file_seconds", c_double), ("copy_wait_seconds", c_double),
                ("direct_bytes", c_int64), ("n_chunks", c_int64), ("slot_bytes", c_int64), ("n_threads", c_int),
                ("direct", c_int)]


IO_DIRECT = 1
IO_FSYNC = 2
IO_TRUNCATE = 4


class LmLayer(ctypes.Structure):
    """RarcLmLayer (include/rarc.h): device pointers of one decoder layer."""
    _fields_ = [(n, c_void_p) for n in ("in_norm", "qkv_w", "q_norm", "k_norm", "o_w", "post_norm", "gate_up_w", "down_w",
                                        "qkv_w_folded", "gate_up_w_folded")]


class LmModel(ctypes.Structure):
    """RarcLmModel (include/rarc.h)."""
    _fields_ = [("hidden", c_int), ("n_layers", c_int), ("n_q_heads", c_int), ("n_kv_heads", c_int), ("head_dim", c_int),
                ("inter", c_int), ("vocab", c_int), ("rms_eps", c_float), ("rope_theta", c_float), ("embed", c_void_p),
                ("lm_head", c_void_p), ("final_norm", c_void_p), ("zero_bias", c_void_p), ("layers", ctypes.POINTER(LmLayer))]


_lock = threading.Lock()
_lib = None

# name -> (restype, argtypes); every symbol include/rarc.h declares is listed here
SIGNATURES = {
    "rarc_version": (c_int, []),
    "rarc_last_error": (ctypes.c_char_p, []),
    "rarc_padded_dim": (c_int, [c_int]),
    "rarc_l2norm_rows_f32": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int, c_void_p]),
    "rarc_cosine_matrix_f32": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p]),
    "rarc_adjacent_cosine_distance_f32": (c_int, [c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p]),
    "rarc_ingest_f16": (c_int, [c_void_p, c_int64, c_void_p, c_int, c_void_p, c_int64, c_int, c_int, c_void_p]),
    "rarc_quant_meta_floats": (c_size_t, [c_int64]),
    "rarc_quant_meta_f16": (c_int, [c_void_p, c_int64, c_int, c_int64, c_void_p, c_void_p]),
    "rarc_debug_q8_scores": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "rarc_query_block_bytes": (c_size_t, [c_int]),
    "rarc_prep_queries": (c_int, [c_void_p, c_int64, c_int, c_int, c_int, c_int, c_float, c_void_p, c_void_p,
                                  c_void_p]),
    "rarc_qblock_set_floor": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "rarc_search_workspace_bytes": (c_size_t, [c_int]),
    "rarc_search_f16": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p, c_int, c_int, c_int,
                                c_int64, c_float, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t,
                                c_int, c_void_p]),
    "rarc_repair_f16": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_int, c_int, c_int64, c_void_p, c_void_p,
                                c_void_p, c_void_p, c_size_t, c_void_p]),
    "rarc_verify_batch": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_int, c_void_p, c_int, c_int, c_int, c_int64, c_void_p,
                                  c_void_p, c_void_p, c_void_p]),
    "rarc_quant_shadow_f16": (c_int, [c_void_p, c_int64, c_int, c_int64, c_void_p, c_void_p, c_void_p]),
    "rarc_search_f16_shadow": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_int, c_int, c_int,
                                       c_int64, c_float, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t,
                                       c_int, c_void_p]),
    "rarc_ingest_f32": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_int, c_void_p, c_int64, c_int, c_int, c_void_p]),
    "rarc_search_f32": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_int, c_int, c_int,
                                c_int64, c_float, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t,
                                c_int, c_void_p]),
    "rarc_repair_f32": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_int, c_int, c_int64, c_void_p, c_void_p,
                                c_void_p, c_void_p, c_size_t, c_void_p]),
    "rarc_padded_dim_f8": (c_int, [c_int]),
    "rarc_ingest_f8": (c_int, [c_void_p, c_int64, c_void_p, c_int, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p]),
    "rarc_quant_meta_floats_f8": (c_size_t, [c_int64]),
    "rarc_quant_meta_f8": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int64, c_void_p, c_void_p]),
    "rarc_search_f8": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_int, c_int, c_int,
                               c_int64, c_float, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t,
                               c_int, c_void_p]),
    "rarc_repair_f8": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p, c_int, c_int, c_int64, c_void_p,
                               c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "rarc_topk_merge": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "rarc_pack_results": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "rarc_topk_merge_packed": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "rarc_rrf_fuse": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_double, c_int, c_void_p, c_void_p,
                              c_void_p, c_void_p]),
    "rarc_rerank_order": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "rarc_mmr_workspace_doubles": (c_size_t, [c_int, c_int]),
    "rarc_mmr_select": (c_int, [c_void_p, c_int64, c_void_p, c_int, c_int, c_int, c_int, c_double, c_void_p, c_void_p,
                                c_void_p]),
    "rarc_synth_rows_f16": (c_int, [c_void_p, c_int, c_int, c_int64, c_int64, c_uint64, c_void_p]),
    "rarc_synth_rows_f32": (c_int, [c_void_p, c_int64, c_int, c_int64, c_int64, c_uint64, c_void_p]),
    "rarc_enc_embed_ln": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_int, c_int,
                                  c_int, c_int, c_void_p, c_void_p]),
    "rarc_enc_gemm": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "rarc_enc_gemm_zero_bias": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "rarc_enc_attention": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "rarc_enc_add_ln": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_int, c_int, c_void_p, c_void_p]),
    "rarc_enc_pool": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "rarc_enc_pool_mean": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "rarc_enc_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "rarc_enc_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_size_t, c_void_p,
                                 c_void_p]),
    "rarc_enc32_split_weight": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "rarc_search_batch": (c_int, [c_void_p, c_void_p]),
    "rarc_stream_read": (c_int, [c_void_p, c_size_t, c_void_p, c_void_p]),
    "rarc_enc32_pack_query_weight": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "rarc_enc32_split_rows": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "rarc_enc32_gemm": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "rarc_enc32_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "rarc_enc32_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_size_t, c_void_p,
                                   c_void_p]),
    "rarc_lm_workspace_bytes": (c_size_t, [c_void_p, c_int]),
    "rarc_lm_yes_no_logits": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_size_t, c_void_p,
                                      c_void_p]),
    "rarc_lm_prefix_cache_bytes": (c_size_t, [c_void_p, c_int]),
    "rarc_lm_prefix_kv": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_size_t, c_void_p, c_size_t, c_void_p]),
    "rarc_lm_yes_no_logits_prefixed": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_int,
                                                c_void_p, c_int, c_int, c_void_p, c_size_t, c_void_p, c_void_p]),
    "rarc_wordpiece_create": (c_int, [ctypes.c_char_p, c_size_t, c_int, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p,
                                      ctypes.c_char_p, ctypes.c_char_p, c_size_t, c_int, ctypes.POINTER(c_void_p)]),
    "rarc_wordpiece_destroy": (None, [c_void_p]),
    "rarc_wordpiece_encode": (c_int, [c_void_p, ctypes.c_char_p, c_void_p, c_int, c_int, c_void_p, c_int64, c_void_p, c_int]),
    "rarc_file_to_device": (c_int, [ctypes.c_char_p, c_int, ctypes.POINTER(c_int64), ctypes.POINTER(c_int64),
                                    ctypes.POINTER(c_int64), c_void_p, c_int64, c_void_p, c_size_t, c_int, c_int, c_void_p,
                                    ctypes.POINTER(IoStats)]),
    "rarc_device_to_file": (c_int, [ctypes.c_char_p, c_int, ctypes.POINTER(c_int64), ctypes.POINTER(c_int64),
                                    ctypes.POINTER(c_int64), c_void_p, c_int64, c_void_p, c_size_t, c_int, c_int, c_void_p,
                                    ctypes.POINTER(IoStats)]),
    "rarc_wide_workspace_bytes": (c_size_t, [c_int, c_int]),
    "rarc_search_wide": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_int, c_float, c_float, c_void_p, c_int, c_int, c_int64,
                                 c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_void_p]),
    "rarc_similar_pairs_workspace_bytes": (c_size_t, [c_int64, c_int, c_int]),
    "rarc_similar_pairs": (c_int, [c_void_p, c_int64, c_int64, c_int, ctypes.c_double, c_void_p, c_size_t, c_int, c_void_p, c_void_p,
                                   c_int64, c_void_p, c_void_p, c_void_p]),
    "rarc_compact_rows": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_size_t, c_void_p]),
    "rarc_vmem_create": (c_int, [c_int, c_size_t, c_size_t, c_size_t, ctypes.POINTER(c_void_p)]),
    "rarc_vmem_grow": (c_int, [c_void_p, c_size_t]),
    "rarc_vmem_base": (c_void_p, [c_void_p]),
    "rarc_vmem_mapped": (c_size_t, [c_void_p]),
    "rarc_vmem_reserved": (c_size_t, [c_void_p]),
    "rarc_vmem_slab": (c_size_t, [c_void_p]),
    "rarc_vmem_granularity": (c_size_t, [c_void_p]),
    "rarc_vmem_destroy": (c_int, [c_void_p]),
    "rarc_profile_begin": (c_int, [c_int]),
    "rarc_profile_end": (c_int, [ctypes.POINTER(c_double), ctypes.POINTER(c_int)]),
}


def library_path() -> str:
    return _LIB


def load_library() -> ctypes.CDLL:
    """Load librarc_hip.so once; raise RarcError if it is not built."""
    global _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(_LIB):
                raise RarcError(
                    f"{_LIB} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                    "(there is no CPU fallback)")
            lib = ctypes.CDLL(_LIB)
            lib.rarc_version.restype = c_int
            if lib.rarc_version() >= 100000 and os.environ.get("RARC_ALLOW_EXPERIMENT") != "1":
                raise RarcError(f"{_LIB} is a MEASUREMENT build (-DRARC_EXPERIMENT: it may contain kernels that return wrong "
                                "results on purpose); set RARC_ALLOW_EXPERIMENT=1 to load it from a bench tool")
            for name, (res, args) in SIGNATURES.items():
                fn = getattr(lib, name)
                fn.restype = res
                fn.argtypes = args
            _lib = lib
    return _lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load_library().rarc_last_error()
        raise RarcError(f"{what or 'librarc_hip'} failed ({rc}): {msg.decode() if msg else '?'}")


def padded_dim(d: int, align: int = DIM_ALIGN) -> int:
    return ((d + align - 1) // align) * align
