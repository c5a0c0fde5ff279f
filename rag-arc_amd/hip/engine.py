"""HBM-resident flat inner-product index (fp16 rows) driven through librarc_hip.so.

This is the device-side replacement for the `faiss.IndexFlatIP` object that
`FaissVectorStore` holds (encapsulation/database/vector_db/VectorStore_Faiss.py:112-136, :199-202,
:262-263): `add` == normalise + index.add, `search` == normalise + index.search.  torch is used for
device memory and streams only; all arithmetic happens in the HIP kernels.
"""
from __future__ import annotations

import ctypes
import os
import threading
from typing import Optional, Tuple

import numpy as np

from . import binding as B

_ROW_ALIGN = 32  # the scan reads whole 32-row tiles
# pipelined contexts (FlatIndexF16._pipeline_context): hold a context's scan back until its partner's batch is complete
# (RarcSearchBatch.gate_event)?  Measured at config 2, 400 steps, same box (tools/r06/gate.sh, profiles/r06_c2_pipeline.txt): one
# context 0.449 ms per batch, two gated 0.442, two UNGATED 0.423 — a cross-stream event costs ~17 µs from its completion to
# the waiting kernel's start, where the scan's workgroups simply take each CU as the partner's finalize leaves it.  Off.
_PIPELINE_GATE = os.environ.get("RARC_PIPELINE_GATE", "0") == "1"
_IO_RING: dict = {}            # the process's pinned staging ring for shard files (FlatIndexF16._io_staging)
_IO_LOCK = threading.Lock()    # one shard-file transfer at a time per process: they share the ring


class _PinSlot:
    """One pinned (ids int64, scores fp32) staging pair on loan from a _PinnedPool.  Whoever holds the slot owns the memory:
    it goes back to the pool when release() is called or when the last reference to the slot dies — never while a search
    handle that was told to copy its answer into it is alive."""

    __slots__ = ("pool", "ids", "scores", "__weakref__")

    def __init__(self, pool, ids, scores):
        self.pool, self.ids, self.scores = pool, ids, scores

    def views(self, nq: int, k: int):
        need = int(nq) * int(k)
        return self.ids[:need].view(nq, k), self.scores[:need].view(nq, k)

    def release(self) -> None:
        pool, self.pool = self.pool, None
        if pool is not None:
            pool._give_back(self.ids, self.scores)

    def __del__(self):
        try:
            self.release()
        except Exception:       # interpreter shutdown
            pass


class _PinnedPool:
    """Pinned host staging for answers.  Allocating pinned memory per search is what NOT to do (hipHostMalloc takes
    milliseconds and waits for the device — measured as the eight scans of a 2048-query call running one after the other),
    and a ring that hands a slot out again after N more searches is what not to do either (a 2048-query call's
    chunk 7 landed on chunk 1's slot before chunk 1 had been read).  So: a free list.  A slot is taken for a search and
    comes back when its handle lets go of it; the pool grows to the number of answers the callers really keep alive at
    once (two or three in the pipelined paths) and allocates nothing after that."""

    def __init__(self):
        self._lock = threading.Lock()
        self._free: list = []
        self.allocated = 0          # slots ever created (diagnostics / tests)

    def acquire(self, torch, nq: int, k: int) -> _PinSlot:
        need = int(nq) * int(k)
        with self._lock:
            for i in range(len(self._free) - 1, -1, -1):
                if self._free[i][0].numel() >= need:
                    ids, scores = self._free.pop(i)
                    return _PinSlot(self, ids, scores)
            self.allocated += 1
        cap = max(need, B.MAX_QUERIES * 128)
        return _PinSlot(self, torch.empty(cap, dtype=torch.int64, pin_memory=True),
                        torch.empty(cap, dtype=torch.float32, pin_memory=True))

    def _give_back(self, ids, scores) -> None:
        with self._lock:
            if len(self._free) < 16:        # (a burst's extra slots are dropped again: torch's host allocator keeps them)
                self._free.append((ids, scores))


class _FlagPool:
    """Pinned int32 status words, one per launch in flight: taken for a launch, given back when its handle has read it (or
    dies).  A free list over blocks of 64 — never a ring: a call of 70 x 256 queries has 70 launches in flight before the
    first result() and a ring of 64 would hand the first launch's word to the 65th."""

    def __init__(self):
        self._lock = threading.Lock()
        self._free: list = []

    def acquire(self, torch):
        with self._lock:
            if not self._free:
                block = torch.zeros(64, dtype=torch.int32, pin_memory=True)
                self._free.extend(block[i: i + 1] for i in range(64))
            return self._free.pop()

    def give_back(self, word) -> None:
        with self._lock:
            self._free.append(word)


def _torch():
    import torch

    if not torch.cuda.is_available():
        raise B.RarcError("no ROCm device visible: the HIP backend has no CPU fallback")
    return torch


class _DeviceArray:
    """What torch.as_tensor reads a raw device pointer from (`__cuda_array_interface__`); torch keeps this object — and
    through it the arena — alive for as long as the tensor's storage lives."""

    def __init__(self, arena, nbytes: int):
        self.arena = arena
        self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(arena.base), False),
                                         "version": 2, "strides": None}


class DeviceArena:
    """A growable HBM buffer on HIP virtual memory (rarc_vmem_*, csrc/vmem.hip): address space reserved up front, backed
    slab by slab as it grows, never moved, never copied.  `view(nbytes)` is a uint8 torch tensor over the first nbytes."""

    def __init__(self, torch, lib, device_index: int, reserve_bytes: int, min_reserve_bytes: int = 0):
        """reserve_bytes of address space; min_reserve_bytes (0 = the same): what the caller can live with when the
        process's space has no range that large left (`reserved` says what the arena got)."""
        import ctypes

        self.torch, self.lib, self.device_index = torch, lib, int(device_index)
        handle = ctypes.c_void_p()
        args = (self.device_index, int(reserve_bytes), int(min_reserve_bytes), 0)   # slab 0 = the library's (one per process)
        with torch.cuda.device(self.device_index):
            rc = lib.rarc_vmem_create(*args, ctypes.byref(handle))
            if rc == -3:        # the process's address space is held by arenas nobody uses any more: collect them, once
                import gc

                gc.collect()
                rc = lib.rarc_vmem_create(*args, ctypes.byref(handle))
            B.check(rc, "rarc_vmem_create")
        self.handle = handle
        self.base = int(lib.rarc_vmem_base(handle))
        self.reserved = int(lib.rarc_vmem_reserved(handle))
        self.granularity = int(lib.rarc_vmem_granularity(handle))
        self.slab = int(lib.rarc_vmem_slab(handle))

    @property
    def mapped(self) -> int:
        return int(self.lib.rarc_vmem_mapped(self.handle))

    def grow(self, nbytes: int) -> None:
        """Back at least nbytes (whole slabs of 16 MiB: what an arena holds beyond its live bytes is under one slab)."""
        nbytes = int(nbytes)
        if nbytes <= self.mapped:
            return
        if nbytes > self.reserved:
            raise B.RarcError(f"the index was created for at most {self.reserved} bytes of rows; {nbytes} asked "
                              "(give a larger max_rows)")
        with self.torch.cuda.device(self.device_index):
            B.check(self.lib.rarc_vmem_grow(self.handle, nbytes), "rarc_vmem_grow")

    def view(self, nbytes: int):
        t = self.torch
        return t.as_tensor(_DeviceArray(self, nbytes), device=t.device("cuda", self.device_index))

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self.torch.cuda.synchronize(self.device_index)      # no kernel may still be reading what gets unmapped
                self.lib.rarc_vmem_destroy(self.handle)
                self.handle = None
        except Exception:  # noqa: BLE001 - interpreter shutdown: the driver reclaims the memory with the process
            pass


class FlatIndexF16:
    """Exact top-k inner-product search over fp16 (or, with storage="f8", fp8 e4m3fn + per-row scale)
    rows resident in HBM.

    metric "cosine": rows and queries are L2-normalised in fp32 before use (faiss.normalize_L2);
    metric "ip": raw inner product.  Returned scores are the canonical fp32 inner products of the
    fp32 query with the stored fp16 rows; ties are ordered by id ascending.
    """

    # rows live in a virtual-memory arena that grows in place (DeviceArena) unless told otherwise: measured on the part,
    # an arena streams like a plain allocation (124M x 768 rows grown by add(): 5.77 TB/s end to end, peak 0.34 GiB over
    # the live rows — tests/test_gpu_growable.py)
    GROWABLE_DEFAULT = True

    def __init__(self, dim: int, metric: str = "cosine", device: int = 0, capacity: int = 0,
                 id_base: int = 0, cand_cap: int = 131072, scan: str = "auto", storage: str = "f16",
                 shadow: bool = False, growable: Optional[bool] = None, max_rows: int = 0):
        if metric not in ("cosine", "ip"):
            raise ValueError(f"unsupported metric: {metric}")
        if dim <= 0:
            raise ValueError("dim must be positive")
        if storage not in ("f16", "f8", "f32"):
            raise ValueError(f"unknown storage format: {storage}")
        self.torch = _torch()
        self.lib = B.load_library()
        self.dim = int(dim)
        # "f16": fp16 rows; "f8": e4m3fn bytes + one fp32 scale per row (BASELINE config 5); "f32": the reference's
        # own storage (fp32 rows, VectorStore_Faiss.py:170) — scores carry no storage rounding; an fp16 image of
        # the rows feeds the scan (6 bytes per element in HBM)
        self.storage = storage
        self.d_pad = B.padded_dim(self.dim, 256 if storage == "f8" else B.DIM_ALIGN)
        if scan not in ("auto", "q8", "mfma16"):
            raise ValueError(f"unknown scan mode: {scan}")
        if storage in ("f8", "f32") and scan == "mfma16":
            raise ValueError("fp8 / fp32 rows are scanned by the int8-prefilter kernel only")
        limit = 768 if scan == "mfma16" else 1024
        # rows beyond 1024 padded dimensions (up to 4096: the reference's OpenAI embeddings are 1536- / 3072-d) take the
        # WIDE path — score GEMM in chunks + select + canonical finalize (csrc/wide.hip) — as does k beyond 1024
        self.wide = self.d_pad > 1024
        if self.wide and (scan != "auto" or storage == "f8" or shadow or self.d_pad > B.WIDE_MAX_DPAD):
            raise B.RarcUnsupported(f"dim {dim} pads to {self.d_pad}: rows wider than 1024 take the wide path (storage 'f16' or "
                                    f"'f32', scan='auto', no shadow image, at most {B.WIDE_MAX_DPAD} padded dimensions)")
        if not self.wide and self.d_pad > limit:
            raise B.RarcUnsupported(f"dim {dim} pads to {self.d_pad} > {limit}: not supported by the {scan} scan kernel")
        # "q8": int8-prefilter scan (HBM-bound; its error margin costs candidates, which only matters on
        # small shards); "mfma16": fp16 MFMA scan + certificate (tight margin, matrix-pipe bound);
        # "auto": q8 from AUTO_Q8_ROWS rows up (and always beyond 768 dims), mfma16 below
        self.scan = scan
        self.metric = metric
        self.device = self.torch.device("cuda", device)
        self.id_base = int(id_base)
        self.cand_cap = int(cand_cap)
        self.ntotal = 0
        self.max_norm = 0.0
        # shadow=True keeps the int8 image of the fp16 rows next to them (+50 % HBM): the int8-prefilter scan
        # then reads the image (half the bytes per search, no conversion); results are unchanged
        self.shadow = bool(shadow)
        if self.shadow and (storage != "f16" or self.d_pad % 256 or scan == "mfma16"):
            raise ValueError("shadow=True needs fp16 storage, a dimension that pads to a multiple of 256, and the q8 scan")
        self._shadow = None  # torch.int8 [capacity][d_pad]
        self._rows = None  # torch.float16 (or uint8 for fp8 storage) [capacity][d_pad]
        self._rowscale = None  # fp8 storage: torch.float32 [capacity]
        self._image16 = None   # fp32 storage: torch.float16 [capacity][d_pad], what the scan kernels read
        self._qmeta = None  # torch.float32 [4 + 2*capacity/32]: quantisation metadata (include/rarc.h)
        self._lock = threading.Lock()  # callers may be pool threads (core/retrieval/base.py:92-96)
        self._ws = None
        self._cap_eff = 0          # candidate capacity the workspace was allocated for (grows with k)
        self._qbuf = None
        self._version = 0          # bumped by every change of the rows (twin() contexts check it)
        self._pair = None          # search_async's two pipelined contexts (_pipeline_context)
        self._partner = None       # a pipelined context: the other one of the pair
        self._fin_event = None     # ... and the event behind its last batch (the partner's next scan waits for it)
        self._parent = None        # twin(): the index whose rows this search context reads
        self._pins = _PinnedPool()  # pinned staging for answers (shared with the index's twins: copy.copy keeps the object)
        self._flags = _FlagPool()   # pinned status words, one per launch in flight
        self._own_stream = None    # twin(): the side stream its searches are enqueued on
        # growable=True: the row buffers live in DeviceArenas (HIP virtual memory) — add() past the capacity maps more
        # memory behind the same pointer instead of allocating a bigger buffer and copying (peak = live rows + one step
        # of at most 1 GiB; the reallocating form peaks at 3x).  max_rows: the address space to reserve (0 = what the
        # device's whole memory could hold); capacity: rows to back right away.
        self.growable = self.GROWABLE_DEFAULT if growable is None else bool(growable)
        self.max_rows = int(max_rows)
        self._arenas: dict = {}
        if capacity:
            self.reserve(capacity)

    # ------------------------------------------------------------------ memory
    def reserve(self, n_rows: int) -> None:
        t = self.torch
        cap = ((int(n_rows) + _ROW_ALIGN - 1) // _ROW_ALIGN) * _ROW_ALIGN
        if self._rows is not None and self._rows.shape[0] >= cap:
            return
        if self.growable:
            if self.max_rows and cap > ((self.max_rows + _ROW_ALIGN - 1) // _ROW_ALIGN) * _ROW_ALIGN:
                raise B.RarcError(f"the index was created for at most {self.max_rows} rows; {int(n_rows)} asked (give a larger max_rows)")
            self._reserve_arenas(cap)
            self._fit_qmeta()
            return
        new = t.zeros((cap, self.d_pad), dtype=self._row_dtype(), device=self.device)
        if self._rows is not None and self.ntotal:
            new[: self.ntotal].copy_(self._rows[: self.ntotal])
        self._rows = new
        if self.shadow:
            ns8 = t.zeros((cap, self.d_pad), dtype=t.int8, device=self.device)
            if self._shadow is not None and self.ntotal:
                ns8[: self.ntotal].copy_(self._shadow[: self.ntotal])
            self._shadow = ns8
        if self.storage == "f8":
            ns = t.ones(cap, dtype=t.float32, device=self.device)
            if self._rowscale is not None and self.ntotal:
                ns[: self.ntotal].copy_(self._rowscale[: self.ntotal])
            self._rowscale = ns
        if self.storage == "f32":
            ni = t.zeros((cap, self.d_pad), dtype=t.float16, device=self.device)
            if self._image16 is not None and self.ntotal:
                ni[: self.ntotal].copy_(self._image16[: self.ntotal])
            self._image16 = ni
        self._fit_qmeta()

    def _arena_view(self, name: str, row_bytes: int, cap: int, dtype, cols: Optional[int]):
        """Grow arena `name` to cap rows of row_bytes each and return (tensor [rows][cols] or [rows], rows backed).  A buffer
        that did not come from the arena (rows adopted by add_rows_f16) is copied into it once."""
        t = self.torch
        arena = self._arenas.get(name)
        if arena is not None and cap * row_bytes > arena.reserved and not self.max_rows:
            arena = None        # it outgrew a range it had to settle for (see `least` below): a new arena, rows copied once
        if arena is None:
            max_rows = self.max_rows or int(t.cuda.get_device_properties(self.device).total_memory // max(row_bytes, 1))
            max_rows = max(((max_rows + _ROW_ALIGN - 1) // _ROW_ALIGN) * _ROW_ALIGN, cap)
            # an explicit max_rows is a promise; "as large as the device" (max_rows = 0) settles for what the space has
            least = max_rows if self.max_rows else max(cap, min(max_rows, 1 << 20))
            arena = self._arenas[name] = DeviceArena(t, self.lib, self.device.index or 0, max_rows * row_bytes,
                                                     least * row_bytes)
        old_mapped = arena.mapped
        arena.grow(cap * row_bytes)
        rows = (arena.mapped // (row_bytes * _ROW_ALIGN)) * _ROW_ALIGN          # whole 32-row tiles of what is backed
        if arena.mapped > old_mapped:
            # fresh memory reads as zeros (padding rows and columns rely on it) — zeroed over the MAPPED byte range, not over
            # the row view: a slab's ragged end (16 MiB is not a whole number of 32-row tiles at d = 768) belongs to rows that
            # only become whole once the next slab is mapped, and that call starts zeroing at old_mapped
            arena.view(arena.mapped)[old_mapped:].zero_()
        flat = arena.view(rows * row_bytes)
        out = flat.view(dtype)
        return (out.view(rows, cols) if cols else out), rows

    def _reserve_arenas(self, cap: int) -> None:
        t = self.torch
        esz = {"f8": 1, "f16": 2, "f32": 4}[self.storage]

        def adopt(name, row_bytes, dtype, cols, current):
            new, _ = self._arena_view(name, row_bytes, cap, dtype, cols)
            if current is not None and current.data_ptr() != new.data_ptr() and self.ntotal:
                new[: self.ntotal].copy_(current[: self.ntotal])
            return new

        self._rows = adopt("rows", self.d_pad * esz, self._row_dtype(), self.d_pad, self._rows)
        if self.shadow:
            self._shadow = adopt("shadow", self.d_pad, t.int8, self.d_pad, self._shadow)
        if self.storage == "f8":
            had = 0 if self._rowscale is None else max(int(self._rowscale.shape[0]), self.ntotal)
            from_arena = "rowscale" in self._arenas
            self._rowscale = adopt("rowscale", 4, t.float32, None, self._rowscale)
            self._rowscale[(had if from_arena else self.ntotal):].fill_(1.0)     # rows not yet written: scale 1, as t.ones() gave
        if self.storage == "f32":
            self._image16 = adopt("image16", self.d_pad * 2, t.float16, self.d_pad, self._image16)
        # every buffer covers the same rows: the smallest backing decides
        n = min(x.shape[0] for x in (self._rows, self._shadow, self._rowscale, self._image16) if x is not None)
        self._rows = self._rows[:n]
        self._shadow = None if self._shadow is None else self._shadow[:n]
        self._rowscale = None if self._rowscale is None else self._rowscale[:n]
        self._image16 = None if self._image16 is None else self._image16[:n]

    def memory_bytes(self) -> dict:
        """HBM held by the row buffers: {'live': bytes of the stored rows, 'backed': bytes physically mapped / allocated}."""
        esz = {"f8": 1, "f16": 2, "f32": 4}[self.storage]
        per_row = self.d_pad * esz + (self.d_pad if self.shadow else 0) + (4 if self.storage == "f8" else 0) \
            + (2 * self.d_pad if self.storage == "f32" else 0)
        if self._arenas:
            backed = sum(a.mapped for a in self._arenas.values())
        else:
            backed = 0 if self._rows is None else int(self._rows.shape[0]) * per_row
        return {"live": int(self.ntotal) * per_row, "backed": int(backed)}

    def _row_dtype(self):
        t = self.torch
        return {"f8": t.uint8, "f16": t.float16, "f32": t.float32}[self.storage]

    def _fit_qmeta(self) -> None:
        """Size the quantisation metadata for the current row buffer (keeps what is already there)."""
        t = self.torch
        if self.wide:
            return
        fn = self.lib.rarc_quant_meta_floats_f8 if self.storage == "f8" else self.lib.rarc_quant_meta_floats
        need = int(fn(self._rows.shape[0]))
        if self._qmeta is None or self._qmeta.numel() < need:
            new = t.zeros(need, dtype=t.float32, device=self.device)
            if self._qmeta is not None:
                new[: self._qmeta.numel()].copy_(self._qmeta)
            self._qmeta = new

    def _requant(self, first_row: int) -> None:
        """(Re)compute tile scales + residual bound for the tiles touched by rows [first_row, ntotal)."""
        if self.wide:
            return            # no int8 prefilter over wide rows: nothing to keep up to date
        self._fit_qmeta()
        if self.storage == "f8":
            B.check(self.lib.rarc_quant_meta_f8(self._rows.data_ptr(), self._rowscale.data_ptr(), self.ntotal, self.d_pad,
                                                int(first_row), self._qmeta.data_ptr(), self._stream()),
                    "rarc_quant_meta_f8")
            return
        if self.storage == "f32":
            B.check(self.lib.rarc_quant_meta_f16(self._image16.data_ptr(), self.ntotal, self.d_pad, int(first_row),
                                                 self._qmeta.data_ptr(), self._stream()), "rarc_quant_meta_f16")
            # qmeta[1] = rho >= max ||d32 - d16||: 2^-11 relative per normal half, 2^-25 absolute per subnormal one
            self._qmeta[1] = float(self.max_norm) * 2.0 ** -11 * 1.001 + 2.0 ** -25 * float(self.d_pad) ** 0.5
            return
        if self.shadow:
            t = self.torch
            if self._shadow is None or self._shadow.shape[0] != self._rows.shape[0]:
                self._shadow = t.zeros((self._rows.shape[0], self.d_pad), dtype=t.int8, device=self.device)
                first_row = 0  # a fresh image: every tile has to be written
            B.check(self.lib.rarc_quant_shadow_f16(self._rows.data_ptr(), self.ntotal, self.d_pad, int(first_row),
                                                   self._qmeta.data_ptr(), self._shadow.data_ptr(), self._stream()),
                    "rarc_quant_shadow_f16")
            return
        B.check(self.lib.rarc_quant_meta_f16(self._rows.data_ptr(), self.ntotal, self.d_pad, int(first_row),
                                             self._qmeta.data_ptr(), self._stream()), "rarc_quant_meta_f16")

    def _stream(self):
        return self.torch.cuda.current_stream(self.device).cuda_stream

    def _workspace(self, k: int = 0, scale: int = 1):
        """Scratch for one search.  The candidate buffer grows with k (the int8 margin lets through a number of
        candidates roughly proportional to k): `cand_cap` is per 128 results — k = 996 takes 8 x 268 MB.
        (The re-run of flagged queries takes a larger workspace of its own, see _repair_rows.)"""
        t = self.torch
        cap = self.cand_cap * max(1, -(-int(k) // 128)) * max(1, int(scale))
        if self._ws is None or cap > self._cap_eff:
            want = max(cap, self._cap_eff)
            try:
                ws = t.empty(self.lib.rarc_search_workspace_bytes(want), dtype=t.uint8, device=self.device)
            except (t.cuda.OutOfMemoryError, RuntimeError) as exc:
                # HBM cannot hold the capacity this index has GROWN to (sticky growth below): fall back to the capacity it
                # had before — such a corpus is then answered through the re-run / exact per-query scan again (slowly), but
                # it is answered.  Anything else (the default capacity does not fit either) is the caller's to see.
                prev = getattr(self, "_cand_cap_before_growth", None)
                if "out of memory" not in str(exc).lower() or prev is None or prev >= self.cand_cap:
                    raise
                self.cand_cap, self._cand_cap_before_growth = prev, None
                self.cand_cap_growth_refused = True
                return self._workspace(k, scale)
            self._ws, self._cap_eff = ws, want
        self._ensure_qbuf()
        return self._ws

    def _ensure_qbuf(self) -> None:
        t = self.torch
        if self._qbuf is None:
            mq = B.MAX_QUERIES
            self._qbuf = dict(
                qblock=t.empty(int(self.lib.rarc_query_block_bytes(self.d_pad)), dtype=t.uint8, device=self.device),
                status=t.zeros(mq + 1, dtype=t.int32, device=self.device),
                found=t.empty(1, dtype=t.int32, device=self.device),
            )

    # "auto": measured crossover of the two scans (768 dims, ms per 256-query batch, int8 vs fp16 MFMA):
    #   1M rows: k=10 0.44 / 0.50, k=50 0.51 / 0.52, k=100 0.56 / 0.53, k=400 0.82 / 0.66;  300K rows: k=10 0.24 / 0.25,
    #   k=100 0.31 / 0.26;  2M rows, k=100: 0.86 / 0.90 — the int8 scan's candidate count grows with k, the fp16 scan's
    #   cost does not, so the switch-over shard size does: 1.5M rows x (k/100)^0.7
    AUTO_Q8_ROWS = 1_500_000

    def _use_q8(self, k: int = 100) -> bool:
        if self.storage in ("f8", "f32") or self.shadow:
            return True
        if self.scan == "auto":
            if self.d_pad > 768 or k > 900:   # the fp16 scan stops at 768 dims; its k' <= 1024 certificate gives out near k = 1000
                return True
            return self.ntotal >= self.AUTO_Q8_ROWS * (max(int(k), 1) / 100.0) ** 0.7
        return self.scan == "q8"

    def _prep(self, q, k: int = 100, force_q8: bool = False) -> None:
        """rarc_prep_queries into the shared query block (caller holds the lock)."""
        norm = 1 if self.metric == "cosine" else 0
        qm = self._qmeta.data_ptr() if ((force_q8 or self._use_q8(k)) and self._qmeta is not None) else 0
        B.check(self.lib.rarc_prep_queries(q.data_ptr(), q.shape[1], q.shape[0], self.dim, self.d_pad, norm,
                                           max(self.max_norm, 1.0) if norm else self.max_norm, qm,
                                           self._qbuf["qblock"].data_ptr(), self._stream()), "rarc_prep_queries")

    @property
    def rows(self):
        """The stored rows as a torch view [ntotal][d_pad] (fp16, or e4m3fn bytes for fp8 storage)."""
        return None if self._rows is None else self._rows[: self.ntotal]

    @property
    def row_scales(self):
        """fp8 storage: the per-row scales [ntotal] (value = scale * decode(byte))."""
        return None if self._rowscale is None else self._rowscale[: self.ntotal]

    # ------------------------------------------------------------------ add
    def add(self, vectors) -> None:
        """index.add: fp32 vectors [n][dim] (numpy or torch) -> normalise (cosine) -> fp16 rows."""
        self._not_a_twin()
        t = self.torch
        with self._lock, t.cuda.device(self.device):
            x = t.as_tensor(vectors, dtype=t.float32).to(self.device).contiguous()
            if x.ndim != 2 or x.shape[1] != self.dim:
                raise ValueError(f"expected [n][{self.dim}] vectors, got {tuple(x.shape)}")
            n = x.shape[0]
            if n == 0:
                return
            if self._rows is None or self.ntotal + n > self._rows.shape[0]:
                # an arena grows in place, slab by slab: ask for what is needed; a plain buffer is reallocated and copied:
                # double it so that a stream of adds copies O(log) times
                self.reserve(self.ntotal + n if self.growable else max(self.ntotal + n, 2 * self.ntotal))
            norm2 = t.empty(n, dtype=t.float32, device=self.device)
            dst = self._rows[self.ntotal: self.ntotal + n]
            if self.storage == "f8":
                B.check(self.lib.rarc_ingest_f8(x.data_ptr(), x.shape[1], dst.data_ptr(), self.d_pad,
                                                self._rowscale[self.ntotal: self.ntotal + n].data_ptr(),
                                                norm2.data_ptr(), n, self.dim,
                                                1 if self.metric == "cosine" else 0, self._stream()),
                        "rarc_ingest_f8")
            elif self.storage == "f32":
                B.check(self.lib.rarc_ingest_f32(x.data_ptr(), x.shape[1], dst.data_ptr(),
                                                 self._image16[self.ntotal: self.ntotal + n].data_ptr(), self.d_pad,
                                                 norm2.data_ptr(), n, self.dim,
                                                 1 if self.metric == "cosine" else 0, self._stream()),
                        "rarc_ingest_f32")
            else:
                B.check(self.lib.rarc_ingest_f16(x.data_ptr(), x.shape[1], dst.data_ptr(), self.d_pad,
                                                 norm2.data_ptr(), n, self.dim,
                                                 1 if self.metric == "cosine" else 0, self._stream()),
                        "rarc_ingest_f16")
            self.max_norm = max(self.max_norm, float(norm2.max().sqrt().item()))
            old = self.ntotal
            self.ntotal += n
            self._version += 1
            self._requant(old)

    def add_rows_f16(self, rows_f16, max_norm: float, n_valid: Optional[int] = None) -> None:
        """Adopt rows that are already in storage format ([n][d_pad] fp16 on this device).  A buffer
        whose length is a multiple of 32 is adopted without a copy when the index is empty;
        `n_valid` (default: all) says how many of its rows are real."""
        self._not_a_twin()
        t = self.torch
        if self.storage != "f16":
            raise B.RarcError("add_rows_f16 needs fp16 storage")
        with self._lock, t.cuda.device(self.device):
            if rows_f16.dtype != t.float16 or rows_f16.shape[1] != self.d_pad:
                raise ValueError("rows must be float16 [n][d_pad]")
            n = rows_f16.shape[0] if n_valid is None else int(n_valid)
            if not (0 <= n <= rows_f16.shape[0]):
                raise ValueError("n_valid out of range")
            old = self.ntotal
            if self.ntotal == 0 and rows_f16.shape[0] % _ROW_ALIGN == 0 and rows_f16.is_contiguous():
                self._rows = rows_f16
                self._qmeta = None
            else:
                self.reserve(self.ntotal + n)
                self._rows[self.ntotal: self.ntotal + n].copy_(rows_f16[:n])
            self.ntotal += n
            self._version += 1
            self.max_norm = max(self.max_norm, float(max_norm))
            self._requant(old)

    def load_rows(self, rows_host, max_norm: float, row_scales=None) -> None:
        """Upload rows already in storage format: host array [n][d_pad] float16 (e.g. a memmap), or for
        fp8 storage uint8 bytes plus their fp32 `row_scales` [n]."""
        self._not_a_twin()
        t = self.torch
        if (self.storage == "f8") != (row_scales is not None):
            raise ValueError("row_scales go with fp8 storage (and only with it)")
        with self._lock, t.cuda.device(self.device):
            n = rows_host.shape[0]
            if rows_host.shape[1] != self.d_pad:
                raise ValueError("rows must be [n][d_pad]")
            self.reserve(self.ntotal + n)
            step = 1 << 20
            for s in range(0, n, step):  # bounded pinned staging instead of one huge host tensor
                chunk = t.from_numpy(np.array(rows_host[s:s + step], copy=True))
                self._rows[self.ntotal + s: self.ntotal + s + chunk.shape[0]].copy_(chunk)
            if row_scales is not None:
                self._rowscale[self.ntotal: self.ntotal + n].copy_(t.from_numpy(np.array(row_scales, dtype=np.float32, copy=True)))
            if self.storage == "f32":   # the scan's image: the rows rounded to fp16 (round to nearest even)
                self._image16[self.ntotal: self.ntotal + n].copy_(self._rows[self.ntotal: self.ntotal + n])
            old = self.ntotal
            self.ntotal += n
            self._version += 1
            self.max_norm = max(self.max_norm, float(max_norm))
            self._requant(old)

    # ------------------------------------------------------------------ shard files
    IO_THREADS = 8
    IO_STAGING_BYTES = 256 << 20   # pinned ring: 2 x IO_THREADS slots of 16 MiB; the host footprint of a save / load

    def _io_staging(self, nbytes: Optional[int] = None):
        """The pinned host ring for rarc_file_to_device / rarc_device_to_file: ONE per process (shared by every index;
        _IO_LOCK serialises its users), sized exactly — torch's pinned allocator rounds requests up to a power of two."""
        t = self.torch
        nbytes = int(nbytes or self.IO_STAGING_BYTES)
        buf = _IO_RING.get("buf")
        if buf is None or buf.numel() < nbytes:
            buf = t.empty(nbytes, dtype=t.uint8, pin_memory=True)
            if buf.data_ptr() % 4096:       # (hipHostMalloc returns page-aligned memory; kept for other allocators)
                buf = t.empty(2 * nbytes, dtype=t.uint8, pin_memory=True)
                buf = buf[(-buf.data_ptr()) % 4096:]
            _IO_RING["buf"] = buf
        return buf[:nbytes]

    def _io_call(self, fn_name: str, path: str, segs, tensor, threads, direct, fsync=False):
        """One native transfer: segs = [(file offset, bytes, device offset in bytes from the tensor's start)]."""
        import ctypes

        if not segs:
            return None
        n = len(segs)
        arr = lambda j: (ctypes.c_int64 * n)(*[int(sg[j]) for sg in segs])   # noqa: E731
        stats = B.IoStats()
        flags = (B.IO_DIRECT if direct else 0) | (B.IO_FSYNC if fsync else 0)
        fn = getattr(self.lib, fn_name)
        with _IO_LOCK:
            staging = self._io_staging()
            B.check(fn(os.fsencode(path), n, arr(0), arr(1), arr(2), tensor.data_ptr(), tensor.numel() * tensor.element_size(),
                       staging.data_ptr(), staging.numel(), int(threads or self.IO_THREADS), flags, self._stream(),
                       ctypes.byref(stats)), fn_name)
        return dict(bytes=int(stats.bytes), seconds=float(stats.seconds), file_seconds=float(stats.file_seconds),
                    copy_wait_seconds=float(stats.copy_wait_seconds), direct_bytes=int(stats.direct_bytes),
                    n_chunks=int(stats.n_chunks), slot_bytes=int(stats.slot_bytes), n_threads=int(stats.n_threads),
                    direct=int(stats.direct), gb_per_s=(stats.bytes / stats.seconds / 1e9 if stats.seconds > 0 else 0.0))

    def save_shard(self, path: str, blocks=None, rank: int = 0, world: int = 1, global_ntotal: Optional[int] = None,
                   threads: Optional[int] = None, direct: bool = True, fsync: bool = False) -> dict:
        """Write this index's stored rows as a `.rarc` shard file (hip/shardfile.py), streaming HBM -> pinned ring -> file:
        no host copy of the shard exists at any point.  Written aside and renamed, so a crash never leaves a header
        without its rows.  `blocks`: the id map [(first global id, rows)] of the local rows (default: one block at id_base).
        Returns the transfer statistics of the row section."""
        from . import shardfile as SF

        self._not_a_twin()
        t = self.torch
        with self._lock, t.cuda.device(self.device):
            n = int(self.ntotal)
            blocks = [(self.id_base, n)] if blocks is None else [(int(g), int(c)) for g, c in blocks]
            if sum(c for _, c in blocks) != n:
                raise ValueError("the id map must cover exactly the stored rows")
            hdr = SF.ShardHeader(n, self.dim, self.d_pad, SF.CODES[self.storage], float(self.max_norm), int(rank), int(world),
                                 int(n if global_ntotal is None else global_ntotal), blocks)
            tmp = path + ".tmp"
            with open(tmp, "wb") as fh:
                fh.write(hdr.pack())
                fh.truncate(hdr.file_bytes)        # sized once: the workers write into place
            stats = None
            if n:
                stats = self._io_call("rarc_device_to_file", tmp, [(hdr.rows_offset, n * hdr.row_bytes, 0)], self._rows,
                                      threads, direct, fsync)
                if self.storage == "f8":
                    self._io_call("rarc_device_to_file", tmp, [(hdr.scales_offset, 4 * n, 0)], self._rowscale, threads, direct,
                                  fsync)
            with open(tmp, "r+b") as fh:
                fh.seek(hdr.idmap_offset)
                fh.write(hdr.idmap_bytes())
                if fsync:
                    fh.flush()
                    os.fsync(fh.fileno())
            os.replace(tmp, path)
            return stats or dict(bytes=0, seconds=0.0, gb_per_s=0.0)

    def load_shard(self, path: str, row_ranges=None, header=None, threads: Optional[int] = None, direct: bool = False) -> dict:
        """Append rows of a `.rarc` shard file to this index, streaming file -> pinned ring -> HBM (two slots per worker:
        one being read into while the other's DMA runs).  `row_ranges`: [(first row in the file, count)] (default: all
        rows).  `direct` (O_DIRECT reads) is off by default: measured on the GPU box a cold file loads at the storage's
        20 GB/s either way, a file still in the page cache at 50 GB/s buffered but 20 GB/s direct
        (profiles/r04_persist_rates.txt).  Returns the transfer statistics of the row section."""
        from . import shardfile as SF

        self._not_a_twin()
        t = self.torch
        hdr = header or SF.read_header(path)
        if hdr.storage != self.storage:
            raise ValueError(f"{path}: stored as {hdr.storage}, index configured for {self.storage}")
        if hdr.d_pad != self.d_pad or hdr.dim != self.dim:
            raise ValueError(f"{path}: rows are [{hdr.dim} -> {hdr.d_pad}], index is [{self.dim} -> {self.d_pad}]")
        ranges = [(0, hdr.n_rows)] if row_ranges is None else [(int(a), int(c)) for a, c in row_ranges]
        for a, c in ranges:
            if a < 0 or c < 0 or a + c > hdr.n_rows:
                raise ValueError(f"{path}: rows [{a}, {a + c}) outside the file's {hdr.n_rows}")
        n_new = sum(c for _, c in ranges)
        with self._lock, t.cuda.device(self.device):
            old = self.ntotal
            if n_new == 0:
                return dict(bytes=0, seconds=0.0, gb_per_s=0.0)
            self.reserve(old + n_new)
            rb = hdr.row_bytes
            segs, ssegs, at = [], [], old
            for a, c in ranges:
                if c:
                    segs.append((hdr.rows_offset + a * rb, c * rb, at * rb))
                    ssegs.append((hdr.scales_offset + 4 * a, 4 * c, 4 * at))
                    at += c
            stats = self._io_call("rarc_file_to_device", path, segs, self._rows, threads, direct)
            if self.storage == "f8":
                self._io_call("rarc_file_to_device", path, ssegs, self._rowscale, threads, direct)
            if self.storage == "f32":   # the scan's image: the rows rounded to fp16 (round to nearest even)
                self._image16[old: old + n_new].copy_(self._rows[old: old + n_new])
            self.ntotal = old + n_new
            self._version += 1
            self.max_norm = max(self.max_norm, float(hdr.max_norm))
            self._requant(old)
            return stats

    COMPACT_TMP_BYTES = 256 << 20      # scratch the rows move through when rows are removed (rarc_compact_rows)

    def remove_rows(self, rows) -> int:
        """Delete stored rows (local row numbers, any order) by stable in-place compaction: the survivors keep their
        order and move down over the holes, so a later row's number drops by the holes before it.  No encoder call — the
        reference re-embeds every surviving text (VectorStore_Faiss.py:374-415).  Tile metadata is recomputed from the
        first hole on.  Returns the number of rows removed."""
        self._not_a_twin()
        t = self.torch
        holes = np.unique(np.asarray(rows, dtype=np.int64).reshape(-1))
        if holes.size == 0:
            return 0
        if holes[0] < 0 or holes[-1] >= self.ntotal:
            raise IndexError(f"rows to remove must lie in [0, {self.ntotal})")
        with self._lock, t.cuda.device(self.device):
            adj = t.from_numpy(holes - np.arange(holes.size, dtype=np.int64)).to(self.device)
            first, n = int(holes[0]), int(self.ntotal)
            bufs = [(self._rows, self._rows.shape[1] * self._rows.element_size())]
            if self._rowscale is not None:
                bufs.append((self._rowscale, 4))
            if self._image16 is not None:
                bufs.append((self._image16, self.d_pad * 2))
            if self._shadow is not None:
                bufs.append((self._shadow, self.d_pad))
            need = max((n - holes.size - first) * rb for _, rb in bufs)
            tmp = t.empty(max(min(self.COMPACT_TMP_BYTES, need), max(rb for _, rb in bufs)), dtype=t.uint8, device=self.device)
            for buf, rb in bufs:
                B.check(self.lib.rarc_compact_rows(buf.data_ptr(), rb, n, adj.data_ptr(), holes.size, first,
                                                   tmp.data_ptr(), tmp.numel(), self._stream()), "rarc_compact_rows")
            self.ntotal = n - int(holes.size)
            self._version += 1
            if self.ntotal == 0:
                self.max_norm = 0.0
                if self._qmeta is not None:
                    self._qmeta[:4].zero_()
            else:
                self._requant(min(first, self.ntotal - 1))
            t.cuda.current_stream(self.device).synchronize()      # `tmp` and `adj` are released on return
        return int(holes.size)

    def reset(self) -> None:
        self._not_a_twin()
        with self._lock:
            self._version += 1
            self.ntotal = 0
            self.max_norm = 0.0
            if self._qmeta is not None:
                self._qmeta[:4].zero_()

    # ------------------------------------------------------------------ search
    def twin(self) -> "FlatIndexF16":
        """A second SEARCH CONTEXT over the same rows: its own query block, workspace and lock, and a side stream its
        searches are enqueued on.  Alternating batches between an index and its twin keeps two searches in flight on
        two streams, so the small kernels either side of one batch's scan (query prep, seed, finalize) run under the
        other's (1M x 768: 0.480 -> 0.463 ms per batch).  Read-only: it refuses to search once the parent's rows have
        changed (take a new twin), and it cannot be added to."""
        import copy

        t = self.torch
        with self._lock:
            other = copy.copy(self)
            other._lock = threading.Lock()
            other._ws, other._qbuf, other._cap_eff = None, None, 0
            other._parent, other._parent_version = self, self._version
            other._pair, other._partner, other._fin_event = None, None, None
            with t.cuda.device(self.device):
                other._own_stream = t.cuda.Stream()
        return other

    # the fp16-scan path (small shards: config 2) spends a sixth of a batch in small dependent kernels either side of the scan
    # — query prep, seed pass, seed threshold in front of it (30 µs), finalize behind it (35 µs) — while the scan itself holds
    # every CU.  search_async (answers that stay on the device) therefore alternates between TWO search contexts of the index
    # (twin(): own query block, workspace and stream each): batch i+1's prep / seed run under batch i's finalize, and its scan's
    # workgroups take each CU as that finalize leaves it (ungated by default: _PIPELINE_GATE above).  RARC_PIPELINE=0 keeps
    # one context.
    PIPELINE_MIN_ROWS = 65536

    def _pipeline_context(self, k: int):
        if self._parent is not None or self._own_stream is not None or os.environ.get("RARC_PIPELINE", "1") == "0":
            return None
        if self.ntotal < self.PIPELINE_MIN_ROWS or (self._use_q8(k) and os.environ.get("RARC_PIPELINE_Q8", "0") != "1"):
            return None
        pair = self.__dict__.get("_pair")
        if pair is None or pair[0]._parent_version != self._version:
            import weakref

            a, b = self.twin(), self.twin()
            # (the pair belongs to this index and to nothing else: no reference cycle through it — contexts see their index and
            #  each other weakly — so an index dropped by its last reference frees its rows at once, not at the cyclic
            #  collector's next pass: the contexts share the row tensors)
            a._partner, b._partner = weakref.proxy(b), weakref.proxy(a)
            a._parent = b._parent = weakref.proxy(self)
            pair = self._pair = [a, b, 0]
        pair[2] ^= 1
        return pair[pair[2]]

    def _rows_version(self) -> int:
        """Version of the rows a search reads (a twin reads its parent's)."""
        if self._parent is None:
            return self._version
        try:
            return self._parent._version
        except ReferenceError:          # a pipelined context whose index is gone: the rows it holds are the ones it was made for
            return self._parent_version

    def _not_a_twin(self) -> None:
        if self._parent is not None:
            raise B.RarcError("a twin() search context is read-only: change the index it was taken from")

    def _check_twin(self) -> None:
        if self._parent is not None and self._rows_version() != self._parent_version:
            raise B.RarcError("the index has changed since twin() was taken: take a new twin")

    @staticmethod
    def kprime_for(k: int) -> int:
        return min(B.MAX_K, max(k + 28, (k * 5 + 3) // 4))

    def search(self, queries, k: int, repair: bool = True) -> Tuple[np.ndarray, np.ndarray]:
        """index.search: returns (scores fp32 [nq][k], ids int64 [nq][k]) like faiss (D, I);
        entries beyond ntotal are (-inf, -1)."""
        if k < 1:
            raise ValueError("k must be >= 1")
        if repair and self.ntotal and not self._takes_wide_path(k):
            # the answer lands in pinned memory straight from the finalize kernel (search_async(to_host=True)): no device
            # copy of it, no copy launches, one event to wait for — what a single embed_query + search call pays per query
            t = self.torch
            q = t.as_tensor(queries, dtype=t.float32)
            return self.search_async(q[None, :] if q.ndim == 1 else q, k, to_host=True).host()
        ids, scores = self.search_device(queries, k, repair=repair)
        return self.to_host(ids, scores)

    def to_host(self, ids, scores) -> Tuple[np.ndarray, np.ndarray]:
        """A device answer as numpy (scores fp32 [nq][k], ids int64 [nq][k]): both copies go to pinned memory behind the
        search on its stream and ONE event is waited for (two pageable `.cpu()` calls are two synchronous staged copies).
        The arrays own their pinned block (torch's caching host allocator recycles it when they die)."""
        t = self.torch
        with t.cuda.device(self.device):        # (callers may be pool threads: each takes a staging slot of its own)
            slot = self._pins.acquire(t, ids.shape[0], ids.shape[1])
            try:
                h_i, h_s = slot.views(ids.shape[0], ids.shape[1])
                h_i.copy_(ids, non_blocking=True)
                h_s.copy_(scores, non_blocking=True)
                done = t.cuda.Event()
                done.record()
                done.synchronize()
                return h_s.numpy().copy(), h_i.numpy().copy()  # (the caller owns its arrays: the slot goes back)
            finally:
                slot.release()

    def search_device(self, queries, k: int, repair: bool = True):
        """Same as search() but returns device tensors (ids int64, scores fp32)."""
        t = self.torch
        if k < 1:
            raise ValueError("k must be >= 1")
        wide = self._takes_wide_path(k)
        self._check_twin()
        with self._lock, t.cuda.device(self.device):
            q = t.as_tensor(queries, dtype=t.float32).to(self.device).contiguous()
            if q.ndim == 1:
                q = q[None, :]
            if q.shape[1] != self.dim:
                raise ValueError(f"expected [nq][{self.dim}] queries, got {tuple(q.shape)}")
            nq = q.shape[0]
            if self.ntotal == 0:        # an empty index answers (-1, -inf) like faiss; no kernel has anything to read
                return (t.full((nq, k), -1, dtype=t.int64, device=self.device),
                        t.full((nq, k), float("-inf"), dtype=t.float32, device=self.device))
            out_ids = t.empty((nq, k), dtype=t.int64, device=self.device)
            out_sc = t.empty((nq, k), dtype=t.float32, device=self.device)
            for s in range(0, nq, B.MAX_QUERIES):
                e = min(nq, s + B.MAX_QUERIES)
                if wide:
                    self._search_wide_chunk(q[s:e], k, out_ids[s:e], out_sc[s:e])
                else:
                    self._search_chunk(q[s:e], k, out_ids[s:e], out_sc[s:e], repair)
            return out_ids, out_sc

    # ------------------------------------------------------------------ wide rows / large k (csrc/wide.hip)
    def _takes_wide_path(self, k: int) -> bool:
        """Rows wider than 1024 padded dimensions, or k beyond the register-resident scans' 1024: the chunked-GEMM path."""
        if not (self.wide or k > B.MAX_K):
            return False
        if k > B.WIDE_MAX_K:
            raise B.RarcUnsupported(f"k={k} exceeds the limit of {B.WIDE_MAX_K} results per query (faiss has none: ask in pages of "
                                    f"{B.WIDE_MAX_K} if a corpus-sized k is really meant)")
        if self.storage == "f8" or self.shadow:
            raise B.RarcUnsupported(f"k={k} > {B.MAX_K} takes the wide path, which reads fp16 / fp32 rows (storage={self.storage}): "
                                    "store the rows as 'f16' or 'f32' for such k")
        return True

    def _rho(self) -> float:
        """fp32 storage: >= max ||row32 - image16|| (2^-11 relative per normal half, 2^-25 absolute per subnormal one)."""
        return float(self.max_norm) * 2.0 ** -11 * 1.001 + 2.0 ** -25 * float(self.d_pad) ** 0.5

    def _search_wide_chunk(self, q, k, out_ids, out_sc) -> None:
        """<= 256 queries through rarc_search_wide.  The candidate capacity starts at max(16640, 4k) entries per query; a
        query whose list filled up (rows within the error margin of its k-th best score: near-duplicates) is answered again
        with four times the capacity, up to the capacity that cannot overflow (one entry per stored row)."""
        t = self.torch
        nq = q.shape[0]
        self._ensure_qbuf()
        norm = 1 if self.metric == "cosine" else 0
        mn = max(self.max_norm, 1.0) if norm else self.max_norm
        stream = self._stream()
        # metric "ip" takes rows and queries of any scale, and the wide path keeps its first chunk's scores as fp16 (finite by
        # assumption: csrc/wide.hip).  |q·d| <= ||q||·max ||row||: where that bound passes 2^15 the QUERIES go in scaled by a
        # power of two that brings it under (exact in fp32, and every later step is linear in q) and the scores come back
        # multiplied by its inverse (exact) — an infinite first-chunk score would set an infinite threshold and drop every
        # later row.  Cosine scores are at most 1: no read-back on that path.
        unscale = 1.0
        if not norm:
            bound = float(q.norm(dim=1).max().item()) * float(self.max_norm) * 1.01
            if bound >= 32768.0:
                e = int(np.ceil(np.log2(bound / 32768.0))) + 1
                q, unscale = q * (2.0 ** -e), 2.0 ** e
        B.check(self.lib.rarc_prep_queries(q.data_ptr(), q.shape[1], nq, self.dim, self.d_pad, norm, mn, 0,
                                           self._qbuf["qblock"].data_ptr(), stream), "rarc_prep_queries")
        first = ((max(16384, 2 * k) + 255) // 256) * 256 + 256   # the library's first chunk of rows (csrc/wide.hip) at its largest
        cap = (max(4 * k, first) + 7) // 8 * 8           # (a query's list is eight sub-lists)
        sure = ((self.ntotal + 255) // 256) * 256 + first    # every stored row + the first chunk: cannot overflow
        status = t.zeros(B.MAX_QUERIES, dtype=t.int32, device=self.device)
        image = self._image16.data_ptr() if self.storage == "f32" else 0
        fmt = 2 if self.storage == "f32" else 0
        rho = self._rho() if self.storage == "f32" else 0.0
        while True:
            nbytes = int(self.lib.rarc_wide_workspace_bytes(self.d_pad, cap))
            if getattr(self, "_wide_ws", None) is None or self._wide_ws.numel() < nbytes:
                self._wide_ws = None
                self._wide_ws = t.empty(nbytes, dtype=t.uint8, device=self.device)
            B.check(self.lib.rarc_search_wide(self._rows.data_ptr(), image, fmt, self.ntotal, self.d_pad, mn, rho,
                                              self._qbuf["qblock"].data_ptr(), nq, k, self.id_base, out_ids.data_ptr(),
                                              out_sc.data_ptr(), status.data_ptr(), self._wide_ws.data_ptr(),
                                              self._wide_ws.numel(), cap, stream), "rarc_search_wide")
            flagged = bool((status[:nq] != 0).any().item())
            self.last_wide_cap = cap          # (diagnostics: the capacity this batch was answered at)
            if not flagged:
                self.last_repaired = []
                if unscale != 1.0:
                    out_sc.mul_(unscale)
                return
            if cap >= sure:
                raise B.RarcError("rarc_search_wide flagged a query at a capacity that holds every row")
            cap = min(4 * cap, sure)

    # ------------------------------------------------------------------ pipelined search
    def search_async(self, queries, k: int, to_host: bool = False) -> "PendingSearch":
        """Enqueue one batch and return at once; `.result()` later performs the status read-back (and the rare
        repair).  Lets a caller keep the GPU queue full: launch batch i+1, then collect batch i.  Results are
        identical to search_device().  More than 256 queries are enqueued as consecutive 256-query launches.
        to_host=True: the answer is written into pinned host memory by the search's own finalize kernel — `.host()` /
        `.host_view()` wait for this batch only and no device copy of the answer exists (`.result()` of such a handle returns
        the pinned host tensors)."""
        t = self.torch
        if k < 1:
            raise ValueError("k out of range")
        if self._takes_wide_path(k):       # (the wide path runs to completion: its handle is born finished)
            return _FinishedSearch(self, *self.search_device(queries, k))
        # (answers bound for the host — the store's batch calls — keep ONE context: behind them the time is python's (mapping
        #  25,600 Documents per batch), a second context only adds its bookkeeping: 1M rows, batch_invoke 266 k vs 258 k q/s,
        #  2048 queries in one call 218 k vs 185 k, tools/r06/api_ab.sh)
        ctx = None if to_host else self._pipeline_context(k)
        if ctx is not None:
            return ctx.search_async(queries, k, to_host)
        self._check_twin()
        if self._own_stream is not None and t.cuda.current_stream(self.device) != self._own_stream:
            # a twin: enqueue on its side stream, behind whatever produced the queries on the caller's stream
            # (result() waits for the batch's own event on the host, so the answer is safe to use on any stream)
            self._own_stream.wait_stream(t.cuda.current_stream(self.device))
            with t.cuda.stream(self._own_stream):
                return self.search_async(queries, k, to_host)
        with self._lock, t.cuda.device(self.device):
            q = t.as_tensor(queries, dtype=t.float32).to(self.device).contiguous()
            if q.ndim != 2 or q.shape[1] != self.dim or q.shape[0] < 1:
                raise ValueError(f"expected [nq][{self.dim}] queries, got {tuple(q.shape)}")
            nq = q.shape[0]
            parts = []
            # the answer's pinned staging slot belongs to the handle returned below until it is released / dies
            slot = self._pins.acquire(t, nq, k) if to_host else None
            h_ids, h_sc = slot.views(nq, k) if to_host else (None, None)
            if to_host:
                # to_host=True: the finalize kernel writes the answer STRAIGHT into the pinned slot (pinned host memory is
                # device-addressable): no device copy of it exists and no copy launches follow the search — behind a
                # neighbouring context's persistent scan a blit kernel does not get a CU until that scan ends, which held
                # chunk i's answer back until chunk i+1 had been scanned (2048 queries in one call: 1.55 ms per chunk where
                # a lone chunk takes 1.05).  result() of such a handle returns these pinned (host) tensors.
                out_ids, out_sc = h_ids, h_sc
            else:
                out_ids = t.empty((nq, k), dtype=t.int64, device=self.device)
                out_sc = t.empty((nq, k), dtype=t.float32, device=self.device)
            for s0 in range(0, nq, B.MAX_QUERIES):
                e0 = min(nq, s0 + B.MAX_QUERIES)
                # this launch's own status words (zeroed by its query-prep kernel) and its one any-flag word in pinned host
                # memory, written by the finalize kernel itself: no fill launch in front of the batch, no copy launch behind
                # it.  The event is the batch's own: result() waits for THIS batch only, not for what was enqueued after it
                status = t.empty(B.MAX_QUERIES + 1, dtype=t.int32, device=self.device)
                flag_h = self._flags.acquire(t)
                gate = None
                if self._partner is not None and _PIPELINE_GATE:
                    try:
                        gate = self._partner._fin_event
                    except ReferenceError:
                        gate = None
                self._search_chunk(q[s0:e0], k, out_ids[s0:e0], out_sc[s0:e0], repair=False, status=status,
                                   flag_host=flag_h.data_ptr(), gate=gate.cuda_event if gate is not None else 0)
                done = t.cuda.Event()
                done.record()
                self._fin_event = done
                parts.append(PendingSearch(self, q[s0:e0], k, out_ids[s0:e0], out_sc[s0:e0], flag_h,
                                           status[:e0 - s0], done, stream=t.cuda.current_stream(self.device),
                                           version=self._rows_version(),
                                           host=(h_ids[s0:e0], h_sc[s0:e0]) if to_host else None, cand_cap=self.cand_cap,
                                           slot=slot))
            return parts[0] if len(parts) == 1 else PendingBatches(parts, out_ids, out_sc,
                                                                   host=(h_ids, h_sc) if to_host else None, slot=slot)

    CAND_CAP_LIMIT = 1 << 21    # sticky growth of cand_cap stops here (4.3 GB of candidate keys per 128 results)

    def _grow_if_segments_overflowed(self, words, launched_with: Optional[int] = None) -> None:
        """A query flagged with RARC_Q_WHY_SEGMENT ran out of room in a scan workgroup's candidate segment: this CORPUS lets
        more rows through the int8 margin than `cand_cap` was sized for (clustered data at 100M rows: a query's whole
        cluster, ~100 K rows).  The re-run below fixes the answer; doubling the capacity for every LATER search fixes the
        cause — otherwise each batch pays a second scan of the shard (76.9 vs 39.6 ms per batch at 100M clustered rows).

        One overflow, one doubling: a batch that was LAUNCHED under a smaller capacity than today's (`launched_with` — with
        search_async several are in flight when the first of them reports) says nothing about today's, and does not
        double it again.  The growth stops where the workspace would take more than a tenth of the device's memory
        (CAND_CAP_LIMIT at most), is undone if that workspace cannot be allocated after all (_workspace), and is the
        capacity the index keeps: it does not decay (a corpus does not un-cluster)."""
        try:
            seg = any(int(w) & 0x100 for w in words)
        except TypeError:
            seg = False
        if not seg or self.cand_cap >= self.CAND_CAP_LIMIT:
            return
        if launched_with is not None and launched_with < self.cand_cap:
            return
        new_cap = min(self.CAND_CAP_LIMIT, 2 * self.cand_cap)
        total = int(self.torch.cuda.get_device_properties(self.device).total_memory)
        if int(self.lib.rarc_search_workspace_bytes(new_cap)) > total // 10:
            return
        self._cand_cap_before_growth = self.cand_cap
        self.cand_cap = new_cap
        self.cand_cap_grown = getattr(self, "cand_cap_grown", 0) + 1
        # the next search allocates the larger workspace.  Launches still in flight keep reading the old one: they are
        # on this index's stream, and the caching allocator hands a freed block to later work of the SAME stream only
        # (a twin's workspace is its own, allocated on the twin's stream)
        self._ws, self._cap_eff = None, 0

    def warm_up(self, k: int = 100, n_queries: int = 64, rounds: int = 6) -> int:
        """Let the index learn its corpus before the first user batch: stored rows (evenly spaced) are searched as queries
        until no candidate segment overflows any more, i.e. until `cand_cap` has settled (see above).  Queries of a
        retrieval workload look like the rows they are meant to find, so on a clustered corpus this pays the
        first-batch cost (at 100M clustered rows: 77 ms instead of 41) once, at load time, instead of in a user's call.
        A no-op on corpora the default capacity already fits.  Returns the number of capacity doublings it caused."""
        t = self.torch
        if self.ntotal == 0 or self.wide:
            return 0
        with t.cuda.device(self.device):
            n = min(int(n_queries), B.MAX_QUERIES, self.ntotal)
            pick = t.linspace(0, self.ntotal - 1, n, device=self.device).long()
            rows = self._rows[pick][:, : self.dim]
            if self.storage == "f8":
                q = rows.view(t.float8_e4m3fn).float() * self._rowscale[pick][:, None]
            else:
                q = rows.float()
            before = getattr(self, "cand_cap_grown", 0)
            kk = max(1, min(int(k), B.MAX_K, self.ntotal))
            for _ in range(max(1, int(rounds))):
                grown = getattr(self, "cand_cap_grown", 0)
                self.search_device(q, kk)
                if getattr(self, "cand_cap_grown", 0) == grown:
                    break
            self.warmed_up = True
            return getattr(self, "cand_cap_grown", 0) - before

    def _repair_rows(self, q, k, out_ids, out_sc, flagged, words=None, launched_with: Optional[int] = None) -> None:
        """Make the flagged rows of (out_ids, out_sc) exact (the shared query buffers may hold a later batch by now).

        First a re-run of the SEARCH for the flagged queries together, started from what the first attempt did
        establish: its k-th entry is the canonical score of a real row, hence a lower bound L of the true k-th
        best score, and a scan that begins at L - eps8 (rarc_qblock_set_floor) keeps only rows within one error
        bound of the final threshold — where the first attempt, whose threshold had to find its way up from a
        sample statistic, overflowed a candidate segment.  One more scan of the shard for all flagged queries
        (27 ms at 100M rows), with four times the candidate capacity.  Only what is STILL flagged after that —
        thousands of rows within eps8 of the k-th best score, i.e. near-duplicate clusters — goes through the
        exact single-query scan (62 ms per query at 100M rows)."""
        t = self.torch
        stream = self._stream()
        left = list(flagged)
        big = None
        cap_then = int(launched_with or self.cand_cap)      # the capacity the flagged batch ran with (before any growth)
        if words is not None:
            self._grow_if_segments_overflowed(words, launched_with)
        if left and self.ntotal:
            # the re-run gets a workspace of ITS OWN with four times the candidate capacity the batch ran with, released
            # afterwards.  If HBM cannot spare it the re-run is skipped — the exact per-query scan below needs no extra
            # memory (62 ms per query at 100M rows: `last_rerun_skipped` says when that happened).
            cap_big = 4 * cap_then * max(1, -(-int(k) // 128))
            try:
                big = t.empty(int(self.lib.rarc_search_workspace_bytes(cap_big)), dtype=t.uint8, device=self.device)
            except (t.cuda.OutOfMemoryError, RuntimeError) as exc:
                if "out of memory" not in str(exc).lower():
                    raise
                big = None
                self.last_rerun_skipped = getattr(self, "last_rerun_skipped", 0) + 1
        if big is not None:
            sel = t.as_tensor(left, dtype=t.long, device=self.device)
            sub_q = q[sel].contiguous()
            prev_i, prev_s = out_ids[sel].contiguous(), out_sc[sel].contiguous()
            self._workspace(k)
            b = self._qbuf
            # (queries the fp16 scan flagged UNCERTAIN — its certificate cannot separate rows closer than its own
            #  error bound, as in tight clusters — go through the int8-prefilter path, which needs no certificate)
            use_q8 = self._use_q8(k) or (self._qmeta is not None and self.storage == "f16")
            self._prep(sub_q, k, force_q8=use_q8)
            B.check(self.lib.rarc_qblock_set_floor(b["qblock"].data_ptr(), self.d_pad, prev_i.data_ptr(), prev_s.data_ptr(),
                                                   k, len(left), stream), "rarc_qblock_set_floor")
            new_i, new_s = t.empty_like(prev_i), t.empty_like(prev_s)
            status = t.zeros(B.MAX_QUERIES + 1, dtype=t.int32, device=self.device)
            lo, hi = self._bins(sub_q)
            kp = k if use_q8 else self.kprime_for(k)
            rows_ptr = self._rows.data_ptr()
            qm = self._qmeta.data_ptr() if (use_q8 and self._qmeta is not None) else 0
            self._call_search(rows_ptr, qm, len(left), k, kp, lo, hi, new_i, new_s, status, big, stream, cap=cap_big)
            st = status[: len(left)].cpu()
            ok = (st == 0)
            if bool(ok.any()):
                good = sel[ok.to(self.device)]
                out_ids[good] = new_i[ok.to(self.device)]
                out_sc[good] = new_s[ok.to(self.device)]
            left = [qi for qi, fine in zip(left, ok.tolist()) if not fine]
            self.last_rerun = len(flagged) - len(left)
            t.cuda.current_stream(self.device).synchronize()   # the re-run has finished with `big` before it is released
            del big
        if not left:
            return
        ws, b = self._workspace(k), self._qbuf
        self._prep(q, k)
        for qi in left:
            self._call_repair(qi, k, out_ids, out_sc, ws, stream)
            if int(b["found"].item()) & 0x80000000:
                raise B.RarcError(f"repair of query {qi} overflowed its scratch list")

    def _call_search(self, rows_ptr, qm, nq, k, kp, lo, hi, out_ids, out_sc, status, ws, stream, cap=None) -> None:
        b = self._qbuf
        cap = self._cap_eff if cap is None else int(cap)
        if self.storage == "f8":
            sc_ptr = self._rowscale.data_ptr() if self._rowscale is not None else 0
            B.check(self.lib.rarc_search_f8(rows_ptr, sc_ptr, self.ntotal, self.d_pad, qm, b["qblock"].data_ptr(), nq, k,
                                            kp, self.id_base, lo, hi, out_ids.data_ptr(), out_sc.data_ptr(),
                                            status.data_ptr(), ws.data_ptr(), ws.numel(), cap, stream),
                    "rarc_search_f8")
        elif self.storage == "f32":
            img = self._image16.data_ptr() if self._image16 is not None else 0
            B.check(self.lib.rarc_search_f32(rows_ptr, img, self.ntotal, self.d_pad, qm, b["qblock"].data_ptr(), nq, k,
                                             kp, self.id_base, lo, hi, out_ids.data_ptr(), out_sc.data_ptr(),
                                             status.data_ptr(), ws.data_ptr(), ws.numel(), cap, stream),
                    "rarc_search_f32")
        elif self.shadow and qm and self._shadow is not None:
            B.check(self.lib.rarc_search_f16_shadow(rows_ptr, self._shadow.data_ptr(), self.ntotal, self.d_pad, qm,
                                                    b["qblock"].data_ptr(), nq, k, kp, self.id_base, lo, hi,
                                                    out_ids.data_ptr(), out_sc.data_ptr(), status.data_ptr(),
                                                    ws.data_ptr(), ws.numel(), cap, stream),
                    "rarc_search_f16_shadow")
        else:
            B.check(self.lib.rarc_search_f16(rows_ptr, self.ntotal, self.d_pad, qm, b["qblock"].data_ptr(), nq, k, kp,
                                             self.id_base, lo, hi, out_ids.data_ptr(), out_sc.data_ptr(),
                                             status.data_ptr(), ws.data_ptr(), ws.numel(), cap, stream),
                    "rarc_search_f16")

    def _call_repair(self, qi, k, out_ids, out_sc, ws, stream) -> None:
        b = self._qbuf
        if self.storage == "f8":
            B.check(self.lib.rarc_repair_f8(self._rows.data_ptr(), self._rowscale.data_ptr(), self.ntotal, self.d_pad,
                                            b["qblock"].data_ptr(), qi, k, self.id_base, out_ids.data_ptr(),
                                            out_sc.data_ptr(), b["found"].data_ptr(), ws.data_ptr(), ws.numel(), stream),
                    "rarc_repair_f8")
        elif self.storage == "f32":
            B.check(self.lib.rarc_repair_f32(self._rows.data_ptr(), self.ntotal, self.d_pad, b["qblock"].data_ptr(), qi, k,
                                             self.id_base, out_ids.data_ptr(), out_sc.data_ptr(), b["found"].data_ptr(),
                                             ws.data_ptr(), ws.numel(), stream), "rarc_repair_f32")
        else:
            B.check(self.lib.rarc_repair_f16(self._rows.data_ptr(), self.ntotal, self.d_pad, b["qblock"].data_ptr(), qi, k,
                                             self.id_base, out_ids.data_ptr(), out_sc.data_ptr(), b["found"].data_ptr(),
                                             ws.data_ptr(), ws.numel(), stream), "rarc_repair_f16")

    def _bins(self, q) -> Tuple[float, float]:
        if self.metric == "cosine":
            return -1.0, 1.0
        bound = float(q.norm(dim=1).max().item()) * max(self.max_norm, 1e-30) * 1.001
        return -bound, bound

    def _search_chunk(self, q, k, out_ids, out_sc, repair, status=None, flag_host: int = 0, gate: int = 0) -> None:
        """One batch of at most 256 queries through rarc_search_batch: query prep + seed + scan + finalize in one foreign
        call; `status` (257 words) is zeroed by the prep kernel, `flag_host` (address of a pinned word, 0 = none) receives
        the any-flag word, `gate` (a hipEvent_t, 0 = none) holds the scan back until a neighbouring context has finished."""
        t = self.torch
        ws = self._workspace(k)
        b = self._qbuf
        if status is None:
            status = b["status"]
        nq = q.shape[0]
        stream = self._stream()
        lo, hi = self._bins(q)
        # the int8 path's threshold proof needs the k-th best approximate score, nothing beyond it (its 2·eps8
        # margin is the slack); the fp16 path's certificate wants k' > k candidates
        kp = k if self._use_q8(k) else self.kprime_for(k)
        rows_ptr = self._rows.data_ptr() if self._rows is not None else 0
        qm = self._qmeta.data_ptr() if (self._use_q8(k) and self._qmeta is not None and self.ntotal) else 0
        norm = 1 if self.metric == "cosine" else 0
        fmt, aux = 0, 0
        if self.storage == "f8":
            fmt, aux = 1, (self._rowscale.data_ptr() if self._rowscale is not None else 0)
        elif self.storage == "f32":
            fmt, aux = 2, (self._image16.data_ptr() if self._image16 is not None else 0)
        elif self.shadow and qm and self._shadow is not None:
            fmt, aux = 3, self._shadow.data_ptr()
        if q.stride(1) != 1:
            q = q.contiguous()
        batch = B.SearchBatch(rows_ptr, aux, fmt, self.ntotal, self.d_pad, qm,
                              q.data_ptr(), q.stride(0), nq, self.dim, norm, max(self.max_norm, 1.0) if norm else self.max_norm,
                              b["qblock"].data_ptr(),
                              k, kp, self.id_base, lo, hi,
                              out_ids.data_ptr(), out_sc.data_ptr(), status.data_ptr(), flag_host,
                              ws.data_ptr(), ws.numel(), self._cap_eff, gate)
        B.check(self.lib.rarc_search_batch(ctypes.byref(batch), stream), "rarc_search_batch")
        self.last_status = status[:nq]
        if not repair or self.ntotal == 0:
            return
        # one 4-byte read-back per batch (syncs); the per-query words are fetched only if it is set
        any_flag = int(status[B.MAX_QUERIES].item())
        words = status[:nq].cpu().tolist() if any_flag else []
        flagged = [i for i, w in enumerate(words) if w]
        self.last_repaired = flagged
        if flagged:
            self._repair_rows(q, k, out_ids, out_sc, flagged, words=[words[i] for i in flagged])

    def neighbors_above(self, queries, threshold: float, k_cap: int = 64):
        """Range query by score: for every query, the stored rows whose canonical score is >= threshold
        (the all-pairs-cosine-with-cut-off pattern of the reference's entity de-duplication,
        encapsulation/database/graph_db/Base_Neo4j.py:542-583, threshold 0.95, and of SemanticChunker's
        neighbour distances, core/file_management/chunker/spliter.py:354-371).  Runs as exact top-k_cap
        searches, 256 queries per scan; a query whose k_cap-th hit still clears the threshold is searched
        again with a larger k (up to the kernel limit) so that nothing is cut off silently.
        Returns a list (one entry per query) of (ids int64 array, scores fp32 array), best first."""
        t = self.torch
        q = t.as_tensor(queries, dtype=t.float32).to(self.device)
        if q.ndim == 1:
            q = q[None, :]
        out = [None] * q.shape[0]
        todo = list(range(q.shape[0]))
        k = max(1, min(int(k_cap), B.MAX_K - 28, max(self.ntotal, 1)))
        while todo:
            sub = q[t.as_tensor(todo, device=self.device)]
            ids, sc = self.search_device(sub, k)
            ids_h, sc_h = ids.cpu().numpy(), sc.cpu().numpy()
            again = []
            for j, qi in enumerate(todo):
                full = k < min(self.ntotal, B.MAX_K - 28) and sc_h[j, k - 1] >= threshold and ids_h[j, k - 1] >= 0
                if full:
                    again.append(qi)
                    continue
                m = (sc_h[j] >= threshold) & (ids_h[j] >= 0)
                out[qi] = (ids_h[j][m], sc_h[j][m])
            todo = again
            if todo:
                k = min(4 * k, B.MAX_K - 28, self.ntotal)
        return out

    def verify_query(self, queries, qi: int, ids, scores) -> int:
        """Run the exact repair scan on row `qi` of (ids, scores) [device tensors from search_device]
        and return how many rows beat the stored k-th entry (0 == the answer was already exact)."""
        t = self.torch
        with self._lock, t.cuda.device(self.device):
            ws = self._workspace()
            b = self._qbuf
            q = t.as_tensor(queries, dtype=t.float32).to(self.device).contiguous()
            c0 = (qi // B.MAX_QUERIES) * B.MAX_QUERIES          # the 256-query launch row qi belongs to
            c1 = min(q.shape[0], c0 + B.MAX_QUERIES)
            self._prep(q[c0:c1])
            self._call_repair(qi - c0, ids.shape[1], ids[c0:c1], scores[c0:c1], ws, self._stream())
            return int(b["found"].item())


    def verify_batch(self, queries, ids, scores, which=None, detail: bool = False):
        """Exact check of whole answers: for the queries `which` (default: all), by a canonical scan that reads every row
        once per EIGHT queries (rarc_verify_batch), (a) count the rows of the shard that beat the stored k-th entry and
        (b) look every row at or above the k-th entry up in the answer, (id, canonical score) compared bit for bit.
        Returns the number of rows beating a k-th entry PLUS the number of answer entries that are not such an exact
        pair (0 == every checked answer is exact, entry by entry); detail=True returns the two figures separately."""
        t = self.torch
        fmt = {"f16": 0, "f8": 1, "f32": 2}[self.storage]
        nq, k = ids.shape
        which = list(range(nq)) if which is None else sorted(set(int(w) for w in which))
        beating = wrong = 0
        with self._lock, t.cuda.device(self.device):
            self._workspace()
            b = self._qbuf
            q = t.as_tensor(queries, dtype=t.float32).to(self.device).contiguous()
            counts = t.zeros(24, dtype=t.int32, device=self.device)
            sc_ptr = self._rowscale.data_ptr() if self._rowscale is not None else 0
            for c0 in range(0, nq, B.MAX_QUERIES):
                c1 = min(nq, c0 + B.MAX_QUERIES)
                todo = [w - c0 for w in which if c0 <= w < c1]
                if not todo:
                    continue
                self._prep(q[c0:c1])
                ids_c, sc_c = ids[c0:c1].contiguous(), scores[c0:c1].contiguous()
                # groups of up to 8 CONSECUTIVE queries (the kernel takes a range)
                i = 0
                while i < len(todo):
                    j = i
                    while j + 1 < len(todo) and todo[j + 1] == todo[j] + 1 and j + 1 - i < 8:
                        j += 1
                    first, n = todo[i], j - i + 1
                    B.check(self.lib.rarc_verify_batch(self._rows.data_ptr(), sc_ptr, fmt, self.ntotal, self.d_pad,
                                                       b["qblock"].data_ptr(), first, n, k, self.id_base, ids_c.data_ptr(),
                                                       sc_c.data_ptr(), counts.data_ptr(), self._stream()),
                            "rarc_verify_batch")
                    c_h = counts.cpu().numpy().astype(np.int64)
                    got, pairs, skipped = c_h[:n], c_h[8:8 + n], c_h[16:16 + n]
                    valid_all = (ids_c[first:first + n] >= 0).sum(dim=1).cpu().numpy().astype(np.int64)
                    valid = (ids_c[first:first + n, : k - 1] >= 0).sum(dim=1).cpu().numpy().astype(np.int64)
                    full = (ids_c[first:first + n, k - 1] >= 0).cpu().numpy()      # (short answers are not pair-checked)
                    beating += int(np.maximum(got - valid, 0).sum())
                    short = np.where(full, np.maximum(valid_all - pairs, 0), 0)
                    if bool(((short > 0) & (skipped > 0)).any()):
                        # the kernel ran out of look-ups for a query (a thread met > 64 rows at or above its k-th entry):
                        # the pair count is a lower bound there — the check is incomplete, which is not a wrong answer
                        raise B.RarcError("rarc_verify_batch could not complete the pair check of queries "
                                          f"{[first + int(i) for i in np.nonzero((short > 0) & (skipped > 0))[0]]}: "
                                          "too many rows at or above their k-th entry fall to one thread")
                    wrong += int(short.sum())
                    i = j + 1
        return (beating, wrong) if detail else beating + wrong


class _FinishedSearch:
    """search_async over the wide path: the answer is complete when the handle is made."""

    def __init__(self, index, ids, scores):
        self.index, self.ids, self.scores, self.repaired = index, ids, scores, []

    def result(self):
        return self.ids, self.scores

    def host(self):
        return self.index.to_host(self.ids, self.scores)

    host_view = host

    def release(self) -> None:
        pass


class PendingSearch:
    """Handle returned by FlatIndexF16.search_async."""

    def __init__(self, index, q, k, ids, scores, flag, status, done=None, stream=None, version=None, host=None, cand_cap=None,
                 slot=None):
        self.index, self.q, self.k, self.ids, self.scores, self.flag, self.status = index, q, k, ids, scores, flag, status
        self.cand_cap = cand_cap   # the candidate capacity this batch was launched with
        self.host_copy = host   # (ids, scores) pinned tensors the finalize kernel wrote the answer into (to_host=True)
        self.slot = slot        # the _PinSlot those tensors are views of: ours (shared with the sibling launches of one call)
        self.done = done        # event recorded behind the copy of the status word into pinned memory (`flag`)
        self.stream = stream    # the stream the search was enqueued on (a twin's side stream, else the caller's)
        self.version = version  # version of the rows when it was enqueued
        self.repaired = None

    def _flag_word(self) -> int:
        """The launch's status word, read once; its pinned word goes back to the pool."""
        if self.done is not None:
            self.done.synchronize()   # this batch only: later batches keep running
            word = int(self.flag[0])
            flag, self.flag = self.flag, None
            self.index._flags.give_back(flag)
            return word
        return int(self.flag.item())

    def __del__(self):
        try:                        # a handle dropped without result(): its launch may still be writing the word
            if self.done is not None and self.flag is not None:
                self.done.synchronize()
                self.index._flags.give_back(self.flag)
        except Exception:
            pass

    def result(self):
        """(ids int64 [nq][k], scores fp32 [nq][k]), exact: device tensors — or, for a to_host=True batch, the pinned host
        tensors its finalize kernel wrote.

        A flagged query is repaired HERE, and the repair reuses the index's query block and workspace — which later
        batches of the same index (the other 256-query chunks of a PendingBatches, a twin's next batch) may still be
        reading.  So the repair is enqueued on the stream the search itself ran on: behind everything already queued
        there, ahead of everything queued later, whatever stream the collecting thread happens to be on; and that
        stream is drained before the answer is handed out, so it is safe to use on any stream."""
        if self.repaired is None:
            t = self.index.torch
            self.repaired = []
            if self._flag_word():
                words = self.status.cpu().tolist()
                self.repaired = [i for i, w in enumerate(words) if w]
                if self.version is not None and self.version != self.index._rows_version():
                    raise B.RarcError("the index rows changed while a search was in flight: its flagged queries cannot be "
                                      "repaired against the rows it scanned (collect results before add() / reset())")
                stream = self.stream if self.stream is not None else t.cuda.current_stream(self.index.device)
                with self.index._lock, t.cuda.device(self.index.device), t.cuda.stream(stream):
                    host_direct = not self.ids.is_cuda        # (to_host=True: the answer sits in pinned memory only)
                    ids = self.ids.to(self.index.device) if host_direct else self.ids
                    scores = self.scores.to(self.index.device) if host_direct else self.scores
                    self.index._repair_rows(self.q, self.k, ids, scores, self.repaired,
                                            words=[words[i] for i in self.repaired], launched_with=self.cand_cap)
                    if host_direct:                          # the repaired rows go back where the caller reads them
                        self.ids.copy_(ids)
                        self.scores.copy_(scores)
                    stream.synchronize()
            self.index.last_repaired = self.repaired
            if self.index._parent is not None:      # (a pipelined context of the index the caller holds)
                try:
                    self.index._parent.last_repaired = self.repaired
                except ReferenceError:
                    pass
        return self.ids, self.scores

    def host_view(self):
        """result() as numpy (scores fp32 [nq][k], ids int64 [nq][k]) WITHOUT a copy when the answer was staged behind the
        search (to_host=True): the arrays are views of this handle's pinned slot — valid until release() or until the
        handle dies, whichever comes first; nothing else can be handed that memory meanwhile.  A repaired batch — rare — is
        copied again (and then owns its arrays)."""
        ids, scores = self.result()
        if self.host_copy is not None:          # (written by the finalize kernel itself; a repair wrote its rows back too)
            return self.host_copy[1].numpy(), self.host_copy[0].numpy()
        t = self.index.torch
        with t.cuda.stream(self.stream if self.stream is not None else t.cuda.current_stream(self.index.device)):
            return self.index.to_host(ids, scores)

    def host(self):
        """result() as numpy arrays the caller owns.  With to_host=True the answer is in pinned memory once the batch's event
        is complete (nothing later on the stream is waited for); the staging slot is released here."""
        scores, ids = self.host_view()
        if self.host_copy is not None:
            scores, ids = scores.copy(), ids.copy()
        self.release()
        return scores, ids

    def release(self) -> None:
        """Give the pinned staging slot back (arrays from host_view() must not be read afterwards)."""
        self.host_copy = None
        slot, self.slot = self.slot, None
        if slot is not None:
            slot.release()


class PendingBatches:
    """search_async over more than 256 queries: one PendingSearch per 256-query launch, one result."""

    def __init__(self, parts, ids, scores, host=None, slot=None):
        self.parts, self.ids, self.scores = parts, ids, scores
        self.host_copy = host
        self.slot = slot
        for p in parts:             # the one slot is this handle's: a part must not hand it back on its own
            p.slot = None
        self.repaired = None

    def result(self):
        if self.repaired is None:
            self.repaired = []
            for i, p in enumerate(self.parts):
                p.result()                                  # repairs write into the shared output views
                self.repaired += [i * B.MAX_QUERIES + r for r in p.repaired]
            self.parts[0].index.last_repaired = self.repaired
        return self.ids, self.scores

    def host_view(self):
        ids, scores = self.result()
        if self.host_copy is not None:
            return self.host_copy[1].numpy(), self.host_copy[0].numpy()
        p = self.parts[0]
        t = p.index.torch
        with t.cuda.stream(p.stream if p.stream is not None else t.cuda.current_stream(p.index.device)):
            return p.index.to_host(ids, scores)

    def host(self):
        scores, ids = self.host_view()
        if self.host_copy is not None:
            scores, ids = scores.copy(), ids.copy()
        self.release()
        return scores, ids

    def release(self) -> None:
        self.host_copy = None
        for p in self.parts:
            p.host_copy = None
        slot, self.slot = self.slot, None
        if slot is not None:
            slot.release()
