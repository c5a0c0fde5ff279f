"""The `.rarc` shard file: the stored rows of one GPU's share of a flat index, laid out so that they stream between
a file and HBM without ever existing as a host array (rarc_file_to_device / rarc_device_to_file, csrc/shard_io.hip).

Counterpart of the `index.faiss` file of the reference (faiss.write_index / faiss.read_index,
encapsulation/database/vector_db/VectorStore_Faiss.py:438, :467); the docstore stays in the pickle next to it, as in
the reference (:441-450, :459-460).  SURVEY.md §8 f1.

Layout, version 3 (little endian):

    [0, 64)      header   magic u64 "RARC" | version u32 = 3 | code u32 (0 fp16, 1 fp8 e4m3fn + row scales, 2 fp32)
                          n_rows i64 | dim u32 | d_pad u32 | max_norm f32 | n_blocks u32 | rank u32 | world u32
                          global_ntotal i64 | rows_offset u32 = 4096 | reserved u32
    [4096, ...)  rows     n_rows x d_pad elements, row-major, exactly as they sit in HBM (page aligned: mmap- and
                          O_DIRECT-able)
    (4096-aligned)        fp8 only: n_rows fp32 row scales
    (8-aligned)  id map   n_blocks x (first global id i64, row count i64): local rows in order, global ids increasing —
                          one block (id_base, n_rows) for a single-GPU store, one per add() call for a rank of the sharded
                          store (hip_sharded._ShardedIndex._blocks)

Version 2 files (round 3: rows at byte 64, scales straight behind them, no id map) are still read.
"""
from __future__ import annotations

import os
import struct
from dataclasses import dataclass, field
from typing import List, Sequence, Tuple

MAGIC = 0x43524152
VERSION = 3
HEADER_BYTES = 64
ROWS_OFFSET = 4096
CODES = {"f16": 0, "f8": 1, "f32": 2}
STORAGE_OF = {v: k for k, v in CODES.items()}
ELEM_BYTES = {0: 2, 1: 1, 2: 4}
_HDR = struct.Struct("<QIIqIIfIIIqII")
assert _HDR.size == HEADER_BYTES


def _align(x: int, a: int) -> int:
    return (x + a - 1) // a * a


@dataclass
class ShardHeader:
    n_rows: int
    dim: int
    d_pad: int
    code: int
    max_norm: float
    rank: int = 0
    world: int = 1
    global_ntotal: int = -1
    blocks: List[Tuple[int, int]] = field(default_factory=list)
    version: int = VERSION

    # -- derived offsets ------------------------------------------------------------------------------------------------
    @property
    def storage(self) -> str:
        return STORAGE_OF[self.code]

    @property
    def row_bytes(self) -> int:
        return self.d_pad * ELEM_BYTES[self.code]

    @property
    def rows_offset(self) -> int:
        return ROWS_OFFSET if self.version >= 3 else HEADER_BYTES

    @property
    def scales_offset(self) -> int:
        end = self.rows_offset + self.n_rows * self.row_bytes
        return _align(end, 4096) if self.version >= 3 else end

    @property
    def idmap_offset(self) -> int:
        end = self.scales_offset + 4 * self.n_rows if self.code == 1 else self.rows_offset + self.n_rows * self.row_bytes
        return _align(end, 8)

    @property
    def file_bytes(self) -> int:
        return self.idmap_offset + 16 * len(self.blocks)

    def pack(self) -> bytes:
        return _HDR.pack(MAGIC, VERSION, self.code, self.n_rows, self.dim, self.d_pad, float(self.max_norm), len(self.blocks),
                         self.rank, self.world, self.global_ntotal, ROWS_OFFSET, 0)

    def idmap_bytes(self) -> bytes:
        return b"".join(struct.pack("<qq", int(g0), int(n)) for g0, n in self.blocks)


def read_header(path: str) -> ShardHeader:
    size = os.path.getsize(path)
    with open(path, "rb") as fh:
        raw = fh.read(HEADER_BYTES)
        if len(raw) < HEADER_BYTES or struct.unpack_from("<Q", raw, 0)[0] != MAGIC:
            raise ValueError(f"{path}: not a rarc shard file")
        version, hi = struct.unpack_from("<II", raw, 8)
        if version in (1, 2) and hi == 0:     # round-3 layout: six int64 words, max_norm as fp32 behind them
            _, _, n, dim, d_pad, code = struct.unpack_from("<6q", raw, 0)
            if version == 1:
                code = 0
            (max_norm,) = struct.unpack_from("<f", raw, 48 if version == 2 else 40)
            hdr = ShardHeader(int(n), int(dim), int(d_pad), int(code), float(max_norm), version=int(version))
            hdr.global_ntotal = hdr.n_rows
            hdr.blocks = [(0, hdr.n_rows)]
        elif version == VERSION:
            (_, _, code, n, dim, d_pad, max_norm, n_blocks, rank, world, gtotal, rows_off, _) = _HDR.unpack(raw)
            if rows_off != ROWS_OFFSET:
                raise ValueError(f"{path}: rows at byte {rows_off}, expected {ROWS_OFFSET}")
            if code not in ELEM_BYTES:
                raise ValueError(f"{path}: unknown row format {code}")
            hdr = ShardHeader(int(n), int(dim), int(d_pad), int(code), float(max_norm), int(rank), int(world), int(gtotal))
            hdr.blocks = [(0, 0)] * int(n_blocks)      # sized first: idmap_offset / file_bytes depend on the count only
            if size < hdr.file_bytes:
                raise ValueError(f"{path}: truncated ({size} bytes, header asks for {hdr.file_bytes})")
            fh.seek(hdr.idmap_offset)
            raw_map = fh.read(16 * int(n_blocks))
            hdr.blocks = [struct.unpack_from("<qq", raw_map, 16 * i) for i in range(int(n_blocks))]
            if sum(n_b for _, n_b in hdr.blocks) != hdr.n_rows:
                raise ValueError(f"{path}: the id map covers {sum(n_b for _, n_b in hdr.blocks)} rows, the file holds {hdr.n_rows}")
        else:
            raise ValueError(f"{path}: unsupported shard file version {version}")
    if hdr.code not in ELEM_BYTES or hdr.n_rows < 0 or hdr.d_pad <= 0 or hdr.dim <= 0 or hdr.dim > hdr.d_pad:
        raise ValueError(f"{path}: inconsistent header")
    data_end = hdr.scales_offset + 4 * hdr.n_rows if hdr.code == 1 else hdr.rows_offset + hdr.n_rows * hdr.row_bytes
    if size < data_end:
        raise ValueError(f"{path}: truncated ({size} bytes, the rows end at byte {data_end})")
    return hdr


def shard_path(folder: str, index_name: str, rank: int = 0, world: int = 1) -> str:
    """One file for a single-GPU store (`<name>.rarc`), one per rank for the sharded store (`<name>.r<rank>of<world>.rarc`)."""
    if world == 1:
        return os.path.join(folder, f"{index_name}.rarc")
    return os.path.join(folder, f"{index_name}.r{rank}of{world}.rarc")


def plan_reshard(file_blocks: Sequence[Sequence[Tuple[int, int]]], rank: int, world: int):
    """Which rows of which saved files rank `rank` of a NEW world size loads.

    `file_blocks[f]` is the id map of saved file f.  Together the blocks tile the global ids [0, N).  The new rank takes the
    contiguous global range shard_range(N, rank, world) — any assignment gives the same search results (canonical scores do
    not depend on the sharding); a contiguous one keeps local row order = global id order, which the tie rule needs.
    Returns (segments, blocks): segments = [(file index, first row in that file, row count)] in global-id order, blocks = the
    new rank's id map."""
    from .sharded import shard_range

    pieces = []   # (g0, n, file, first row in file)
    for f, blocks in enumerate(file_blocks):
        row = 0
        for g0, n in blocks:
            if n:
                pieces.append((int(g0), int(n), f, row))
            row += int(n)
    pieces.sort()
    pos = 0
    for g0, n, _, _ in pieces:
        if g0 != pos:
            raise ValueError(f"the saved shards do not tile the global ids: expected a block at {pos}, found one at {g0}")
        pos += n
    lo, hi = shard_range(pos, rank, world)
    segments, blocks = [], []
    for g0, n, f, row in pieces:
        a, b = max(g0, lo), min(g0 + n, hi)
        if a < b:
            segments.append((f, row + (a - g0), b - a))
            if blocks and blocks[-1][0] + blocks[-1][1] == a:
                blocks[-1] = (blocks[-1][0], blocks[-1][1] + (b - a))
            else:
                blocks.append((a, b - a))
    return segments, blocks, pos
