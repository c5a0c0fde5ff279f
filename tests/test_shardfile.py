"""The `.rarc` shard file and the stores' save_local / load_local around it, on CPU: header and id map
round trips, round-3 (version 2) files still read, the re-sharding plan, and a two-process gloo run of the sharded store
— save as 2 ranks, load as 2 ranks / 1 process / 3 ranks' worth of plans — whose answers equal the single-shard ones.
The engines here are the oracle-backed doubles of tests/helpers.py (host arrays); the byte mover itself
(rarc_file_to_device / rarc_device_to_file) needs HBM and is covered by tests/test_gpu_persistence.py."""
import os
import pickle
import struct
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp
from hypothesis import given, settings
from hypothesis import strategies as st

from rag_arc_amd.hip import shardfile as SF
from rag_arc_amd.hip.sharded import shard_range

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_round_trip_and_offsets(tmp_path):
    hdr = SF.ShardHeader(1000, 384, 384, SF.CODES["f8"], 1.25, rank=3, world=8, global_ntotal=8000,
                         blocks=[(375, 125), (1375, 875)])
    assert len(hdr.pack()) == 64
    assert hdr.rows_offset == 4096 and hdr.scales_offset % 4096 == 0 and hdr.idmap_offset % 8 == 0
    assert hdr.scales_offset >= 4096 + 1000 * 384 and hdr.idmap_offset >= hdr.scales_offset + 4000
    path = str(tmp_path / "x.rarc")
    with open(path, "wb") as fh:
        fh.write(hdr.pack())
        fh.truncate(hdr.file_bytes)
        fh.seek(hdr.idmap_offset)
        fh.write(hdr.idmap_bytes())
    back = SF.read_header(path)
    assert (back.n_rows, back.dim, back.d_pad, back.storage, back.rank, back.world, back.global_ntotal) == \
        (1000, 384, 384, "f8", 3, 8, 8000)
    assert back.blocks == [(375, 125), (1375, 875)] and back.max_norm == 1.25 and back.version == 3
    # damage is reported, not read through
    with open(path, "r+b") as fh:
        fh.truncate(hdr.file_bytes - 8)
    with pytest.raises(ValueError, match="truncated"):
        SF.read_header(path)
    with open(path, "r+b") as fh:
        fh.write(b"\0" * 8)
    with pytest.raises(ValueError, match="not a rarc shard file"):
        SF.read_header(path)


def test_round3_files_are_still_read(tmp_path):
    """Version 2 (round 3): six int64 words, max_norm at byte 48, rows at byte 64, fp8 scales straight behind the rows."""
    path = str(tmp_path / "old.rarc")
    n, dim, d_pad = 10, 100, 256
    with open(path, "wb") as fh:
        fh.write(np.array([SF.MAGIC, 2, n, dim, d_pad, 1], dtype=np.int64).tobytes())
        fh.write(np.float32(0.75).tobytes())
        fh.write(b"\0" * 12)
        fh.write(bytes(n * d_pad))
        fh.write(np.ones(n, np.float32).tobytes())
    hdr = SF.read_header(path)
    assert (hdr.version, hdr.n_rows, hdr.dim, hdr.d_pad, hdr.storage, hdr.max_norm) == (2, n, dim, d_pad, "f8", 0.75)
    assert hdr.rows_offset == 64 and hdr.scales_offset == 64 + n * d_pad and hdr.blocks == [(0, n)]


@settings(max_examples=200, deadline=None)
@given(adds=st.lists(st.integers(0, 40), min_size=1, max_size=6), saved=st.integers(1, 5), new=st.integers(1, 6))
def test_reshard_plan_covers_every_row_once_in_order(adds, saved, new):
    # the saved layout is what HipShardedFlatVectorStore.add_texts produces: every add() call split over the ranks
    file_blocks = [[] for _ in range(saved)]
    start = 0
    for n_block in adds:
        for r in range(saved):
            lo, hi = shard_range(n_block, r, saved)
            if hi > lo:
                file_blocks[r].append((start + lo, hi - lo))
        start += n_block
    total = start
    # global id of (file, row)
    gid = [np.concatenate([np.arange(g, g + c) for g, c in fb]) if fb else np.zeros(0, np.int64) for fb in file_blocks]
    seen = []
    for r in range(new):
        segs, blocks, tot = SF.plan_reshard(file_blocks, r, new)
        assert tot == total
        ids = np.concatenate([gid[f][a:a + c] for f, a, c in segs]) if segs else np.zeros(0, np.int64)
        lo, hi = shard_range(total, r, new)
        assert ids.tolist() == list(range(lo, hi))                   # exactly its range, ascending
        assert sum(c for _, c in blocks) == hi - lo
        assert np.concatenate([np.arange(g, g + c) for g, c in blocks]).tolist() == ids.tolist() if len(ids) else not blocks
        seen += ids.tolist()
    assert seen == list(range(total))


def test_reshard_plan_rejects_a_missing_file():
    with pytest.raises(ValueError, match="do not tile"):
        SF.plan_reshard([[(0, 5)], [(10, 5)]], 0, 1)


# ---- the sharded store, two gloo ranks, oracle engines --------------------------------------------------------------------
def _engine(dim, metric, device):
    from tests.helpers import OracleIndex

    return OracleIndex(dim, metric)


def _cpu_merge(ids, scores, k):
    import torch

    from oracle import cpu_ref

    i, s = cpu_ref.topk_merge(ids.numpy(), scores.numpy(), k)
    return torch.from_numpy(i), torch.from_numpy(s)


def _texts(n, first=0):
    return [f"passage {first + i} about topic {(first + i) % 17}" for i in range(n)]


def _worker(rank, world, port, folder, out_path):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    import torch.distributed as dist

    from rag_arc_amd.encapsulation.database.vector_db.hip_sharded import HipShardedFlatVectorStore
    from tests.helpers import HashEmbeddings

    dist.init_process_group("gloo", rank=rank, world_size=world)
    emb = HashEmbeddings(96)
    store = HipShardedFlatVectorStore(emb, engine_factory=_engine, merge_fn=_cpu_merge)
    store.add_texts(_texts(301), ids=[f"a{i}" for i in range(301)])
    store.add_texts(_texts(77, 301), ids=[f"a{301 + i}" for i in range(77)])       # a second block: the id map has two entries
    queries = ["passage 5 about topic 5", "passage 350 about topic 10", "something else entirely"]
    before = [[(d.id, s) for d, s in store.similarity_search_with_score(q, k=12)] for q in queries]
    store.save_local(folder)
    again = HipShardedFlatVectorStore.load_local(folder, emb, engine_factory=_engine, merge_fn=_cpu_merge)
    assert again.shard == store.shard and again.index._blocks == store.index._blocks
    after = [[(d.id, s) for d, s in again.similarity_search_with_score(q, k=12)] for q in queries]
    assert after == before
    # an emptied store saved into the same folder leaves no rank file behind
    if rank == 0:
        with open(out_path, "wb") as fh:
            pickle.dump(dict(before=before, shard=store.shard, blocks=store.index._blocks), fh)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_store_saves_loads_and_reshards(tmp_path, oracle):
    from rag_arc_amd.encapsulation.database.vector_db import HipFlatVectorStore
    from tests.helpers import HashEmbeddings

    folder, out = str(tmp_path / "idx"), str(tmp_path / "r0.pkl")
    mp.spawn(_worker, args=(2, 29533, folder, out), nprocs=2, join=True)
    got = pickle.load(open(out, "rb"))
    assert sorted(os.listdir(folder)) == ["index.pkl", "index.r0of2.rarc", "index.r1of2.rarc"]
    h0, h1 = (SF.read_header(os.path.join(folder, f"index.r{r}of2.rarc")) for r in range(2))
    assert (h0.rank, h0.world, h0.global_ntotal) == (0, 2, 378) and (h1.rank, h1.world) == (1, 2)
    assert h0.blocks == [(0, 151), (301, 39)] and h1.blocks == [(151, 150), (340, 38)]
    # the same save loaded by ONE process (all rows, both files): same answers as the two ranks gave, and as a store
    # built in one piece gives
    emb = HashEmbeddings(96)
    one = HipFlatVectorStore.load_local(folder, emb, engine_factory=_engine)
    assert one.ntotal == 378
    built = HipFlatVectorStore.from_texts(_texts(378), emb, ids=[f"a{i}" for i in range(378)], engine_factory=_engine)
    assert np.array_equal(one.index.rows, built.index.rows)
    for q, want in zip(["passage 5 about topic 5", "passage 350 about topic 10", "something else entirely"], got["before"]):
        assert [(d.id, s) for d, s in one.similarity_search_with_score(q, k=12)] == want
        assert [(d.id, s) for d, s in built.similarity_search_with_score(q, k=12)] == want
    # saving the single store into the same folder replaces the rank files
    one.save_local(folder)
    assert sorted(os.listdir(folder)) == ["index.pkl", "index.rarc"]
    # a rank file that went missing is noticed
    mp.spawn(_worker, args=(2, 29534, folder, out), nprocs=2, join=True)
    os.unlink(os.path.join(folder, "index.r1of2.rarc"))
    with pytest.raises(ValueError, match="missing"):
        HipFlatVectorStore.load_local(folder, emb, engine_factory=_engine)
