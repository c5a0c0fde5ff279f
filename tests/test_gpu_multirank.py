"""The sharded path across REAL separate ranks, on one GPU: N processes (2 and 8), each owning one row shard of the
corpus on cuda:0, local HIP search -> rarc_pack_results -> all-gather of the packed (id, score) records ->
rarc_topk_merge_packed on every rank.  The collective's transport here is gloo (RCCL refuses two ranks on one device;
with one GPU per box that is the only way to put several ranks on real kernels) — everything either side of it is
the product path `bench.py --gpus N` runs over RCCL.  The merged answer must equal the single-shard answer bit for bit
on every rank, pipelined (search_async / finish) and synchronous alike.

The SAME workers also run over the **nccl** backend (RCCL over xGMI), one device per rank, for 2 / 4 / 8 ranks — on a box
that has that many GPUs; with fewer the RCCL cases skip (a lease of this pool has one GPU, so the first multi-GPU contact
is the driver's; these tests are what it meets).  What differs between the two transports is one argument of
init_process_group and the device index: everything else is exercised here on every run."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _init(rank, world, port, backend):
    """One rank's process group: gloo with every rank on cuda:0, or nccl (= RCCL) with rank r on cuda:r."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist

    dev = rank if backend == "nccl" else 0
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    return torch, dist, dev


def _need_gpus(world, backend):
    import torch

    if backend == "nccl" and torch.cuda.device_count() < world:
        pytest.skip(f"RCCL with {world} ranks needs {world} GPUs, this box has {torch.cuda.device_count()}")


def _worker(rank, world, port, n, d, nq, k, out_dir, backend="gloo"):
    torch, dist, dev = _init(rank, world, port, backend)

    from rag_arc_amd.hip import binding as B
    from rag_arc_amd.hip.engine import FlatIndexF16
    from rag_arc_amd.hip.sharded import ShardedFlatSearch, shard_range

    lib = B.load_library()
    lo, hi = shard_range(n, rank, world)
    cap = ((hi - lo + 31) // 32) * 32
    rows = torch.zeros((max(cap, 32), d), dtype=torch.float16, device="cuda")
    B.check(lib.rarc_synth_rows_f16(rows.data_ptr(), d, d, lo, hi - lo, 1234, 0))
    wide = B.padded_dim(d) > 1024 or k > 1024                                            # (the wide path takes scan="auto" only)
    if n in _F32_SIZES:       # storage="f32": the reference's own row format (the one mode inside 1e-5 on arbitrary inputs), sharded
        x = torch.zeros((hi - lo, d), dtype=torch.float32, device="cuda")
        B.check(lib.rarc_synth_rows_f32(x.data_ptr(), d, d, lo, hi - lo, 1234, 0))
        idx = FlatIndexF16(d, device=dev, id_base=lo, storage="f32")
        idx.add(x)
    else:
        idx = FlatIndexF16(d, device=dev, id_base=lo, scan="q8" if rank % 2 and not wide else "auto")     # mixed scan kernels across ranks
        idx.add_rows_f16(rows, 1.001, n_valid=hi - lo)
    q = torch.zeros((nq, d), dtype=torch.float32, device="cuda")
    B.check(lib.rarc_synth_rows_f32(q.data_ptr(), d, d, 0, nq, 4321, 0))
    s = ShardedFlatSearch(idx)
    assert s.world == world and s.rank == rank
    ids, sc = s.search_device(q, k)
    h1, h2 = s.search_async(q, k), s.search_async(q, k)                       # two batches in flight
    ids2, sc2 = s.finish(h1, k)
    ids3, sc3 = s.finish(h2, k)
    assert torch.equal(ids, ids2) and torch.equal(ids, ids3) and torch.equal(sc, sc2) and torch.equal(sc, sc3)
    np.savez(os.path.join(out_dir, f"r{rank}.npz"), ids=ids.cpu().numpy(), sc=sc.cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()


# shapes: the register-resident scans; rows of 1536 dimensions and k = 1500 (the wide path, round 5) behind the same exchange
_SHAPES = {"narrow": (300_001, 256, 96, 50), "wide_rows": (60_001, 1536, 40, 50), "wide_k": (40_001, 256, 24, 1500),
           "f32_rows": (200_003, 384, 64, 100)}
_F32_SIZES = {200_003}      # (the worker tells the fp32-storage case by its row count)


@pytest.mark.parametrize("world,backend,shape", [(2, "gloo", "narrow"), (8, "gloo", "narrow"), (2, "gloo", "wide_rows"),
                                                 (4, "gloo", "wide_k"), (4, "gloo", "f32_rows"), (2, "nccl", "f32_rows"),
                                                 (2, "nccl", "narrow"), (4, "nccl", "narrow"),
                                                 (8, "nccl", "narrow"), (2, "nccl", "wide_rows")])
def test_sharded_search_across_real_ranks(tmp_path, world, backend, shape):
    import torch
    import torch.multiprocessing as mp

    from rag_arc_amd.hip import binding as B
    from rag_arc_amd.hip.engine import FlatIndexF16

    _need_gpus(world, backend)
    n, d, nq, k = _SHAPES[shape]
    port = 29650 + world + (20 if backend == "nccl" else 0) + 40 * list(_SHAPES).index(shape)
    mp.spawn(_worker, args=(world, port, n, d, nq, k, str(tmp_path), backend), nprocs=world, join=True)
    lib = B.load_library()
    if n in _F32_SIZES:
        x = torch.zeros((n, d), dtype=torch.float32, device="cuda")
        B.check(lib.rarc_synth_rows_f32(x.data_ptr(), d, d, 0, n, 1234, 0))
        single = FlatIndexF16(d, storage="f32")
        single.add(x)
    else:
        rows = torch.zeros((((n + 31) // 32) * 32, d), dtype=torch.float16, device="cuda")
        B.check(lib.rarc_synth_rows_f16(rows.data_ptr(), d, d, 0, n, 1234, 0))
        single = FlatIndexF16(d)
        single.add_rows_f16(rows, 1.001, n_valid=n)
    q = torch.zeros((nq, d), dtype=torch.float32, device="cuda")
    B.check(lib.rarc_synth_rows_f32(q.data_ptr(), d, d, 0, nq, 4321, 0))
    D, I = single.search(q, k)
    for r in range(world):
        got = np.load(tmp_path / f"r{r}.npz")
        assert np.array_equal(got["ids"], I), f"rank {r}: ids differ from the single-shard answer"
        assert np.array_equal(got["sc"].view(np.uint32), D.view(np.uint32)), f"rank {r}: scores differ"


# ---- the registered sharded store across ranks: build, search, save, load --------------------------------------------------
def _store_worker(rank, world, port, folder, out_dir, backend):
    torch, dist, dev = _init(rank, world, port, backend)
    import pickle

    from rag_arc_amd.encapsulation.database.vector_db.hip_sharded import HipShardedFlatVectorStore
    from tests.helpers import HashEmbeddings

    emb = HashEmbeddings(192)
    store = HipShardedFlatVectorStore(emb, device=dev, storage="f8" if world == 4 else "f16")
    texts = [f"entry {i} of group {i % 11}" for i in range(4003)]
    store.add_texts(texts, ids=[f"e{i}" for i in range(4003)])
    queries = ["entry 17 of group 6", "entry 4000 of group 7", "nothing like the rest"]
    before = [[(d.id, s) for d, s in store.similarity_search_with_score(q, k=25)] for q in queries]
    batch = store.batch_similarity_search_with_score(queries, k=25)
    assert [[(d.id, s) for d, s in one] for one in batch] == before
    store.save_local(folder)                                   # one shard file per rank, rank 0 writes the docstore
    again = HipShardedFlatVectorStore.load_local(folder, emb, device=dev, storage=store.storage)
    assert again.shard == store.shard
    after = [[(d.id, s) for d, s in again.similarity_search_with_score(q, k=25)] for q in queries]
    assert after == before
    with open(os.path.join(out_dir, f"s{rank}.pkl"), "wb") as fh:
        pickle.dump(dict(before=before, shard=store.shard), fh)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,backend", [(2, "gloo"), (2, "nccl"), (4, "nccl"), (8, "nccl")])
def test_sharded_store_across_real_ranks(tmp_path, world, backend):
    import pickle

    import torch.multiprocessing as mp

    from rag_arc_amd.encapsulation.database.vector_db import HipFlatVectorStore
    from tests.helpers import HashEmbeddings

    _need_gpus(world, backend)
    folder = str(tmp_path / "idx")
    mp.spawn(_store_worker, args=(world, 29700 + world + (20 if backend == "nccl" else 0), folder, str(tmp_path), backend),
             nprocs=world, join=True)
    emb = HashEmbeddings(192)
    storage = "f8" if world == 4 else "f16"
    texts = [f"entry {i} of group {i % 11}" for i in range(4003)]
    single = HipFlatVectorStore.from_texts(texts, emb, ids=[f"e{i}" for i in range(4003)], storage=storage)
    want = [[(d.id, s) for d, s in single.similarity_search_with_score(q, k=25)]
            for q in ["entry 17 of group 6", "entry 4000 of group 7", "nothing like the rest"]]
    rows = 0
    for r in range(world):
        got = pickle.load(open(tmp_path / f"s{r}.pkl", "rb"))
        assert got["before"] == want, f"rank {r}: the merged answer differs from the single-GPU store's"
        assert got["shard"][:2] == (r, world) and got["shard"][3] == 4003
        rows += got["shard"][2]
    assert rows == 4003
    # the rank files, loaded into ONE GPU: the same store again
    merged = HipFlatVectorStore.load_local(folder, emb, storage=storage)
    assert merged.ntotal == 4003
    assert [[(d.id, s) for d, s in merged.similarity_search_with_score(q, k=25)]
            for q in ["entry 17 of group 6", "entry 4000 of group 7", "nothing like the rest"]] == want


@pytest.mark.parametrize("world", [2, 4, 8])
def test_bench_over_rccl(world):
    """`bench.py --gpus N` as the driver runs it at round end, at a small size: N RCCL ranks, every rank's exhaustive check
    clean, the line carries n_gpus = rccl_ranks = N."""
    import json
    import subprocess

    _need_gpus(world, "nccl")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--rows", "4000000", "--steps", "3",
                           "--warmup", "1", "--c5-rows", "2000000", "--c5-layers", "2", "--no-cpu-baseline", "--verify-queries", "32"],
                          capture_output=True, text=True, timeout=1200, env=env, cwd=ROOT)
    assert proc.returncode == 0, proc.stderr[-3000:]
    line = json.loads(proc.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == world and line["rccl_ranks"] == world and line["scaling"] == "strong"
    assert line["config"]["full_size_check"]["rows_beating_kth"] == 0 and line["config"]["full_size_check"]["ranks_checked"] == world
    assert line["config"]["rows_per_gpu"] * world >= 4000000 and line["value"] > 0
    assert line["c5"]["value"] > 0
