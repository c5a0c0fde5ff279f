"""The sharded path across REAL separate ranks, on one GPU: N processes (2 and 8), each owning one row shard of the
corpus on cuda:0, local HIP search -> rarc_pack_results -> all-gather of the packed (id, score) records ->
rarc_topk_merge_packed on every rank.  The collective's transport here is gloo (RCCL refuses two ranks on one device;
with one GPU per box that is the only way to put several ranks on real kernels) — everything either side of it is
the product path `bench.py --gpus N` runs over RCCL.  The merged answer must equal the single-shard answer bit for bit
on every rank, pipelined (search_async / finish) and synchronous alike."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, n, d, nq, k, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    import torch
    import torch.distributed as dist

    from rag_arc_amd.hip import binding as B
    from rag_arc_amd.hip.engine import FlatIndexF16
    from rag_arc_amd.hip.sharded import ShardedFlatSearch, shard_range

    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    lib = B.load_library()
    lo, hi = shard_range(n, rank, world)
    cap = ((hi - lo + 31) // 32) * 32
    rows = torch.zeros((max(cap, 32), d), dtype=torch.float16, device="cuda")
    B.check(lib.rarc_synth_rows_f16(rows.data_ptr(), d, d, lo, hi - lo, 1234, 0))
    idx = FlatIndexF16(d, id_base=lo, scan="q8" if rank % 2 else "auto")     # mixed scan kernels across ranks
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


@pytest.mark.parametrize("world", [2, 8])
def test_sharded_search_across_real_ranks(tmp_path, world):
    import torch
    import torch.multiprocessing as mp

    from rag_arc_amd.hip import binding as B
    from rag_arc_amd.hip.engine import FlatIndexF16

    n, d, nq, k = 300_001, 256, 96, 50
    mp.spawn(_worker, args=(world, 29650 + world, n, d, nq, k, str(tmp_path)), nprocs=world, join=True)
    lib = B.load_library()
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
