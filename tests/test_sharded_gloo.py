"""N>1 path on CPU: two processes (gloo), each owning half of the rows, local top-k -> all-gather ->
merge must equal the single-shard answer bit for bit (the merge and the local search are supplied by
the oracle here; on the GPU box the same ShardedFlatSearch drives the HIP kernels over RCCL)."""
import os
import sys

import numpy as np
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class _LocalOracleIndex:
    def __init__(self, rows, lo):
        self.rows, self.lo = rows, lo

    def search_device(self, q, k):
        from oracle import cpu_ref

        ids, sc, _ = cpu_ref.flat_search_f16(self.rows, np.asarray(q, np.float32), k, id_base=self.lo)
        return torch.from_numpy(ids), torch.from_numpy(sc)


def _cpu_merge(ids, scores, k):
    from oracle import cpu_ref

    i, s = cpu_ref.topk_merge(ids.numpy(), scores.numpy(), k)
    return torch.from_numpy(i), torch.from_numpy(s)


def _worker(rank, world, port, n, d, k, out_path):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    import torch.distributed as dist
    from oracle import cpu_ref
    from rag_arc_amd.hip.sharded import ShardedFlatSearch, shard_range

    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_range(n, rank, world)
    rows = cpu_ref.synth_rows_f16(hi - lo, d, first_row=lo)
    q = cpu_ref.synth_rows_f32(6, d)
    s = ShardedFlatSearch(_LocalOracleIndex(rows, lo), merge_fn=_cpu_merge)
    ids, sc = s.search_device(q, k)
    if rank == 0:
        np.savez(out_path, ids=ids.numpy(), sc=sc.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_search_equals_single_shard(tmp_path, oracle):
    n, d, k = 5001, 128, 20
    out = str(tmp_path / "r0.npz")
    mp.spawn(_worker, args=(2, 29517, n, d, k, out), nprocs=2, join=True)
    got = np.load(out)
    rows = oracle.synth_rows_f16(n, d)
    ids, sc, _ = oracle.flat_search_f16(rows, oracle.synth_rows_f32(6, d), k)
    assert np.array_equal(got["ids"], ids) and np.array_equal(got["sc"].view(np.uint32), sc.view(np.uint32))


def test_shard_ranges_cover_everything():
    from rag_arc_amd.hip.sharded import shard_range

    for n, g in ((100_000_000, 8), (10, 3), (7, 8), (0, 2)):
        spans = [shard_range(n, r, g) for r in range(g)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    assert shard_range(100_000_000, 3, 8) == (37_500_000, 50_000_000)


def test_pack_unpack_round_trip():
    from rag_arc_amd.hip.sharded import pack_results, unpack_results

    ids = torch.tensor([[5, -1, 2 ** 40 + 3]], dtype=torch.int64)
    sc = torch.tensor([[0.5, float("-inf"), -1.25]], dtype=torch.float32)
    i2, s2 = unpack_results(torch, pack_results(torch, ids, sc))
    assert torch.equal(i2, ids) and torch.equal(s2, sc)
