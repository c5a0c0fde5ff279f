"""The device side of the shard exchange (pack -> RCCL all-gather -> merge over the packed records) with the
collective forced on a single rank: must hand back exactly what the local search returned.  (Two ranks are
covered on CPU by test_sharded_gloo.py; a GPU box here has one GPU.)"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_forced_collective_round_trip(oracle):
    import torch
    import torch.distributed as dist

    from rag_arc_amd.hip.engine import FlatIndexF16
    from rag_arc_amd.hip.sharded import ShardedFlatSearch

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29561")
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        n, d, k = 20_000, 384, 50
        rows = oracle.synth_rows_f16(n, d)
        idx = FlatIndexF16(d, metric="cosine", device=0, id_base=1_000_000_000_000)   # ids beyond 32 bits
        idx.load_rows(rows.view(np.float16), 1.001)                                  # host rows in storage format
        q = torch.from_numpy(oracle.synth_rows_f32(9, d)).cuda()
        li, ls = idx.search_device(q, k)
        s = ShardedFlatSearch(idx, force_collective=True)
        gi, gs = s.search_device(q, k)
        assert torch.equal(gi, li) and torch.equal(gs.view(torch.int32), ls.view(torch.int32))
        h = s.search_async(q, k)                                                     # pipelined form
        pi, ps = s.finish(h, k)
        assert torch.equal(pi, li) and torch.equal(ps.view(torch.int32), ls.view(torch.int32))
    finally:
        if created:
            dist.destroy_process_group()
