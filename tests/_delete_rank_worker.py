"""One rank of tests/test_gpu_delete.py::test_two_rank_sharded_delete_equals_single_store (gloo, both ranks on cuda:0)."""
import sys

import torch
import torch.distributed as dist

rank, world = int(sys.argv[1]), int(sys.argv[2])
dist.init_process_group("gloo", rank=rank, world_size=world)

from rag_arc_amd.encapsulation.database.vector_db import HipFlatVectorStore, HipShardedFlatVectorStore  # noqa: E402
from tests.helpers import HashEmbeddings  # noqa: E402

emb = HashEmbeddings(96)
texts = [f"record {i} in group {i % 11}" for i in range(3001)]
ids = [f"r{i}" for i in range(3001)]
sharded = HipShardedFlatVectorStore(emb, device=0)
sharded.add_texts(texts[:2000], ids=ids[:2000])
sharded.add_texts(texts[2000:], ids=ids[2000:])
single = HipFlatVectorStore(emb, device=0)
single.add_texts(texts, ids=ids)
gone = ["r0", "r1499", "r1500", "r3000"] + [f"r{i}" for i in range(900, 1100)] + [f"r{i}" for i in range(2500, 2600)]
assert sharded.delete(gone + ["missing"]) is False and sharded.ntotal == 3001
assert sharded.delete(gone) is True and single.delete(gone) is True
assert sharded.ntotal == single.ntotal == 3001 - len(gone)
assert sum(c for _, c in sharded.index._blocks) == sharded.index.local.ntotal
queries = [texts[5], texts[1000], texts[1501], texts[2999], "record 2550 in group 9"]
for q in queries:
    a = [(d.id, s) for d, s in sharded.similarity_search_with_score(q, k=15)]
    b = [(d.id, s) for d, s in single.similarity_search_with_score(q, k=15)]
    assert a == b, (rank, q, a[:3], b[:3])
sharded.add_texts(["a new record"], ids=["new"])
single.add_texts(["a new record"], ids=["new"])
assert sharded.similarity_search("a new record", k=1)[0].id == "new"
assert [d.id for d in sharded.similarity_search(texts[2000], k=20)] == [d.id for d in single.similarity_search(texts[2000], k=20)]
assert sharded.delete(["new", "r2"]) is True and single.delete(["new", "r2"]) is True
assert [d.id for d in sharded.similarity_search(texts[3], k=20)] == [d.id for d in single.similarity_search(texts[3], k=20)]
dist.barrier()
if rank == 0:
    print("DELETE_OK")
dist.destroy_process_group()
