"""Save / load of a corpus-scale shard in a FRESH process, with the host memory high-water mark taken from /proc.

usage: python tools/persist_probe.py --rows 10000000 --dim 768 --storage f16 --dir /tmp/x [--threads 8] [--no-direct]
Prints one JSON line: rows, bytes, save / load seconds and GB/s (native transfer phase and wall), VmHWM before / after
(kB), whether the search after the load equals the search before it bit for bit, and the exhaustive on-device check of a
query sample against the loaded rows (rarc_verify_batch: rows beating a k-th entry, answer entries that are not exact
pairs).  Used by tests/test_gpu_persistence.py and for the figures in DESIGN.md.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def vm(field: str) -> int:
    with open("/proc/self/status") as fh:
        for line in fh:
            if line.startswith(field + ":"):
                return int(line.split()[1])
    return -1


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--storage", default="f16")
    ap.add_argument("--dir", required=True)
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--no-direct", action="store_true", help="save through the page cache instead of O_DIRECT")
    ap.add_argument("--direct-load", action="store_true", help="load with O_DIRECT reads (default: buffered)")
    ap.add_argument("--nq", type=int, default=256)
    ap.add_argument("--k", type=int, default=100)
    ap.add_argument("--verify", type=int, default=16, help="queries re-scanned exhaustively after the load")
    ap.add_argument("--keep", action="store_true")
    ap.add_argument("--cold", action="store_true", help="drop the file's pages from the page cache before loading (posix_fadvise)")
    a = ap.parse_args()

    import torch

    import bench
    from rag_arc_amd.hip import binding as B
    from rag_arc_amd.hip.engine import FlatIndexF16

    lib = B.load_library()
    torch.cuda.init()
    torch.zeros(1, device="cuda")
    out = dict(rows=a.rows, dim=a.dim, storage=a.storage, threads=a.threads or FlatIndexF16.IO_THREADS,
               save_direct=not a.no_direct, load_direct=a.direct_load, cold=a.cold, rss_kb_start=vm("VmRSS"))
    idx = bench.build_index(torch, lib, B, FlatIndexF16, 0, a.dim, 0, a.rows, storage=a.storage)
    q = torch.empty((a.nq, a.dim), dtype=torch.float32, device="cuda")
    B.check(lib.rarc_synth_rows_f32(q.data_ptr(), a.dim, a.dim, 0, a.nq, 4321, 0))
    I0, D0 = idx.search_device(q, a.k)
    torch.cuda.synchronize()
    out["rss_kb_before_save"] = vm("VmRSS")
    path = os.path.join(a.dir, "probe.rarc")
    os.makedirs(a.dir, exist_ok=True)
    t0 = time.perf_counter()
    st = idx.save_shard(path, threads=a.threads or None, direct=not a.no_direct)
    out["save_wall_s"] = time.perf_counter() - t0
    out["save"] = st
    out["file_bytes"] = os.path.getsize(path)
    out["hwm_kb_after_save"] = vm("VmHWM")
    max_norm = idx.max_norm
    del idx
    torch.cuda.empty_cache()
    if a.cold:
        fd = os.open(path, os.O_RDONLY)
        os.fsync(fd)
        os.posix_fadvise(fd, 0, 0, os.POSIX_FADV_DONTNEED)
        os.close(fd)
    idx2 = FlatIndexF16(a.dim, metric="cosine", device=0, storage=a.storage)
    t0 = time.perf_counter()
    st = idx2.load_shard(path, threads=a.threads or None, direct=a.direct_load)
    torch.cuda.synchronize()
    out["load_wall_s"] = time.perf_counter() - t0
    out["load"] = st
    out["hwm_kb_after_load"] = vm("VmHWM")
    out["max_norm_kept"] = bool(abs(idx2.max_norm - max_norm) < 1e-6)
    I1, D1 = idx2.search_device(q, a.k)
    out["search_identical"] = bool(torch.equal(I0, I1) and torch.equal(D0.view(torch.int32), D1.view(torch.int32)))
    if a.verify:
        step = max(1, a.nq // a.verify)
        beating, wrong = idx2.verify_batch(q, I1, D1, which=range(0, a.nq, step), detail=True)
        out["verify_queries"], out["rows_beating_kth"], out["inexact_pairs"] = len(range(0, a.nq, step)), beating, wrong
    # spot sample for the caller's oracle check: the first answers' ids and score bits
    out["sample_ids"] = I1[:4, :8].cpu().tolist()
    out["sample_score_bits"] = D1[:4, :8].view(torch.int32).cpu().tolist()
    out["hwm_kb_end"] = vm("VmHWM")
    if not a.keep:
        os.unlink(path)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
