#!/usr/bin/env python3
"""bench.py — queries/sec of the dense-retrieval hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--rows R] [--dim D] [--batch B] [--k K]

`--gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N ranks itself (a child
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py ...`,
spawned BEFORE this process touches the GPU) and relays rank 0's JSON line; launched under torchrun
(WORLD_SIZE set) it is one of the ranks.

One "step" = one batch of B queries through the whole path with everything resident in HBM:
prep_queries (L2-normalise + fp16/int8 copies) -> fused MFMA scan + pruning -> canonical fp32 rescore +
sort -> [N>1: pack, ONE RCCL all-gather of (id, score), merge].  The corpus is FIXED (strong
scaling): `--rows` rows of dimension `--dim`, N(0,1) directions, row-sharded over the N ranks.  Default
workload = BASELINE.json config 4's corpus (100M x 768 fp16) when it fits the ranks' HBM, otherwise the
largest power-of-ten row count that does.  The same line also carries, as objects of their own:
  c2  config 2 (1M x 768, one GPU)                                   [N = 1]
  c3  config 3 end to end: c2 + the cross-encoder's LM forward on 256 x 100 prompts + score->order + RRF   [N = 1]
  c5  config 5 end to end: bge-large encoder forward (24 layers, seeded weights) -> fp8 100M x 1024
      sharded scan -> RRF with the supplied lexical list             [every N]
  cpu_baseline  the CPU oracle timed on the host cores               [N = 1]
  wide  10M x 1536 (OpenAI-sized embeddings) through the wide path: chunked score GEMM + select + canonical finalize   [N = 1]
  f32  storage="f32" (the reference's row format): 10M x 768, the only mode inside 1e-5 on arbitrary embeddings   [N = 1]
  api  the same indexes reached THROUGH the plugin surface: retriever.batch_invoke / invoke, texts -> Documents   [N = 1]
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 achievable
MFMA_F16_PEAK_TF = 2500.0  # dense fp16/bf16 MFMA peak (same guide)
HBM_MEASURED = {"read_GBps": None, "copy_GBps": None}   # this box's ceilings, measured in this process (measure_hbm_peak)


def measure_hbm_peak(torch, lib, B, dev, buf_bytes=8 << 30):
    """report against nominal AND measured.  Two figures of THIS box, in THIS process, timed with events on the
    current stream: (a) rarc_stream_read — a read-only sweep of 8 GiB by one persistent 512-thread workgroup per CU, eight 16-byte
    loads in flight per lane: the ceiling of a kernel shaped like the scan; (b) a device-to-device copy of 4 GiB (read + write
    bytes counted), the guide's 6.3 TB/s figure.  peak_measured = (a)."""
    free = torch.cuda.mem_get_info(dev)[0]
    buf_bytes = int(min(buf_bytes, max(1 << 28, free // 4)))
    buf = torch.empty(buf_bytes, dtype=torch.uint8, device=dev)
    buf.zero_()
    sink = torch.zeros(1, dtype=torch.int64, device=dev)
    st = torch.cuda.current_stream(dev).cuda_stream

    def best_of(fn, n=5):
        out = []
        for _ in range(n):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            e1.synchronize()
            out.append(e0.elapsed_time(e1))
        return min(out[1:])          # (the first pass pages the buffer in)

    ms_r = best_of(lambda: B.check(lib.rarc_stream_read(buf.data_ptr(), buf_bytes, sink.data_ptr(), st), "rarc_stream_read"))
    half = buf_bytes // 2
    ms_c = best_of(lambda: buf[half:2 * half].copy_(buf[:half]))
    HBM_MEASURED["read_GBps"] = round(buf_bytes / (ms_r * 1e-3) / 1e9, 1)
    HBM_MEASURED["copy_GBps"] = round(2 * half / (ms_c * 1e-3) / 1e9, 1)
    del buf
    torch.cuda.empty_cache()
    return dict(HBM_MEASURED, bytes=buf_bytes, how="rarc_stream_read over 8 GiB (best of 4 after a warm-up pass); copy = torch "
                "device-to-device copy of half of it, read + write bytes")


def hbm_roofline(achieved_GBps, **extra):
    """An HBM roofline object: nominal peak and the peak measured on this box (read-only stream), a fraction of each."""
    out = {"bound": "hbm"}
    out.update(extra)
    out.update({"achieved": round(achieved_GBps, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved_GBps / HBM_PEAK_GBS, 4)})
    if HBM_MEASURED["read_GBps"]:
        out["peak_measured"] = HBM_MEASURED["read_GBps"]
        out["frac_of_measured"] = round(achieved_GBps / HBM_MEASURED["read_GBps"], 4)
    return out


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rows", type=int, default=0, help="total corpus rows (0 = auto: 100M if it fits)")
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--k", type=int, default=100)
    ap.add_argument("--storage", choices=("f16", "f8", "f32"), default="f16",
                    help="row storage: fp16 (default), fp8 e4m3fn + per-row scale (BASELINE config 5) or fp32 (the reference's own row format: the one mode inside 1e-5 on arbitrary embeddings)")
    ap.add_argument("--shadow", action="store_true",
                    help="keep the int8 image of the fp16 rows (+50%% HBM): the prefilter scan reads it instead")
    ap.add_argument("--scan", choices=("auto", "q8", "mfma16"), default="auto",
                    help="scan kernel (auto: the engine's choice by shard size)")
    ap.add_argument("--twin", action="store_true",
                    help="alternate the batches between the index and a twin() search context on a side stream: the small "
                         "kernels either side of one batch's scan (prep, seed, finalize, exchange) run under the other's")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-c2", action="store_true")
    ap.add_argument("--no-c3", action="store_true")
    ap.add_argument("--no-c5", action="store_true")
    ap.add_argument("--no-wide", action="store_true", help="skip the wide-row leg (10M x 1536 through rarc_search_wide)")
    ap.add_argument("--no-pairs", action="store_true", help="skip the all-pairs cosine leg (100k x 1024 entities, SURVEY 8(f) rank 4)")
    ap.add_argument("--no-f32", action="store_true", help="skip the storage=f32 leg (fp32 rows at the largest size one GPU holds)")
    ap.add_argument("--f32-rows", type=int, default=0, help="rows of the storage=f32 leg (0 = auto: 45M x 768 when 207 GB are free)")
    ap.add_argument("--no-api", action="store_true", help="skip the legs through the registered retriever (texts -> Documents)")
    ap.add_argument("--no-lm", action="store_true", help="config 3 without its cross-encoder's LM forward (seeded logits instead)")
    ap.add_argument("--c3-steps", type=int, default=2, help="timed steps of the config-3 leg (one step = 256 x 100 LM prompts, seconds)")
    ap.add_argument("--c3-chunk", type=int, default=640, help="(query, document) prompts per LM call in the config-3 leg")
    ap.add_argument("--c3-no-prefix-sharing", action="store_true",
                    help="config 3: run every (query, document) prompt whole (the reference's way) instead of the shared part once per query")
    ap.add_argument("--c5-rows", type=int, default=0, help="rows of the config-5 corpus (0 = auto: 100M if it fits)")
    ap.add_argument("--c5-layers", type=int, default=24, help="encoder depth of the config-5 leg (bge-large: 24)")
    ap.add_argument("--encoder-precision", choices=("fp32", "fp16"), default="fp32",
                    help="config-5 encoder arithmetic: fp32 = the reference's (SentenceTransformer default; split-operand MFMA "
                         "GEMMs, ids equal to an fp32 forward's), fp16 = the 1e-3-class fast forward")
    ap.add_argument("--encoder-overlap", action="store_true",
                    help="config 5: embed batch i+1 on a side stream under the scan of batch i (measured: no gain on one GPU — "
                         "the persistent scan kernel holds every CU's whole register file, so the forward's kernels wait for it)")
    ap.add_argument("--c5-split", choices=("slice", "rotate"), default="rotate",
                    help="config 5 on N > 1 ranks, who runs the query encoder: 'slice' = every rank embeds batch/N queries of "
                         "every batch (one all-gather); 'rotate' = rank (i mod N) embeds ALL the queries of batch i and broadcasts "
                         "them (a 256-query forward every N batches instead of a latency-bound 32-query one every batch: measured faster at "
                         "every N — one rank of 8: 5.74 vs 7.63 ms per step — and the default)")
    ap.add_argument("--emulate-world", type=int, default=0,
                    help="config 5 on ONE GPU: give the encoder the share one rank of this many would carry (the other queries' "
                         "embeddings are taken as already received) — one rank's step of an N-rank run, collectives excluded")
    ap.add_argument("--no-c5-alt", action="store_true", help="config 5: skip the second timed loop in the other encoder precision")
    ap.add_argument("--no-persist", action="store_true", help="skip the shard-file leg (save / load of a 10M-row shard)")
    ap.add_argument("--persist-rows", type=int, default=10_000_000)
    ap.add_argument("--persist-dir", default="", help="where the shard file goes (default: a temporary directory)")
    ap.add_argument("--no-ingest", action="store_true", help="skip the ingest leg (texts -> tokeniser -> encoder -> stored rows)")
    ap.add_argument("--ingest-docs", type=int, default=0, help="documents per ingest configuration (0 = sized for ~1-2 s each)")
    ap.add_argument("--verify-queries", type=int, default=256,
                    help="queries whose answer is re-checked by an exact canonical re-scan of the whole shard (eight queries "
                         "per pass over the rows: the default checks the whole batch, ~2 s at 100M rows)")
    ap.add_argument("--backend", choices=("nccl", "gloo"), default="nccl",
                    help="collective backend of the ranks: nccl (= RCCL, the measured configuration) or gloo (rehearsals)")
    ap.add_argument("--one-device", action="store_true",
                    help="rehearsal on a one-GPU box: every rank uses cuda:0 (needs --backend gloo: RCCL refuses two ranks "
                         "on one device).  The N-rank code path runs for real, on real kernels; the timings mean nothing")
    ap.add_argument("--dry-run", action="store_true",
                    help="CPU rehearsal of the multi-rank plumbing (gloo): shard ranges, the (id, score) all-gather, the "
                         "merge and the max-over-ranks timing, with a stand-in local search; no GPU, no kernel")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------ launcher
def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def count_gpus_sysfs():
    """GPUs of this node per the amdkfd topology (nodes with simd_count > 0), honouring HIP/ROCR_VISIBLE_DEVICES as a count;
    None when the topology is not readable."""
    base = "/sys/class/kfd/kfd/topology/nodes"
    try:
        n = 0
        for node in os.listdir(base):
            with open(os.path.join(base, node, "properties")) as fh:
                props = dict(ln.split()[:2] for ln in fh if len(ln.split()) >= 2)
            n += int(props.get("simd_count", "0")) > 0
    except OSError:
        return None
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None and v.strip() != "":
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def launch_ranks(a, argv) -> int:
    """Start a.gpus ranks of this script under torch.distributed.run and relay rank 0's JSON line.
    The parent never touches the GPU runtime: the devices are counted from the kernel driver's topology in sysfs
    (/sys/class/kfd: one node per agent, GPUs are those with SIMDs), not through torch / HIP — a device count through the
    runtime can bring the runtime up in this process.  When sysfs cannot tell, the ranks themselves report a
    missing device."""
    if a.one_device and a.backend != "gloo":
        print("bench.py: --one-device needs --backend gloo (RCCL refuses two ranks on one device)", file=sys.stderr)
        return 2
    if not a.dry_run and not a.one_device:
        n_dev = count_gpus_sysfs()
        if n_dev is not None and n_dev < a.gpus:
            print(f"bench.py: --gpus {a.gpus} but only {n_dev} GPU(s) visible on this node", file=sys.stderr)
            return 2
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC (RCCL across processes on this host driver)
    env.setdefault("OMP_NUM_THREADS", "1")
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env)
    line = None
    for out in proc.stdout:
        out = out.rstrip("\n")
        if out.startswith('{"metric"'):
            line = out
        elif out:
            print(out, file=sys.stderr)
    rc = proc.wait()
    if line is not None:
        print(line, flush=True)
    if rc == 0 and line is None:
        print("bench.py: the ranks exited cleanly but rank 0 printed no result line", file=sys.stderr)
        return 3
    return rc


def dry_run(a) -> None:
    """The N-rank control flow on CPU (gloo): what SCALE exercises, minus the kernels."""
    import numpy as np
    import torch
    import torch.distributed as dist

    from rag_arc_amd.hip.sharded import ShardedFlatSearch, shard_range

    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rows = a.rows or 100_000_000
    lo, hi = shard_range(rows, rank, world)

    class _StandIn:  # local "search": row r of the shard scores 1 - r/rows for every query (best rows first)
        def search_device(self, q, k):
            ids = torch.arange(lo, lo + k, dtype=torch.int64).repeat(q.shape[0], 1)
            ids = torch.where(ids < hi, ids, torch.full_like(ids, -1))
            sc = torch.where(ids >= 0, 1.0 - ids.double() / rows, torch.tensor(float("-inf"), dtype=torch.float64))
            return ids, sc.float()

        def search_async(self, q, k):
            return self

        def result(self):
            return self.search_device(self._q, a.k)

    def merge(ids, scores, k):  # (score desc, id asc) over the gathered lists
        G, nq, kk = ids.shape
        i2, s2 = ids.permute(1, 0, 2).reshape(nq, G * kk), scores.permute(1, 0, 2).reshape(nq, G * kk)
        key = np.lexsort((i2.numpy(), -s2.numpy().astype(np.float64)), axis=1)[:, :k]
        key = torch.from_numpy(key)
        return torch.gather(i2, 1, key), torch.gather(s2, 1, key)

    local = _StandIn()
    local._q = torch.zeros((a.batch, 8))
    s = ShardedFlatSearch(local, merge_fn=merge)
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        ids, sc = s.search_device(local._q, a.k)
    dist.barrier()
    dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    # every rank's shard size, gathered the way a real run's ranks would report them
    sizes = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([hi - lo], dtype=torch.int64))
    if rank == 0:
        want = torch.arange(0, a.k, dtype=torch.int64)   # the global best k rows are rows 0..k-1 (all on shard 0)
        ms = float(dt.item()) / max(1, a.steps) * 1e3
        # the SAME keys a measured line carries (main() below), so that whatever parses SCALE_rNN.json can be rehearsed on it;
        # value / roofline are null: nothing was measured
        print(json.dumps({"metric": "queries/sec at fixed (N_corpus, d), exact top-k (ids bit-exact vs CPU oracle)",
                          "value": None, "unit": "queries/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
                          "ms_per_step": round(ms, 4), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
                          "dtype": a.storage, "data": "none (dry run: stand-in local search, gloo, no GPU, no kernel)",
                          "dry_run": True, "ranks": dist.get_world_size(), "backend": "gloo", "rccl_ranks": 0,
                          "rows_per_gpu": hi - lo, "merged_ids_ok": bool((ids == want).all()),
                          "config": {"workload": f"DRY RUN of {rows}x{a.dim} row-sharded over {world} ranks, batch {a.batch}, top-{a.k}",
                                     "n_corpus": rows, "d": a.dim, "batch": a.batch, "k": a.k, "rows_per_gpu": hi - lo,
                                     "rows_per_rank": [int(x.item()) for x in sizes],
                                     "exchange_ms_per_step": round(ms, 4),
                                     "exchange": "one gloo all-gather of (id, score) + merge (the whole step of a dry run)"},
                          "roofline": {"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None,
                                       "traffic": None},
                          "cpu_baseline": None}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


# ------------------------------------------------------------------------------------------------ pieces
def build_index(torch, lib, B, FlatIndexF16, dev_index, dim, lo, hi, seed=1234, scan="auto", storage="f16", shadow=False):
    """HBM-resident shard holding global rows [lo, hi) of the synthetic corpus."""
    n = hi - lo
    if storage in ("f8", "f32"):  # synthetic fp32 rows -> ingest kernel (normalise; e4m3fn + row scale, or fp32 + fp16 image), in slabs
        idx = FlatIndexF16(dim, metric="cosine", device=dev_index, id_base=lo, storage=storage, capacity=n)
        slab = 1 << 20
        buf = torch.empty((min(slab, max(n, 1)), dim), dtype=torch.float32, device=torch.device("cuda", dev_index))
        for s0 in range(0, n, slab):
            m = min(slab, n - s0)
            B.check(lib.rarc_synth_rows_f32(buf.data_ptr(), dim, dim, lo + s0, m, seed, 0), "rarc_synth_rows_f32")
            idx.add(buf[:m])
        del buf
        return idx
    d_pad = B.padded_dim(dim)
    cap = ((n + 31) // 32) * 32
    rows = torch.empty((max(cap, 32), d_pad), dtype=torch.float16, device=torch.device("cuda", dev_index))
    B.check(lib.rarc_synth_rows_f16(rows.data_ptr(), d_pad, dim, lo, n, seed, 0), "rarc_synth_rows_f16")
    if cap > n:
        rows[n:].zero_()
    idx = FlatIndexF16(dim, metric="cosine", device=dev_index, id_base=lo, scan=scan, shadow=shadow)
    idx.add_rows_f16(rows, 1.001, n_valid=n)  # adopts the buffer (no copy); unit rows rounded to fp16
    return idx


def timed_loop(torch, dist, step_begin, step_end, steps, warmup, use_dist):
    """W untimed + K timed steps, two in flight (step i+1 is enqueued before step i is collected, so the host side
    of one step overlaps the kernels of the next); barrier + synchronize on both sides; max over ranks."""
    last = None
    for _ in range(warmup):
        last = step_end(step_begin())
    torch.cuda.synchronize()
    if steps <= 0:
        return 0.0, last
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pending = None
    for _ in range(steps):
        nxt = step_begin()
        if pending is not None:
            last = step_end(pending)
        pending = nxt
    last = step_end(pending)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt, last


def scan_profile(lib, B, ctypes, fn, max_launches):
    """Run fn() with the library's HIP-event brackets around every scan launch; returns (result, total ms, launches)."""
    B.check(lib.rarc_profile_begin(max_launches), "rarc_profile_begin")
    out = fn()
    tot_ms, n_l = ctypes.c_double(0), ctypes.c_int(0)
    B.check(lib.rarc_profile_end(ctypes.byref(tot_ms), ctypes.byref(n_l)), "rarc_profile_end")
    return out, tot_ms.value, n_l.value


def recorded_traffic(kernel, rows_per_launch, dim, storage="f16"):
    """HBM bytes per scan from the committed PMC pass of the same shape (profiles/traffic_r06.json, falling back to the earlier rounds' files), or None."""
    for name in ("traffic_r06.json", "traffic_r05.json", "traffic_r04.json", "traffic_r03.json", "traffic_r02.json", "traffic_r01.json"):
        path = os.path.join(ROOT, "profiles", name)
        if not os.path.exists(path):
            continue
        try:
            for ent in json.load(open(path)).get("entries", []):
                if (ent.get("rows_per_launch") == rows_per_launch and ent.get("dim") == dim
                        and ent.get("storage", "f16") == storage
                        and ent.get("kernel", "rarc_scan_f16_kernel") == kernel):
                    return ent.get("hbm_bytes_per_launch"), f"recorded PMC pass (profiles/{name}), not measured in this run"
        except Exception:
            pass
    return None, None


def lexical_lists(torch, ids, n_corpus, k, seed=777):
    """The "supplied BM25 rank list" (SURVEY §8d): per query a seeded list of k ids, ~30 % of them dense hits."""
    g = torch.Generator(device=ids.device)
    g.manual_seed(seed)
    nq = ids.shape[0]
    n_over = (3 * k) // 10
    pick = torch.argsort(torch.rand((nq, ids.shape[1]), generator=g, device=ids.device), dim=1)[:, :n_over]
    over = torch.gather(ids, 1, pick)
    # k - n_over other ids: distinct within the query by construction (a random odd stride through the id space)
    start = torch.randint(0, n_corpus, (nq, 1), generator=g, device=ids.device)
    rest = (start + torch.arange(k - n_over, device=ids.device)[None, :] * 7919) % n_corpus
    lex = torch.cat([over, rest], dim=1)
    perm = torch.argsort(torch.rand((nq, k), generator=g, device=ids.device), dim=1)
    return torch.gather(lex, 1, perm).contiguous()


# ------------------------------------------------------------------------------------------------ main
def main():
    argv = sys.argv[1:]
    a = parse(argv)
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(a, argv))
    if a.dry_run:
        return dry_run(a)

    import ctypes

    # stdout carries ONE line, the result.  Libraries print there too (RCCL writes a version banner to stdout when
    # its communicator goes up or down): from here on file descriptor 1 is stderr, the result goes to the real one.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = 0 if a.one_device else int(os.environ.get("LOCAL_RANK", "0"))
    use_dist = world > 1 or os.environ.get("RARC_FORCE_DIST") == "1"   # the env var exercises RCCL with one rank
    if torch.cuda.device_count() <= local_rank:    # (a rank is its own process: counting devices here is this rank's business)
        print(f"bench.py: --gpus {a.gpus} but only {torch.cuda.device_count()} GPU(s) visible on this node", file=sys.stderr)
        sys.exit(2)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if a.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    if a.gpus != world and rank == 0:
        print(f"# note: --gpus {a.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from rag_arc_amd.hip import binding as B
    from rag_arc_amd.hip.engine import FlatIndexF16
    from rag_arc_amd.hip.sharded import ShardedFlatSearch, shard_range, split_range

    lib = B.load_library()
    measure_hbm_peak(torch, lib, B, dev)      # (every rank: their work stays symmetric; HBM_MEASURED feeds every hbm roofline below)
    d_pad = B.padded_dim(a.dim, 256 if a.storage == "f8" else 128)
    esize = 1 if a.storage == "f8" else 2               # bytes per element the SCAN streams (fp32 rows: their fp16 image)
    held = 6 if a.storage == "f32" else esize            # bytes per element the index holds in HBM
    rows = a.rows
    if rows <= 0:  # auto: config 4's corpus if every rank's shard (+ slack) fits its HBM
        free = torch.cuda.mem_get_info(dev)[0]
        rows = 100_000_000
        while rows > 1_000_000 and (rows / world) * d_pad * (held + (1 if a.shadow else 0)) > 0.85 * free:
            rows //= 10
    lo, hi = shard_range(rows, rank, world)
    idx = build_index(torch, lib, B, FlatIndexF16, local_rank, a.dim, lo, hi, scan=a.scan, storage=a.storage,
                      shadow=a.shadow)
    searcher = ShardedFlatSearch(idx, force_collective=use_dist)
    if a.twin:
        class _Alternating:
            """search_async goes to the index and its twin in turn (two search contexts, two streams)."""

            def __init__(self, contexts):
                self.contexts, self.i = contexts, 0

            def search_async(self, queries, k):
                self.i += 1
                return self.contexts[self.i % len(self.contexts)].search_async(queries, k)

            def __getattr__(self, name):
                return getattr(self.contexts[0], name)

        searcher = ShardedFlatSearch(_Alternating([idx, idx.twin()]), force_collective=use_dist)
    q = torch.empty((a.batch, a.dim), dtype=torch.float32, device=dev)
    B.check(lib.rarc_synth_rows_f32(q.data_ptr(), a.dim, a.dim, 0, a.batch, 4321, 0), "rarc_synth_rows_f32")
    torch.cuda.synchronize()
    passes_per_step = (a.batch + 255) // 256

    # ---- timed region: K steps, every scan launch bracketed by its own HIP events ---------------
    timed_loop(torch, dist, lambda: searcher.search_async(q, a.k), lambda h: searcher.finish(h, a.k), 0, a.warmup, use_dist)
    searcher.measure_exchange(True)
    (dt, (ids, scores)), tot_ms, n_l = scan_profile(
        lib, B, ctypes,
        lambda: timed_loop(torch, dist, lambda: searcher.search_async(q, a.k), lambda h: searcher.finish(h, a.k),
                           a.steps, 0, use_dist),
        8 * a.steps * passes_per_step + 16)
    exch_ms = searcher.exchange_ms() / max(1, a.steps)
    searcher.measure_exchange(False)
    scan_ms = tot_ms / max(1, n_l)
    # algorithmic bytes of one scan launch on this rank (shadow mode: the scan reads the int8 image).  A large
    # shard is scanned in up to four launches of the same kernel (stretches ending at 1/512, 1/64, 1/8 of the shard
    # and at its end, an exact tightening pass between them): per-launch figures are averages over all launches,
    # like the AverageNs of the kernel in the rocprofv3 CSV; scan_ms_per_pass is their sum for one whole scan
    passes = a.steps * passes_per_step
    launches_per_pass = max(1, round(n_l / max(1, passes)))
    shard_bytes = (hi - lo) * d_pad * (1 if a.shadow else esize) / launches_per_pass
    flagged = len(getattr(idx, "last_repaired", []))
    # full-size exactness property on this rank's shard: the exact repair scan (canonical fp32 scores of EVERY row)
    # must find no row beating the returned k-th entry (local results, before the cross-shard merge); and, on fp16
    # rows, the fp16-MFMA scan path (a different kernel with a different bound) must return the same ids
    l_ids, l_sc = idx.search_device(q, a.k)
    nver = max(0, min(a.verify_queries, a.batch))
    vq = sorted(set(np.linspace(0, a.batch - 1, nver).astype(int).tolist())) if nver else []
    beat = idx.verify_batch(q, l_ids, l_sc, vq) if vq else 0
    check = {"queries_verified_by_exact_rescan": len(vq), "rows_beating_kth": int(beat)}
    if a.storage == "f16" and not a.shadow and d_pad <= 768 and idx._use_q8(a.k):
        idx.scan = "mfma16"
        m_ids, m_sc = idx.search_device(q, a.k)
        idx.scan = a.scan
        check["queries_cross_checked_vs_fp16_mfma_scan"] = a.batch
        check["queries_differing"] = int((~((m_ids == l_ids).all(dim=1) & (m_sc == l_sc).all(dim=1))).sum().item())
    if use_dist:   # every rank's own shard check, summed
        t = torch.tensor([check["rows_beating_kth"], check.get("queries_differing", 0)], dtype=torch.int64, device=dev)
        dist.all_reduce(t)
        check["rows_beating_kth"] = int(t[0].item())
        if "queries_differing" in check:
            check["queries_differing"] = int(t[1].item())
        check["ranks_checked"] = world

    result = None
    kname = "rarc_scan_q8_kernel" if idx._use_q8(a.k) else "rarc_scan_f16_kernel"
    if rank == 0:
        qps = a.batch * a.steps / dt
        ach = shard_bytes / (scan_ms * 1e-3) / 1e9
        traffic, tsrc = recorded_traffic(kname, hi - lo, a.dim, a.storage)
        if traffic:
            traffic = int(traffic / launches_per_pass)
        store_txt = {"f8": "fp8 (e4m3fn + row scale)", "f32": "fp32 (+ fp16 image for the scan)"}.get(a.storage, "fp16")
        result = {
            "metric": "queries/sec at fixed (N_corpus, d), exact top-k (ids bit-exact vs CPU oracle)",
            "value": round(qps, 1), "unit": "queries/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(dt / a.steps * 1e3, 4), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": a.storage, "data": "synthetic",
            "rccl_ranks": dist.get_world_size() if (use_dist and a.backend == "nccl") else 0,
            **({"rehearsal": f"{world} ranks over {a.backend}" + (" sharing cuda:0: timings are not a measurement" if a.one_device else "")}
               if (use_dist and (a.backend != "nccl" or a.one_device)) else {}),
            "config": {"workload": f"{rows}x{a.dim} {store_txt} corpus of N(0,1) directions resident in HBM, row-sharded "
                                   f"over {world} GPU(s), batch {a.batch} queries, cosine top-{a.k}, exact (canonical fp32 rescore)",
                       "n_corpus": rows, "d": a.dim, "batch": a.batch, "k": a.k, "rows_per_gpu": hi - lo,
                       "generator": "counter hash -> 52-bit uniform -> inverse normal CDF (AS 241), rows L2-normalised",
                       "int8_shadow_image": bool(a.shadow),
                       "exchange_ms_per_step": round(exch_ms, 4) if use_dist else 0.0,
                       "exchange": (f"pack + one {'RCCL' if a.backend == 'nccl' else a.backend} all-gather of (id, score) + merge") if use_dist else "none (one shard)",
                       "repaired_queries_last_step": flagged,
                       "full_size_check": check},
            "roofline": hbm_roofline(ach, traffic=traffic, traffic_source=tsrc, kernel=kname, avg_launch_ms=round(scan_ms, 4),
                                     algorithmic_bytes_per_launch=int(shard_bytes), launches_timed=n_l,
                                     launches_per_scan=launches_per_pass, scan_ms_per_pass=round(scan_ms * launches_per_pass, 4),
                                     hbm_measured=dict(HBM_MEASURED)),
        }

    # ---- config 2 / 3 (1M x 768, one GPU) and the CPU baseline on the same sample -----------------
    if world == 1 and a.storage == "f16" and not (a.no_c2 and a.no_c3):
        n2 = min(1_000_000, rows)
        idx2 = idx if rows == n2 else build_index(torch, lib, B, FlatIndexF16, local_rank, a.dim, 0, n2)
        steps2 = max(a.steps, 400)      # (0.45 ms each: a fifth of a second; short runs of this leg vary by 5 % with the clocks the
        w2 = max(a.warmup, 20)          #  100M-row scan in front of it left behind)
        (dt2, (ids2, sc2)), tot2, nl2 = scan_profile(
            lib, B, ctypes,
            lambda: timed_loop(torch, dist, lambda: idx2.search_async(q, a.k), lambda h: h.result(), steps2, w2, False),
            8 * (steps2 + w2) * passes_per_step + 16)
        scan2 = tot2 / max(1, steps2 + w2)   # per scan pass (the warm-up passes are inside the profiling window too)
        k2 = "rarc_scan_q8_kernel" if idx2._use_q8(a.k) else "rarc_scan_f16_kernel"
        bytes2 = n2 * B.padded_dim(a.dim) * 2
        if not a.no_c2:
            result["c2"] = {"workload": f"{n2}x{a.dim} fp16, 1 GPU, batch {a.batch}, top-{a.k}",
                            "value": round(a.batch * steps2 / dt2, 1), "unit": "queries/s",
                            "ms_per_step": round(dt2 / steps2 * 1e3, 4), "scan_ms": round(scan2, 4),
                            "scan_GBps": round(bytes2 / (scan2 * 1e-3) / 1e9, 1),
                            "pipeline": "two search contexts on two streams (engine.py _pipeline_context): a batch's prep / seed run under the previous batch's finalize",
                            "roofline": hbm_roofline(bytes2 / (scan2 * 1e-3) / 1e9, kernel=k2,
                                                     mfma_TFLOPs=round(2.0 * a.batch * n2 * a.dim / (scan2 * 1e-3) / 1e12, 1),
                                                     end_to_end_frac=round(bytes2 / (dt2 / steps2) / 1e9 / HBM_PEAK_GBS, 4))}
        if not a.no_c3:
            result["c3"] = leg_c3(torch, dist, lib, B, ctypes, np, idx2, q, ids2, n2, a, steps2, w2, passes_per_step, bytes2, k2,
                                  dev, local_rank)
            torch.cuda.empty_cache()
        if not a.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(np, idx2, q, ids2, sc2, n2, a)
        if not a.no_api:      # the plugin surface over config 2's index, via the registry (saved index + JSON config)
            eng2 = {"value": round(a.batch * steps2 / dt2, 1), "unit": "queries/s", "ms_per_batch": round(dt2 / steps2 * 1e3, 4)}
            result["api"] = {"c2": leg_api(torch, np, lib, B, a, dev, local_rank, idx2, n2, eng2, via_registry=True)}
            if idx2 is not idx:     # ... and over the headline index (100M rows: columnar docstore, adopted engine)
                eng = {"value": result["value"], "unit": "queries/s", "ms_per_batch": result["ms_per_step"]}
                result["api"]["headline"] = leg_api(torch, np, lib, B, a, dev, local_rank, idx, hi - lo, eng, via_registry=False)
        if idx2 is not idx:
            del idx2

    # ---- rows wider than 1024 dimensions: 10M x 1536 through the wide path (one GPU) ----------------------
    if world == 1 and a.storage == "f16" and not a.no_wide:
        if torch.cuda.mem_get_info(dev)[0] > 45 * (1 << 30):
            result["wide"] = leg_wide(torch, lib, B, FlatIndexF16, a, dev, local_rank)
        else:
            result["wide"] = {"skipped": "needs 45 GiB of free HBM"}
    # ---- all pairs with cosine >= 0.95 among 100k entity embeddings (the graph store's dedup step), one GPU --------
    if world == 1 and a.storage == "f16" and not a.no_pairs:
        result["pairs"] = leg_pairs(torch, np, lib, B, a, dev)
    # ---- config 5 end to end (every N): encoder forward -> fp8 sharded scan -> RRF ----------------
    if not a.no_c5 and a.storage == "f16":
        del searcher, idx, l_ids, l_sc
        import gc

        gc.collect()
        torch.cuda.empty_cache()
        c5 = leg_c5(torch, dist, lib, B, ctypes, np, FlatIndexF16, ShardedFlatSearch, shard_range, split_range, a,
                    world, rank, local_rank, dev, use_dist)
        if rank == 0:
            result["c5"] = c5
    # ---- storage="f32": the reference's own row format AT THE SCALE ONE GPU HOLDS (one GPU; after the headline index and
    #      config 5's corpus are gone): fp32 rows + their fp16 image = 6 bytes per element — 45M x 768 = 207 GB of the 288 ----
    if world == 1 and a.storage == "f16" and not a.no_f32:
        import gc

        searcher = idx = l_ids = l_sc = None
        gc.collect()
        torch.cuda.empty_cache()
        free = torch.cuda.mem_get_info(dev)[0]
        per_row = B.padded_dim(a.dim) * 6 + 64
        rows_f32 = a.f32_rows or int(min(45_000_000, max(1_000_000, (0.80 * free - (6 << 30)) // per_row)) // 1_000_000 * 1_000_000)
        if free > rows_f32 * per_row + (4 << 30):
            result["f32"] = leg_f32(torch, dist, lib, B, ctypes, FlatIndexF16, a, dev, local_rank, rows=rows_f32)
        else:
            result["f32"] = {"skipped": f"{rows_f32} fp32 rows need {rows_f32 * per_row >> 30} GiB of HBM, {free >> 30} GiB free"}
        torch.cuda.empty_cache()
    # ---- shard files: a 10M x dim fp16 shard streamed HBM -> file -> HBM by the library, one GPU --------------
    if world == 1 and not a.no_persist and a.storage == "f16":
        torch.cuda.empty_cache()
        try:
            result["persistence"] = leg_persist(torch, lib, B, FlatIndexF16, a, local_rank)
        except OSError as exc:      # (no room for the file: the leg reports why instead of failing the bench)
            result["persistence"] = {"skipped": str(exc)}
    # ---- ingest: texts -> host WordPiece -> encoder -> normalise / quantise / append, one GPU ----------
    if world == 1 and not a.no_ingest and a.storage == "f16":
        torch.cuda.empty_cache()
        result["ingest"] = leg_ingest(torch, np, a, dev, local_rank)
    if rank == 0:
        os.write(real_stdout, (json.dumps(_ordered_for_readers(result)) + "\n").encode())
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


# Token counts of the reference's prompt around a (query, document) pair (core/rerank/Reranker_Qwen3.py:16-17,23-27), counted by
# words and punctuation marks: Qwen's vocabulary does not ship here (no network), so these are estimates of what its BPE yields.
C3_TPL = {"prefix": 39,      # <|im_start|>system\nJudge whether the Document meets ... "yes" or "no".<|im_end|>\n<|im_start|>user\n
          "instruct": 19,    # "<Instruct>: Given a web search query, retrieve relevant passages that answer the query\n"
          "query_hdr": 4,    # "<Query>: "
          "query": 12,       # a short web-search query
          "doc_hdr": 5,      # "\n<Document>: "
          "suffix": 9}       # <|im_end|>\n<|im_start|>assistant\n<think>\n\n</think>\n\n
C3_DOC_MIN, C3_DOC_SPAN = 64, 129      # document chunk length in tokens: 64 + hash(doc id) % 129  ->  64 .. 192


def _mix64(torch, key, pos):
    """Counter hash on int64 tensors (wrapping arithmetic): the synthetic token at position `pos` of the text `key`."""
    x = key * 6364136223846793005 + pos * 1442695040888963407 + 0x632BE59BD9B4E019
    x = x ^ (x >> 29)
    x = x * (-4658895280553007687)
    x = x ^ (x >> 32)
    return (x >> 17) & 0x7FFFFFFF


def _c3_fixed():
    t = C3_TPL
    q0 = t["prefix"] + t["instruct"] + t["query_hdr"]            # first query token
    d0 = q0 + t["query"] + t["doc_hdr"]                          # first document token = length of the part shared by a query's prompts
    return q0, d0, d0 + t["suffix"]


def c3_pair_lengths(torch, doc):
    """prompt length of each (query, document) pair"""
    return _c3_fixed()[2] + C3_DOC_MIN + _mix64(torch, doc, torch.zeros_like(doc) - 1) % C3_DOC_SPAN


def c3_pair_tokens(torch, qidx, doc, Lc, vocab, p_lo=0):
    """LEFT-padded token ids [n][Lc] (int32) + first-real-token index [n] (int32) of prompt positions [p_lo, length) of the
    pairs (query qidx[i], document doc[i]) — p_lo = 0: the whole prompts; p_lo = the shared length: what follows the part all
    prompts of a query have in common.  Template tokens depend on the position only (the same ids in every pair, like a real
    template), query tokens on (query, position), document tokens on (document id, position); lengths per C3_TPL / C3_DOC_*."""
    t = C3_TPL
    q0, d0, fixed = _c3_fixed()
    dl = C3_DOC_MIN + _mix64(torch, doc, torch.zeros_like(doc) - 1) % C3_DOC_SPAN
    ell = fixed + dl
    pos = torch.arange(Lc, device=doc.device)[None, :] - (Lc - (ell - p_lo))[:, None] + p_lo      # < p_lo: padding
    in_q = (pos >= q0) & (pos < q0 + t["query"])
    in_d = (pos >= d0) & (pos < d0 + dl[:, None])
    key = torch.where(in_q, (qidx[:, None] + 1) * 1_000_003, torch.where(in_d, doc[:, None] + (1 << 40), torch.zeros_like(pos)))
    # (suffix positions are counted from the end so that the suffix ids do not depend on the document length)
    ppos = torch.where(pos >= d0 + dl[:, None], pos - ell[:, None] + (1 << 20), pos)
    tok = 10 + _mix64(torch, key, ppos) % (vocab - 10)
    ids = torch.where(pos >= p_lo, tok, torch.zeros_like(tok)).int().contiguous()
    return ids, (Lc - (ell - p_lo)).int().contiguous(), ell


def c3_shared_tokens(torch, nq, P, vocab, dev):
    """The part every prompt of a query shares (chat prefix, instruction, query, "<Document>: "): [nq][P] LEFT padded + starts."""
    q0, d0, _ = _c3_fixed()
    t = C3_TPL
    pos = (torch.arange(P, device=dev)[None, :] - (P - d0)).expand(nq, P)
    qidx = torch.arange(nq, device=dev)
    in_q = (pos >= q0) & (pos < q0 + t["query"])
    key = torch.where(in_q, ((qidx[:, None] + 1) * 1_000_003).expand(nq, P), torch.zeros_like(pos))
    tok = 10 + _mix64(torch, key, pos) % (vocab - 10)
    ids = torch.where(pos >= 0, tok, torch.zeros_like(tok)).int().contiguous()
    return ids, torch.full((nq,), P - d0, dtype=torch.int32, device=dev)


LM_GEOM = dict(H=1024, LAYERS=28, NQ=16, NKV=8, DH=128, I=3072, V=151_669)   # Qwen3-Reranker-0.6B


def build_reranker_lm(torch, dev, local_rank, want_host):
    from rag_arc_amd.core.rerank import HipCausalLM

    G = LM_GEOM
    H, LAYERS, NQ, NKV, DH, I, V = (G[k] for k in ("H", "LAYERS", "NQ", "NKV", "DH", "I", "V"))
    g = torch.Generator(device=dev)
    g.manual_seed(28)
    rnd = lambda *shape: torch.randn(shape, generator=g, device=dev) * 0.03
    sd = {"model.embed_tokens.weight": rnd(V, H), "model.norm.weight": 1.0 + rnd(H)}
    for i in range(LAYERS):
        p = f"model.layers.{i}."
        sd[p + "self_attn.q_proj.weight"], sd[p + "self_attn.k_proj.weight"] = rnd(NQ * DH, H), rnd(NKV * DH, H)
        sd[p + "self_attn.v_proj.weight"], sd[p + "self_attn.o_proj.weight"] = rnd(NKV * DH, H), rnd(H, NQ * DH)
        sd[p + "self_attn.q_norm.weight"], sd[p + "self_attn.k_norm.weight"] = 1.0 + rnd(DH), 1.0 + rnd(DH)
        sd[p + "mlp.gate_proj.weight"], sd[p + "mlp.up_proj.weight"], sd[p + "mlp.down_proj.weight"] = rnd(I, H), rnd(I, H), rnd(H, I)
        sd[p + "input_layernorm.weight"], sd[p + "post_attention_layernorm.weight"] = 1.0 + rnd(H), 1.0 + rnd(H)
    lm = HipCausalLM(sd, NQ, NKV, DH, device=local_rank)
    # the oracle sees the weights the device holds (fp16 storage, as the reference's torch_dtype=float16), computes in fp32
    sd_host = {k: v.half().float().cpu().numpy() for k, v in sd.items()} if want_host else None
    return lm, sd_host


def leg_c3(torch, dist, lib, B, ctypes, np, idx2, q, dense_ids, n2, a, steps, warmup, passes_per_step, scan_bytes, kname,
           dev, local_rank):
    """BASELINE config 3 end to end on one GPU, AS CONFIGURED (reference path: core/retrieval/mutipath.py:37-93 ->
    VectorStore_Faiss.py:240,258-263 -> Reranker_Qwen3.py:23-49,57-74 -> core/utils/Fusion.py:45-76): per step, batch-256
    dense top-100 over 1M x 768 -> the cross-encoder's LM forward on all 256 x 100 (query, document) prompts (templated,
    left padded, Qwen3-Reranker-0.6B geometry with seeded fp16 weights) -> p_yes -> stable order -> RRF with the supplied
    lexical list.  Token ids are synthesised on the device from (query, document id) (no vocabulary ships offline): the
    LM's work depends on the lengths only.  The pairs of a step are sorted by length and run in chunks of `--c3-chunk`
    pairs, each chunk left padded to its own longest prompt.
    Sub-field `without_lm_forward`: the same loop with seeded logits in place of the LM (round 2's c3 figure)."""
    from rag_arc_amd.core.rerank import HipLogitReranker
    from rag_arc_amd.core.utils import HipRRFusion

    K, nq = a.k, a.batch
    g = torch.Generator(device=dev)
    g.manual_seed(99)
    zn = (torch.randn((nq, K), generator=g, device=dev) * 3).half()
    zy = (torch.randn((nq, K), generator=g, device=dev) * 3).half()
    lex = lexical_lists(torch, dense_ids, n2, K)
    lens = torch.full((nq, 2), K, dtype=torch.int32, device=dev)
    rr, fuse = HipLogitReranker(lambda *_: None, device=local_rank), HipRRFusion(device=local_rank)

    def end_seeded(h):
        ids, _ = h.result()
        _, perm = rr.score_order(zn, zy)
        keys = torch.stack([torch.gather(ids, 1, perm.long()), lex], dim=1)
        return fuse.fuse_ids(keys, lens, K)

    (dt0, (fk, fs, fn)), tot, nl = scan_profile(
        lib, B, ctypes, lambda: timed_loop(torch, dist, lambda: idx2.search_async(q, K), end_seeded, steps, warmup, False),
        8 * (steps + warmup) * passes_per_step + 16)
    scan = tot / max(1, steps + warmup)
    scan_roof = hbm_roofline(scan_bytes / (scan * 1e-3) / 1e9, kernel=kname)
    no_lm = {"workload": "the same loop with seeded fp16 (no, yes) logits in place of the LM forward: scan -> rerank score->order -> RRF",
             "value": round(nq * steps / dt0, 1), "unit": "queries/s", "ms_per_step": round(dt0 / steps * 1e3, 4),
             "scan_ms": round(scan, 4), "roofline": scan_roof}
    if a.no_lm:
        return {"workload": f"config 3 WITHOUT its cross-encoder (--no-lm): {n2}x{a.dim} fp16 scan top-{K} -> rerank score->order "
                            f"(seeded logits) -> RRF, batch {nq}, 1 GPU", **{k: v for k, v in no_lm.items() if k != "workload"},
                "fused_entries_per_query": int(fn.min().item())}

    G = LM_GEOM
    H, LAYERS, NQ, NKV, DH, I, V = (G[k] for k in ("H", "LAYERS", "NQ", "NKV", "DH", "I", "V"))
    lm, sd_host = build_reranker_lm(torch, dev, local_rank, want_host=not a.no_cpu_baseline)
    NO_ID, YES_ID = 1, 2
    n_pairs, CH = nq * K, max(4, a.c3_chunk - a.c3_chunk % 4)
    qidx_all = torch.arange(nq, device=dev).repeat_interleave(K)
    z_all = torch.empty((n_pairs, 2), dtype=torch.float16, device=dev)
    lm_ev, stats = [], {}

    share = not a.c3_no_prefix_sharing
    q0_, d0_, fixed_ = _c3_fixed()
    # the shared part as it is (79 tokens), left padded only as far as nq x P needs to be a multiple of 128 (a padded prefix
    # costs its pad tokens in the prefix pass and, worse, counts toward the key range that decides whether a chunk's attention
    # fits the LDS-resident kernel: 96 + 201 keys did not, 79 + 201 do)
    P_sh = d0_
    while (nq * P_sh) % 128:
        P_sh += 1
    pre_ids, pre_start = c3_shared_tokens(torch, nq, P_sh, V, dev)

    def rerank_fuse(ids):
        """the cross-encoder over every (query, top-k document) pair of the batch, then order and fuse.  With prefix sharing
        (default) the part the 100 prompts of a query have in common — chat prefix, instruction, query: 79 of ~216 tokens —
        runs through the LM once per query (rarc_lm_prefix_kv) and only what follows it per pair."""
        doc = ids.reshape(-1)
        ell = c3_pair_lengths(torch, doc)
        p_lo = d0_ if share else 0
        order = torch.argsort(ell, stable=True)
        ell_s = ell[order]
        starts = list(range(0, n_pairs, CH))
        cmax = [int(v) for v in torch.stack([ell_s[min(s0 + CH, n_pairs) - 1] for s0 in starts]).cpu().tolist()]   # one read-back
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        handle = lm.prefix_kv_device(pre_ids, pre_start) if share else None
        tokens = nq * P_sh if share else 0
        for s0, mx in zip(starts, cmax):
            sel = order[s0:s0 + CH]
            if sel.numel() % 4:      # (n_pairs is a multiple of 4 whenever k is: not hit at the defaults)
                sel = torch.cat([sel, sel[-1:].expand(4 - sel.numel() % 4)])
            Lc = mx - p_lo               # the chunk's longest remainder: tokens = pairs x Lc must be a multiple of 128 — it is for
            if (int(sel.numel()) * Lc) % 128:                                  # chunks of 640 pairs; otherwise round the length up
                Lc = -(-Lc // 32) * 32
            t_ids, t_start, _ = c3_pair_tokens(torch, qidx_all[sel], doc[sel], Lc, V, p_lo)
            if share:
                z_all[sel] = lm.yes_no_logits_device(t_ids, t_start, NO_ID, YES_ID, prefix=handle, prefix_of=qidx_all[sel].int())
            else:
                z_all[sel] = lm.yes_no_logits_device(t_ids, t_start, NO_ID, YES_ID)
            tokens += int(sel.numel()) * Lc
        e1.record()
        lm_ev.append((e0, e1))
        stats.update(padded_tokens=tokens, ell=ell)
        z = z_all.view(nq, K, 2)
        _, perm = rr.score_order(z[:, :, 0].contiguous(), z[:, :, 1].contiguous())
        keys = torch.stack([torch.gather(ids, 1, perm.long()), lex], dim=1)
        return fuse.fuse_ids(keys, lens, K)

    def end(h):
        ids, _ = h.result()
        return rerank_fuse(ids)

    st3, w3 = max(1, a.c3_steps), 1
    timed_loop(torch, dist, lambda: idx2.search_async(q, K), end, 0, w3, False)
    lm_ev.clear()
    dt, (fk, fs, fn) = timed_loop(torch, dist, lambda: idx2.search_async(q, K), end, st3, 0, False)
    torch.cuda.synchronize()
    lm_ms = sum(x.elapsed_time(y) for x, y in lm_ev) / max(1, len(lm_ev))
    ell = stats["ell"].double()
    g_mix = 2.0 * H * (NQ + 2 * NKV) * DH
    g_rest = 2.0 * (NQ * DH * H + H * 2 * I + I * H)
    att = 2.0 * NQ * DH          # flops per (query token, visible key) pair and layer, per product (q.k, p.v)
    # algorithmic flops of one step's LM work AS THE PATH RUNS IT: projections on the real tokens (the last layer's output
    # projection and MLP on the last position only, as rarc_lm_yes_no_logits computes them), causal attention over real keys
    if share:
        own = ell - d0_                                   # tokens of a pair after the shared part
        real_tokens = float(own.sum().item()) + nq * d0_
        shared_flops = nq * ((LAYERS - 1) * (d0_ * (g_mix + g_rest) + att * 2.0 * d0_ * (d0_ + 1) / 2.0) + d0_ * g_mix)
        pair_keys = float((own * d0_ + own * (own + 1) / 2.0).sum().item())          # visible keys summed over a pair's own tokens
        flops = (shared_flops + LAYERS * (float(own.sum().item()) * g_mix + att * 2.0 * pair_keys)
                 + (LAYERS - 1) * float(own.sum().item()) * g_rest + n_pairs * g_rest)
    else:
        real_tokens = float(ell.sum().item())
        flops = (LAYERS * (real_tokens * g_mix + att * float((ell * (ell + 1)).sum().item()))
                 + (LAYERS - 1) * real_tokens * g_rest + n_pairs * g_rest)
    flops_unshared = (LAYERS * (float(ell.sum().item()) * g_mix + att * float((ell * (ell + 1)).sum().item()))
                      + (LAYERS - 1) * float(ell.sum().item()) * g_rest + n_pairs * g_rest)
    out = {"workload": f"config 3 end to end, cross-encoder included: {n2}x{a.dim} fp16 scan top-{K} -> LM forward on {nq} x {K} = "
                       f"{n_pairs} templated (query, document) prompts of {int(ell.min().item())}-{int(ell.max().item())} tokens "
                       f"(Qwen3-Reranker-0.6B geometry: {LAYERS} layers, hidden {H}, {NQ}/{NKV} heads of {DH}, ffn {I}, vocabulary {V}; "
                       f"seeded fp16 weights) -> p_yes -> stable order -> RRF with a supplied lexical list, batch {nq}, 1 GPU",
           "value": round(nq * st3 / dt, 2), "unit": "queries/s", "ms_per_step": round(dt / st3 * 1e3, 3), "steps": st3, "warmup": w3,
           "pairs_per_s": round(n_pairs * st3 / dt, 1), "lm_ms_per_step": round(lm_ms, 3),
           "prefix_sharing": (f"the {d0_} tokens every prompt of a query shares run through the LM once per query (k | v cached, "
                              f"rarc_lm_prefix_kv); each pair runs its remaining {C3_DOC_MIN + C3_TPL['suffix']}-"
                              f"{C3_DOC_MIN + C3_DOC_SPAN - 1 + C3_TPL['suffix']} tokens") if share else "off (--c3-no-prefix-sharing)",
           "prompt_tokens": {"template": C3_TPL, "document": f"{C3_DOC_MIN}..{C3_DOC_MIN + C3_DOC_SPAN - 1} by a hash of the document id",
                             "prompt_tokens_per_step": int(ell.sum().item()),
                             "tokens_through_the_lm_per_step": int(real_tokens), "padded_tokens_per_step": int(stats["padded_tokens"]),
                             "chunk_pairs": CH, "note": "template token counts are word-count estimates (Qwen's vocabulary does not ship offline)"},
           "fused_entries_per_query": int(fn.min().item()),
           "roofline": {"bound": "mfma", "kernel": "rarc_gemm256_f16_kernel<16 | 19 | 96> (LM projections with RMSNorm folded in: row scale, row scale + SwiGLU, "
                                                        "residual add + sums of squares), rarc_lm_attention_resident_kernel<128, NP>",
                        "achieved": round(flops / (lm_ms * 1e-3) / 1e12, 1), "peak": MFMA_F16_PEAK_TF, "unit": "TFLOP/s",
                        "frac": round(flops / (lm_ms * 1e-3) / 1e12 / MFMA_F16_PEAK_TF, 4), "flops_per_step": flops,
                        "flops_per_step_without_prefix_sharing": flops_unshared,
                        "timed": "HIP events around the LM calls of a step (token synthesis, chunk gather and scatter included)",
                        "end_to_end_frac": round(flops / (dt / st3) / 1e12 / MFMA_F16_PEAK_TF, 4)},
           "without_lm_forward": no_lm}
    if sd_host is not None:
        out["cpu_baseline"], out["lm_parity_vs_oracle"] = cpu_baseline_lm(torch, np, lm, sd_host, dense_ids, qidx_all, V, NO_ID, YES_ID, K,
                                                                          (pre_ids, pre_start, d0_) if share else None)
    return out


def cpu_baseline_lm(torch, np, lm, sd_host, dense_ids, qidx_all, V, no_id, yes_id, K, shared=None):
    """The reranker's LM forward on the host (numpy fp32 oracle, pinned to transformers.Qwen3ForCausalLM) on four of the
    step's prompts — WHOLE prompts, the reference's way —, timed; and the device logits of the same four prompts (through
    the shared-prefix path when the leg uses it) checked against it."""
    from oracle import cpu_ref

    G = LM_GEOM
    doc = dense_ids.reshape(-1)
    ell = c3_pair_lengths(torch, doc)
    order = torch.argsort(ell, stable=True)
    pick = order[torch.tensor([0, len(order) // 3, 2 * len(order) // 3, len(order) - 1], device=order.device)]
    Lc = -(-int(ell[pick].max().item()) // 32) * 32
    t_ids, t_start, _ = c3_pair_tokens(torch, qidx_all[pick], doc[pick], Lc, V)
    if shared is None:
        got = lm.yes_no_logits_device(t_ids, t_start, no_id, yes_id).float().cpu().numpy()
    else:
        pre_ids, pre_start, d0 = shared
        Ls = -(-int((ell[pick] - d0).max().item()) // 32) * 32
        s_ids, s_start, _ = c3_pair_tokens(torch, qidx_all[pick], doc[pick], Ls, V, d0)
        got = lm.yes_no_logits_device(s_ids, s_start, no_id, yes_id, prefix=lm.prefix_kv_device(pre_ids, pre_start),
                                      prefix_of=qidx_all[pick].int()).float().cpu().numpy()
    ids_h, start_h = t_ids.cpu().numpy(), t_start.cpu().numpy()
    mask = (np.arange(Lc)[None, :] >= start_h[:, None]).astype(np.int64)
    t0 = time.perf_counter()
    want = cpu_ref.qwen3_last_logits_f32(sd_host, dict(num_attention_heads=G["NQ"], num_key_value_heads=G["NKV"], head_dim=G["DH"],
                                                       rms_norm_eps=1e-6, rope_theta=1e6), ids_h, mask, [no_id, yes_id])
    t_cpu = time.perf_counter() - t0
    p_got = 1.0 / (1.0 + np.exp(-(got[:, 1] - got[:, 0]).astype(np.float64)))
    p_want = 1.0 / (1.0 + np.exp(-(want[:, 1] - want[:, 0]).astype(np.float64)))
    n = len(pick)
    base = {"value": round(n / t_cpu / K, 4), "unit": "queries/s", "cores": os.cpu_count(), "kind": "port",
            "sample": f"numpy fp32 LM forward (oracle/cpu_ref.qwen3_last_logits_f32) of {n} prompts of up to {Lc} tokens in {t_cpu:.2f} s "
                      f"= {n / t_cpu:.2f} pairs/s; a query is {K} pairs; the scan and the fusion are not in this figure (they are "
                      f"three orders of magnitude cheaper)"}
    parity = {"prompts_checked": n, "max_abs_dlogit": float(np.max(np.abs(got - want))), "max_abs_logit": float(np.max(np.abs(want))),
              "max_abs_dp_yes": float(np.max(np.abs(p_got - p_want))),
              "tolerance": "fp16 model as the reference's torch_dtype=float16: |dlogit| <= 1e-1 at 28 layers, |dp_yes| <= 3e-2 "
                           "(tests/test_gpu_reranker_lm.py::test_bench_geometry_full_depth_and_vocabulary)",
              "within_tolerance": bool(np.max(np.abs(got - want)) <= 1e-1 and np.max(np.abs(p_got - p_want)) <= 3e-2)}
    return base, parity


def cpu_baseline(np, idx2, q, ids2, sc2, n2, a):
    """The CPU baseline as SURVEY.md 8(d) defines it, config 2 in full, on this box's host cores, + parity of the GPU answer.
    faiss is not on the box, so both figures are restatements (kind "port"):
      * `value` / `nq1_value`: numpy fp32 — normalised queries, `Q @ D_chunkᵀ` through the host's BLAS (all its threads), top-k
        by argpartition (oracle/cpu_ref.flat_search_blas_f32): batched B = 256 and the reference's own one-query-per-call
        pattern (VectorStore_Faiss.py:258-263).  This is the sgemm-backed shape faiss's IndexFlatIP has for nq >= 20.
      * `port_value` / `port_nq1_value`: the canonical-order oracle (oracle/rarc_oracle.c, AVX2 + OpenMP) — the parity checker,
        whose ids and score bits the GPU answer is compared with.
    Both score the SAME stored rows (the index's fp16 rows widened to fp32)."""
    from oracle import cpu_ref

    rows_h = idx2.rows.cpu().numpy().view(np.uint16)
    qn = cpu_ref.normalize_L2(q.cpu().numpy())
    t0 = time.perf_counter()
    ref_i, ref_s, nthreads = cpu_ref.flat_search_f16(rows_h, qn, a.k)
    tcpu = time.perf_counter() - t0
    # the reference itself searches one query per call (VectorStore_Faiss.py:258-263): same port, nq = 1
    n1 = min(8, a.batch)
    t1 = time.perf_counter()
    for qi in range(n1):
        cpu_ref.flat_search_f16(rows_h, qn[qi:qi + 1], a.k)
    t_nq1 = (time.perf_counter() - t1) / n1
    # the numpy / BLAS path over the same rows as fp32 (3 GB at 1M x 768)
    rows32 = rows_h.view(np.float16)[:, :a.dim].astype(np.float32)
    cpu_ref.flat_search_blas_f32(rows32[:65536], qn[:8], min(a.k, 10))             # (BLAS thread pool up, pages touched)
    t2 = time.perf_counter()
    bl_i, bl_s, blas_threads = cpu_ref.flat_search_blas_f32(rows32, qn, a.k)
    t_blas = time.perf_counter() - t2
    t3 = time.perf_counter()
    for qi in range(n1):
        cpu_ref.flat_search_blas_f32(rows32, qn[qi:qi + 1], a.k)
    t_blas1 = (time.perf_counter() - t3) / n1
    del rows32
    gpu_i = ids2.cpu().numpy()
    recall = float(np.mean([len(np.intersect1d(ref_i[b], gpu_i[b])) / float(a.k) for b in range(a.batch)]))
    recall_blas = float(np.mean([len(np.intersect1d(bl_i[b], gpu_i[b])) / float(a.k) for b in range(a.batch)]))
    return {"value": round(a.batch / t_blas, 1), "unit": "queries/s", "cores": blas_threads, "kind": "port",
            "sample": f"numpy fp32 Q @ D_chunk^T (host BLAS, {blas_threads} threads) + argpartition top-{a.k}: {a.batch} queries x {n2} rows x "
                      f"{a.dim} (config 2 in full) in {t_blas:.2f} s; {os.cpu_count()} host cpus; faiss is not installed on the box",
            "nq1_value": round(1.0 / t_blas1, 1),
            "nq1_sample": f"the same path, one query per call as the reference issues them, {n1} queries, {t_blas1 * 1e3:.1f} ms each",
            "port_value": round(a.batch / tcpu, 1), "port_cores": nthreads,
            "port_sample": f"oracle/rarc_oracle.c canonical-order flat search (AVX2 + OpenMP, {nthreads} threads), the same {a.batch} x {n2} x {a.dim}, {tcpu:.2f} s",
            "port_nq1_value": round(1.0 / t_nq1, 1),
            "blas_vs_canonical": {"max_abs_score_diff": float(np.abs(bl_s - ref_s).max()), f"recall_at_{a.k}_vs_gpu": round(recall_blas, 6)},
            "parity_vs_gpu": {"ids_bit_exact": bool(np.array_equal(ref_i, gpu_i)),
                              "scores_bit_exact": bool(np.array_equal(ref_s.view(np.uint32), sc2.cpu().numpy().view(np.uint32))),
                              f"recall_at_{a.k}": round(recall, 6)}}


def leg_f32(torch, dist, lib, B, ctypes, FlatIndexF16, a, dev, local_rank, rows=10_000_000):
    """storage="f32": the reference's own row format (fp32 rows in faiss, VectorStore_Faiss.py:170) — the one mode whose
    scores sit within north_star's 1e-5 of the float64 cosine on ARBITRARY fp32 embeddings (tests/test_gpu_storage_precision.py).
    The scan streams the rows' fp16 image (2 bytes per element, int8 prefilter with the image's own error bound added to
    the margin); survivors are rescored canonically from the fp32 rows.  Round 6: at the size ONE GPU HOLDS — 45M x 768 = 138 GB of
    fp32 rows + 69 GB of image (main() frees the other corpora first) —, batch 256, k = 100, 32 answers re-scanned exhaustively."""
    idx = FlatIndexF16(a.dim, metric="cosine", device=local_rank, storage="f32", capacity=rows)
    slab = 1 << 20
    buf = torch.empty((slab, a.dim), dtype=torch.float32, device=dev)
    t0 = time.perf_counter()
    for s0 in range(0, rows, slab):
        m = min(slab, rows - s0)
        B.check(lib.rarc_synth_rows_f32(buf.data_ptr(), a.dim, a.dim, s0, m, 1234, 0), "rarc_synth_rows_f32")
        idx.add(buf[:m])
    torch.cuda.synchronize()
    t_ingest = time.perf_counter() - t0
    del buf
    q = torch.empty((a.batch, a.dim), dtype=torch.float32, device=dev)
    B.check(lib.rarc_synth_rows_f32(q.data_ptr(), a.dim, a.dim, 0, a.batch, 4321, 0), "rarc_synth_rows_f32")
    steps, warm = max(a.steps, 30), max(a.warmup, 10)   # (the leg runs behind config 5's encoder: ten batches to settle the clocks)
    passes = (a.batch + 255) // 256
    (dt, (ids, sc)), tot, nl = scan_profile(
        lib, B, ctypes,
        lambda: timed_loop(torch, dist, lambda: idx.search_async(q, a.k), lambda h: h.result(), steps, warm, False),
        8 * (steps + warm) * passes + 16)
    scan_ms = tot / max(1, steps + warm)
    d_pad = B.padded_dim(a.dim)
    bytes_img = rows * d_pad * 2
    beat = idx.verify_batch(q, ids, sc, list(range(0, a.batch, 8)))
    out = {"workload": f"{rows}x{a.dim} fp32 rows (+ fp16 image for the scan: 6 bytes per element in HBM), batch {a.batch}, top-{a.k}",
           "value": round(a.batch * steps / dt, 1), "unit": "queries/s", "ms_per_step": round(dt / steps * 1e3, 4),
           "scan_ms_per_pass": round(scan_ms, 4), "ingest_rows_per_s": round(rows / t_ingest, 1),
           "rows_beating_kth": int(beat), "queries_verified_by_exact_rescan": len(range(0, a.batch, 8)),
           "roofline": hbm_roofline(bytes_img / (scan_ms * 1e-3) / 1e9, kernel="rarc_scan_q8_kernel",
                                    algorithmic_bytes_per_pass=int(bytes_img),
                                    end_to_end_frac=round(bytes_img / (dt / steps) / 1e9 / HBM_PEAK_GBS, 4))}
    del idx
    torch.cuda.empty_cache()
    return out


def leg_wide(torch, lib, B, FlatIndexF16, a, dev, local_rank, rows=10_000_000, dim=1536):
    """Rows wider than the register-resident scans take (the reference's OpenAI embeddings: 1536-d, openai_llm.py:139-161):
    the wide path (csrc/wide.hip) — score GEMM on the encoder's MFMA tiles, one chunk of 131072 rows at a time, select against
    a rigorous threshold, canonical finalize.  10M x 1536 fp16, batch 256, k = 100 and k = 2000."""
    d_pad = B.padded_dim(dim)
    buf = torch.empty((rows, d_pad), dtype=torch.float16, device=dev)
    B.check(lib.rarc_synth_rows_f16(buf.data_ptr(), d_pad, dim, 0, rows, 1234, 0), "rarc_synth_rows_f16")
    idx = FlatIndexF16(dim, metric="cosine", device=local_rank, growable=False)
    idx.add_rows_f16(buf, 1.001)
    q = torch.empty((a.batch, dim), dtype=torch.float32, device=dev)
    B.check(lib.rarc_synth_rows_f32(q.data_ptr(), dim, dim, 0, a.batch, 4321, 0), "rarc_synth_rows_f32")
    out = {"workload": f"{rows}x{dim} fp16, batch {a.batch}: rarc_search_wide (chunked score GEMM -> select -> canonical finalize)"}
    bytes_rows = rows * d_pad * 2
    flops = 2.0 * 256 * rows * d_pad
    for k in (a.k, 2000):
        for _ in range(2):
            ids, sc = idx.search_device(q, k)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 5
        for _ in range(reps):
            ids, sc = idx.search_device(q, k)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / reps * 1e3
        out[f"k{k}"] = {"value": round(a.batch / (ms * 1e-3), 1), "unit": "queries/s", "ms_per_step": round(ms, 3),
                        "roofline": {"bound": "mfma", "achieved": round(flops / (ms * 1e-3) / 1e12, 1), "peak": MFMA_F16_PEAK_TF,
                                     "unit": "TFLOP/s", "frac": round(flops / (ms * 1e-3) / 1e12 / MFMA_F16_PEAK_TF, 4),
                                     "hbm_frac_of_row_bytes": round(bytes_rows / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
                        "candidate_capacity_per_query": int(getattr(idx, "last_wide_cap", 0))}
    # full-size property (parity itself: tests/test_gpu_wide.py, bit-exact vs the oracle up to 200k rows at these widths): an
    # exact top-k under a total order is a prefix of the exact top-k' for k' > k — two independent runs, different
    # thresholds and candidate sets, must agree bit for bit on the first k entries
    ids_h, sc_h = ids.cpu().numpy(), sc.cpu().numpy()
    ids_k, sc_k = idx.search_device(q, a.k)
    out["top_k_is_prefix_of_top_2000"] = bool((ids_k.cpu().numpy() == ids_h[:, : a.k]).all()
                                              and (sc_k.cpu().numpy().view("uint32") == sc_h[:, : a.k].view("uint32")).all())
    out["descending"] = bool((sc_h[:, :-1] >= sc_h[:, 1:]).all())
    del idx, buf
    torch.cuda.empty_cache()
    return out


def leg_pairs(torch, np, lib, B, a, dev, n=100_000, d=1024, dup=2000):
    """the entity de-duplication step of the reference's graph store (Base_Neo4j.py:559-583: sklearn
    cosine_similarity over all entity embeddings + a python loop over i < j) as `similar_pairs`: n synthetic embeddings with
    `dup` planted near-duplicates, every pair with cosine >= 0.95.  CPU side: the same function on a bounded sample."""
    from rag_arc_amd.encapsulation.database.graph_db import similar_pairs

    x = torch.empty((n, d), dtype=torch.float32, device=dev)
    B.check(lib.rarc_synth_rows_f32(x.data_ptr(), d, d, 0, n, 777, 0), "rarc_synth_rows_f32")
    g = torch.Generator(device=dev); g.manual_seed(3)
    src = torch.randint(0, n, (dup,), generator=g, device=dev)
    dst = torch.randint(0, n, (dup,), generator=g, device=dev)
    x[dst] = x[src] + 0.05 * torch.randn((dup, d), generator=g, device=dev) * x[src].norm(dim=1, keepdim=True) / d ** 0.5
    for _ in range(3):                                  # warm-up (kernel attributes, allocator, clocks)
        similar_pairs(x, 0.95)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 10
    for _ in range(reps):
        pairs = similar_pairs(x, 0.95)
    dt = (time.perf_counter() - t0) / reps
    d_pad = max(256, (d + 63) // 64 * 64)
    # the GEMM multiplies rows [0, end of the column super-block) with each super-block of 4096 columns
    n_pad = (n + 255) // 256 * 256
    flops = sum(2.0 * min(n_pad, c0 + 4096) * min(4096, n_pad - c0) * d_pad for c0 in range(0, n, 4096))
    out = {"workload": f"{n} embeddings x {d} dims, {dup} planted near-duplicates: every pair i < j with cosine >= 0.95 "
                       "(rag_arc_amd...graph_db.similar_pairs -> rarc_similar_pairs: fp16 score GEMM with the select in its epilogue, "
                       "exact float64 rescoring of the nominated pairs)",
           "value": round(n / dt, 1), "unit": "entities/s", "ms_per_call": round(dt * 1e3, 2), "pairs_found": len(pairs),
           "roofline": {"bound": "mfma", "achieved": round(flops / dt / 1e12, 1), "peak": MFMA_F16_PEAK_TF, "unit": "TFLOP/s",
                        "frac": round(flops / dt / 1e12 / MFMA_F16_PEAK_TF, 4),
                        "note": "whole call, host side included (upload of nothing: the embeddings are resident; download + sort of the pairs)"}}
    if not a.no_cpu_baseline:
        from oracle import cpu_ref

        m = 6000
        xs = x[:m].cpu().numpy().astype(np.float64)
        t0 = time.perf_counter()
        want = cpu_ref.similar_pairs_f64(xs, 0.95)
        cpu_s = time.perf_counter() - t0
        got = similar_pairs(x[:m], 0.95)
        out["cpu_baseline"] = {"value": round(m / cpu_s, 1), "unit": "entities/s", "cores": os.cpu_count(), "kind": "port",
                               "sample": f"numpy float64 restatement (oracle/cpu_ref.similar_pairs_f64) of the first {m} embeddings in {cpu_s:.2f} s "
                                         f"(cost grows with n^2: {n} entities are {(n / m) ** 2:.0f}x that, and the reference's python "
                                         "double loop is ~100x slower than the vectorised restatement)"}
        out["parity_vs_oracle_on_sample"] = {"pairs": len(want), "same_pairs": [(i, j) for i, j, _ in got] == [(i, j) for i, j, _ in want],
                                             "max_abs_dscore": float(max([abs(a_[2] - b_[2]) for a_, b_ in zip(got, want)] or [0.0]))}
    del x
    torch.cuda.empty_cache()
    return out


# ------------------------------------------------------------------------------------------------ the plugin surface
def _ordered_for_readers(result):
    """The same line, keys ordered so that whoever keeps only the END of a long line still sees what matters most: the
    contract's own keys first (a JSON parser reads them wherever they are), then the legs from the bulkiest / least
    asked-about (ingest table, shard files, config 3) to the ones the last review asked for (f32, wide, config 5, the
    plugin surface), and a one-glance summary of every leg as the very last key."""
    def pick(d, *path):
        for k in path:
            if not isinstance(d, dict) or k not in d:
                return None
            d = d[k]
        return d
    late = ["ingest", "persistence", "c3", "pairs", "c2", "f32", "wide", "c5", "api"]
    out = {k: v for k, v in result.items() if k not in late}
    for k in late:
        if k in result:
            out[k] = result[k]
    summary = {
        "headline_q_per_s": result.get("value"), "headline_scan_frac_of_hbm": pick(result, "roofline", "frac"),
        "c2_q_per_s": pick(result, "c2", "value"), "c2_scan_frac": pick(result, "c2", "roofline", "frac"),
        "c3_q_per_s": pick(result, "c3", "value"), "c3_lm_frac_of_mfma": pick(result, "c3", "roofline", "frac"),
        "c5_q_per_s": pick(result, "c5", "value"), "c5_scan_frac": pick(result, "c5", "roofline", "frac"),
        "c5_encoder_frac_of_mfma": pick(result, "c5", "encoder_roofline", "frac"),
        "f32_q_per_s": pick(result, "f32", "value"), "f32_scan_frac": pick(result, "f32", "roofline", "frac"),
        "wide_1536_k100_q_per_s": pick(result, "wide", "k100", "value"), "wide_1536_k2000_q_per_s": pick(result, "wide", "k2000", "value"),
        "api_1M_batch_invoke_q_per_s": pick(result, "api", "c2", "batch_invoke_256", "value"),
        "api_1M_vs_engine": pick(result, "api", "c2", "batch_invoke_256", "vs_engine"),
        "api_100M_batch_invoke_q_per_s": pick(result, "api", "headline", "batch_invoke_256", "value"),
        "api_100M_vs_engine": pick(result, "api", "headline", "batch_invoke_256", "vs_engine"),
        "api_100M_two_callers_vs_engine": pick(result, "api", "headline", "batch_invoke_256_two_callers", "vs_engine"),
        "api_256_coroutines_ms": [pick(result, "api", "c2", "coroutines_256_ainvoke", "wall_ms"),
                                  pick(result, "api", "headline", "coroutines_256_ainvoke", "wall_ms")],
        "all_pairs_100k_x_1024_ms": pick(result, "pairs", "ms_per_call"),
        "cpu_q_per_s": pick(result, "cpu_baseline", "value"),
        "shard_file_GBps_save_load_cached": [pick(result, "persistence", "save_GBps"), pick(result, "persistence", "load_GBps_from_storage"),
                                            pick(result, "persistence", "load_GBps_from_page_cache")],
    }
    out["summary"] = {k: v for k, v in summary.items() if v is not None and v != [None, None] and v != [None, None, None]}
    return out


def _percentile(xs, p):
    xs = sorted(xs)
    return xs[min(len(xs) - 1, int(round(p / 100.0 * (len(xs) - 1))))]


def leg_api(torch, np, lib, B, a, dev, local_rank, idx, n_rows, engine, via_registry):
    """Throughput and latency THROUGH the registered backend — `retriever.batch_invoke` / `.invoke` of the reference's call
    chain (core/retrieval/mutipath.py:37-93 -> core/retrieval/dense.py:122-174 -> VectorStore_Faiss.py:225-274), texts in,
    lists of Documents out — next to the engine-level figure of the same index (`engine`: q/s and ms per 256-query batch
    at FlatIndexF16.search_async).  The docstore is real: one python Document per row up to 2M rows, byte columns beyond
    (docstore.py).  Query texts map to the synthetic query vectors through a lookup-table provider whose table sits in HBM
    (the encoder is timed on its own in c5 / ingest)."""
    import tempfile
    import threading

    from rag_arc_amd.config.app_registration import registrator
    from rag_arc_amd.config.modules import VectorStoreRetrieverConfig
    from rag_arc_amd.core.retrieval.base import BaseRetriever
    from rag_arc_amd.core.retrieval.dense import VectorStoreRetriever
    from rag_arc_amd.core.retrieval.multipath import MultiPathRetriever
    from rag_arc_amd.core.utils.data_model import Document
    from rag_arc_amd.core.utils.fusion import HipRRFusion
    from rag_arc_amd.encapsulation.database.vector_db.docstore import ColumnarDocstore
    from rag_arc_amd.encapsulation.database.vector_db.hip_flat import HipFlatVectorStore
    from rag_arc_amd.encapsulation.embeddings.table import TableEmbeddings

    import gc

    # what the earlier legs of THIS process left on the heap (the LM's and the encoder's python objects, result dicts, ...) is not
    # part of a retrieval service: out of the cyclic collector's way before the leg builds its own docstore, back at the end
    # (tools/lab/gc_probe.py: in a process that holds the store and nothing else an answer's collection costs ~1 ms per 256 x 100)
    gc.collect()
    gc.freeze()
    K, NB, NQT = a.k, 256, 2048
    out = {"workload": f"{n_rows}x{a.dim} fp16, batch {NB} query TEXTS -> lists of {K} Documents, through "
                       f"{'registrator.get_object(...)' if via_registry else 'VectorStoreRetriever(HipFlatVectorStore)'}",
           "engine": engine,
           "gc": "enabled; the objects of the bench's earlier legs were frozen out of its generations before this leg built its docstore"}
    t0 = time.perf_counter()
    width = len(str(n_rows - 1))
    if n_rows <= 2_000_000:
        docs = [Document(content=f"{i:0{width}d}", metadata={}, id=f"{i:0{width}d}") for i in range(n_rows)]
        out["docstore"] = f"{n_rows} python Documents (list + the reference's two dicts)"
    else:
        docs = ColumnarDocstore.decimal(n_rows)
        out["docstore"] = f"{n_rows} rows as byte columns (ColumnarDocstore): a Document is built when a search names its row"
    out["docstore_build_s"] = round(time.perf_counter() - t0, 2)
    qv = torch.empty((NQT, a.dim), dtype=torch.float32, device=dev)
    B.check(lib.rarc_synth_rows_f32(qv.data_ptr(), a.dim, a.dim, 0, NQT, 4321, 0), "rarc_synth_rows_f32")
    texts = [f"q{i}" for i in range(NQT)]
    qv_h = qv.cpu().numpy()
    tmp = None
    if via_registry:
        # the reference's own route: a saved index + a JSON config, registered and fetched by name (framework/register.py:15-26)
        tmp = tempfile.TemporaryDirectory(prefix="rarc_api_")
        HipFlatVectorStore(TableEmbeddings(texts, qv_h), device=local_rank).adopt(idx, docs).save_local(tmp.name)
        np.savez(os.path.join(tmp.name, "queries.npz"), texts=np.array(texts), vectors=qv_h)
        cfg = {"type": "vectorstore_retriever", "search_type": "similarity",
               "vectorstore": {"type": "hip_flat_vectorstore", "metric": "cosine", "device": local_rank,
                               "index_path": tmp.name,
                               "embedding": {"type": "table_embeddings", "path": os.path.join(tmp.name, "queries.npz"),
                                             "device": local_rank}}}
        with open(os.path.join(tmp.name, "retriever.json"), "w") as fh:
            json.dump(cfg, fh)
        t0 = time.perf_counter()
        registrator.register(os.path.join(tmp.name, "retriever.json"), "rarc_api_bench", VectorStoreRetrieverConfig)
        retriever = registrator.get_object("rarc_api_bench")
        out["registry_build_s"] = round(time.perf_counter() - t0, 2)
        store = retriever.vectorstore
    else:
        store = HipFlatVectorStore(TableEmbeddings(texts, qv_h).to_device(local_rank), device=local_rank).adopt(idx, docs)
        retriever = VectorStoreRetriever(store)
    big = n_rows > 20_000_000
    reps = 8 if big else 40

    def rate(fn, n_queries, reps, warm=2):
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        dt = time.perf_counter() - t0
        return {"value": round(n_queries * reps / dt, 1), "unit": "queries/s", "ms_per_call": round(dt / reps * 1e3, 3)}

    # (a) one caller, batch after batch of 256 texts; host-side phases added up by the store
    first = texts[:NB]
    answers = retriever.batch_invoke(first, k=K)
    ref_ids = idx.search_device(qv[:NB], K)[0].cpu().numpy()
    base = int(getattr(idx, "id_base", 0))
    out["answers_equal_engine_ids"] = bool(all([int(d.id) for d in docs_q] == (ref_ids[i] - base).tolist()
                                               for i, docs_q in enumerate(answers)))
    one = [retriever.invoke(t, k=K) for t in first[:4]]
    out["batch_equals_invoke"] = bool(all([d.id for d in x] == [d.id for d in y] for x, y in zip(one, answers[:4])))
    store.timing = {}
    r = rate(lambda: retriever.batch_invoke(first, k=K), NB, reps, warm=0)
    tm, store.timing = store.timing, None
    r["host_ms_per_call"] = {key[:-2] + "_ms": round(v / reps * 1e3, 3) for key, v in tm.items()}
    r["vs_engine"] = round(r["value"] / engine["value"], 3)
    out["batch_invoke_256"] = r
    # (b) the same with the cyclic garbage collector's old generations frozen: a million-Document docstore is what every
    # young collection's promotions are measured against (gc.freeze() after building a store is the usual remedy)
    import gc
    gc.collect()
    gc.freeze()
    r = rate(lambda: retriever.batch_invoke(first, k=K), NB, reps)
    r["vs_engine"] = round(r["value"] / engine["value"], 3)
    out["batch_invoke_256_gc_frozen"] = r
    # (b2) two callers, each batch after batch of 256: one's scan runs while the other's answer becomes Documents
    def two_callers(n_each):
        def loop():
            for _ in range(n_each):
                retriever.batch_invoke(first, k=K)
        th = [threading.Thread(target=loop) for _ in range(2)]
        t0 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        return time.perf_counter() - t0

    two_callers(1)
    dt2c = two_callers(max(2, reps // 2))
    out["batch_invoke_256_two_callers"] = {"value": round(2 * max(2, reps // 2) * NB / dt2c, 1), "unit": "queries/s",
                                           "vs_engine": round(2 * max(2, reps // 2) * NB / dt2c / engine["value"], 3)}
    # (c) one call with 2048 texts: eight scans, each running while the answer before it is mapped
    r = rate(lambda: retriever.batch_invoke(texts, k=K), NQT, max(2, reps // 8), warm=1)
    r["vs_engine"] = round(r["value"] / engine["value"], 3)
    out["batch_invoke_2048"] = r
    # (d) with (Document, score) pairs: batch_similarity_search_with_score (VectorStore_Faiss.py:250-274 for a batch)
    r = rate(lambda: store.batch_similarity_search_with_score(first, K), NB, reps)
    out["batch_with_scores_256"] = r
    # (e) 256 threads, one invoke each (the reference's ainvoke pattern, core/retrieval/base.py:82-96): the coalescer
    def storm():
        go = threading.Barrier(NB + 1)
        res = [None] * NB

        def call(i):
            go.wait()
            res[i] = retriever.invoke(first[i], k=K)

        th = [threading.Thread(target=call, args=(i,)) for i in range(NB)]
        for t in th:
            t.start()
        l0 = store.coalesced_launches[0]
        t0 = time.perf_counter()
        go.wait()
        for t in th:
            t.join()
        return time.perf_counter() - t0, store.coalesced_launches[0] - l0, res

    storm()
    runs = [storm() for _ in range(3 if big else 5)]
    dt, launches, res = min(runs, key=lambda x: x[0])
    out["threads_256_invoke"] = {"value": round(NB / dt, 1), "unit": "queries/s", "wall_ms": round(dt * 1e3, 3),
                                 "scans": launches,
                                 "equal_batch": bool(all([d.id for d in x] == [d.id for d in y] for x, y in zip(res, answers)))}
    co = getattr(store, "_coalescer", None)
    if co is not None:          # the same burst with the store's `coalesce_window_us` = 200: an idle-index leader waits that long for company
        w0, co.window_s = co.window_s, 200e-6
        storm()
        dtw, launches_w, _ = min((storm() for _ in range(3 if big else 5)), key=lambda x: x[0])
        co.window_s = w0
        out["threads_256_invoke"]["with_coalesce_window_200us"] = {"value": round(NB / dtw, 1), "wall_ms": round(dtw * 1e3, 3),
                                                                   "scans": launches_w}
    # (e2) 256 coroutines, one ainvoke each, from one event loop: no thread per caller (hip_flat._AsyncFront)
    import asyncio

    async def astorm():
        return await asyncio.gather(*[retriever.ainvoke(t, k=K) for t in first])

    asyncio.run(astorm())
    l0 = store._async_front().launches
    t0 = time.perf_counter()
    n_rep = 3 if big else 10
    for _ in range(n_rep):
        ares = asyncio.run(astorm())
    dta = (time.perf_counter() - t0) / n_rep
    out["coroutines_256_ainvoke"] = {"value": round(NB / dta, 1), "unit": "queries/s", "wall_ms": round(dta * 1e3, 3),
                                     "scans": round((store._async_front().launches - l0) / n_rep, 2),
                                     "equal_batch": bool(all([d.id for d in x] == [d.id for d in y] for x, y in zip(ares, answers)))}
    # (f) one caller, one query at a time: latency
    lat = []
    for i in range(30 if big else 300):
        t0 = time.perf_counter()
        retriever.invoke(texts[i % NQT], k=K)
        lat.append((time.perf_counter() - t0) * 1e3)
    lat = lat[5:]
    eng = []
    for i in range(10 if big else 50):
        t0 = time.perf_counter()
        idx.search(qv[i:i + 1], K)
        eng.append((time.perf_counter() - t0) * 1e3)
    out["invoke_latency_ms"] = {"p50": round(_percentile(lat, 50), 3), "p99": round(_percentile(lat, 99), 3),
                                "mean": round(sum(lat) / len(lat), 3), "calls": len(lat),
                                "queries_per_s": round(1e3 / (sum(lat) / len(lat)), 1),
                                "engine_nq1_p50": round(_percentile(eng[2:], 50), 3)}
    # (f2) the reference-shaped call WITH THE ENCODER IN IT: invoke(text) = tokenise -> embed_query (the
    # fp32-class forward of a model of this dimension, 32 padded tokens: the query path) -> search -> Documents
    out["invoke_latency_with_encoder_ms"] = api_encoder_latency(torch, np, a, dev, local_rank, idx, docs, K, big)
    # (g) MultiPathRetriever: dense + the supplied lexical list, RRF over all 256 queries in one launch
    lex = lexical_lists(torch, torch.from_numpy(ref_ids).to(dev), base + n_rows, K).cpu().numpy() - base
    lex_docs = {t: [Document(content=f"{r:0{width}d}", metadata={"path": "lexical"}, id=f"L{r}") for r in row.tolist()]
                for t, row in zip(first, lex)}

    class SuppliedLists(BaseRetriever):
        """The "supplied BM25 rank list"  as a retriever: query text -> its ordered Documents."""

        def _get_relevant_documents(self, query, **kw):
            return lex_docs[query][: kw.get("k", 5)]

        def batch_invoke(self, inputs, **kw):
            return [lex_docs[q][: kw.get("k", 5)] for q in inputs]

    mp = MultiPathRetriever([retriever, SuppliedLists()], HipRRFusion(device=local_rank), top_k_per_retriever=K)
    fused = mp.batch_invoke(first, top_k=K)
    single = [mp.invoke(t, top_k=K) for t in first[:4]]
    r = rate(lambda: mp.batch_invoke(first, top_k=K), NB, reps)
    r["vs_engine"] = round(r["value"] / engine["value"], 3)
    r["batch_equals_invoke"] = bool(all([d.content for d in x] == [d.content for d in y] for x, y in zip(single, fused[:4])))
    out["multipath_batch_invoke_256"] = r
    gc.unfreeze()
    if tmp is not None:
        registrator.registrations.pop("rarc_api_bench", None)
        tmp.cleanup()
    # the store, its coalescer and the retrievers hold each other in cycles: collect them now, or the index they adopted
    # (153.6 GB at 100M rows) stays allocated under the legs that follow
    del store, retriever, mp, fused, single, answers, docs, lex_docs
    gc.collect()
    return out


ENC_GEOMETRY = {384: ("bge-small", 384, 12, 1536, 12), 768: ("bge-base", 768, 12, 3072, 12), 1024: ("bge-large", 1024, 16, 4096, 24)}


def seeded_bert_state_dict(torch, dev, H, FFN, LAYERS, VOCAB=30522, seed=5):
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    rnd = lambda *s: torch.randn(s, generator=g, device=dev) * 0.05
    sd = {"embeddings.word_embeddings.weight": rnd(VOCAB, H), "embeddings.position_embeddings.weight": rnd(512, H),
          "embeddings.token_type_embeddings.weight": rnd(2, H), "embeddings.LayerNorm.weight": 1.0 + rnd(H),
          "embeddings.LayerNorm.bias": rnd(H)}
    for i in range(LAYERS):
        p = f"encoder.layer.{i}."
        for nm, (o, c) in {"attention.self.query": (H, H), "attention.self.key": (H, H), "attention.self.value": (H, H),
                           "attention.output.dense": (H, H), "intermediate.dense": (FFN, H), "output.dense": (H, FFN)}.items():
            sd[p + nm + ".weight"], sd[p + nm + ".bias"] = rnd(o, c), rnd(o)
        for nm in ("attention.output.LayerNorm", "output.LayerNorm"):
            sd[p + nm + ".weight"], sd[p + nm + ".bias"] = 1.0 + rnd(H), rnd(H)
    return sd


def api_encoder_latency(torch, np, a, dev, local_rank, idx, docs, K, big):
    """One query at a time, as the reference issues them (VectorStore_Faiss.py:240 `embed_query`, :258-263 search;
    core/file_management/embeddings/huggingface.py:136-145), with a REAL encoder forward in the line: a seeded model of the
    index's dimension (bge-base geometry for 768-d, bge-large for 1024-d, bge-small for 384-d), fp32-class precision, query
    texts of 8..24 tokens (32 padded: the encoder's query path).  Reported: the whole `retriever.invoke` and its two halves
    on their own."""
    from rag_arc_amd.core.retrieval.dense import VectorStoreRetriever
    from rag_arc_amd.encapsulation.database.vector_db.hip_flat import HipFlatVectorStore
    from rag_arc_amd.encapsulation.embeddings.hip_bert import HipBertEmbeddings, HipBertEncoder

    if a.dim not in ENC_GEOMETRY:
        return {"skipped": f"no encoder geometry of dimension {a.dim}"}
    name, H, HEADS, FFN, LAYERS = ENC_GEOMETRY[a.dim]
    enc = HipBertEncoder(seeded_bert_state_dict(torch, dev, H, FFN, LAYERS), num_heads=HEADS, device=local_rank, precision="fp32")

    def tokenize(text):          # [CLS] + 6..22 word pieces decided by the text + [SEP]  (no vocabulary ships offline)
        h = int.from_bytes(text.encode()[-8:].rjust(8, b"0"), "little")
        n = 6 + h % 17
        return [101] + [1000 + (h * (j + 3) * 2654435761 >> 7) % 28000 for j in range(n)] + [102]

    emb = HipBertEmbeddings(enc, tokenize, max_length=64)
    store = HipFlatVectorStore(emb, device=local_rank).adopt(idx, docs)
    retriever = VectorStoreRetriever(store)
    qtexts = [f"what does passage {i} say about retrieval" for i in range(64)]
    for t_ in qtexts[:6]:
        retriever.invoke(t_, k=K)
    # the answers are the engine's answers for the encoder's embedding
    ok = True
    for t_ in qtexts[:3]:
        e = emb.embed_queries_device([t_])
        want = idx.search_device(e, K)[0].cpu().numpy()[0] - int(getattr(idx, "id_base", 0))
        ok = ok and [int(d.id) for d in retriever.invoke(t_, k=K)] == want.tolist()
    n = 20 if big else 200
    lat, enc_only, enc_list, search_only = [], [], [], []
    for i in range(n):
        t0 = time.perf_counter()
        retriever.invoke(qtexts[i % 64], k=K)
        lat.append((time.perf_counter() - t0) * 1e3)
    for i in range(min(n, 100)):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e = emb.embed_queries_device([qtexts[i % 64]])
        torch.cuda.synchronize()
        enc_only.append((time.perf_counter() - t0) * 1e3)
        t0 = time.perf_counter()
        idx.search(e, K)
        search_only.append((time.perf_counter() - t0) * 1e3)
        t0 = time.perf_counter()
        emb.embed_query(qtexts[i % 64])            # the reference's contract: a python list of floats
        enc_list.append((time.perf_counter() - t0) * 1e3)
    out = {"encoder": f"{name} geometry ({LAYERS} layers x {H}, seeded weights), fp32-class, query path "
                      f"({'on' if enc.query_path else 'off'}): 32 padded tokens per query",
           "p50": round(_percentile(lat[5:], 50), 3), "p99": round(_percentile(lat[5:], 99), 3),
           "mean": round(sum(lat[5:]) / len(lat[5:]), 3), "calls": len(lat) - 5,
           "queries_per_s": round(1e3 / (sum(lat[5:]) / len(lat[5:])), 1),
           "embed_queries_device_p50": round(_percentile(enc_only[3:], 50), 3),
           "embed_query_as_python_list_p50": round(_percentile(enc_list[3:], 50), 3),
           "search_nq1_p50": round(_percentile(search_only[3:], 50), 3),
           "answers_equal_engine_on_the_embedding": bool(ok)}
    del store, retriever, emb, enc
    import gc
    gc.collect()
    torch.cuda.empty_cache()
    return out


def leg_c5(torch, dist, lib, B, ctypes, np, FlatIndexF16, ShardedFlatSearch, shard_range, split_range, a, world, rank,
           local_rank, dev, use_dist):
    """BASELINE config 5 end to end (reference path: huggingface.py:122-126 -> VectorStore_Faiss.py:258-263 ->
    mutipath.py:37-93 -> Fusion.py:45-76): token ids resident in HBM -> bge-large-geometry encoder forward
    (hidden 1024, 16 heads, ffn 4096, `--c5-layers` layers of seeded weights, CLS pooling, L2 norm) -> fp8
    (e4m3fn + row scale) 1024-d corpus, row-sharded -> top-k -> [N>1: all-gather + merge] -> RRF with the supplied
    lexical list.  With N ranks each rank embeds batch/N of the queries and ONE all-gather rebuilds the batch."""
    from rag_arc_amd.core.utils import HipRRFusion
    from rag_arc_amd.encapsulation.embeddings.hip_bert import HipBertEncoder

    H, HEADS, FFN, LAYERS, L, VOCAB, K, nq = 1024, 16, 4096, a.c5_layers, 32, 30522, a.k, a.batch
    g = torch.Generator(device=dev)
    g.manual_seed(5)

    def rnd(*shape, scale=0.05):
        return torch.randn(shape, generator=g, device=dev) * scale

    sd = {"embeddings.word_embeddings.weight": rnd(VOCAB, H), "embeddings.position_embeddings.weight": rnd(512, H),
          "embeddings.token_type_embeddings.weight": rnd(2, H),
          "embeddings.LayerNorm.weight": 1.0 + rnd(H), "embeddings.LayerNorm.bias": rnd(H)}
    for i in range(LAYERS):
        p = f"encoder.layer.{i}."
        for nm, (o, c) in {"attention.self.query": (H, H), "attention.self.key": (H, H), "attention.self.value": (H, H),
                           "attention.output.dense": (H, H), "intermediate.dense": (FFN, H), "output.dense": (H, FFN)}.items():
            sd[p + nm + ".weight"], sd[p + nm + ".bias"] = rnd(o, c), rnd(o)
        for nm in ("attention.output.LayerNorm", "output.LayerNorm"):
            sd[p + nm + ".weight"], sd[p + nm + ".bias"] = 1.0 + rnd(H), rnd(H)
    enc = HipBertEncoder(sd, num_heads=HEADS, device=local_rank, precision=a.encoder_precision)
    other = "fp16" if a.encoder_precision == "fp32" else "fp32"
    enc_other = HipBertEncoder(sd, num_heads=HEADS, device=local_rank, precision=other)   # timed alone, beside the leg
    sd_host = None
    if rank == 0 and not a.no_cpu_baseline and world == 1:
        sd_host = {k: v.cpu().numpy() for k, v in sd.items()}
    del sd
    tok = torch.randint(1, VOCAB, (nq, L), generator=g, device=dev).int()
    lens = torch.randint(4, L + 1, (nq,), generator=g, device=dev).int()
    tok = torch.where(torch.arange(L, device=dev)[None, :] < lens[:, None], tok, torch.zeros_like(tok))
    tok_h, lens_h = tok.cpu().numpy(), lens.cpu().numpy()
    tok, lens = tok.contiguous(), lens.contiguous()

    rows = a.c5_rows
    if rows <= 0:
        free = torch.cuda.mem_get_info(dev)[0]
        rows = 100_000_000
        while rows > 1_000_000 and (rows / world) * (H + 4) > 0.85 * free - (6 << 30):
            rows //= 10
    lo, hi = shard_range(rows, rank, world)
    idx = build_index(torch, lib, B, FlatIndexF16, local_rank, H, lo, hi, storage="f8")
    searcher = ShardedFlatSearch(idx, force_collective=use_dist)
    fuse = HipRRFusion(device=local_rank)
    q_lo, q_hi = split_range(nq, rank, world)
    split = use_dist and nq % world == 0
    if not split:
        q_lo, q_hi = 0, nq
    # who embeds what on N ranks (--c5-split); --emulate-world W plays ONE rank of W on this GPU (no collective: the other
    # ranks' embeddings are a copy of what the first forward produced)
    emu = a.emulate_world if (world == 1 and a.emulate_world > 1 and nq % a.emulate_world == 0) else 0
    rotate = a.c5_split == "rotate" and (split or emu) and world * max(emu, 1) > 1
    if emu:
        q_lo, q_hi = split_range(nq, 0, emu)
    if rotate:
        split = False
    # The encoder of batch i+1 runs on a SIDE stream under the scan of batch i (the scan leaves the matrix pipe about half
    # idle and the encoder barely touches HBM); the scan's stream waits on the event that closes the forward.  Two result
    # buffers alternate: a search copies its queries into the index's query block first thing, so a buffer is free again
    # long before its next use with at most two batches in flight.
    overlap = a.encoder_overlap
    main_stream = torch.cuda.current_stream(dev)
    enc_stream = torch.cuda.Stream(device=dev) if overlap else main_stream
    mine_buf = [torch.empty((q_hi - q_lo, H), dtype=torch.float32, device=dev) for _ in range(2)]
    full_buf = [torch.empty((nq, H), dtype=torch.float32, device=dev) for _ in range(2)]
    enc_ev, n_emb, active = [], [0], [a.encoder_precision]

    def embed():
        slot = n_emb[0] & 1
        n_emb[0] += 1
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(enc_stream):
            e0.record()
            if rotate:
                # batch i belongs to rank i mod N: it embeds all the queries, the others receive them
                n_ranks = emu or world
                turn = (n_emb[0] - 1) % n_ranks
                out = full_buf[slot]
                if turn == (0 if emu else rank):
                    encoders[active[0]].forward_device(tok, lens, normalize=True, out=out)
                elif emu:
                    out.copy_(emb_all[0])
                if not emu:
                    dist.broadcast(out, src=turn)
            else:
                mine = encoders[active[0]].forward_device(tok[q_lo:q_hi], lens[q_lo:q_hi], normalize=True, out=mine_buf[slot])   # token ids already in HBM
                if split:
                    dist.all_gather_into_tensor(full_buf[slot], mine)
                    out = full_buf[slot]
                elif emu:
                    out = full_buf[slot]
                    out.copy_(emb_all[0])
                    out[q_lo:q_hi].copy_(mine)
                else:
                    out = mine
            e1.record()
        if overlap:
            main_stream.wait_event(e1)
        enc_ev.append((e0, e1))
        return out

    def time_alone(e, reps=5):   # forward of this rank's share, nothing else on the GPU
        for _ in range(2):
            e.forward_device(tok[q_lo:q_hi], lens[q_lo:q_hi], normalize=True, out=mine_buf[0])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            e.forward_device(tok[q_lo:q_hi], lens[q_lo:q_hi], normalize=True, out=mine_buf[0])
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3

    enc_alone_ms, enc_other_ms = time_alone(enc), time_alone(enc_other)
    encoders = {a.encoder_precision: enc, other: enc_other}
    emb_all = [enc.forward_device(tok, lens, normalize=True).clone()] if emu else [None]

    emb0 = embed().clone()
    ids0, _ = searcher.search_device(emb0, K)
    lex = lexical_lists(torch, ids0, rows, K)
    lens2 = torch.full((nq, 2), K, dtype=torch.int32, device=dev)

    def begin():
        return searcher.search_async(embed(), K)

    def end(h):
        ids, _ = searcher.finish(h, K)
        return fuse.fuse_ids(torch.stack([ids, lex], dim=1), lens2, K)

    steps, warmup = a.steps, a.warmup
    timed_loop(torch, dist, begin, end, 0, warmup, use_dist)
    enc_ev.clear()
    searcher.measure_exchange(True)
    (dt, (fk, fs, fn)), tot, nl = scan_profile(lib, B, ctypes, lambda: timed_loop(torch, dist, begin, end, steps, 0, use_dist),
                                               8 * steps + 16)
    exch = searcher.exchange_ms() / max(1, steps)
    torch.cuda.synchronize()
    enc_ms = sum(x.elapsed_time(y) for x, y in enc_ev) / max(1, len(enc_ev))
    scan_ms = tot / max(1, steps)
    alt = None
    if not a.no_c5_alt:   # the same loop with the other encoder arithmetic, for the record (not the leg's value)
        active[0] = other
        timed_loop(torch, dist, begin, end, 0, 2, use_dist)
        enc_ev.clear()
        dt_alt, _ = timed_loop(torch, dist, begin, end, steps, 0, use_dist)
        torch.cuda.synchronize()
        alt = {"encoder_precision": other, "value": round(nq * steps / dt_alt, 1), "unit": "queries/s",
               "ms_per_step": round(dt_alt / steps * 1e3, 4),
               "encoder_ms": round(sum(x.elapsed_time(y) for x, y in enc_ev) / max(1, len(enc_ev)), 4),
               "note": "same loop, other encoder arithmetic" + ("; ids are NOT the fp32 reference's (1e-3-class embeddings)" if other == "fp16" else "")}
        active[0] = a.encoder_precision
    # full-size property of the fp8 shard: exact re-scan of a few queries
    l_ids, l_sc = idx.search_device(emb0, K)
    vq = list(range(min(nq, max(0, a.verify_queries))))
    beat = idx.verify_batch(emb0, l_ids, l_sc, vq) if vq else 0
    if rank != 0:
        return None
    n_tok = (q_hi - q_lo) * L
    gemm_flops = LAYERS * n_tok * 2.0 * (3 * H * H + H * H + 2 * H * FFN)
    flops = gemm_flops + LAYERS * n_tok * 4.0 * L * H
    mfma_flops = flops + (2.0 * gemm_flops if a.encoder_precision == "fp32" else 0.0)
    scan_bytes = (hi - lo) * (H + 4)
    prec_txt = ("fp32-class forward (the reference's precision: split-operand fp16 MFMA GEMMs, fp32 elsewhere)"
                if a.encoder_precision == "fp32" else "fp16 forward (1e-3 class)")
    out = {"workload": f"config 5 end to end: {nq} queries x {L} tokens -> bge-large geometry encoder ({LAYERS} layers, seeded "
                       f"weights, {prec_txt}) -> {rows}x{H} fp8 (e4m3fn + row scale) corpus over {world} GPU(s), top-{K} -> RRF "
                       f"with a supplied lexical list",
           "encoder_precision": a.encoder_precision,
           "encoder_overlap": "side stream under the previous batch's scan" if overlap else "none (same stream)",
           "encoder_split": ("rotate: rank i mod N embeds the whole batch i and broadcasts it" if rotate else
                             "slice: every rank embeds batch/N queries of every batch") if (use_dist or emu) else "one rank",
           **({"emulated_world": emu, "note_emulation": f"ONE rank of {emu} played on this GPU: its shard, its share of the encoder "
               "work; the other ranks' embeddings are copies; no collective — a per-rank step time for the projection, not a measurement of N GPUs"} if emu else {}),
           "encoder_alone_ms": round(enc_alone_ms, 4), f"encoder_alone_ms_{other}": round(enc_other_ms, 4),
           "value": round(nq * steps / dt, 1), "unit": "queries/s", "ms_per_step": round(dt / steps * 1e3, 4),
           "n_gpus": world, "rows_per_gpu": hi - lo, "queries_embedded_per_gpu": q_hi - q_lo,
           "encoder_ms": round(enc_ms, 4), "scan_ms": round(scan_ms, 4), "exchange_ms_per_step": round(exch, 4),
           "fused_entries_per_query": int(fn.min().item()),
           "full_size_check": {"queries_verified_by_exact_rescan": len(vq), "rows_beating_kth": int(beat)},
           "roofline": hbm_roofline(scan_bytes / (scan_ms * 1e-3) / 1e9, kernel="rarc_scan_q8_kernel<1024, fp8>",
                                    algorithmic_bytes_per_scan=int(scan_bytes),
                                    int8_TOPs=round(2.0 * nq * (hi - lo) * H / (scan_ms * 1e-3) / 1e12, 1)),
           "encoder_roofline": {"bound": "mfma", "achieved": round(mfma_flops / (enc_alone_ms * 1e-3) / 1e12, 1),
                                "peak": MFMA_F16_PEAK_TF, "unit": "TFLOP/s",
                                "frac": round(mfma_flops / (enc_alone_ms * 1e-3) / 1e12 / MFMA_F16_PEAK_TF, 4),
                                "flops_per_forward": flops, "mfma_flops_per_forward": mfma_flops, "tokens_per_forward": n_tok,
                                "timed": "the forward alone (encoder_alone_ms); encoder_ms is its duration inside the step, under the scan",
                                "note": "fp32 mode issues 3 fp16 MFMA products per model flop (split operands): achieved counts the MFMA flops issued"
                                        if a.encoder_precision == "fp32" else "fp16 MFMA flops = model flops",
                                "includes": "embedding + all layers + pooling" + (" + all-gather of the embeddings" if split else "")}}
    if alt is not None:
        out["other_encoder_precision"] = alt
    if sd_host is not None:
        out["cpu_baseline"] = cpu_baseline_c5(np, sd_host, tok_h, lens_h, HEADS, idx, emb0, K, fuse, lex)
        out["encoder_vs_host_fp32_forward"] = out["cpu_baseline"].pop("encoder_check")
    return out


def leg_persist(torch, lib, B, FlatIndexF16, a, local_rank):
    """SURVEY 8 f1 — save_local / load_local of a corpus-scale shard (VectorStore_Faiss.py:432-482 -> faiss.write_index /
    read_index): `--persist-rows` x dim fp16 rows written from HBM and read back into HBM by rarc_device_to_file /
    rarc_file_to_device (pinned ring, no host copy), GB/s of row bytes for each direction, the process's resident-memory
    high-water mark, and whether the search after the load equals the search before the save bit for bit."""
    import shutil
    import tempfile

    def vm(field):
        with open("/proc/self/status") as fh:
            for line in fh:
                if line.startswith(field + ":"):
                    return int(line.split()[1])
        return -1

    n, d = a.persist_rows, a.dim
    need = n * B.padded_dim(d) * 2
    folder = a.persist_dir or tempfile.mkdtemp(prefix="rarc_bench_")
    if shutil.disk_usage(folder).free < need * 1.1:
        raise OSError(f"{folder}: less than {need * 1.1 / 1e9:.0f} GB free for the shard file")
    path = os.path.join(folder, "bench.rarc")
    try:
        idx = build_index(torch, lib, B, FlatIndexF16, local_rank, d, 0, n)
        q = torch.empty((a.batch, d), dtype=torch.float32, device=torch.device("cuda", local_rank))
        B.check(lib.rarc_synth_rows_f32(q.data_ptr(), d, d, 0, a.batch, 4321, 0), "rarc_synth_rows_f32")
        i0, s0 = idx.search_device(q, a.k)
        torch.cuda.synchronize()
        rss0 = vm("VmRSS")
        t0 = time.perf_counter()
        st_save = idx.save_shard(path)
        t_save = time.perf_counter() - t0
        del idx
        torch.cuda.empty_cache()
        # first load: the file was written with O_DIRECT, so nothing of it is in the page cache — it comes from storage
        idx2 = FlatIndexF16(d, metric="cosine", device=local_rank)
        t0 = time.perf_counter()
        st_cold = idx2.load_shard(path)
        torch.cuda.synchronize()
        t_cold = time.perf_counter() - t0
        i1, s1 = idx2.search_device(q, a.k)
        same = bool(torch.equal(i0, i1) and torch.equal(s0.view(torch.int32), s1.view(torch.int32)))
        del idx2
        torch.cuda.empty_cache()
        # second load: the (buffered) first one left the file in the page cache — the rate of the pinned ring + PCIe alone
        idx3 = FlatIndexF16(d, metric="cosine", device=local_rank)
        t0 = time.perf_counter()
        st_hot = idx3.load_shard(path)
        torch.cuda.synchronize()
        t_hot = time.perf_counter() - t0
        del idx3
        torch.cuda.empty_cache()
        return {"workload": f"{n}x{d} fp16 shard ({need / 1e9:.1f} GB), rarc_device_to_file / rarc_file_to_device, 8 threads, 256 MB pinned ring",
                "save_GBps": round(st_save["gb_per_s"], 2), "save_s": round(t_save, 3), "save_o_direct": bool(st_save.get("direct")),
                "load_GBps_from_storage": round(st_cold["gb_per_s"], 2), "load_s_from_storage": round(t_cold, 3),
                "load_GBps_from_page_cache": round(st_hot["gb_per_s"], 2), "load_s_from_page_cache": round(t_hot, 3),
                "pcie_gen5_x16_GBps": 63.0, "search_after_load_identical": same,
                "host_rss_kb_before": rss0, "host_hwm_kb_after": vm("VmHWM"), "directory": folder}
    finally:
        if os.path.exists(path):
            os.unlink(path)
        if not a.persist_dir:
            shutil.rmtree(folder, ignore_errors=True)


INGEST_GEOMS = {"bge-base": dict(H=768, HEADS=12, FFN=3072, LAYERS=12), "bge-large": dict(H=1024, HEADS=16, FFN=4096, LAYERS=24)}


def ingest_vocab(n_words=24000, n_tails=5000, seed=11):
    """A synthetic uncased WordPiece vocabulary of BERT's size class (no vocabulary ships offline): specials, single
    characters and their "##" forms, punctuation, `n_words` pseudo-words of 3-9 letters, `n_tails` "##" continuations."""
    import random

    rnd = random.Random(seed)
    letters = "abcdefghijklmnopqrstuvwxyz"
    vocab = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"] + list(letters) + ["##" + c for c in letters] + list("0123456789") + \
            ["##" + c for c in "0123456789"] + list(".,;:!?()-'\"")
    seen = set(vocab)
    words, tails = [], []
    while len(words) < n_words:
        w = "".join(rnd.choice(letters) for _ in range(rnd.randint(3, 9)))
        if w not in seen:
            seen.add(w)
            words.append(w)
    while len(tails) < n_tails:
        w = "##" + "".join(rnd.choice(letters) for _ in range(rnd.randint(2, 5)))
        if w not in seen:
            seen.add(w)
            tails.append(w)
    return vocab + words + tails, words, [t[2:] for t in tails]


def ingest_texts(n_docs, target_tokens, words, tails, seed):
    """Word-salad documents of roughly `target_tokens` WordPiece tokens: vocabulary words, a fifth of them with a "##" tail
    glued on, some capitalised, punctuation in between (every text runs into the max_length truncation, like real chunks
    cut to the model's window)."""
    import random

    rnd = random.Random(seed)
    out = []
    for _ in range(n_docs):
        parts, n = [], 0
        while n < target_tokens + 8:
            w = rnd.choice(words)
            r = rnd.random()
            if r < 0.2:
                w, n = w + rnd.choice(tails), n + 1
            if r > 0.9:
                w = w.capitalize()
            if r > 0.93:
                w, n = w + rnd.choice(".,;!?"), n + 1
            parts.append(w)
            n += 1
        out.append(" ".join(parts))
    return out


def leg_ingest(torch, np, a, dev, local_rank):
    """SURVEY 8 f2 — `add_texts` (VectorStore_Faiss.py:156-210) with the HIP encoder as the provider
    (huggingface.py:105-134): raw texts -> the library's WordPiece threads -> token-budget encoder calls (fp32-class
    forward, the reference's precision; fp16 beside it) -> normalise + fp16 rows appended in HBM.  docs/s end to end, with
    the tokeniser's share and the encoder's MFMA fraction (issued flops: the fp32-class forward runs 3x the model's GEMM
    flops on the fp16 MFMA)."""
    from rag_arc_amd.encapsulation.database.vector_db import HipFlatVectorStore
    from rag_arc_amd.encapsulation.embeddings.hip_bert import HipBertEmbeddings, HipBertEncoder
    from rag_arc_amd.encapsulation.embeddings.wordpiece import WordPieceTokenizer

    vocab, words, tails = ingest_vocab()
    out = {"unit": "documents/s", "path": "texts -> rarc_wordpiece_encode (host threads) -> rarc_enc32_forward / rarc_enc_forward -> "
                                          "rarc_ingest_f16 + rarc_quant_meta_f16 (HipFlatVectorStore.add_texts)",
           "vocabulary": f"synthetic, {len(vocab)} entries", "host_cpus": os.cpu_count(), "configs": []}
    g = torch.Generator(device=dev)
    g.manual_seed(17)

    def rnd(*shape, scale=0.05):
        return torch.randn(shape, generator=g, device=dev) * scale

    for name, G in INGEST_GEOMS.items():
        H, FFN, LAYERS = G["H"], G["FFN"], G["LAYERS"]
        sd = {"embeddings.word_embeddings.weight": rnd(len(vocab), H), "embeddings.position_embeddings.weight": rnd(512, H),
              "embeddings.token_type_embeddings.weight": rnd(2, H),
              "embeddings.LayerNorm.weight": 1.0 + rnd(H), "embeddings.LayerNorm.bias": rnd(H)}
        for i in range(LAYERS):
            p = f"encoder.layer.{i}."
            for nm, (o, c) in {"attention.self.query": (H, H), "attention.self.key": (H, H), "attention.self.value": (H, H),
                               "attention.output.dense": (H, H), "intermediate.dense": (FFN, H), "output.dense": (H, FFN)}.items():
                sd[p + nm + ".weight"], sd[p + nm + ".bias"] = rnd(o, c), rnd(o)
            for nm in ("attention.output.LayerNorm", "output.LayerNorm"):
                sd[p + nm + ".weight"], sd[p + nm + ".bias"] = 1.0 + rnd(H), rnd(H)
        gemm_flops_per_token = LAYERS * 2.0 * (4 * H * H + 2 * H * FFN)
        for precision in ("fp32", "fp16"):
            enc = HipBertEncoder(sd, num_heads=G["HEADS"], device=local_rank, precision=precision)
            for L in (128, 512):
                tok = WordPieceTokenizer({t: i for i, t in enumerate(vocab)}, max_length=L)
                emb = HipBertEmbeddings(enc, tok, max_length=L, pad_id=tok.pad)
                est = {"bge-base": 9000, "bge-large": 3000}[name] * (128 / L) * (3 if precision == "fp16" else 1)
                n_docs = a.ingest_docs or int(max(512, min(16384, 1.5 * est)) // 256 * 256)
                texts = ingest_texts(n_docs, L, words, tails, seed=L + len(name))
                HipFlatVectorStore(emb).add_texts(texts[: min(n_docs, 512)])        # warm-up (allocations, first launches)
                torch.cuda.synchronize()
                store = HipFlatVectorStore(emb)
                t0 = time.perf_counter()
                store.add_texts(texts, ids=[str(i) for i in range(n_docs)])
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                st = dict(emb.last_stats)
                # the encoder alone on the same token arrays (ids already tokenised): what the GPU side sustains
                ids, lens = tok.encode_batch(texts)
                order = np.argsort(-lens, kind="stable")
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for s0, e0 in emb._batches(lens[order]):
                    sel = order[s0:e0]
                    enc.forward(ids[sel, : int(lens[sel].max())], lens[sel], True, non_blocking=True)
                torch.cuda.synchronize()
                dt_enc = time.perf_counter() - t1
                # the python tokeniser on a sample, for scale
                t2 = time.perf_counter()
                for x in texts[:64]:
                    tok(x)
                py_tok_s = sum(int(v) for v in lens[:64]) / (time.perf_counter() - t2)
                issued = gemm_flops_per_token * st["padded_tokens"] * (3.0 if precision == "fp32" else 1.0)
                out["configs"].append({
                    "model": f"{name} geometry ({LAYERS} layers, hidden {H}), seeded weights", "precision": precision,
                    "tokens_per_text": L, "documents": n_docs, "value": round(n_docs / dt, 1),
                    "tokens_per_s": round(st["tokens"] / dt, 0), "encoder_calls": st["encoder_calls"],
                    "sequences_per_call": round(n_docs / max(1, st["encoder_calls"]), 1),
                    "tokenize_share": round(st["tokenize_seconds"] / dt, 3),
                    "tokenizer_tokens_per_s": round(st["tokens"] / max(st["tokenize_seconds"], 1e-9), 0),
                    "python_tokenizer_tokens_per_s": round(py_tok_s, 0),
                    "encoder_only_documents_per_s": round(n_docs / dt_enc, 1),
                    "encoder_mfma_TFLOPs_issued": round(issued / dt_enc / 1e12, 1),
                    "encoder_mfma_frac_of_2500": round(issued / dt_enc / 1e12 / 2500.0, 4),
                    "stored_rows": int(store.ntotal)})
                del store, emb
            del enc
            torch.cuda.empty_cache()
        del sd
        torch.cuda.empty_cache()
    return out


def cpu_baseline_c5(np, sd_host, tok_h, lens_h, heads, idx, emb, K, fuse, lex):
    """CPU path for config 5 on a bounded sample: fp32 numpy encoder forward (the oracle pinned to
    transformers.BertModel) on 16 of the queries, the oracle's fp8 flat search on a 1M-row slice for all of them, RRF."""
    from oracle import cpu_ref

    ns = min(16, tok_h.shape[0])
    t0 = time.perf_counter()
    o_emb = cpu_ref.bert_forward_f32(sd_host, tok_h[:ns], lens_h[:ns], heads)
    t_enc = (time.perf_counter() - t0) / ns
    dist_l2 = np.linalg.norm(emb[:ns].cpu().numpy().astype(np.float64) - o_emb.astype(np.float64), axis=1)
    n_s = min(1_000_000, idx.ntotal)
    rows_h = idx.rows[:n_s].cpu().numpy()
    sc_h = idx.row_scales[:n_s].cpu().numpy()
    qn = cpu_ref.normalize_L2(emb.cpu().numpy())
    t1 = time.perf_counter()
    ref_i, _, nthreads = cpu_ref.flat_search_f8(rows_h, sc_h, qn, K)
    t_scan = time.perf_counter() - t1
    lex_h = lex.cpu().numpy()
    t2 = time.perf_counter()
    for b in range(qn.shape[0]):
        cpu_ref.rrf_fuse([ref_i[b].tolist(), lex_h[b].tolist()], 60.0, K)
    t_rrf = (time.perf_counter() - t2) / qn.shape[0]
    per_q_1m = t_enc + t_scan / qn.shape[0] + t_rrf
    return {"value": round(1.0 / per_q_1m, 2), "unit": "queries/s", "cores": nthreads, "kind": "port",
            "sample": f"numpy fp32 encoder forward on {ns} queries ({t_enc * 1e3:.1f} ms each) + oracle fp8 flat search of "
                      f"{qn.shape[0]} queries over a {n_s}-row slice ({t_scan:.2f} s) + python RRF ({t_rrf * 1e6:.0f} us each); "
                      f"the rate is for the {n_s}-row slice, not extrapolated to the full corpus",
            "encoder_check": {"queries": ns, "max_l2_distance_to_numpy_fp32_forward": float(dist_l2.max()),
                              "note": "device embeddings vs the host's fp32 forward from the same token ids (bounds every cosine-score difference)"}}


if __name__ == "__main__":
    main()
