"""`bench.py --gpus N` end to end with N real ranks on real kernels, on a one-GPU box: --backend gloo --one-device puts
every rank on cuda:0 (RCCL refuses that; gloo moves the same tensors).  Exercises what the driver's scaling run
exercises — launcher, shard ranges, per-rank index build, pipelined steps, packed exchange + merge, the full-size
check reduced over ranks, and the config-5 leg with the encoder split over the ranks — minus RCCL itself.  Timings
of ranks that share a GPU mean nothing and are not looked at."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("world,split", [(2, "slice"), (8, "slice"), (4, "rotate")])
def test_bench_runs_with_n_ranks_on_one_device(world, split):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--backend", "gloo", "--one-device",
                        "--rows", "4000000", "--c5-rows", "2000000", "--c5-layers", "2", "--steps", "3", "--warmup", "1",
                        "--verify-queries", "4", "--c5-split", split], capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == world and out["config"]["rows_per_gpu"] == 4_000_000 // world
    chk = out["config"]["full_size_check"]
    assert chk["rows_beating_kth"] == 0 and chk.get("queries_differing", 0) == 0 and chk["ranks_checked"] == world
    assert out["config"]["repaired_queries_last_step"] == 0 and out["config"]["exchange_ms_per_step"] > 0
    c5 = out["c5"]
    assert c5["n_gpus"] == world and c5["queries_embedded_per_gpu"] == 256 // world and c5["fused_entries_per_query"] == 100
    assert c5["encoder_split"].startswith(split)
    assert c5["full_size_check"]["rows_beating_kth"] == 0


def test_bench_with_fp32_rows_sharded_over_four_ranks():
    """`--storage f32` (the reference's own row format, the one mode inside north_star's 1e-5 on arbitrary embeddings) through the
    same N-rank path: every rank's shard checked exhaustively, the merged line reports fp32 storage and the streamed image's bytes."""
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--backend", "gloo", "--one-device",
                        "--storage", "f32", "--rows", "2000000", "--steps", "3", "--warmup", "1", "--verify-queries", "8"],
                       capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    out = json.loads([ln for ln in p.stdout.splitlines() if ln.strip()][-1])
    assert out["n_gpus"] == 4 and out["dtype"] == "f32" and out["config"]["rows_per_gpu"] == 500_000
    assert "fp32" in out["config"]["workload"]
    chk = out["config"]["full_size_check"]
    assert chk["rows_beating_kth"] == 0 and chk["ranks_checked"] == 4
    assert out["roofline"]["algorithmic_bytes_per_launch"] * out["roofline"]["launches_per_scan"] == 500_000 * 768 * 2
