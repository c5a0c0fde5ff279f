"""`bench.py --gpus N` starts its own N ranks (no torchrun around it) and relays rank 0's one JSON line.
On CPU this is rehearsed with --dry-run (gloo): shard ranges, the (id, score) all-gather, the merge and the
max-over-ranks timing run for real, the local search is a stand-in."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*args, env=None):
    e = dict(os.environ)
    e.pop("WORLD_SIZE", None)
    e.pop("RANK", None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True,
                          timeout=600, env=e)


def test_two_rank_dry_run_relays_one_json_line():
    p = _run("--gpus", "2", "--dry-run", "--steps", "3", "--rows", "1001", "--k", "10", "--batch", "4")
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["ranks"] == 2 and out["dry_run"] is True
    assert out["rows_per_gpu"] == 501 and out["merged_ids_ok"] is True


def test_more_gpus_than_devices_fails_loudly():
    import torch

    have = torch.cuda.device_count()
    p = _run("--gpus", str(have + 2), "--steps", "1")
    assert p.returncode != 0 and "GPU(s) visible" in p.stderr and p.stdout.strip() == ""
