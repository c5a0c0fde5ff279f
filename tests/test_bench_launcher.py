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


CONTRACT_KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                 "dtype", "data", "config", "roofline", "cpu_baseline"}


def test_eight_rank_dry_run_emits_the_contract_line():
    """no 8-GPU box has run this tree, so the first one must not fail on plumbing.  `bench.py --gpus 8`
    (here --dry-run: gloo, stand-in local search) starts its own eight ranks and relays ONE line with every key of the
    driver's contract, n_gpus = 8, every rank's shard size (they cover the corpus exactly) and an exchange_ms_per_step."""
    p = _run("--gpus", "8", "--dry-run", "--steps", "2", "--warmup", "1", "--rows", "100000003", "--k", "100", "--batch", "16")
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert CONTRACT_KEYS <= set(out), CONTRACT_KEYS - set(out)
    assert out["n_gpus"] == 8 and out["ranks"] == 8 and out["steps"] == 2 and out["warmup"] == 1 and out["scaling"] == "strong"
    cfg = out["config"]
    assert cfg["rows_per_gpu"] == 12500001 and len(cfg["rows_per_rank"]) == 8 and sum(cfg["rows_per_rank"]) == 100000003
    assert max(cfg["rows_per_rank"]) - min(cfg["rows_per_rank"]) <= 8
    assert isinstance(cfg["exchange_ms_per_step"], float) and out["merged_ids_ok"] is True
    assert out["roofline"]["bound"] == "hbm" and out["value"] is None


def test_a_measured_line_has_the_same_contract_keys_as_the_dry_run():
    """The keys of the dry-run line are held against the literal result dict in bench.py's main(): if one side gains or loses a
    contract key the other must follow."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    main = src[src.index("def main():"):]
    block = main[main.index("result = {"):main.index("# ---- config 2 / 3")]
    for key in CONTRACT_KEYS - {"cpu_baseline"}:
        assert f'"{key}":' in block, key
    assert 'result["cpu_baseline"]' in main
    for key in ("rows_per_gpu", "exchange_ms_per_step", "n_corpus"):
        assert f'"{key}":' in block, key
