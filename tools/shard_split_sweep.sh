#!/bin/bash
# experiment: where the int8-prefilter scan of a SHARD (default 12.5M rows = 100M / 8) is cut for its tightening passes
# (RARC_SPLIT_DIVS = up to three divisors of the shard, ascending cut positions; 0 = no cut): ms per step, scan ms
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
rows=${1:-12500000}; shift
for divs in "${@:-512,64,8}"; do
  RARC_SPLIT_DIVS=$divs RARC_FORCE_DIST=1 python3 bench.py --rows $rows --steps 60 --warmup 5 --no-c2 --no-c3 --no-c5 --no-persist --no-ingest --no-api --no-f32 --no-wide --no-pairs --no-cpu-baseline --verify-queries 8 2>/dev/null > /tmp/split.json
  python3 - "$divs" <<'PY'
import json, sys
j = json.load(open("/tmp/split.json"))
print(f"divs {sys.argv[1]:12s}: {j['ms_per_step']:.4f} ms/step  scan {j['roofline']['scan_ms_per_pass']:.4f} ms in {j['roofline']['launches_per_scan']} launches  check {j['config']['full_size_check']}")
PY
done
