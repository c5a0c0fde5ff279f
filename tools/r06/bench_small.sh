#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; cd "$R"; O=$R/gpurun_out/r06_bench_small; mkdir -p "$O"
timeout 1200 python3 bench.py --rows 1000000 --no-c3 --no-c5 --no-persist --no-ingest --no-f32 --no-wide --no-pairs "$@" > "$O/bench.json" 2> "$O/bench.err"
echo rc=$?; tail -5 "$O/bench.err"
python3 - "$O/bench.json" <<'PY'
import json, sys
j = json.load(open(sys.argv[1]))
print(json.dumps({k: j[k] for k in ("value", "ms_per_step", "roofline")}, indent=0)[:900])
print("c2", json.dumps(j.get("c2"))[:700])
print("cpu_baseline", json.dumps(j.get("cpu_baseline"), indent=0)[:1800])
api = j.get("api", {}).get("c2", {})
for k in ("batch_invoke_256", "batch_with_scores_256", "threads_256_invoke", "coroutines_256_ainvoke", "invoke_latency_ms", "invoke_latency_with_encoder_ms"):
    print(k, json.dumps(api.get(k))[:600])
PY
