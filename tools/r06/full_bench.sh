#!/bin/bash
# the driver's command (python bench.py, no flags) timed by wall clock, + selected new tests in front
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; cd "$R"; O=$R/gpurun_out/r06_full_bench; mkdir -p "$O"
if [ "${1:-tests}" = tests ]; then
  timeout 900 python3 -m pytest tests/test_gpu_wide.py tests/test_gpu_multirank.py tests/test_gpu_batching.py -x -q -m gpu -rs 2>&1 | tail -4 | tee "$O/pytest_tail.txt"
fi
t0=$(date +%s.%N)
timeout 1500 python3 bench.py > "$O/bench.json" 2> "$O/bench.err"; echo "rc=$? wall=$(echo "$(date +%s.%N) - $t0" | bc) s"
tail -3 "$O/bench.err"
python3 - "$O/bench.json" <<'PY'
import json, sys
j = json.load(open(sys.argv[1]))
print(json.dumps(j["summary"], indent=0))
print("roofline", json.dumps(j["roofline"])[:900])
print("f32", json.dumps(j.get("f32"))[:900])
print("c2", json.dumps(j.get("c2"))[:600])
print("cpu_baseline", json.dumps(j.get("cpu_baseline"))[:900])
for leg in ("c2", "headline"):
    a = j.get("api", {}).get(leg, {})
    print(leg, "invoke", json.dumps(a.get("invoke_latency_ms")), "with encoder", json.dumps(a.get("invoke_latency_with_encoder_ms"))[:500])
PY
