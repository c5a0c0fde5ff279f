#!/bin/bash
# round 6, the judged measurements on one box: whole GPU suite (raw log), the driver's command unprofiled and under rocprofv3 --kernel-trace --stats,
# config 2 and the 12.5M-row shard step with per-kernel traces, a PMC pass (FETCH_SIZE) of the headline scan.   usage: tools/r06_final.sh [parts...]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd "$R"; O=$R/gpurun_out/r06_final; mkdir -p "$O"
parts=${@:-tests bench prof c2 shard pmc}
for part in $parts; do case $part in
  tests) timeout 1500 python3 -m pytest tests -m gpu -x -q -rs > "$O/pytest_gpu.log" 2>&1; tail -3 "$O/pytest_gpu.log";;
  bench) s=$(date +%s); timeout 1500 python3 bench.py > "$O/bench_driver_style.json" 2> "$O/bench_driver_style.err"; echo "driver-style bench rc=$? wall=$(( $(date +%s) - s )) s"
         python3 -c "import json; j=json.load(open('$O/bench_driver_style.json')); print(json.dumps(j['summary']))";;
  prof)  tools/prof.sh bench | tail -20;;
  c2)    tools/prof.sh c2;;
  shard) tools/prof.sh shard 12500000 --steps 100 --warmup 10;;
  pmc)   tools/prof.sh pmc 100000000 768 f16;;
esac; done
find "$R/gpurun_out" -name "*.db" -delete 2>/dev/null
