#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; cd "$R"
QUIET="--no-c2 --no-c3 --no-c5 --no-persist --no-ingest --no-api --no-f32 --no-wide --no-pairs --no-cpu-baseline"
for rep in 1 2; do
for mode in "0 1" "1 1" "1 0"; do
  set -- $mode
  RARC_PIPELINE=$1 RARC_PIPELINE_GATE=$2 python3 bench.py --rows 1000000 --steps 400 --warmup 20 $QUIET --verify-queries 8 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('C2 pipeline=$1 gate=$2 ms/step', j['ms_per_step'], 'scan ms', j['roofline']['scan_ms_per_pass'], j['config']['full_size_check'])"
done; done
