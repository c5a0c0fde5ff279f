#!/bin/bash
# the int8-prefilter path with two pipelined contexts (RARC_PIPELINE_Q8=1) against one: the 12.5M-row shard step (exchange forced), 25M, 100M
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; cd "$R"
QUIET="--no-c2 --no-c3 --no-c5 --no-persist --no-ingest --no-api --no-f32 --no-wide --no-pairs --no-cpu-baseline"
for rows in ${@:-12500000 100000000}; do for mode in 0 1 0 1; do
  RARC_FORCE_DIST=1 RARC_PIPELINE_Q8=$mode python3 bench.py --rows $rows --steps $((rows > 50000000 ? 20 : 100)) --warmup 5 $QUIET --verify-queries 16 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('ROWS $rows q8-pipeline=$mode ms/step', j['ms_per_step'], 'scan ms', j['roofline']['scan_ms_per_pass'], 'exch', j['config']['exchange_ms_per_step'], j['config']['full_size_check'])"
done; done
