#!/bin/bash
# the int8-prefilter scan on fewer workgroups than CUs, alone and with two pipelined contexts (the finalize of one batch on the CUs the
# other's scan leaves free): 12.5M-row shard step (exchange forced) and the 100M-row headline
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; cd "$R"
QUIET="--no-c2 --no-c3 --no-c5 --no-persist --no-ingest --no-api --no-f32 --no-wide --no-pairs --no-cpu-baseline"
for rows in ${ROWS:-12500000}; do for cfg in "256 0" "248 0" "240 0" "224 0" "248 1" "240 1" "232 1" "224 1" "256 0"; do set -- $cfg
  RARC_FORCE_DIST=1 RARC_SCAN_Q8_WGS=$1 RARC_PIPELINE_Q8=$2 python3 bench.py --rows $rows --steps $((rows > 50000000 ? 20 : 100)) --warmup 5 $QUIET --verify-queries 16 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('ROWS $rows wgs=$1 q8-pipeline=$2 ms/step', j['ms_per_step'], 'scan ms', j['roofline']['scan_ms_per_pass'], j['config']['full_size_check']['rows_beating_kth'], j['config']['full_size_check'].get('queries_differing'))"
done; done
