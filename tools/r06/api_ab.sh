#!/bin/bash
# the plugin-surface legs at 1M rows with one search context (RARC_PIPELINE=0) and with two
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; cd "$R"
for mode in 0 1; do
  RARC_PIPELINE=$mode timeout 900 python3 bench.py --rows 1000000 --no-c3 --no-c5 --no-persist --no-ingest --no-f32 --no-wide --no-pairs --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; j=json.loads(sys.stdin.read()); a=j['api']['c2']
print('PIPELINE=$mode c2 ms/step', j['c2']['ms_per_step'], '| batch_invoke', a['batch_invoke_256']['value'], a['batch_invoke_256']['host_ms_per_call'], '| two callers', a['batch_invoke_256_two_callers']['value'], '| 2048', a['batch_invoke_2048']['value'], '| scores', a['batch_with_scores_256']['value'], '| threads', a['threads_256_invoke']['value'], '| coroutines', a['coroutines_256_ainvoke']['value'], '| invoke p50', a['invoke_latency_ms']['p50'], '| multipath', a['multipath_batch_invoke_256']['value'])"
done
