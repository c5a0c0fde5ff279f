#!/bin/bash
# rarc_search_batch + pipelined contexts: the whole GPU suite, then config 2 with and without the pipeline, the 12.5M-row shard step, timelines
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; cd "$R"; O=$R/gpurun_out/r06_pipe; mkdir -p "$O"
if [ "${1:-full}" = full ]; then timeout 1500 python3 -m pytest tests -m gpu -x -q -rs > "$O/pytest.log" 2>&1; tail -4 "$O/pytest.log"; fi
QUIET="--no-c2 --no-c3 --no-c5 --no-persist --no-ingest --no-api --no-f32 --no-wide --no-pairs --no-cpu-baseline"
for mode in 0 1; do
  RARC_PIPELINE=$mode python3 bench.py --rows 1000000 --steps 200 --warmup 20 $QUIET --verify-queries 32 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('C2 RARC_PIPELINE=$mode ms/step', j['ms_per_step'], 'scan ms', j['roofline']['scan_ms_per_pass'], j['config']['full_size_check'])"
done
RARC_FORCE_DIST=1 python3 bench.py --rows 12500000 --steps 100 --warmup 10 $QUIET --verify-queries 32 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('SHARD 12.5M ms/step', j['ms_per_step'], 'scan ms', j['roofline']['scan_ms_per_pass'], 'frac', j['roofline']['frac'], j['config']['full_size_check'])"
timeout 900 rocprofv3 --kernel-trace --output-format csv -d "$O/kt" -- python3 bench.py --rows 1000000 --steps 30 $QUIET --verify-queries 8 > "$O/c2.json" 2> "$O/c2.err"
python3 - "$O" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/kt/*/*kernel_trace.csv')[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'rarc_scan_f16' in r['Kernel_Name']]
i0 = idx[len(idx) // 2]
base = int(rows[i0 - 4]['Start_Timestamp'])
for r in rows[i0 - 4:i0 + 18]:
    s, e = int(r['Start_Timestamp']) - base, int(r['End_Timestamp']) - base
    print(f"{s/1e3:9.1f} {e/1e3:9.1f} {(e-s)/1e3:7.1f}  q{r.get('Queue_Id','?')} {r['Kernel_Name'][:44]}")
PY
find "$O" -name "*.db" -delete 2>/dev/null
