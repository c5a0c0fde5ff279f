#!/bin/bash
# config 2 (1M x 768) with two search contexts on two streams (--twin): kernel timeline
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; cd "$R"; O=$R/gpurun_out/r06_twin_c2; mkdir -p "$O"
QUIET="--no-c2 --no-c3 --no-c5 --no-persist --no-ingest --no-api --no-f32 --no-wide --no-pairs --no-cpu-baseline"
python3 bench.py --rows 1000000 --steps 200 --warmup 20 $QUIET --verify-queries 8 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('PLAIN ms/step', j['ms_per_step'])"
python3 bench.py --rows 1000000 --steps 200 --warmup 20 --twin $QUIET --verify-queries 8 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('TWIN ms/step', j['ms_per_step'])"
timeout 900 rocprofv3 --kernel-trace --output-format csv -d "$O/kt" -- python3 bench.py --rows 1000000 --steps 30 --twin $QUIET --verify-queries 8 > "$O/twin.json" 2> "$O/twin.err"
python3 - "$O" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/kt/*/*kernel_trace.csv')[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'rarc_scan_f16' in r['Kernel_Name']]
i0 = idx[len(idx) // 2]
base = int(rows[i0 - 4]['Start_Timestamp'])
for r in rows[i0 - 4:i0 + 22]:
    s, e = int(r['Start_Timestamp']) - base, int(r['End_Timestamp']) - base
    print(f"{s/1e3:9.1f} {e/1e3:9.1f} {(e-s)/1e3:7.1f}  q{r.get('Queue_Id','?')} {r['Kernel_Name'][:44]}")
PY
find "$O" -name "*.db" -delete 2>/dev/null
