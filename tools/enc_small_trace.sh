#!/bin/bash
# per-kernel time of the fp32-class bge-large forward at small batches (PROBE_SEQS sequences of 32 tokens): rocprofv3 kernel stats
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd "$R"
for n in ${@:-1 32}; do
  O=$R/gpurun_out/prof_enc_small/n$n; rm -rf "$O"; mkdir -p "$O"
  PROBE_SEQS=$n RARC_ENC_PRECISION=fp32 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$O" -- python3 tools/enc_only.py > "$O/run.log" 2>&1
  echo "== $n sequences x 32 tokens (6 forwards of 24 layers)"; grep ENC "$O/run.log"
  f=$(ls "$O"/*/*kernel_stats.csv | head -1); python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(int(r["TotalDurationNs"]) for r in rows)
for r in rows[:14]:
    print(f'{r["Name"][:74]:74s} calls {int(r["Calls"]):5d}  avg {float(r["AverageNs"]) / 1e3:7.1f} us  {int(r["TotalDurationNs"]) / tot * 100:5.1f} %')
print(f"kernel time per forward: {tot / 6 / 1e6:.3f} ms")
PY
done
