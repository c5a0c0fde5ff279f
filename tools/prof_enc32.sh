#!/bin/bash
# kernel trace of the bge-large forward at 32 sequences x 32 tokens (one rank's share of a 256-query batch on 8 GPUs)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_enc32; mkdir -p $O; cd $R
export PROBE_SEQS=32
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 tools/enc_only.py > $O/kt.log 2>&1
f=$(ls -t $O/kt/*/*kernel_stats.csv | head -1); cp $f $O/enc32_kernel_stats.csv; grep "rarc_" $f | cut -c1-130
t=$(ls -t $O/kt/*/*kernel_trace.csv | head -1)
python3 - "$t" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "rarc_" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last = rows[-200:]
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in last)
span = int(last[-1]["End_Timestamp"]) - int(last[0]["Start_Timestamp"])
gaps = [int(b["Start_Timestamp"]) - int(a["End_Timestamp"]) for a, b in zip(last, last[1:])]
print(f"last {len(last)} kernels: span {span/1e3:.1f} us, busy {busy/1e3:.1f} us, mean gap {sum(gaps)/len(gaps)/1e3:.2f} us")
PY
find $O -name "*.db" -delete
