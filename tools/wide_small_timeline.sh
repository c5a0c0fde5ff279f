cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out/prof_wide_small; rm -rf $O; mkdir -p $O; cd $R
PROBE_SIZES=100000 rocprofv3 --kernel-trace --output-format csv -d $O -- python3 tools/wide_size_sweep.py > $O/log.txt 2>&1
f=$(ls $O/*/*kernel_trace.csv | head -1)
python3 - $f <<'PY'
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1]))); rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "wide_eps_kernel" in r["Kernel_Name"]]
a, b = marks[-3], marks[-2]
prev = None
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = re.sub(r"^void |\(.*", "", r["Kernel_Name"])[:44]
    print(f"{name:46s} wgs {int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X'])):5d} {(e - s) / 1e3:8.1f} us  gap {0 if prev is None else (s - prev) / 1e3:6.1f}")
    prev = e
print("span", (int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"])) / 1e3)
PY
