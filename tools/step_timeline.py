"""One steady-state search step out of a rocprofv3 kernel trace: every kernel between two consecutive
rarc_prep_queries_kernel launches, with its duration and the gap since the previous kernel ended.
    python3 tools/step_timeline.py <..._kernel_trace.csv> [which step from the end, default 3]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
marks = [i for i, r in enumerate(rows) if "rarc_prep_queries_kernel" in r["Kernel_Name"]]
a, b = marks[-back - 1], marks[-back]
prev_end, busy, gaps = None, 0, 0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = 0 if prev_end is None else s - prev_end
    name = re.sub(r"^void |\(.*", "", r["Kernel_Name"])[:56]
    wgs = int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))
    print(f"{name:58s} wgs {wgs:6d}  {(e - s) / 1e3:9.1f} us   gap {gap / 1e3:7.1f} us")
    busy += e - s; gaps += max(gap, 0); prev_end = max(e, prev_end or e)
span = int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"])
print(f"step: {span / 1e3:.1f} us = kernels {busy / 1e3:.1f} + gaps {gaps / 1e3:.1f} (+ gap to the next step's first kernel), {b - a} launches")
