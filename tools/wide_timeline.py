"""One rarc_search_wide call out of a rocprofv3 kernel trace, kernels aggregated by name (calls, total, mean) + the finalize.
    python3 tools/wide_timeline.py <..._kernel_trace.csv> [which call from the end, default 2]"""
import collections, csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "wide_eps_kernel" in r["Kernel_Name"]]
print(len(marks), "wide searches in the trace")
for back in ([int(sys.argv[2])] if len(sys.argv) > 2 else range(1, min(len(marks), 40), 1)):
    a = marks[-back - 1] if back < len(marks) else None
    if a is None: break
    b = marks[-back]
    agg = collections.OrderedDict()
    for r in rows[a:b]:
        name = re.sub(r"^void |\(.*", "", r["Kernel_Name"])[:40]
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        c = agg.setdefault(name, [0, 0]); c[0] += 1; c[1] += d
    span = int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"])
    print(f"call -{back}: span {span / 1e3:9.1f} us  " + "  ".join(f"{n.split('_kernel')[0][-18:]}:{c[0]}x{c[1] / 1e3:.0f}us" for n, c in agg.items()))
