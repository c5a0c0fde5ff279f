"""Summarise tools/power_sample.sh output (<out>.smi: rocm-smi socket power + sclk every ~0.25 s):
    python3 tools/power_summary.py <out> [label]"""
import re, statistics, sys
rows, p, c = [], None, None
for ln in open(sys.argv[1] + ".smi"):
    m = re.search(r"sclk clock level: \S+ \((\d+)Mhz\)", ln)
    if m: c = int(m.group(1))
    m = re.search(r"Package Power \(W\): ([\d.]+)", ln)
    if m: p = float(m.group(1))
    if ln.startswith("---") and p is not None:
        rows.append((p, c or 0)); p = c = None
top = max(r[0] for r in rows)
busy = [r for r in rows if r[0] >= 0.93 * top]
label = sys.argv[2] if len(sys.argv) > 2 else sys.argv[1]
print(f"== {label}: {len(rows)} samples, {len(busy)} in steady state (within 7 % of the highest power): socket power "
      f"{min(b[0] for b in busy):.0f}-{max(b[0] for b in busy):.0f} W (median {statistics.median(b[0] for b in busy):.0f}), "
      f"sclk {min(b[1] for b in busy)}-{max(b[1] for b in busy)} MHz (median {statistics.median(b[1] for b in busy):.0f})")
print("   all samples (W, MHz):", " ".join(f"({p:.0f},{c})" for p, c in rows))
