"""Per-kernel stats from a rocprofv3 --kernel-trace sqlite db: python tools/kstats.py <dir-or-db>"""
import sqlite3, sys, glob, os
path = sys.argv[1]
dbs = [path] if path.endswith(".db") else glob.glob(os.path.join(path, "**", "*.db"), recursive=True)
for db in dbs:
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
    ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
    q = f"select s.kernel_name, count(*), avg(d.end-d.start)/1000.0, min(d.end-d.start)/1000.0, max(d.end-d.start)/1000.0, sum(d.end-d.start)/1000.0 from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name order by 6 desc"
    print("%-64s %6s %10s %10s %10s %12s" % ("kernel", "calls", "avg_us", "min_us", "max_us", "total_us"))
    for r in c.execute(q):
        print("%-64s %6d %10.1f %10.1f %10.1f %12.1f" % (r[0][:64], r[1], r[2], r[3], r[4], r[5]))
