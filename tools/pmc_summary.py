"""Summarise rocprofv3 --pmc output (sqlite .db) per kernel: mean counter value per dispatch."""
import sqlite3, sys, glob, os
root = sys.argv[1]
for db in glob.glob(os.path.join(root, "**", "*.db"), recursive=True):
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    pmc = [t for t in tabs if t.startswith("rocpd_pmc_event")]
    info = [t for t in tabs if t.startswith("rocpd_info_pmc")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")]
    if not (pmc and info and kd and ks):
        print(db, "tables:", sorted(set(t.split('_0000')[0] for t in tabs))); continue
    q = f"""select s.kernel_name, i.name, count(*), avg(e.value) from {pmc[0]} e
            join {info[0]} i on e.pmc_id = i.id join {kd[0]} d on e.event_id = d.event_id
            join {ks[0]} s on d.kernel_id = s.id group by s.kernel_name, i.name"""
    for r in c.execute(q):
        if "rarc_scan" in r[0] or len(sys.argv) > 2:
            print("%-40s %-34s n=%3d mean=%.4g" % (r[0][:40], r[1], r[2], r[3]))
