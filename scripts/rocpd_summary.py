#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd SQLite result (kernel trace): per-kernel calls, total/avg/min/max duration.
Usage: rocpd_summary.py results.db [> profiles/xyz_kernel_stats.txt]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
cols = [r[1] for r in cur.execute(f"pragma table_info({disp})")]
scol = [r[1] for r in cur.execute(f"pragma table_info({sym})")]
namecol = "kernel_name" if "kernel_name" in scol else ("display_name" if "display_name" in scol else scol[-1])
q = f"""select s.{namecol}, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start)
        from {disp} d join {sym} s on d.kernel_id = s.id group by s.{namecol} order by 3 desc"""
rows = list(cur.execute(q))
tot = sum(r[2] for r in rows) or 1
print(f"{'kernel':<64s} {'calls':>7s} {'total_ms':>12s} {'avg_us':>12s} {'min_us':>10s} {'max_us':>12s} {'pct':>6s}")
for name, n, t, a, mn, mx in rows:
    short = name.split("(")[0]
    print(f"{short[:64]:<64s} {n:7d} {t/1e6:12.3f} {a/1e3:12.2f} {mn/1e3:10.2f} {mx/1e3:12.2f} {100*t/tot:6.2f}")
if len(sys.argv) > 2 and sys.argv[2] == "--dispatches":
    k = sys.argv[3]
    for r in cur.execute(f"select d.start, d.end-d.start from {disp} d join {sym} s on d.kernel_id=s.id where s.{namecol} like ? order by d.start", (f"%{k}%",)):
        print(r[1] / 1e3)
