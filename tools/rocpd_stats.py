"""Per-kernel summary (calls, total, average) of a rocprofv3 rocpd database: python tools/rocpd_stats.py results.db [steps]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]; ks = [t for t in tabs if "info_kernel_symbol" in t][0]
rows = list(cur.execute(f"select s.kernel_name, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start) from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name order by 3 desc"))
tot = sum(r[2] for r in rows)
print('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs","ms_per_step"')
for n, c, t, a, mn, mx in rows:
    print('"%s",%d,%d,%.1f,%.2f,%d,%d,%.3f' % (n[:110], c, t, a, 100.0 * t / tot, mn, mx, t / steps * 1e-6))
print('"TOTAL",,%d,,,,,%.3f' % (tot, tot / steps * 1e-6))
