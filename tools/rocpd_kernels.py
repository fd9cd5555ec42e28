"""Per-kernel launch counts and durations from rocprofv3's default (rocpd / SQLite) output:
rocpd_kernels.py <dir or .db> [name filter]."""
import sqlite3, glob, os, sys

paths = []
for a in sys.argv[1:2]:
    paths += [a] if a.endswith(".db") else sorted(glob.glob(os.path.join(a, "**", "*_results.db"), recursive=True))
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for d in paths:
    c = sqlite3.connect(d)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    rows = c.execute(f"select s.kernel_name, count(*), avg(k.end-k.start), min(k.end-k.start) from {kd} k join {ks} s "
                     f"on k.kernel_id=s.id group by s.kernel_name order by sum(k.end-k.start) desc").fetchall()
    print("==", d)
    for n, cnt, avg, mn in rows:
        if flt in n:
            print(f"   {n[:70]:70s} x{cnt:4d} avg {avg/1e3:9.1f} us  min {mn/1e3:9.1f}")
