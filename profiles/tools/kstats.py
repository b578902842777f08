"""Per-kernel totals of a rocprofv3 --kernel-trace rocpd database:  kstats.py <db> <steps> [rows]."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); steps = float(sys.argv[2])
c = db.cursor()
rows = list(c.execute("select name, count(*), sum(end-start)/1e6, avg(end-start)/1e3 from kernels group by name order by 3 desc limit %d" % int(sys.argv[3] if len(sys.argv) > 3 else 25)))
tot = list(c.execute("select sum(end-start)/1e6 from kernels"))[0][0]
for r in rows: print(f"{r[0][:80]:80s} {r[1]/steps:7.1f}/step {r[2]/steps:9.2f} ms/step {r[3]:9.1f} us")
print("total/step", tot / steps)
