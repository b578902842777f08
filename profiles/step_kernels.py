"""Kernels of ONE steady-state training step (between the last two optim_adamw_kernel launches of a rocprofv3 --kernel-trace run
of bench.py): per-kernel count / total / average for that step, then the launch sequence with durations.

  python profiles/step_kernels.py out/trace/t_results.db > profiles/r03_step_kernels.txt
"""
import sqlite3
import sys
from collections import OrderedDict

c = sqlite3.connect(sys.argv[1]).cursor()
rows = list(c.execute("select name, start, end from kernels order by start"))
short = lambda s: s.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:70]
ends = [i for i, r in enumerate(rows) if "optim_adamw_kernel" in r[0]]
a, b = ends[-2], ends[-1]
seg = rows[a + 1 : b + 1]
agg = OrderedDict()
for n, s, e in seg:
    d = agg.setdefault(short(n), [0, 0])
    d[0] += 1
    d[1] += e - s
print(f"# {len(seg)} kernels, sum of durations {sum(e - s for _, s, e in seg) / 1e6:.2f} ms, wall {(seg[-1][2] - rows[a][2]) / 1e6:.2f} ms")
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{n:5d} {t / 1e3:10.1f} us {t / n / 1e3:9.1f} avg  {k}")
print("# sequence")
t0 = rows[a][2]
for n, s, e in seg:
    print(f"{(s - t0) / 1e3:10.1f} {(e - s) / 1e3:9.1f}  {short(n)}")
