"""GPU idle time inside the timed steps of a rocprofv3 --kernel-trace run of bench.py (rocpd sqlite output).

  python profiles/gap_analysis.py out/trace/t_results.db [n_steps]

Takes the last `n_steps` optimiser launches (optim_adamw_kernel ends a step) as step boundaries, and for each step prints
wall time, the union of kernel-busy intervals (all streams), and the largest gaps with the kernels on either side.
"""
import sqlite3
import sys
from collections import Counter

c = sqlite3.connect(sys.argv[1]).cursor()
n_steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
rows = list(c.execute("select name, start, end from kernels order by start"))
short = lambda s: s.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:48]
ends = [i for i, r in enumerate(rows) if "optim_adamw_kernel" in r[0]]
if len(ends) < n_steps + 1:
    sys.exit("not enough steps in the trace")
for a, b in zip(ends[-n_steps - 1 : -1], ends[-n_steps:]):
    seg = rows[a + 1 : b + 1]
    t0, t1 = rows[a][2], seg[-1][2]
    busy, cur_end, gaps = 0, t0, []
    prev = rows[a][0]
    for name, s, e in seg:
        if s > cur_end:
            gaps.append((s - cur_end, short(prev), short(name)))
            busy += e - s
            cur_end = e
            prev = name
        else:
            if e > cur_end:
                busy += e - cur_end
                cur_end = e
                prev = name
    wall = t1 - t0
    print(f"step: wall {wall / 1e6:.2f} ms, busy (union over streams) {busy / 1e6:.2f} ms, idle {(wall - busy) / 1e6:.2f} ms in {len(gaps)} gaps, "
          f"{len(seg)} kernels, sum of kernel durations {sum(e - s for _, s, e in seg) / 1e6:.2f} ms")
    hist = Counter()
    for g, p, n in gaps:
        hist[min(int(g / 1e3) // 5 * 5, 100)] += g
    print("  idle by gap length (us bucket -> ms):", {k: round(v / 1e6, 2) for k, v in sorted(hist.items())})
    by_prev = Counter()
    for g, p, n in gaps:
        by_prev[(p, n)] += g
    for (p, n), g in by_prev.most_common(14):
        print(f"  {g / 1e3:8.1f} us  after {p}  before {n}")
