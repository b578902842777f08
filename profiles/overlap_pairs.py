"""Which kernels run BESIDE which in one steady-state training step (between the last two optim_adamw_kernel launches of a
rocprofv3 --kernel-trace run of bench.py): for every kernel family, its total duration and the part of it during which a kernel of
another family was running too.

  python profiles/overlap_pairs.py out/trace/t_results.db > profiles/r04_overlap_pairs.txt
"""
import sqlite3
import sys
from collections import defaultdict

c = sqlite3.connect(sys.argv[1]).cursor()
rows = list(c.execute("select name, start, end from kernels order by start"))
short = lambda s: s.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("<")[0][:40]
ends = [i for i, r in enumerate(rows) if "optim_adamw_kernel" in r[0]]
seg = [(short(n), s, e) for n, s, e in rows[ends[-2] + 1 : ends[-1] + 1]]
tot = defaultdict(float)
pair = defaultdict(float)
for i, (n, s, e) in enumerate(seg):
    tot[n] += e - s
    for m, s2, e2 in seg[i + 1 :]:
        if s2 >= e:
            break
        ov = min(e, e2) - s2
        if ov > 0:
            pair[(n, m)] += ov
            pair[(m, n)] += ov
wall = seg[-1][2] - seg[0][1]
busy, cur_s, cur_e = 0, None, None
for n, s, e in seg:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print(f"# wall {wall / 1e6:.2f} ms, busy {busy / 1e6:.2f} ms, sum of durations {sum(tot.values()) / 1e6:.2f} ms")
for n, t in sorted(tot.items(), key=lambda kv: -kv[1])[:24]:
    others = sorted(((m, v) for (a, m), v in pair.items() if a == n), key=lambda kv: -kv[1])[:3]
    print(f"{t / 1e3:10.1f} us  {n:36s} beside: " + ", ".join(f"{m} {v / 1e3:.0f}" for m, v in others))
