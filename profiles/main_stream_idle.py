"""Where the MAIN stream of a steady-state training step sits idle (a rocprofv3 --kernel-trace run of bench.py, last step between two
optim_adamw_kernel launches): kernels are split by hardware queue, the queue with the most kernel time is the main stream, and every gap between two
consecutive kernels of that queue is listed with what ran on the other queue(s) meanwhile.

  python profiles/main_stream_idle.py out/trace/t_results.db [min_gap_us=20]
"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
c = db.cursor()
min_gap = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
qcol = next((k for k in ("queue_id", "stream_id", "queue", "stream") if k in cols), None)
if qcol is None:
    print("columns:", cols)
    sys.exit("no queue / stream column in the kernels view")
rows = list(c.execute(f"select name, start, end, {qcol} from kernels order by start"))
short = lambda s: s.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]
ends = [i for i, r in enumerate(rows) if "optim_adamw_kernel" in r[0]]
a, b = ends[-2], ends[-1]
seg = rows[a + 1: b + 1]
t0 = rows[a][2]
by_q = {}
for n, s, e, q in seg:
    by_q.setdefault(q, []).append((s, e, n))
busy = {q: sum(e - s for s, e, _ in v) for q, v in by_q.items()}
main = max(busy, key=busy.get)
wall = seg[-1][2] - t0
print(f"# step wall {wall / 1e6:.2f} ms; queues ({qcol}): " + ", ".join(f"{q}: {len(v)} kernels, {busy[q] / 1e6:.2f} ms busy" for q, v in by_q.items()) + f"; main = {main}")
others = sorted((s, e, n) for q, v in by_q.items() if q != main for s, e, n in v)
mk = sorted(by_q[main])
gaps = []
prev_end, prev_name = t0, "optim_adamw_kernel (previous step)"
for s, e, n in mk:
    if s > prev_end:
        gaps.append((s - prev_end, prev_end, s, prev_name, n))
    if e > prev_end:
        prev_end, prev_name = e, n
total = sum(g[0] for g in gaps)
small = sum(g[0] for g in gaps if g[0] < min_gap * 1e3)
print(f"# main-stream idle {total / 1e6:.2f} ms in {len(gaps)} gaps; {small / 1e6:.2f} ms of it in gaps below {min_gap:.0f} us (kernel boundaries)")
print("# gaps >= %.0f us: start(us) gap(us)  after -> before   | other queues busy during the gap (us)" % min_gap)
for g, s0, s1, pn, nn in gaps:
    if g < min_gap * 1e3:
        continue
    ob = sum(max(0, min(e, s1) - max(s, s0)) for s, e, _ in others)
    names = sorted({short(n) for s, e, n in others if min(e, s1) > max(s, s0)})
    print(f"{(s0 - t0) / 1e3:10.1f} {g / 1e3:8.1f}  {short(pn)} -> {short(nn)}   | {ob / 1e3:.1f} {', '.join(names)[:120]}")

# ---- the other direction: when is the SIDE stream idle while the main stream runs, and what does the main stream run then
mfma = ("tapconv6_kernel", "tapconv5_kernel", "wgrad3_kernel", "wgrad2_kernel", "tapconv4_kernel", "tapconv2_kernel", "tapconv_kernel", "pointwise_kernel")
side_iv = sorted((s, e) for s, e, _ in others)
merged = []
for s, e in side_iv:
    if merged and s <= merged[-1][1]:
        merged[-1][1] = max(merged[-1][1], e)
    else:
        merged.append([s, e])
def side_busy(s0, s1):
    return sum(max(0, min(e, s1) - max(s, s0)) for s, e in merged)
alone = {"mfma": 0.0, "hbm": 0.0}
shared = {"mfma": 0.0, "hbm": 0.0}
for s, e, n in mk:
    k = "mfma" if short(n).startswith(mfma) else "hbm"
    b = side_busy(s, e)
    shared[k] += b
    alone[k] += (e - s) - b
print(f"# main-stream kernel time WITH the side stream busy: conv-type {shared['mfma'] / 1e6:.2f} ms, bandwidth-type {shared['hbm'] / 1e6:.2f} ms; "
      f"with the side stream IDLE: conv-type {alone['mfma'] / 1e6:.2f} ms, bandwidth-type {alone['hbm'] / 1e6:.2f} ms")
print("# stretches of >= 300 us with the side stream idle: start(us) length(us): what the main stream ran")
edges = [t0] + [x for iv in merged for x in iv] + [seg[-1][2]]
for i in range(0, len(edges), 2):
    s0, s1 = edges[i], edges[i + 1]
    if s1 - s0 < 300e3:
        continue
    ran = {}
    for s, e, n in mk:
        o = max(0, min(e, s1) - max(s, s0))
        if o > 0:
            ran[short(n)] = ran.get(short(n), 0) + o
    top = ", ".join(f"{k} {v / 1e3:.0f}" for k, v in sorted(ran.items(), key=lambda kv: -kv[1])[:5])
    print(f"{(s0 - t0) / 1e3:10.1f} {(s1 - s0) / 1e3:8.1f}: {top}")
