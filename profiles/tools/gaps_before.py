"""Idle time in front of each kernel family in a rocprofv3 kernel trace (sqlite .db): for every dispatch, start - max(end of all earlier dispatches).

    python profiles/tools/gaps_before.py trace_results.db [name-substring ...]
"""
import collections, sqlite3, sys

db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, start, end from kernels order by start"))
keys = sys.argv[2:] or ["tapconv6", "bn_reduce_finalize", "ew_combine_rows", "tapconv4"]
prev_end, prev_name = 0, ""
gaps = collections.defaultdict(list)
for name, s, e in rows:
    if prev_end:
        for k in keys:
            if k in name:
                gaps[(k, prev_name.split("(")[0].split("::")[-1][:40])].append((s - prev_end) / 1e3)
    if e > prev_end:
        prev_end, prev_name = e, name
for (k, p), v in sorted(gaps.items(), key=lambda kv: -len(kv[1])):
    if len(v) >= 6:
        v.sort()
        print(f"{k:22s} after {p:42s} n={len(v):4d}  median gap {v[len(v) // 2]:7.2f} us   mean {sum(v) / len(v):7.2f}")
