"""Per-kernel summary (calls, total, average, share) of a rocprofv3 --kernel-trace run (rocpd sqlite output).

  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --stats -d out/trace -o t -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline
  python profiles/kernel_stats.py out/trace/t_results.db > profiles/r01_bench_kernel_stats.csv
"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1]).cursor()
rows = list(c.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name order by 3 desc"))
total = sum(r[2] for r in rows)
print("kernel,calls,total_ms,avg_us,min_us,max_us,percent")
for name, n, tot, avg, lo, hi in rows:
    short = name.replace("(anonymous namespace)::", "").replace("void ", "")
    short = short.split("(")[0] if not short.startswith("at::") else short[:80]
    print(f"\"{short}\",{n},{tot / 1e6:.3f},{avg / 1e3:.1f},{lo / 1e3:.1f},{hi / 1e3:.1f},{100.0 * tot / total:.2f}")
