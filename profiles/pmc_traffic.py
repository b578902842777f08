"""Per-kernel HBM traffic from two separate rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) over bench.py.

  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d out/pmc_FETCH_SIZE -o p -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline
  rocprofv3 --pmc WRITE_SIZE --kernel-trace -d out/pmc_WRITE_SIZE -o p -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline
  python profiles/pmc_traffic.py out/pmc_FETCH_SIZE/p_results.db out/pmc_WRITE_SIZE/p_results.db > profiles/r01_pmc_traffic.json

Corrections (MI355X_MICROARCH.md, HBM section): both counters are in KiB; on gfx950 FETCH_SIZE reports exactly half of
the bytes of wide (16 B / lane) coalesced reads, which is how every kernel here reads -> fetched bytes = 2 x FETCH_SIZE;
WRITE_SIZE is exact for 16-byte stores.  Infinity-Cache hits are counted (these are L2 memory-side requests).
"""
import json
import sqlite3
import sys


def per_kernel(path):
    c = sqlite3.connect(path).cursor()
    out = {}
    for name, n, tot in c.execute("select kernel_name, count(*), sum(value) from counters_collection group by kernel_name"):
        out[name] = (n, tot)
    return out


args = [a for a in sys.argv[1:] if not a.startswith("--")]
opts = dict(a[2:].split("=", 1) for a in sys.argv[1:] if a.startswith("--") and "=" in a)
fetch, write = per_kernel(args[0]), per_kernel(args[1])
res = {}
for name in fetch:
    n, f = fetch[name]
    w = write.get(name, (n, 0.0))[1] * n / max(write.get(name, (n, 0.0))[0], 1)
    short = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    res[short] = {"launches": n, "fetch_bytes_per_launch": 2.0 * 1024.0 * f / n, "write_bytes_per_launch": 1024.0 * w / n,
                  "hbm_bytes_per_launch": (2.0 * 1024.0 * f + 1024.0 * w) / n, "total_gb": (2.0 * 1024.0 * f + 1024.0 * w) / 1e9}
total_all = sum(v["total_gb"] for v in res.values())
res = dict(sorted(res.items(), key=lambda kv: -kv[1]["total_gb"])[:40])
out = {"corrections": "FETCH_SIZE KiB x 1024 x 2 (gfx950 wide-read undercount), WRITE_SIZE KiB x 1024", "kernels": res,
       # every kernel of the run (not only the rows kept above) and the training steps the profiled command ran (--steps=N):
       # bench.py's whole_step.traffic_ratio = total_gb_all_kernels / steps_profiled over the algorithmic bytes of a step
       "total_gb_all_kernels": total_all, "steps_profiled": int(opts["steps"]) if "steps" in opts else None}
if "into" in opts:  # --section=NAME --into=FILE: store this run as a named section of an existing file (the rv-waymo passes)
    with open(opts["into"]) as fh:
        top = json.load(fh)
    top[opts["section"]] = out
    out = top
print(json.dumps(out, indent=1))
