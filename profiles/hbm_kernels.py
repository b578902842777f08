"""Table of the bandwidth-bound kernels of a step: HBM-side bytes (PMC passes: FETCH_SIZE x 2 + WRITE_SIZE per launch, profiles/pmc_traffic.py) over the
kernel's OWN duration (rocprofv3 --kernel-trace of the same command under RV3D_OVERLAP=off), per step.  The round-5 review's item 3: no kernel above
0.3 ms per step should sit below 5 TB/s.

    python profiles/hbm_kernels.py profiles/r06_bench_kernel_stats_one_stream.csv profiles/r06_pmc_traffic.json [steps=3] [section] > profiles/r06_hbm_kernels.md
"""
import csv
import json
import sys

MFMA = ("tapconv6_kernel", "tapconv5_kernel", "wgrad3_kernel")  # compute-bound families: listed for completeness, not held to the bandwidth bar


def main():
    stats, pmc = sys.argv[1], json.load(open(sys.argv[2]))
    steps = float(sys.argv[3]) if len(sys.argv) > 3 else 3.0
    section = sys.argv[4] if len(sys.argv) > 4 else None
    rows_pmc = pmc[section]["kernels"] if section else pmc["kernels"]
    rows = []
    with open(stats) as f:
        for r in csv.DictReader(f):
            name = r["kernel"]
            p = rows_pmc.get(name)
            if p is None:
                continue
            ms_step = float(r["total_ms"]) / steps
            gb_launch = p["hbm_bytes_per_launch"] / 1e9
            tbs = gb_launch / (float(r["avg_us"]) * 1e-6) / 1e3
            rows.append((ms_step, name, float(r["calls"]) / steps, float(r["avg_us"]), gb_launch, tbs))
    rows.sort(reverse=True)
    print("| kernel | launches / step | ms / step (one stream) | avg us | HBM-side GB / launch (PMC) | TB/s | |")
    print("|---|---|---|---|---|---|---|")
    for ms, name, n, us, gb, tbs in rows:
        if ms < 0.05:
            continue
        kind = "MFMA-bound" if name.startswith(MFMA) else ("**< 5 TB/s, > 0.3 ms**" if (tbs < 5.0 and ms > 0.3) else "")
        print(f"| `{name[:70]}` | {n:.1f} | {ms:.3f} | {us:.1f} | {gb:.3f} | {tbs:.2f} | {kind} |")


if __name__ == "__main__":
    main()
