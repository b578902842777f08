"""Per-kernel sums of arbitrary rocprofv3 PMC counters (one rocpd sqlite db per pass; several dbs may be given).

  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES \
            --kernel-trace -d out/pmc_sq_a -o p -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline
  python profiles/pmc_counters.py out/pmc_sq_a/p_results.db [more.db ...] > profiles/r02_mfma_counters.json

Per kernel: launches, average duration (ns, under the profiler: dispatches are serialised), and for each counter the
average value per launch.  Derived figures (MI355X_MICROARCH.md, rocprofv3 PMC slots / cycle constants):
  SQ_WAVE_CYCLES, SQ_WAIT_*, SQ_ACTIVE_INST_* count quad-cycles summed over waves;
  wait_any_frac      = SQ_WAIT_ANY / SQ_WAVE_CYCLES          (wave parked at s_waitcnt / s_barrier)
  wait_inst_frac     = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES     (issue stall: MFMA RAW / pipe busy)
  active_inst_frac   = SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES
  mfma_busy_frac     = SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CYCLES x 4 SIMDs / n_xcd-normalisation) -- reported raw as
                       SQ_VALU_MFMA_BUSY_CYCLES / (4 * SQ_BUSY_CU_CYCLES) when SQ_BUSY_CU_CYCLES is present.
"""
import json
import sqlite3
import sys


def short_name(name):
    s = name.replace("(anonymous namespace)::", "").replace("void ", "")
    if s.startswith("at::") or s.startswith("std::"):
        return s[:60]
    return s.split("(")[0]


res = {}
for path in sys.argv[1:]:
    c = sqlite3.connect(path).cursor()
    tables = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
    q = ("select kernel_name, counter_name, count(*), sum(value), avg(end - start) from counters_collection "
         "group by kernel_name, counter_name")
    for name, counter, n, tot, dur in c.execute(q):
        d = res.setdefault(short_name(name), {"launches": n, "avg_ns": dur, "counters": {}})
        d["counters"][counter] = tot / n
for d in res.values():
    k = d["counters"]
    wc = k.get("SQ_WAVE_CYCLES")
    if wc:
        for key, out in (("SQ_WAIT_ANY", "wait_any_frac"), ("SQ_WAIT_INST_ANY", "wait_inst_frac"), ("SQ_ACTIVE_INST_ANY", "active_inst_frac"),
                         ("SQ_ACTIVE_INST_LDS", "active_lds_frac"), ("SQ_WAIT_INST_LDS", "wait_inst_lds_frac"),
                         ("SQ_ACTIVE_INST_VALU", "active_valu_frac"), ("SQ_ACTIVE_INST_VMEM", "active_vmem_frac")):
            if key in k:
                d[out] = k[key] / wc
    if "SQ_VALU_MFMA_BUSY_CYCLES" in k and "SQ_BUSY_CU_CYCLES" in k and k["SQ_BUSY_CU_CYCLES"]:
        d["mfma_busy_per_cu_cycle"] = k["SQ_VALU_MFMA_BUSY_CYCLES"] / k["SQ_BUSY_CU_CYCLES"]
    if "SQ_VALU_MFMA_BUSY_CYCLES" in k and "GRBM_GUI_ACTIVE" in k and k["GRBM_GUI_ACTIVE"]:
        # GRBM_GUI_ACTIVE sums the 8 XCDs; MFMA busy cycles sum over 1024 SIMDs
        d["mfma_util"] = k["SQ_VALU_MFMA_BUSY_CYCLES"] / (k["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
    if "SQ_LDS_BANK_CONFLICT" in k and k.get("SQ_LDS_IDX_ACTIVE"):
        d["lds_conflict_frac"] = k["SQ_LDS_BANK_CONFLICT"] / k["SQ_LDS_IDX_ACTIVE"]
top = dict(sorted(res.items(), key=lambda kv: -kv[1]["launches"] * kv[1]["avg_ns"])[:20])
print(json.dumps({"note": "averages per launch; SQ_* cycle counters are quad-cycles summed over waves; durations are under the profiler (serialised dispatches)",
                  "kernels": top}, indent=1))
