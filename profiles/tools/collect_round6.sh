#!/bin/bash
# usage: collect_round6.sh <tag> [quick]   (on the GPU box, from the repo root)
# The judged evidence of a round.  Every profiler pass runs `bench.py --timed-only` (warm-up 1 + 2 timed steps, nothing else):
# exactly the configuration that produces ms_per_step, so the per-kernel averages are the in-run ones (roofline.achieved).
#   gpurun_out/<tag>_bench_kernel_stats.csv             rocprofv3 --kernel-trace --stats, per-kernel summary (two streams, as run)
#   gpurun_out/<tag>_bench_kernel_stats_one_stream.csv  the same under RV3D_OVERLAP=off (roofline.isolated)
#   gpurun_out/<tag>_pmc_traffic.json                   FETCH_SIZE / WRITE_SIZE passes (separate): headline rows + section "rv_waymo"
#   gpurun_out/<tag>_mfma_counters.json                 SQ wait / issue / MFMA-busy / LDS / GRBM passes (separate), per kernel  [skipped with "quick"]
tag=$1
quick=$2
root=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$root/gpurun_out"
cd /tmp && export TMPDIR=/tmp
STEPS=3   # warm-up 1 + timed 2
AV2="--steps 2 --warmup 1 --timed-only"
WAYMO="--steps 2 --warmup 1 --timed-only --widths rv-waymo --features 6 --classes 3 --width 2656"
pass() {  # pass <dir> <bench args> -- <rocprofv3 args...>
  d=/tmp/prof/${tag}_$1; shift
  bargs=$1; shift
  mkdir -p $d
  rocprofv3 "$@" -d $d -o p -- python3 "$root/bench.py" $bargs > $d/log.txt 2>&1
  echo "$d rc=$? $(grep -c . $d/log.txt) log lines"
}
db() { find /tmp/prof/${tag}_$1 -name '*_results.db' | head -1; }
pass trace "$AV2" --kernel-trace --stats
python3 "$root/profiles/kernel_stats.py" "$(db trace)" > "$root/gpurun_out/${tag}_bench_kernel_stats.csv"
RV3D_OVERLAP=off pass trace1 "$AV2" --kernel-trace --stats
python3 "$root/profiles/kernel_stats.py" "$(db trace1)" > "$root/gpurun_out/${tag}_bench_kernel_stats_one_stream.csv"
pass fetch "$AV2" --pmc FETCH_SIZE --kernel-trace
pass write "$AV2" --pmc WRITE_SIZE --kernel-trace
python3 "$root/profiles/pmc_traffic.py" "$(db fetch)" "$(db write)" --steps=$STEPS > "$root/gpurun_out/${tag}_pmc_traffic_av2.json"
pass wfetch "$WAYMO" --pmc FETCH_SIZE --kernel-trace
pass wwrite "$WAYMO" --pmc WRITE_SIZE --kernel-trace
python3 "$root/profiles/pmc_traffic.py" "$(db wfetch)" "$(db wwrite)" --steps=$STEPS --section=rv_waymo --into="$root/gpurun_out/${tag}_pmc_traffic_av2.json" > "$root/gpurun_out/${tag}_pmc_traffic.json"
pass wtrace "$WAYMO" --kernel-trace --stats
python3 "$root/profiles/kernel_stats.py" "$(db wtrace)" > "$root/gpurun_out/${tag}_bench_kernel_stats_waymo.csv"
if [ -z "$quick" ]; then
  pass sqa "$AV2" --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 --kernel-trace
  pass sqb "$AV2" --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --kernel-trace
  pass grbm "$AV2" --pmc GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_VALU_MFMA_COEXEC_CYCLES --kernel-trace
  python3 "$root/profiles/pmc_counters.py" "$(db sqa)" "$(db sqb)" "$(db grbm)" > "$root/gpurun_out/${tag}_mfma_counters.json"
fi
head -12 "$root/gpurun_out/${tag}_bench_kernel_stats.csv"
