#!/bin/bash
# usage: collect_round.sh <tag>   (on the GPU box, from the repo root)
# The judged evidence of a round, all over the same bench.py command:
#   gpurun_out/<tag>_bench_kernel_stats.csv   rocprofv3 --kernel-trace --stats, per-kernel summary
#   gpurun_out/<tag>_pmc_traffic.json         FETCH_SIZE / WRITE_SIZE passes (separate), per-kernel HBM bytes per launch
#   gpurun_out/<tag>_mfma_counters.json       SQ wait / issue / MFMA-busy / LDS / GRBM passes (separate), per kernel
# Copy what is to be judged into profiles/.
tag=$1
root=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$root/gpurun_out"
cd /tmp && export TMPDIR=/tmp
pass() {  # pass <dir> <rocprofv3 args...>
  d=/tmp/prof/${tag}_$1; shift
  mkdir -p $d
  rocprofv3 "$@" -d $d -o p -- python3 "$root/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-extra > $d/log.txt 2>&1
  echo "$d rc=$? $(grep -c . $d/log.txt) log lines"
}
db() { find /tmp/prof/${tag}_$1 -name '*_results.db' | head -1; }
pass trace --kernel-trace --stats
python3 "$root/profiles/kernel_stats.py" "$(db trace)" > "$root/gpurun_out/${tag}_bench_kernel_stats.csv"
# the same with every kernel on ONE stream: the kernels' own durations (bench.py's roofline.achieved; with the weight gradients
# free-running on the side stream a launch's begin-to-end time includes the time its workgroups wait for CUs: roofline.live)
RV3D_OVERLAP=off pass trace1 --kernel-trace --stats
python3 "$root/profiles/kernel_stats.py" "$(db trace1)" > "$root/gpurun_out/${tag}_bench_kernel_stats_one_stream.csv"
pass fetch --pmc FETCH_SIZE --kernel-trace
pass write --pmc WRITE_SIZE --kernel-trace
python3 "$root/profiles/pmc_traffic.py" "$(db fetch)" "$(db write)" > "$root/gpurun_out/${tag}_pmc_traffic.json"
pass sqa --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 --kernel-trace
pass sqb --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --kernel-trace
pass grbm --pmc GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_VALU_MFMA_COEXEC_CYCLES --kernel-trace
python3 "$root/profiles/pmc_counters.py" "$(db sqa)" "$(db sqb)" "$(db grbm)" > "$root/gpurun_out/${tag}_mfma_counters.json"
head -12 "$root/gpurun_out/${tag}_bench_kernel_stats.csv"
