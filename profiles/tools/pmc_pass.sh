#!/bin/bash
# usage: pmc_pass.sh <name> <counters...>
# One rocprofv3 PMC pass over a short bench.py run; the rocpd db stays in /tmp on the GPU box, the per-kernel
# aggregate (profiles/pmc_counters.py) goes to gpurun_out/<name>.json and the run's log to gpurun_out/<name>.log.
name=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$root/gpurun_out" /tmp/prof/$name
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc "$@" --kernel-trace -d /tmp/prof/$name -o p -- python3 "$root/bench.py" --steps 2 --warmup 1 --no-cpu-baseline ${PMC_BENCH_ARGS} > "$root/gpurun_out/$name.log" 2>&1
echo "pass $name rc=$?"
db=$(find /tmp/prof/$name -name '*_results.db' | head -1)
python3 "$root/profiles/pmc_counters.py" "$db" > "$root/gpurun_out/$name.json" && echo "wrote gpurun_out/$name.json"
