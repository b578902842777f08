#!/bin/bash
# usage: trace_bench.sh <name> [bench args]  -- rocprofv3 kernel trace of a short bench.py run; per-kernel CSV -> gpurun_out/<name>.csv
name=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$root/gpurun_out" /tmp/prof/$name
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof/$name -o t -- python3 "$root/bench.py" --steps 5 --warmup 2 --no-cpu-baseline "$@" > "$root/gpurun_out/$name.log" 2>&1
echo "trace $name rc=$?"
db=$(find /tmp/prof/$name -name '*_results.db' | head -1)
python3 "$root/profiles/kernel_stats.py" "$db" > "$root/gpurun_out/$name.csv" && head -40 "$root/gpurun_out/$name.csv"
