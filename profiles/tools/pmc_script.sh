#!/bin/bash
# usage: pmc_script.sh <name> <script.py> <counters...>  -- one rocprofv3 PMC pass over a python micro-benchmark; per-kernel
# aggregate (profiles/pmc_counters.py) -> gpurun_out/<name>.json
name=$1; script=$2; shift; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$root/gpurun_out" /tmp/prof/$name
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc "$@" --kernel-trace -d /tmp/prof/$name -o p -- python3 "$root/$script" > "$root/gpurun_out/$name.log" 2>&1
echo "pass $name rc=$?"
db=$(find /tmp/prof/$name -name '*_results.db' | head -1)
python3 "$root/profiles/pmc_counters.py" "$db" > "$root/gpurun_out/$name.json" && echo "wrote gpurun_out/$name.json"
