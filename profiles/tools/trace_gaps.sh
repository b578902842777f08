#!/bin/bash
# kernel trace of a short bench.py run -> GPU idle analysis (profiles/gap_analysis.py) in gpurun_out/<name>.txt
name=${1:-gaps}; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$root/gpurun_out" /tmp/prof/$name
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d /tmp/prof/$name -o t -- python3 "$root/bench.py" --steps 4 --warmup 2 --no-cpu-baseline --no-extra "$@" > "$root/gpurun_out/$name.log" 2>&1
echo "trace rc=$?"
db=$(find /tmp/prof/$name -name '*_results.db' | head -1)
python3 "$root/profiles/gap_analysis.py" "$db" 3 > "$root/gpurun_out/$name.txt" && cat "$root/gpurun_out/$name.txt"
