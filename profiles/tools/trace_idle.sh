#!/bin/bash
# kernel trace of a short bench.py run -> idle gaps of the main stream in one steady-state step (profiles/main_stream_idle.py) in gpurun_out/<name>.txt
name=${1:-main_idle}; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$root/gpurun_out" /tmp/prof/$name
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d /tmp/prof/$name -o t -- python3 "$root/bench.py" --steps 3 --warmup 2 --no-cpu-baseline --no-extra --timed-only "$@" > "$root/gpurun_out/$name.log" 2>&1
echo "trace rc=$?"
db=$(find /tmp/prof/$name -name '*_results.db' | head -1)
python3 "$root/profiles/main_stream_idle.py" "$db" > "$root/gpurun_out/$name.txt" && head -60 "$root/gpurun_out/$name.txt"
