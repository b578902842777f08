#!/bin/bash
# kernel trace of a short bench.py run -> profiles/overlap_pairs.py summary in gpurun_out/<name>.txt (env is passed through)
name=${1:-overlap}; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$root/gpurun_out" /tmp/prof/$name
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d /tmp/prof/$name -o t -- python3 "$root/bench.py" --steps 3 --warmup 2 --no-cpu-baseline --no-extra "$@" > "$root/gpurun_out/$name.log" 2>&1
echo "trace rc=$?"
db=$(find /tmp/prof/$name -name '*_results.db' | head -1)
python3 "$root/profiles/overlap_pairs.py" "$db" > "$root/gpurun_out/$name.txt" && head -30 "$root/gpurun_out/$name.txt"
