#!/bin/bash
# rocprofv3 kernel trace of the inference path (eval forward + decode + weighted NMS); per-kernel CSV -> gpurun_out/<name>.csv
name=${1:-r02_infer_trace}
root=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$root/gpurun_out" /tmp/prof/$name
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof/$name -o t -- python3 "$root/profiles/tools/infer_time.py" > "$root/gpurun_out/$name.log" 2>&1
echo "trace $name rc=$?"
grep "sweeps/s" "$root/gpurun_out/$name.log"
db=$(find /tmp/prof/$name -name '*_results.db' | head -1)
python3 "$root/profiles/kernel_stats.py" "$db" > "$root/gpurun_out/$name.csv" && head -45 "$root/gpurun_out/$name.csv"
