#!/bin/bash
# GPU_MAX_HW_QUEUES (HIP runtime: hardware queues per process, default 4) against the stream layout of the step (same box, ms per step)
p() { python3 -c "import json,sys; j=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('$1', round(j['ms_per_step'],2))"; }
A="--steps 20 --warmup 5 --no-extra --no-cpu-baseline"
D="RV3D_FORCE_DIST=1 RV3D_SYNC_WORLD1=1"
for rnd in 1 2; do
python3 bench.py $A 2>/dev/null | p local_default
GPU_MAX_HW_QUEUES=8 python3 bench.py $A 2>/dev/null | p local_hwq8
GPU_MAX_HW_QUEUES=2 python3 bench.py $A 2>/dev/null | p local_hwq2
env $D python3 bench.py $A 2>/dev/null | p dist_c10d_default
env $D GPU_MAX_HW_QUEUES=8 python3 bench.py $A 2>/dev/null | p dist_c10d_hwq8
env $D GPU_MAX_HW_QUEUES=2 python3 bench.py $A 2>/dev/null | p dist_c10d_hwq2
env $D GPU_MAX_HW_QUEUES=8 RV3D_DIRECT_RCCL=1 python3 bench.py $A 2>/dev/null | p dist_direct_hwq8
done
