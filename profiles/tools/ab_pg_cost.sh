#!/bin/bash
# What does an initialised RCCL process group cost a one-rank training step?  (same box, ms per step)
p() { python3 -c "import json,sys; j=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('$1', round(j['ms_per_step'],2))"; }
A="--steps 20 --warmup 5 --no-extra --no-cpu-baseline"
D="RV3D_FORCE_DIST=1 RV3D_SYNC_WORLD1=1"
python3 bench.py $A 2>/dev/null | p local
env $D RV3D_NO_GRADSYNC=1 python3 bench.py $A --no-sync-bn 2>/dev/null | p pg_only
env $D RV3D_NO_GRADSYNC=1 TORCH_NCCL_ENABLE_MONITORING=0 TORCH_NCCL_ASYNC_ERROR_HANDLING=0 python3 bench.py $A --no-sync-bn 2>/dev/null | p pg_only_nowatchdog
env $D RV3D_NO_GRADSYNC=1 GPU_MAX_HW_QUEUES=8 python3 bench.py $A --no-sync-bn 2>/dev/null | p pg_only_hwq8
env $D RV3D_NO_GRADSYNC=1 GPU_MAX_HW_QUEUES=2 python3 bench.py $A --no-sync-bn 2>/dev/null | p pg_only_hwq2
env $D RV3D_NO_GRADSYNC=1 RV3D_LAZY_PG=1 python3 bench.py $A --no-sync-bn 2>/dev/null | p pg_only_lazy_init
env $D RV3D_NO_GRADSYNC=1 RV3D_DIRECT_RCCL=1 python3 bench.py $A 2>/dev/null | p pg+syncbn_direct_no_gradsync
env $D RV3D_DIRECT_RCCL=1 python3 bench.py $A 2>/dev/null | p gradsync+syncbn_direct
python3 bench.py $A 2>/dev/null | p local
