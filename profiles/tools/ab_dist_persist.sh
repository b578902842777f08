#!/bin/bash
# A/B for a multi-GPU node: leave CUs free of the persistent tap-conv workgroups (one per CU, ~150 KB of LDS each) so that
# RCCL's kernels (the SyncBN all-reduces on the critical path, DDP's bucket all-reduces beside the backward) find a CU at once.
#   usage: profiles/tools/ab_dist_persist.sh <n_gpus>        (from the repo root, on the node)
# RV3D_TC_PERSIST = workgroups of a persistent tapconv5 / tapconv4 launch (default 256 = every CU; multiples of 8: XCD-aligned).
# On ONE GPU with a one-rank process group (RV3D_FORCE_DIST=1 RV3D_SYNC_WORLD1=1) the cost of giving CUs up is measurable, the
# benefit is not: profiles/r03_syncbn_collectives.md.
n=${1:-8}
for p in 256 248 240 224; do
  for rep in 1 2; do
    if [ "$n" = 1 ]; then
      RV3D_TC_PERSIST=$p RV3D_FORCE_DIST=1 RV3D_SYNC_WORLD1=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29690 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 \
        python bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-extra 2>/dev/null
    else
      RV3D_TC_PERSIST=$p python -m torch.distributed.run --nnodes=1 --nproc-per-node "$n" --master-addr 127.0.0.1 --master-port 29690 \
        bench.py --gpus "$n" --steps 10 --warmup 3 2>/dev/null
    fi | python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('persist_blocks=$p rep=$rep n_gpus=%d: %.2f ms/step, %.2f sweeps/s, sync collectives/step %s' % (d['n_gpus'], d['ms_per_step'], d['value'], d['config']['collectives']['per_step']['sync_bn_all_reduce']['calls']))"
  done
done
