export MASTER_ADDR=127.0.0.1 MASTER_PORT=29671 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1
A="--gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-extra"
run() { python bench.py $A "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$TAG', round(d['ms_per_step'],2), d['config']['loss'], d['config']['collectives']['per_step']['sync_bn_all_reduce'])"; }
TAG=local run
TAG=ddp_only RV3D_FORCE_DIST=1 RV3D_DIST_BACKEND=nccl run --no-sync-bn
TAG=ddp+syncbn RV3D_FORCE_DIST=1 RV3D_SYNC_WORLD1=1 RV3D_DIST_BACKEND=nccl run
TAG=ddp+syncbn_nogroup RV3D_NO_GROUP_SYNC_BN=1 RV3D_FORCE_DIST=1 RV3D_SYNC_WORLD1=1 RV3D_DIST_BACKEND=nccl run
