# One-rank cost of the N > 1 code path (RCCL process group of one rank, every BatchNorm on the synchronised path), same box.
p() { python -c "import json,sys; j=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('$1', round(j['ms_per_step'],2), j['config']['sync_bn'], j['config']['collectives']['per_step']['sync_bn_all_reduce']['calls'], j['config']['loss'])"; }
A="--steps 20 --warmup 5 --no-extra --no-cpu-baseline"
python bench.py $A 2>/dev/null | p local
RV3D_FORCE_DIST=1 RV3D_SYNC_WORLD1=1 python bench.py $A 2>/dev/null | p gradsync+syncbn_c10d
RV3D_FORCE_DIST=1 RV3D_SYNC_WORLD1=1 RV3D_DIRECT_RCCL=1 python bench.py $A 2>/dev/null | p gradsync+syncbn_direct
RV3D_FORCE_DIST=1 RV3D_SYNC_WORLD1=1 python bench.py $A --no-sync-bn 2>/dev/null | p gradsync_only
python bench.py $A 2>/dev/null | p local
RV3D_FORCE_DIST=1 RV3D_SYNC_WORLD1=1 python bench.py $A 2>/dev/null | p gradsync+syncbn_c10d
