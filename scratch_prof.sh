export MASTER_ADDR=127.0.0.1 MASTER_PORT=29671 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1
root=$(pwd)
cd /tmp && export TMPDIR=/tmp && mkdir -p /tmp/prof
RV3D_FORCE_DIST=1 RV3D_SYNC_WORLD1=1 RV3D_DIST_BACKEND=nccl rocprofv3 --kernel-trace --stats -d /tmp/prof/sync -o p -- python3 $root/bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --no-extra > /tmp/prof/sync.log 2>&1
python3 $root/profiles/kernel_stats.py $(find /tmp/prof/sync -name '*_results.db' | head -1) > $root/gpurun_out/r03_sync_kernel_stats.csv
head -40 $root/gpurun_out/r03_sync_kernel_stats.csv | cut -c1-140
