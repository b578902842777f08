#!/bin/bash
# Diagnostic builds of the library (NOT products: built on the GPU box into gpurun_out/diag/, selected with RV3D_LIB) for two bounds the
# round-5 review asked to be MEASURED instead of estimated:
#   no_remap     -DRV_DIAG_NO_XCD_REMAP   wgrad3's workgroups dealt round-robin: every K slice is fetched into all eight L2s instead of one or two
#                                         (item 4: how does the kernel's time move with its L2-miss traffic?)
#   skip_reduce  -DRV_DIAG_SKIP_REDUCE    no split-K reduction launch behind wgrad2 / wgrad3 (WRONG gradients: the upper bound of item 8 --
#                                         whatever folds the 76 reduce launches into a consumer cannot return more than deleting them does)
# usage (one gpurun call): bash profiles/tools/diag_wgrad.sh <out-prefix>
set -e
pre=${1:-r06_diag}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/diag
mkdir -p $out
cd $root/range_view_3d_detection_amd/csrc
objs=$(ls build/*.o | grep -v wgrad.o)
for v in NO_XCD_REMAP SKIP_REDUCE; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -I../../include -DRV_DIAG_$v -c wgrad.hip -o $out/wgrad_$v.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/librv3d_$v.so $objs $out/wgrad_$v.o
done
cd $root
echo "built: $(ls $out/*.so)" | tee gpurun_out/$pre.txt
# (1) the kernel alone on random data: 512 <-> 512 and 256 <-> 256 3x3 at 4 x 64 x 2048, product library against no_remap, interleaved
for rnd in 1 2 3; do
  for lib in product NO_XCD_REMAP; do
    if [ $lib = product ]; then unset RV3D_LIB; else export RV3D_LIB=$out/librv3d_$lib.so; fi
    python3 profiles/tools/mb_wgrad.py $lib 2>/dev/null | tee -a gpurun_out/$pre.txt
  done
done
# (2) the training step: product against skip_reduce and no_remap, interleaved (ms per step of bench.py --timed-only)
for rnd in 1 2 3; do
  for lib in product SKIP_REDUCE NO_XCD_REMAP; do
    if [ $lib = product ]; then unset RV3D_LIB; else export RV3D_LIB=$out/librv3d_$lib.so; fi
    ms=$(python3 bench.py --steps 20 --warmup 5 --timed-only 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | grep -o '[0-9.]*$')
    echo "step round $rnd $lib: $ms ms" | tee -a gpurun_out/$pre.txt
  done
done
