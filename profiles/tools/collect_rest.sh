#!/bin/bash
# usage: collect_rest.sh <tag>: the remaining evidence of a round (per-shape table, per-step kernel lists, idle analysis, default bench line)
tag=$1
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$root"
python3 profiles/tools/shape_prof.py > gpurun_out/${tag}_shapes.txt 2>&1 && echo shapes ok
python3 profiles/tools/shape_prof.py --widths rv-waymo --width 2656 --features 6 --classes 3 > gpurun_out/${tag}_shapes_waymo.txt 2>&1 && echo shapes waymo ok
bash profiles/tools/trace_step.sh ${tag}_step_kernels > /dev/null 2>&1 && echo step ok
bash profiles/tools/trace_step.sh ${tag}_step_kernels_waymo --widths rv-waymo --width 2656 --features 6 --classes 3 > /dev/null 2>&1 && echo step waymo ok
bash profiles/tools/trace_gaps.sh ${tag}_gap_analysis > /dev/null 2>&1 && echo gaps ok
python3 bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err && echo bench ok
