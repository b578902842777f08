#!/bin/bash
# usage: ab_step.sh <out-prefix> <model: av2|waymo> "ENV=.. ENV=.." "ENV=.." ...   (each quoted argument: one arm; "-" = defaults)
# Interleaved A/B of whole training steps on ONE box: every arm twice, alternating; prints ms per step of each run.
pre=$1; model=$2; shift 2
root=${GRAFT_REPO_ROOT:-$(pwd)}
if [ "$model" = waymo ]; then args="--widths rv-waymo --width 2656 --features 6 --classes 3"; else args=""; fi
for rnd in 1 2; do
  i=0
  for arm in "$@"; do
    i=$((i+1))
    envs=""; [ "$arm" != "-" ] && envs="$arm"
    ms=$(env $envs python3 "$root/bench.py" --steps 20 --warmup 5 --no-extra --no-cpu-baseline $args 2>>"$root/gpurun_out/${pre}.err" | grep -o '"ms_per_step": [0-9.]*' | grep -o '[0-9.]*$')
    echo "$model round $rnd arm $i [$arm]: $ms ms" | tee -a "$root/gpurun_out/${pre}.txt"
  done
done
