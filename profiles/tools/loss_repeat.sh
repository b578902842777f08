#!/bin/bash
# usage: loss_repeat.sh <n> ENV=.. ENV=..   -> config.loss of n runs of the 25-step bench under that environment (must be identical)
n=$1; shift
for i in $(seq $n); do
  env "$@" python3 bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; j=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print(round(j['ms_per_step'],2), j['config']['loss'])"
done
