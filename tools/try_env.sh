#!/bin/bash
# usage: tools/try_env.sh VAR v1 v2 ... : bench value and per-kernel times with VAR=v (GPU box)
VAR=$1; shift
for v in "$@"; do
  echo "== $VAR=$v"
  env $VAR=$v timeout -k 10 300 python bench.py --nx ${NX:-4096} --model ${MODEL:-coupled} --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"per_kernel_ms_per_step": {[^}]*}' | tr '\n' ' '
  echo
done
