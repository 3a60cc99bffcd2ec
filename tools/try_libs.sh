#!/bin/bash
# usage: tools/try_libs.sh lib1.so lib2.so ... : run the bench with each library variant
for l in "$@"; do
  cp niwqg_amd/$l niwqg_amd/libniwqg_amd.so
  echo "== $l"
  timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | grep -o '"value": [0-9.]*\|"avg_launch_ms": [0-9.]*' | tr '\n' ' '
  echo
done
