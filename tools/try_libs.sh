#!/bin/bash
# usage: tools/try_libs.sh lib1.so lib2.so ... : A/B of library variants (files under niwqg_amd/) on the GPU box:
# per-kernel-class HIP-event times of the 4096^2 Coupled step and the bench value with each
cp niwqg_amd/libniwqg_amd.so /tmp/libniwqg_amd.keep
for l in "$@"; do
  cp niwqg_amd/$l niwqg_amd/libniwqg_amd.so
  echo "== $l"
  timeout -k 10 300 python tools/kernel_times.py ${NX:-4096} 10 ${MODEL:-coupled} 2>&1 | grep "launches/step\|sum"
  timeout -k 10 300 python bench.py --nx ${NX:-4096} --model ${MODEL:-coupled} --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' | tr '\n' ' '
  echo
done
cp /tmp/libniwqg_amd.keep niwqg_amd/libniwqg_amd.so
