#!/bin/bash
# What ONE rank of the P-rank slab partition costs, measured on ONE GPU (run from the repo root ON THE GPU BOX):
#   tools/rank_of_sweep.sh TAG [NX ...]        default sizes: 4096 8192
# bench.py --rank-of P builds rank 0 of the decomposition alone with the library's null link (every launch, stream, event and
# row chunk of a real rank; nothing on the wire) and times its step next to the single-GPU step of the same run.  Sweep:
# P = 2, 4, 8 x --chunks 1|2|4; then a rocprofv3 --kernel-trace of P = 8 (program directly after `--`).
# Output: gpurun_out/rank_of_TAG/*.json (one bench line each) and TAG_rank_of_8_NX_kernel_stats.csv.
set -u
TAG=${1:-r04}
shift || true
SIZES=${*:-"4096 8192"}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/rank_of_$TAG
mkdir -p "$OUT"
for nx in $SIZES; do
  for P in 2 4 8; do
    for ch in 1 2 4; do
      python3 bench.py --nx $nx --rank-of $P --chunks $ch --steps 30 --warmup 5 2> "$OUT/err_${nx}_${P}_${ch}.log" | grep '^{' > "$OUT/rank_of_${P}_${nx}_chunks${ch}.json"
      python3 - "$OUT/rank_of_${P}_${nx}_chunks${ch}.json" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    c = d["config"]
    print("nx %s P %d chunks %s: rank %.3f ms  single %.3f ms  ideal %.3f ms  rank/ideal %.3f   per class %s" % (
        d["metric"].split("Model ")[1].split("^")[0], c["rank_of"], sys.argv[1].split("chunks")[1][0], c["rank_compute_ms_per_step"],
        c["single_gpu_ms_per_step_same_run"], c["ideal_ms_per_step"], c["rank_compute_over_ideal"], d["roofline"]["per_kernel_ms_per_step"]))
except Exception as e:
    print("FAILED", sys.argv[1], e)
PY
    done
  done
  ( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d "$OUT/prof_$nx" -o run --output-format csv -- python3 "$ROOT/bench.py" --nx $nx --rank-of 8 --rank-only --chunks 2 --steps 10 --warmup 2 > "$OUT/prof_$nx.log" 2>&1 )
  cp "$(find "$OUT/prof_$nx" -name '*kernel_stats.csv' | head -1)" "$OUT/${TAG}_rank_of_8_${nx}_kernel_stats.csv" 2>/dev/null
  rm -rf "$OUT/prof_$nx"
done
