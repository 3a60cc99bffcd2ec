#!/bin/bash
# rocprofv3 evidence for ONE workload of bench.py (run from the repo root ON THE GPU BOX):
#   tools/profile_grid.sh TAG NX [MODEL] [STEPS]     e.g. tools/profile_grid.sh r03_coupled8192 8192 coupled 5
# 1. kernel-trace statistics of a short timed run, 2. the four --pmc passes (each in its own run, with --kernel-trace only),
# 3. summaries under gpurun_out/profiles_TAG/ (copy the ones to be judged into profiles/).
set -u
TAG=$1; NX=$2; MODEL=${3:-coupled}; STEPS=${4:-5}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
DST=$ROOT/gpurun_out/profiles_$TAG
mkdir -p "$OUT" "$DST"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$OUT/stats" -o run --output-format csv -- python3 "$ROOT/bench.py" --model $MODEL --nx $NX --steps $STEPS --warmup 2 --no-cpu-baseline > "$OUT/bench_line.json" 2> "$OUT/bench_stderr.log"
echo "stats rc=$?"
cp "$(find "$OUT/stats" -name '*kernel_stats.csv' | head -1)" "$DST/${TAG}_kernel_stats.csv"
grep '^{' "$OUT/bench_line.json" | tail -1 > "$DST/${TAG}_bench_line.json"
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  name=$(echo $pass | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $pass -d "$OUT/pmc_$name" -o run --output-format csv -- python3 "$ROOT/bench.py" --model $MODEL --nx $NX --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/pmc_$name.log" 2>&1
  echo "pmc $name rc=$?"
done
cd "$ROOT"
python3 tools/pmc_summary.py "$DST/${TAG}_pmc_summary.json" \
  "rocprofv3 --kernel-trace --pmc <counters>, separate passes (FETCH_SIZE | WRITE_SIZE | SQ_LDS_* SQ_INSTS_* | SQ_WAVE/BUSY/WAIT_INST | SQ_WAIT_ANY SQ_ACTIVE_*), python3 bench.py --model $MODEL --nx $NX --steps 2 --warmup 1 --no-cpu-baseline, tag $TAG; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 64 B per 128-B request); values are averages per launch" \
  "$OUT"/pmc_*
