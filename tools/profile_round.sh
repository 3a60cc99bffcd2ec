#!/bin/bash
# One gpurun call's worth of evidence for the headline workload (run from the repo root ON THE GPU BOX):
#   tools/profile_round.sh TAG        e.g. TAG=r02_a
# 1. rocprofv3 --kernel-trace --stats of `python3 bench.py --steps 20 --warmup 5` (kernel stats csv + the bench line)
# 2. rocprofv3 --pmc passes (FETCH_SIZE | WRITE_SIZE | LDS | waits), each in its own run with --kernel-trace only
# 3. summaries copied to profiles/ (and mirrored to gpurun_out/profiles_TAG/ so that they travel back)
set -u
TAG=${1:-r02}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT" "$ROOT/gpurun_out/profiles_$TAG"
cd /tmp && export TMPDIR=/tmp
# the headline line itself, un-profiled, with the measured CPU sample
python3 "$ROOT/bench.py" > "$OUT/bench_plain.json" 2> "$OUT/bench_plain_stderr.log"
grep '^{' "$OUT/bench_plain.json" | tail -1 > "$ROOT/gpurun_out/profiles_$TAG/${TAG}_bench_line_plain.json"
rocprofv3 --kernel-trace --stats -d "$OUT/stats" -o run --output-format csv -- python3 "$ROOT/bench.py" --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/bench_line.json" 2> "$OUT/bench_stderr.log"
echo "stats rc=$?"
cp "$(find "$OUT/stats" -name '*kernel_stats.csv' | head -1)" "$ROOT/gpurun_out/profiles_$TAG/${TAG}_kernel_stats.csv"
grep '^{' "$OUT/bench_line.json" | tail -1 > "$ROOT/gpurun_out/profiles_$TAG/${TAG}_bench_line.json"
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS"; do
  name=$(echo $pass | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $pass -d "$OUT/pmc_$name" -o run --output-format csv -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/pmc_$name.log" 2>&1
  echo "pmc $name rc=$?"
done
cd "$ROOT"
python3 tools/pmc_summary.py "gpurun_out/profiles_$TAG/${TAG}_pmc_summary.json" \
  "rocprofv3 --kernel-trace --pmc <counters>, separate passes (FETCH_SIZE | WRITE_SIZE | SQ_LDS_* SQ_INSTS_* | SQ_WAVE/BUSY/WAIT), python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline, CoupledModel 4096^2 budgets on, tag $TAG; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 64 B per 128-B request); values are averages per launch" \
  "$OUT"/pmc_*
