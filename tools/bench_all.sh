#!/bin/bash
# The other BASELINE configurations and secondary workloads on ONE GPU, each with its bench line AND a rocprofv3 kernel-stats
# csv, kept per config under gpurun_out/profiles_TAG/ (copy into profiles/ to commit).  Per config the exit status and
# stderr are kept too (`*.rc`, `*.stderr`): an empty line can no longer hide whether a run timed out or failed.
#   tools/bench_all.sh TAG [config ...]      configs: qg2048 qg256 uncoupled1024 coupled2048 coupled8192 ybj4096 ensemble8
set -u
TAG=${1:-r02}
shift || true
CONFIGS=${*:-"qg2048 qg256 uncoupled1024 coupled2048 coupled8192 ybj4096 ensemble8"}
ROOT=$(pwd)
DST=$ROOT/gpurun_out/profiles_$TAG
mkdir -p "$DST"
cd /tmp && export TMPDIR=/tmp
for cfg in $CONFIGS; do
  case $cfg in
    qg2048) args="--model qg --nx 2048 --steps 200 --warmup 20";;
    qg256) args="--model qg --nx 256 --steps 2000 --warmup 100";;
    uncoupled1024) args="--model uncoupled --nx 1024 --steps 200 --warmup 20";;
    coupled2048) args="--model coupled --nx 2048 --steps 100 --warmup 10";;
    coupled8192) args="--model coupled --nx 8192 --steps 20 --warmup 3";;
    ybj4096) args="--model ybj --nx 4096 --steps 50 --warmup 5";;
    ensemble8) args="--members 8 --steps 50 --warmup 5";;
    *) echo "unknown config $cfg"; continue;;
  esac
  echo "== $cfg: bench.py $args"
  W=/tmp/ball_$cfg
  rm -rf "$W"
  # the quoted number: a plain run (rocprofv3 slows launch-bound configurations down by 10-30 %) ...
  timeout -k 10 420 python3 "$ROOT/bench.py" $args --no-cpu-baseline > "$W.out" 2> "$DST/${TAG}_${cfg}.stderr"
  rc=$?
  echo $rc > "$DST/${TAG}_${cfg}.rc"
  grep '^{' "$W.out" | tail -1 > "$DST/${TAG}_${cfg}_bench_line.json"
  # ... and the kernel statistics of the same command under rocprofv3
  timeout -k 10 420 rocprofv3 --kernel-trace --stats -d "$W" -o run --output-format csv -- python3 "$ROOT/bench.py" $args --no-cpu-baseline > "$W.prof.out" 2>> "$DST/${TAG}_${cfg}.stderr"
  echo "$rc $?" > "$DST/${TAG}_${cfg}.rc"
  f=$(find "$W" -name '*kernel_stats.csv' 2>/dev/null | head -1)
  [ -n "$f" ] && cp "$f" "$DST/${TAG}_${cfg}_kernel_stats.csv"
  echo "   rc=$rc $(python3 -c "
import json,sys
try:
    d=json.load(open('$DST/${TAG}_${cfg}_bench_line.json')); print('value %.1f %s, %.3f ms/step, step_frac_of_peak %.3f' % (d['value'], d['unit'], d['ms_per_step'], d['roofline'].get('step_frac_of_peak', d['roofline']['frac'])))
except Exception as e: print('NO JSON LINE:', e)
")"
  [ $rc -ne 0 ] && tail -5 "$DST/${TAG}_${cfg}.stderr"
done
exit 0
