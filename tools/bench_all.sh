#!/bin/bash
# benches of the other BASELINE configs on one GPU (not the headline): QGModel 2048^2, UnCoupledModel 1024^2
for spec in "qg 2048" "qg 256" "uncoupled 1024" "coupled 2048" "coupled 8192"; do
  set -- $spec
  echo "== $1 $2"
  timeout -k 10 300 python bench.py --model $1 --nx $2 --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"step_frac_of_peak": [0-9.]*' | tr '\n' ' '
  echo
done
