#!/bin/bash
# What bracketing the dominant kernel class with HIP events INSIDE bench.py's timed region costs: every launch (--region-stride 1,
# rounds 1-3) against every 7th (default), single GPU and one rank of eight.  Run from the repo root on the GPU box.
show() { python3 -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']; c = d['config']
print('$1', 'value %.2f steps/s  %.4f ms/step  dominant %s %.5f ms/launch frac %.4f (%d launches bracketed)' % (d['value'], d['ms_per_step'], r['kernel'], r['avg_launch_ms'], r['frac'], r['launches']),
      ('rank %.4f ms = %.3f x ideal, exchange stream %.4f ms/step' % (c['rank_compute_ms_per_step'], c['rank_compute_over_ideal'], c['exchange_ms_per_step'])) if 'rank_of' in c else '')
"; }
for st in 1 7 1 7; do python3 bench.py --steps 100 --warmup 20 --no-cpu-baseline --region-stride $st 2>/dev/null | show "single GPU stride $st:"; done
for st in 1 7; do for ch in 1 2; do python3 bench.py --rank-of 8 --chunks $ch --steps 30 --warmup 5 --region-stride $st 2>/dev/null | show "rank-of 8 chunks $ch stride $st:"; done; done
for st in 1 7; do python3 bench.py --rank-of 8 --nx 8192 --chunks 1 --steps 10 --warmup 3 --region-stride $st 2>/dev/null | show "rank-of 8 8192 chunks 1 stride $st:"; done
