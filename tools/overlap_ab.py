#!/usr/bin/env python
"""A/B of the dual-stream step: NIWQG_AMD_OVERLAP_CUS=<n> python tools/overlap_ab.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
m = bench.build_model("coupled", 4096, 0)
c = m._ctx
c.step(5); c.sync()
best = 1e9
for _ in range(3):
    t0 = time.perf_counter(); c.step(20); c.sync(); best = min(best, (time.perf_counter() - t0) / 20)
print("overlap_cus=%s: %.3f ms/step = %.1f steps/s" % (os.environ.get("NIWQG_AMD_OVERLAP_CUS", "0"), best * 1e3, 1 / best))
