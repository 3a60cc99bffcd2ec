#!/usr/bin/env python
"""Per-kernel-class HIP-event times of the CoupledModel step (nq_profile_*): python tools/kernel_times.py [nx] [steps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
nx = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
model = sys.argv[3] if len(sys.argv) > 3 else "coupled"
m = bench.build_model(model, nx, 0)
ctx = m._ctx
ctx.step(3)
ctx.sync()
tot = 0.0
for name, cls in sorted(ctx.KERNEL_CLASSES.items(), key=lambda kv: kv[1]):
    ctx.profile_enable(cls)
    ctx.step(steps)
    n, ms = ctx.profile_read()
    ctx.profile_enable(-1)
    print("%-12s %4d launches/step  %8.1f us/launch  %7.3f ms/step" % (name, n // steps, 1e3 * ms / max(n, 1), ms / steps))
    tot += ms / steps
print("sum %.3f ms/step" % tot)
import time
ctx.sync()
if model == "qg":
    sys.exit(0)
ctx.diagnostic_sums()
t0 = time.perf_counter()
for _ in range(5):
    ctx.diagnostic_sums()
print("diagnostics tick (nq_diagnostics, 32 sums, blocking): %.3f ms" % ((time.perf_counter() - t0) / 5 * 1e3))
