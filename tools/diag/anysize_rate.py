"""Step rate of the any-size path (grids without a fused plan) next to the fused path on the neighbouring power of two:
    python tools/diag/anysize_rate.py [nx ...]"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import logging
import numpy as np
logging.disable(logging.CRITICAL)
import niwqg_amd
import bench

for nx in [int(a) for a in sys.argv[1:]] or [96, 100, 192, 384, 1000, 1536, 3072]:
    for kind, mod in (("coupled", niwqg_amd.CoupledModel), ("qg", niwqg_amd.QGModel)):
        kw = bench.c3_kwargs(nx, kind)
        t0 = time.perf_counter()
        m = mod.Model(**kw)
        t_build = time.perf_counter() - t0
        rng = np.random.default_rng(0)
        m.set_q(1e-5 * rng.standard_normal((nx, nx)))
        if kind == "coupled":
            m.set_phi(0.05 * (rng.standard_normal((nx, nx)) + 1j * rng.standard_normal((nx, nx))))
        for _ in range(2):
            m._step_etdrk4()
        m._ctx.sync()
        n = 3 if nx >= 1000 else 10
        t0 = time.perf_counter()
        for _ in range(n):
            m._step_etdrk4()
        m._ctx.sync()
        dt = (time.perf_counter() - t0) / n
        print("%-8s nx %5d  %s  constructor %.2f s  step %.2f ms  = %.1f steps/s   device bytes %.2f GB" % (
            kind, nx, "any-size" if getattr(m, "_any_size", False) else "fused   ", t_build, 1e3 * dt, 1.0 / dt, m._ctx.device_bytes() / 1e9), flush=True)
        del m
