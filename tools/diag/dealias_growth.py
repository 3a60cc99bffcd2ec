"""Does round-off grow at the edge of the 2/3 mask?  The failing 8192^2 draw (UnCoupledModel, dealias=True, U = 0.05, nu = 20,
the size's dt and nu4: advective CFL ~0.4) scaled to a grid the oracle can run: device and oracle side by side, band-limited
state, max |qh| outside the band per step.      python tools/diag/dealias_growth.py NX [steps] [kind]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import logging
import numpy as np
logging.disable(logging.CRITICAL)
import test_gpu_models as T
from test_gpu_models import L, TE, U0, MZ, NB, F0, O

nx = int(sys.argv[1])
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
kind = sys.argv[3] if len(sys.argv) > 3 else "uncoupled"
kw = dict(L=L, nx=nx, tmax=1e30, dt=0.025 * TE * 128 / nx, twrite=10 ** 9, tdiags=10 ** 9, use_filter=False, dealias=True, U=0.05,
          nu4=5e11 * (128.0 / nx) ** 4 * 0.35, nu=20.0, mu=0.0, m=0.5 * MZ, N=NB, f=F0, nuw=0.0, nu4w=0.035 * 5e11 * (128.0 / nx) ** 4,
          muw=2e-8)
M = T.models()
cls = {"coupled": M.CoupledModel, "uncoupled": M.UnCoupledModel}[kind]
m, o = cls.Model(**kw), O.NIWQGOracle(kind, **kw)
q1, phi1 = T._random_band_limited_state(o.grid.x, o.grid.y, 18106, True)
for x in (m, o):
    x.set_q(q1)
    x.set_phi(phi1)
band = np.zeros(nx, bool)
band[np.r_[0:13, nx - 12:nx]] = True
out = ~(band[:, None] & band[None, :])
for n in range(nsteps):
    m._step_forward()
    o._step_forward()
    a, b = np.abs(m.qh) * out, np.abs(o.qh) * out
    i, j = np.unravel_index(np.argmax(a), a.shape), np.unravel_index(np.argmax(b), b.shape)
    print("step %2d  outside the band: device max|qh| %.2e at %s   oracle %.2e at %s   rel diff of qh %.1e" % (
        n + 1, a[i], tuple(int(v) for v in i), b[j], tuple(int(v) for v in j), T.rel(m.qh, o.qh)), flush=True)
    if not np.isfinite(a[i]) or a[i] > 1e20:
        break
