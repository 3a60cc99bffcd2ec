"""Replay one draw of test_randomly_drawn_configurations_at_size_through_resolution_independence on the device only and print the
growth of the state per step (is a blow-up there with the contour patch off? at the other size?).
    python tools/diag/at_size_seed.py NX SEED [steps]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import logging
import numpy as np
logging.disable(logging.CRITICAL)
import test_gpu_models as T
from test_gpu_models import L, TE, U0, MZ, NB, F0

nx, seed = int(sys.argv[1]), int(sys.argv[2])
nsteps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
rng = np.random.default_rng(17000 + seed)
kind = ["coupled", "qg", "uncoupled", "ybj"][seed % 4]
filt = int(rng.integers(0, 3))
if kind == "qg" and filt == 1:
    filt = 0
dt = 0.025 * TE * 128 / nx * float(rng.choice([0.5, 1.0]))
kw = dict(L=L, nx=nx, tmax=1e30, dt=dt, twrite=10 ** 9, tdiags=10 ** 9, use_filter=filt == 0, dealias=filt == 1,
          U=float(rng.choice([0.0, -U0, 0.5 * U0])), nu4=5e11 * (128.0 / nx) ** 4 * float(rng.uniform(0.2, 2.0)),
          nu=float(rng.choice([0.0, 20.0])), mu=float(rng.choice([0.0, 1e-8])))
M = T.models()
if kind == "qg":
    kw.update(beta=float(rng.choice([0.0, 2e-11])), passive_scalar=bool(rng.integers(0, 2)), nu4c=kw["nu4"] * 0.5, nuc=2.0, muc=1e-8)
    cls = M.QGModel
else:
    kw.update(m=MZ * float(rng.choice([0.5, 1.0, 2.0])), N=NB, f=F0, nuw=float(rng.choice([0.0, 50.0])),
              nu4w=float(rng.choice([0.0, 0.1])) * kw["nu4"], muw=float(rng.choice([0.0, 2e-8])))
    cls = {"coupled": M.CoupledModel, "uncoupled": M.UnCoupledModel, "ybj": M.YBJModel}[kind]
# NOTE: the draw above was the one of the FIRST version of the test (nx = 128 in the nu4 / dt formulas via the size argument)
kw["dt"] = 0.025 * TE * 128 / nx * (kw["dt"] / (0.025 * TE * 128 / nx))
print(kind, {k: v for k, v in kw.items() if k not in ("L", "tmax", "twrite", "tdiags")}, "patch", os.environ.get("NIWQG_AMD_CONTOUR_PATCH", "1"), flush=True)
m = cls.Model(**kw)
print("contour entries patched:", getattr(m._ctx, "contour_patched", None), flush=True)
q1, phi1 = T._random_band_limited_state(m.x, m.y, 18000 + seed, kind != "qg")
m.set_q(q1)
if kind != "qg":
    m.set_phi(phi1)
for n in range(nsteps):
    m._step_forward()
    qh = np.abs(m.qh)
    i = np.unravel_index(np.argmax(qh), qh.shape)
    line = "step %2d  max|qh| %.3e at (l, k) = %s" % (n + 1, qh[i], tuple(int(v) for v in i))
    if kind != "qg":
        ph = np.abs(m.phih)
        j = np.unravel_index(np.argmax(ph), ph.shape)
        line += "   max|phih| %.3e at %s" % (ph[j], tuple(int(v) for v in j))
    print(line, flush=True)
    if not np.isfinite(qh[i]) or qh[i] > 1e30:
        break
