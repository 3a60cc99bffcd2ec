"""Where does the 8e-12 of QGModel 2048^2 random-q (config 2) after 2 steps come from?  Error growth, error spectrum,
coefficient planes.  Run on the GPU box: python tools/diag/qg2048_error.py [nx] [nsteps]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
import niwqg_amd
from oracle import niwqg_oracle as O

nx = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
ns = int(sys.argv[2]) if len(sys.argv) > 2 else 10
kw = bench.c3_kwargs(nx, "qg")
if nx != 2048:
    kw.update(nu4=7.5e8 / 64 * (2048.0 / nx) ** 4)
q0 = 1e-5 * np.random.default_rng(0).standard_normal((nx, nx))
m = niwqg_amd.QGModel.Model(**kw)
o = O.QGOracle(coeff_chunk=8, workers=16, **kw)
rel = lambda a, b: np.linalg.norm(np.ravel(a) - np.ravel(b)) / np.linalg.norm(np.ravel(b))
names = ["E", "Eh", "Q", "f0", "fab", "fc"]
for i, nm in enumerate(names):
    mine, ref = m._ctx.coeff(0, i), o.coef_q[nm]
    d = np.abs(mine - ref) / np.maximum(np.abs(ref), 1e-300)
    j = np.unravel_index(np.argmax(d), d.shape)
    print("coef %-3s max rel diff %.2e at (l,k)=%s  ref=%s  l2 rel %.2e" % (nm, d.max(), j, ref[j], rel(mine, ref)))
for x in (m, o):
    x.set_q(q0)
print("after set_q: rel qh %.2e ph %.2e" % (rel(m.qh, o.qh), rel(m.ph, o.ph)))
for n in range(1, ns + 1):
    m._step_forward(); o._step_forward()
    e = m.qh - o.qh
    print("step %d rel q %.2e qh %.2e |q| %.3e" % (n, rel(m.q, o.q), rel(m.qh, o.qh), np.linalg.norm(o.q)), flush=True)
    if n in (1, 2, ns):
        k = np.sqrt(o.k ** 2 + o.l ** 2) / o.kk[1]
        for lo, hi in ((0, 8), (8, 64), (64, 256), (256, 512), (512, 700), (700, 1024), (1024, 2000)):
            sel = (k >= lo) & (k < hi)
            if sel.any():
                print("   band %4d-%4d: |err| %.2e |ref| %.2e rel %.2e" % (lo, hi, np.linalg.norm(e[sel]), np.linalg.norm(o.qh[sel]), np.linalg.norm(e[sel]) / max(np.linalg.norm(o.qh[sel]), 1e-300)))
