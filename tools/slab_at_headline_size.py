"""CoupledModel 4096^2 (the headline grid) on 8, 4 and 2 peer ranks of ONE GPU against the single-context model: three steps,
fields and budgets (run on the GPU box: python tools/slab_at_headline_size.py)."""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import niwqg_amd
from niwqg_amd import InitialConditions as ic
from test_oracle_golden import notebook_kwargs, rel, L, K0, U0
kw = notebook_kwargs(4096, True)
w = niwqg_amd.CoupledModel.Model(slab=False, **kw)
q0 = ic.LambDipole(w, U=U0, R=2 * np.pi / K0)
phi0 = (np.ones((4096, 4096)) + 1j) * (2 * U0) / np.sqrt(2)
w.set_q(q0); w.set_phi(phi0)
for _ in range(3): w._step_forward()
qw, pw = w.q, w.phi
bw = [w.Ke, w.Pw, w.Kw]
del w
for P, nch in ((8, 2), (4, 4), (2, 1)):
    s = niwqg_amd.CoupledModel.Model(slab=P, nchunks=nch, **kw)
    s.set_q(q0); s.set_phi(phi0)
    t0 = time.time()
    for _ in range(3): s._step_forward()
    print(P, nch, rel(s.q, qw), rel(s.phi, pw), np.allclose([s.Ke, s.Pw, s.Kw], bw, rtol=1e-10), round(time.time() - t0, 2), flush=True)
    assert rel(s.q, qw) < 1e-13 and rel(s.phi, pw) < 1e-13
    s._ctx.close()
    del s
print("headline-size slab runs agree")
