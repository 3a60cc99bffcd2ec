import sys, time, logging
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
logging.disable(logging.CRITICAL)
import numpy as np
import niwqg_amd
from niwqg_amd import InitialConditions as ic
from test_oracle_golden import notebook_kwargs, K0, U0
for nx in (2048, 4096, 8192):
    t0 = time.time(); m = niwqg_amd.CoupledModel.Model(**notebook_kwargs(nx, True)); t1 = time.time()
    q0 = ic.LambDipole(m, U=U0, R=2 * np.pi / K0); t2 = time.time()
    phi0 = (np.ones((nx, nx)) + 1j) * (2 * U0) / np.sqrt(2)
    m.set_q(q0); t3 = time.time(); m.set_phi(phi0); t4 = time.time()
    m._step_forward(); t5 = time.time()
    q = m.q; t6 = time.time()
    print("nx %d: construct %.2f s (contour entries patched: %s), LambDipole %.2f s, set_q %.2f s, set_phi %.2f s, first step %.2f s, read q %.2f s" % (nx, t1 - t0, m._ctx.contour_patched, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t5), flush=True)
    m._ctx.close(); del m
