"""Which attributes of the four model classes differ from the reference's after steps WITHOUT a diagnostics tick (golden g19):
    python tools/diag/attr_diff_g19.py"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, logging
logging.disable(logging.CRITICAL)
import niwqg_amd
from niwqg_amd import InitialConditions as ic
from test_oracle_golden import notebook_kwargs, L, K0, U0, TE
g = np.load(os.path.join(ROOT, "tests", "golden", "g19_attributes_after_steps_without_ticks.npz"))
for tag, cls in (("coupled", niwqg_amd.CoupledModel), ("uncoupled", niwqg_amd.UnCoupledModel), ("qg", niwqg_amd.QGModel), ("ybj", niwqg_amd.YBJModel)):
    if tag == "qg":
        m = cls.Model(L=L, nx=64, tmax=1e30, dt=0.05 * TE * 2, twrite=10**9, nu4=7.5e8 * 16, nu=5.0, mu=1e-8, use_filter=True, U=-U0,
                      tdiags=3, beta=2e-11, passive_scalar=True, nu4c=3e9, nuc=2.0, muc=1e-8, save_to_disk=False)
    else:
        kw = notebook_kwargs(64, True, tdiags=3)
        kw.update(nu4w=1e10, mu=1e-8, muw=2e-8, twrite=10**9)
        m = cls.Model(**kw)
    rng = np.random.default_rng(19)
    m.set_q(ic.LambDipole(m, U=U0, R=2 * np.pi / K0) + 1e-6 * rng.standard_normal((64, 64)))
    if tag == "qg":
        m.set_c(1.0 + 0.3 * rng.standard_normal((64, 64)))
    else:
        m.set_phi(ic.WavePacket(m, k=2 * K0, l=K0, R=L / 6, x0=L / 2, y0=L / 2) * 0.1
                  + 0.01 * (rng.standard_normal((64, 64)) + 1j * rng.standard_normal((64, 64))))
    while m.tc < 3:
        m._step_forward()
    for name, want in zip(g[tag + "_num_names"], g[tag + "_num_values"]):
        name = str(name)
        if not hasattr(m, name):
            print(tag, name, "MISSING"); continue
        got = float(getattr(m, name))
        if not np.isclose(got, float(want), rtol=1e-7, atol=1e-30):
            print(tag, "num", name, got, float(want))
    for name, shape, dtype, cs in zip(g[tag + "_arr_names"], g[tag + "_arr_shapes"], g[tag + "_arr_dtypes"], g[tag + "_arr_checksums"]):
        name = str(name)
        if name in ("qh0", "qh1", "phih0", "phih1", "ch0", "ch1"): continue
        if not hasattr(m, name):
            print(tag, name, "MISSING"); continue
        a = np.asarray(getattr(m, name))
        z = a.astype(complex).ravel(); w = np.cos(0.37 * np.arange(z.size)); scale = np.abs(z).sum() + 1e-300
        e1, e2 = abs(z.sum() - cs[0]) / scale, abs((z * w).sum() - cs[1]) / scale
        if e1 > 1e-9 or e2 > 1e-9:
            print(tag, "arr", name, "%.2e %.2e" % (e1, e2))
