import sys, time, logging
sys.path.insert(0, '/root/repo')
import numpy as np, bench, niwqg_amd
nx = 4096
kw = bench.c3_kwargs(nx, "coupled")
kw.update(twrite=500, tdiags=100, tmax=2000.5 * kw["dt"])
m = niwqg_amd.CoupledModel.Model(**kw)
q, phi = bench.initial_fields("coupled", nx, m)
m.set_q(q); m.set_phi(phi)
ke0, kw0, pw0 = m._calc_ke_qg(), m._calc_ke_niw(), m._calc_pe_niw()
t = time.time(); m.run(); el = time.time() - t
print("steps", m.tc, "wall %.1f s = %.1f steps/s incl. ticks" % (el, m.tc / el))
print("Ke %.6e -> %.6e (budget %.6e)  Kw %.6e -> %.6e (budget %.6e)  Pw %.3e -> %.3e (budget %.3e)" % (ke0, m._calc_ke_qg(), m.Ke, kw0, m._calc_ke_niw(), m.Kw, pw0, m._calc_pe_niw(), m.Pw))
print("finite:", np.isfinite(m.q).all() and np.isfinite(m.phi).all(), "cfl %.3f" % m._calc_cfl())
