import sys, time, numpy as np
sys.path.insert(0, "/root/repo")
import bench
from niwqg_amd import YBJModel, InitialConditions as ic
nx = 4096
kw = bench.c3_kwargs(nx, "uncoupled")
m = YBJModel.Model(nx=nx, **{k: v for k, v in kw.items() if k != "nx"})
m.set_q(ic.LambDipole(m, U=bench.U0, R=2 * np.pi / bench.K0))
m.set_phi((np.ones((nx, nx)) + 1j) * (2 * bench.U0) / np.sqrt(2))
c = m._ctx
c.step(3); c.sync()
t0 = time.perf_counter(); c.step(20); c.sync(); dt = time.perf_counter() - t0
print("YBJModel 4096^2: %.1f steps/s (%.2f ms/step)" % (20 / dt, dt / 20 * 1e3))
