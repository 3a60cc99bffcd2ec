import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np
from oracle import niwqg_oracle as O
from test_oracle_golden import notebook_kwargs, rel, L, K0, U0
from test_gpu_primitives import make_ctx
from niwqg_amd import _lib
nx=64
ctx, orc = make_ctx("coupled", nx, use_filter=False)
q0 = O.lamb_dipole(orc.grid, U=U0, R=2 * np.pi / K0)
phi0 = 0.2 * O.wave_packet(orc.grid, k=3 * K0, l=K0, R=L / 6, x0=L / 2, y0=L / 2)
orc.set_q(q0); orc.set_phi(phi0); ctx.set_q(q0); ctx.set_phi(phi0)
orc._step_forward(); ctx.step(1)
a = ctx.field(_lib.F_PHIH); d = np.abs(a - orc.phih)
print('phih rel', rel(a, orc.phih), 'norm', np.linalg.norm(orc.phih))
idx = np.argsort(d.ravel())[::-1][:12]
for i in idx:
    l,k = np.unravel_index(i, d.shape)
    print(l,k, d[l,k], abs(orc.phih[l,k]), d[l,k]/abs(orc.phih[l,k]))
print('row sums of err^2 top', np.argsort((d**2).sum(1))[::-1][:5], 'col', np.argsort((d**2).sum(0))[::-1][:5])
a = ctx.field(_lib.F_QH); h=33; d = np.abs(a - orc.qh[:,:h])
idx = np.argsort(d.ravel())[::-1][:8]
print('qh rel', rel(a, orc.qh[:,:h]))
for i in idx:
    l,k = np.unravel_index(i, d.shape)
    print(l,k, d[l,k], abs(orc.qh[l,k]))
