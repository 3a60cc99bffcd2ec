import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np
from oracle import niwqg_oracle as O
from oracle import reduced_pipeline as R
from test_oracle_golden import notebook_kwargs, rel, L, K0, U0
from test_gpu_primitives import make_ctx
from niwqg_amd import _lib
nx=64
for use_filter in (False, True):
    ctx, orc = make_ctx("coupled", nx, use_filter=use_filter)
    names = ["E", "Eh", "Q", "f0", "fab", "fc"]
    for eq,co in ((0,orc.coef_q),(1,orc.coef_w)):
        for i,nm in enumerate(names):
            mine = ctx.coeff(eq,i); ref = co[nm][:, :33] if eq==0 else co[nm]
            e = np.abs(mine-ref)/np.abs(ref)
            print(use_filter, eq, nm, 'max rel', e.max(), 'count>1e-13', (e>1e-13).sum())
    q0 = O.lamb_dipole(orc.grid, U=U0, R=2 * np.pi / K0)
    phi0 = 0.2 * O.wave_packet(orc.grid, k=3 * K0, l=K0, R=L / 6, x0=L / 2, y0=L / 2)
    kw = notebook_kwargs(nx, use_filter)
    red = R.ReducedNIWQG("coupled", **kw)
    orc.set_q(q0); orc.set_phi(phi0); ctx.set_q(q0); ctx.set_phi(phi0); red.set_q(q0); red.set_phi(phi0)
    for n in range(10):
        orc._step_forward(); ctx.step(1); red.step()
        print(n+1, 'gpu-vs-ref q', rel(ctx.field(_lib.F_Q), orc.q), 'phi', rel(ctx.field(_lib.F_PHI), orc.phi),
              '| numpy-reduced-vs-ref q', rel(red.q, orc.q), 'phi', rel(red.phi, orc.phi), '|phi|', np.abs(orc.phi).max())
