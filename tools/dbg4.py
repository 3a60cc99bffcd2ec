import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np
from oracle import niwqg_oracle as O
from test_oracle_golden import notebook_kwargs, rel, L, K0, U0
from test_gpu_primitives import make_ctx
from niwqg_amd import _lib
nx=64
def run(tag, qscale=1.0, kpk=3, lpk=1, **over):
    ctx, orc = make_ctx("coupled", nx, use_filter=False, **over)
    q0 = qscale*O.lamb_dipole(orc.grid, U=U0, R=2 * np.pi / K0)
    phi0 = 0.2 * O.wave_packet(orc.grid, k=kpk * K0, l=lpk*K0, R=L / 6, x0=L / 2, y0=L / 2)
    orc.set_q(q0); orc.set_phi(phi0); ctx.set_q(q0); ctx.set_phi(phi0)
    orc._step_forward(); ctx.step(1)
    a = ctx.field(_lib.F_PHIH); d = np.abs(a - orc.phih)
    i = np.unravel_index(d.argmax(), d.shape)
    print(tag, 'phih rel', rel(a, orc.phih), 'worst', i, d[i], abs(orc.phih[i]), 'q rel', rel(ctx.field(_lib.F_Q), orc.q))
run('base')
run('q=0', qscale=0.0)
run('U=0', U=0.0)
run('packet k=1', kpk=1, lpk=0.5)
run('packet k=2', kpk=2, lpk=1)
run('nuw=0', nuw=0.0)
run('q small', qscale=1e-3)
