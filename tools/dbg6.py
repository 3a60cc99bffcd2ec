import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np
import niwqg_amd
from niwqg_amd import _lib
def run(tag, phi_amp=0.0, nx=128, **kw):
    base = dict(use_filter=False, nu4=1e11, nu4w=0., nu=0, tdiags=10**9, nx=nx)
    base.update(kw)
    m = niwqg_amd.CoupledModel.Model(**base)
    k, l = 2 * np.pi * 5 / 5e5, 2 * np.pi * 9 / 5e5
    qi = np.sin(k * m.x + l * m.y)
    m.set_q(qi); m.set_phi(qi*phi_amp)
    out=[]
    for n in range(4):
        m._ctx.step(1); m._dirty()
        out.append((np.isfinite(m.qh).all(), np.isfinite(m.phih).all(), float(np.abs(m.q).max()) if np.isfinite(m.q).all() else 'nan'))
    print(tag, out)
run('base')
run('nobud', budgets=False)
run('phi', phi_amp=0.01)
run('nx64', nx=64)
run('nx256', nx=256)
run('nu4small', nu4=1e9)
