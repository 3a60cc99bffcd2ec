import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np
from oracle import niwqg_oracle as O
from test_oracle_golden import rel, L, K0, U0, TE
import niwqg_amd
for kw in (dict(nu4=7.5e8), dict(nu4=7.5e8, nu=5.0, mu=1e-8, beta=2e-11)):
    base = dict(L=L, nx=64, tmax=1e30, dt=0.1*TE, twrite=10**9, use_filter=False, U=-U0, tdiags=10**9)
    base.update(kw)
    o = O.QGOracle(**base); m = niwqg_amd.QGModel.Model(**base)
    q0 = O.lamb_dipole(o.grid, U=U0, R=2*np.pi/K0)
    o.set_q(q0); m.set_q(q0)
    print('Ke0', o.Ke, m.Ke)
    for n in range(3):
        ke_before = o.Ke
        o._step_forward(); m._step_forward()
        print(n, 'oracle dKe', o.Ke-ke_before, 'Ke', o.Ke, 'gpu Ke', m.Ke, 'ratio of increments', (m.Ke - (m.Ke if False else 0)) )
