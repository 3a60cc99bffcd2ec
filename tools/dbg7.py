import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np
import niwqg_amd
from test_oracle_golden import notebook_kwargs
g = np.load('tests/golden/g2_coupled_64_nofilter.npz')
for kw in (dict(budgets=False), dict(budgets=True)):
    m = niwqg_amd.CoupledModel.Model(**notebook_kwargs(64, False), **kw)
    m.set_q(g["q0"]); m.set_phi(g["phi0"])
    print('set ok', kw, flush=True)
    m._ctx.step(1); m._ctx.sync(); print('step ok', flush=True)
    if kw['budgets']: print(m._ctx.take_budget_increments(), flush=True)
