"""Young & Ben Jelloul waves on a STEADY barotropic QG flow, on the MI355X stepper.

Drop-in for ``niwqg.YBJModel.Model`` (ref: niwqg/YBJModel.py:4-158).  Only phi is stepped
(YBJModel.py:52-87); u, v, q_psi = q are those of ``set_q``.  Quirks of the reference kept exactly:
* the refraction factor ``phi`` is refreshed only after the step, ``phix``/``phiy`` before every stage, so that after a
  step ``phix``/``phiy`` are the gradients of the stage-2 result, not of the new state (the device keeps them in the
  buffer the last stage read them from; ``_calc_pe_niw`` refreshes them, as everywhere in the Kernel family);
* ``jacobian_psi_phi`` does not zero its [0,0] entry here (YBJModel.py:123-133, unlike Kernel.py:468);
* the step accumulates no energy budgets: ``Ke, Pw, Kw`` stay at their ``set_q``/``set_phi`` values;
* the physical streamfunction ``p`` is allocated but never filled (YBJModel.py:45-46, :141-146), so the ``ep_psi``
  diagnostic (Kernel.py:635-640) sees p = 0 and keeps its nu4 term only.
"""
import numpy as np

from . import Kernel, _lib


class Model(Kernel.Kernel):
    model_id = _lib.YBJ

    def __init__(self, **kwargs):
        self.model = " YBJ Model (Steady QG flow)"
        kwargs["budgets"] = False
        super(Model, self).__init__(**kwargs)

    def jacobian_psi_phi(self):
        """F[u phix + v phiy] with [0,0] left alone (ref: niwqg/YBJModel.py:123-133)"""
        return self._ctx.jacobian_psi_phi()

    def _calc_grad_phi(self):
        """ref: niwqg/YBJModel.py:135-139"""
        self._ctx.refresh_grad_phi()
        self._cache.pop("phix", None)
        self._cache.pop("phiy", None)

    def _invert(self):
        """ref: niwqg/YBJModel.py:141-146 (psi is steady: nothing to do after set_q)"""
        pass

    @property
    def p(self):
        """the reference allocates p as zeros and its purely spectral _invert never fills it (niwqg/YBJModel.py:43, :141-146)"""
        return np.zeros((self.ny, self.nx))

    def _calc_ep_psi(self):
        """ref: niwqg/Kernel.py:635-640 with the reference's p = 0 (see module docstring)"""
        return self.nu4 * self._dsums()[12] / self._M2
