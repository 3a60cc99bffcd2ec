"""Young & Ben Jelloul waves on an evolving barotropic QG flow, on the MI355X stepper.

Drop-in for ``niwqg.UnCoupledModel.Model`` (ref: niwqg/UnCoupledModel.py:5-76).  Like the reference,
phix/phiy are refreshed only by ``_calc_pe_niw`` (set_phi, status lines, diagnostics ticks) and stay
frozen in between; the device keeps them in separate buffers for exactly that.
"""
from . import Kernel, _lib


class Model(Kernel.Kernel):
    model_id = _lib.UNCOUPLED

    def __init__(self, **kwargs):
        self.model = " Uncoupled Model"
        super(Model, self).__init__(**kwargs)
