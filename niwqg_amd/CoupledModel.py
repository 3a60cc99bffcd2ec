"""Xie & Vanneste coupled NIW-QG model on the MI355X stepper.

Drop-in for ``niwqg.CoupledModel.Model`` (ref: niwqg/CoupledModel.py:5-152): same constructor
keywords, ``set_q``/``set_phi``/``run`` and attributes; the wave-PV inversion and the ETDRK4 stages
run in HIP kernels (niwqg_amd/csrc/nq_step.hpp: k_x_wavepv, k_s_invert, k_x_products, k_s_q, k_s_phi).
"""
from . import Kernel, _lib
from .Diagnostics import add_diagnostic


class Model(Kernel.Kernel):
    model_id = _lib.COUPLED

    def __init__(self, **kwargs):
        self.model = " Coupled Model"
        super(Model, self).__init__(**kwargs)

    def jacobian_phic_phi(self):
        """F[Re i(phix* phiy - phiy* phix)], [0,0] = 0; refreshes phix, phiy from the current phih.
        ref: niwqg/CoupledModel.py:59-73"""
        self._cache.pop("phix", None)
        self._cache.pop("phiy", None)
        return self._ctx.jacobian_phic_phi()

    # Fields the reference leaves behind after a diagnostics tick (CoupledModel.py:105-112): formed on demand from the
    # gathered spectra with the device FFT seam, never inside the tick (whose scalars come from the device sums).
    _TICK_FIELDS = ("phq", "phw", "uq", "vq", "uw", "vw")

    def __getattr__(self, name):
        if name in Model._TICK_FIELDS:
            stale = self.__dict__.get("_tick_stale")            # steps since the last tick: the tick's spectra (Kernel._tick_field)
            if name == "phq":
                return -self.wv2i * (self._tick_field("qh") if stale else self.qh)
            if name == "phw":
                return self.wv2i * (self._tick_field("qwh") if stale else self.qwh)
            ph = self.phq if name[1] == "q" else self.phw
            return self.ifft((-self.il if name[0] == "u" else self.ik) * ph).real
        return super(Model, self).__getattr__(name)

    def _calc_ke_qg_decomp(self):
        """ref: niwqg/CoupledModel.py:99-113 from the half-spectrum sums of the device tick (Parseval)"""
        s = self._dsums()
        self.ke_qg_q = 0.5 * s[8] / self._M2
        self.ke_qg_w = 0.5 * s[9] / self._M2
        self.ke_qg_qw = -s[10] / self._M2

    def _initialize_class_diagnostics(self):
        """ref: niwqg/CoupledModel.py:115-136"""
        add_diagnostic(self, 'ke_qg_q', description='Quasigeostrophic Kinetic Energy, q-flow',
                       units=r'm^2 s^{-2}', types='scalar', function=(lambda self: self.ke_qg_q))
        add_diagnostic(self, 'ke_qg_w', description='Quasigeostrophic Kinetic Energy, w-flow',
                       units=r'm^2 s^{-2}', types='scalar', function=(lambda self: self.ke_qg_w))
        add_diagnostic(self, 'ke_qg_qw', description='Quasigeostrophic Kinetic Energy, cross-term q-w',
                       units=r'm^2 s^{-2}', types='scalar', function=(lambda self: self.ke_qg_qw))

    def _calc_class_derived_fields(self):
        self._calc_ke_qg_decomp()
