"""Barotropic QG model on the half spectrum, on the MI355X stepper.

Drop-in for ``niwqg.QGModel.Model`` (ref: niwqg/QGModel.py:10-737), with or without its passive scalar:
spectral arrays have shape (ny, nx//2+1), ``fft``/``ifft`` have rfft2/irfft2 semantics.  The scalar c is stepped by
the same kernels as q (it travels through the row kernel paired with q in one complex transform) with its own linear
operator; ``cvar`` accumulates on the device like ``Ke`` (QGModel.py:350-394, including its use of nu for gradC2).
"""
import logging

import numpy as np
from numpy import pi

from . import _lib
from .Diagnostics import add_diagnostic, increment_diagnostics
from .Saving import (initialize_save_snapshots, save_setup, save_snapshots, save_diagnostics, flush_snapshots, flush_pending_quietly)


class Model(object):

    def __new__(cls, *args, **kwargs):
        """grids without a fused plan: the any-size mix-in in front of the class (niwqg_amd/_anysize.py; see Kernel.Kernel.__new__)"""
        nx = kwargs.get("nx", args[0] if args else 128)
        if not _lib.has_fused_plan(nx) and not getattr(cls, "_any_size", False):
            from . import _anysize
            if not _anysize.supported(nx):
                raise RuntimeError("nx = %r: the fused kernels take powers of two in [64, 8192], the any-size path even nx in "
                                   "[4, %d] and 16384" % (nx, _anysize.NX_MAX))
            cls = _anysize.specialise(cls, _anysize.QGFamily)
        return object.__new__(cls)

    def __init__(self, nx=128, ny=None, L=5e5, dt=10000., twrite=1000, tswrite=10, tmax=250000.,
                 use_filter=True, U=.0, nu4=5.e9, nu=0, mu=0, beta=0, passive_scalar=False, nu4c=5.e9,
                 nuc=0, muc=0, dealias=False, save_to_disk=False, overwrite=True, tsave_snapshots=10,
                 tdiags=10, path='output/', use_mkl=False, nthreads=1, device=0, budgets=True, slab=None, nchunks=2):
        # ref: niwqg/QGModel.py:93-139
        self.nx = nx
        self.ny = nx
        self.dtype_real, self.dtype_cplx = np.dtype('float64'), np.dtype('complex128')      # ref: niwqg/QGModel.py:147-150
        self.shape_real, self.shape_cplx = (self.ny, self.nx), (self.ny, self.nx // 2 + 1)
        self.L = L
        self.W = L
        self.dt, self.twrite, self.tswrite, self.tmax, self.tdiags = dt, twrite, tswrite, tmax, tdiags
        self.passive_scalar = passive_scalar
        self.dealias = dealias
        self.U, self.beta, self.nu4, self.nu, self.mu = U, beta, nu4, nu, mu
        self.nu4c, self.nuc, self.muc = nu4c, nuc, muc
        self.save_to_disk, self.overwrite, self.tsnaps, self.path = save_to_disk, overwrite, tsave_snapshots, path
        self.use_filter = use_filter
        self.use_mkl, self.nthreads = use_mkl, nthreads
        if dealias and not use_filter:
            raise TypeError("dealias=True: the reference itself fails here (float slice indices, "
                            "niwqg/QGModel.py:295-296)")
        self._initialize_logger()
        self._initialize_grid()
        self._initialize_filter()
        import os
        if slab is None:                 # under torch.distributed.run the model is slab-decomposed over the ranks (Kernel.py)
            slab = int(os.environ.get("WORLD_SIZE", "1")) > 1 and _lib.has_fused_plan(nx)
        phys = dict(U=U, nu=nu, nu4=nu4, mu=mu, beta=beta, passive_scalar=passive_scalar, nu4c=nu4c, nuc=nuc, muc=muc)
        self._cache, self._user = {}, {}
        self._ctx = self._create_context(phys, budgets, device, slab, nchunks)
        self.t, self.tc = 0, 0
        initialize_save_snapshots(self, self.path)      # ref: niwqg/QGModel.py:133-134; raises if no writer exists
        save_setup(self)
        self.cflmax = .5
        self.Ke = 0.0
        self._initialize_diagnostics()

    def _create_context(self, phys, budgets, device, slab, nchunks):
        if slab:
            from .slab import SlabContext
            return SlabContext(_lib.QG, self.nx, self.kk, self.ll, self.filtr, self.dt, peers=(slab if slab is not True else None),
                               nchunks=nchunks, device=(device if slab is not True else None), budgets=budgets, **phys)
        return _lib.Context(_lib.QG, self.nx, self.kk, self.ll, self.filtr, self.dt, budgets=budgets, device=device, **phys)

    def _initialize_logger(self):
        self.logger = logging.getLogger(__name__)
        if not self.logger.handlers:
            h = logging.StreamHandler()
            h.setFormatter(logging.Formatter('%(levelname)s: %(message)s'))
            self.logger.addHandler(h)
        self.logger.setLevel(10)
        self.logger.propagate = False
        self.logger.info(' Logger initialized')

    def _initialize_grid(self):
        """ref: niwqg/QGModel.py:232-269"""
        self.dk = self.dl = 2. * pi / self.L
        self.nl = self.ny
        self.nk = self.nx // 2 + 1
        self.ll = self.dl * np.append(np.arange(0., self.nx / 2), np.arange(-self.nx / 2, 0.))
        self.kk = self.dk * np.arange(0., self.nk)
        self.dx = self.L / self.nx
        self.dy = self.W / self.ny
        self.M = self.nx * self.ny

    _LAZY = ("x", "y", "k", "l", "ik", "il", "wv2", "wv", "wv4", "wv2i")
    # ETDRK4 coefficient planes (ref: niwqg/QGModel.py:426-461), from the device on demand; the "c" ones with the scalar
    _COEFF = dict(expch=(0, 0), expch_h=(0, 1), Qh=(0, 2), f0=(0, 3), fab=(0, 4), fc=(0, 5), expch2=(0, 6),
                  expchc=(2, 0), expch_hc=(2, 1), Qhc=(2, 2), f0c=(2, 3), fabc=(2, 4), fcc=(2, 5), expch2c=(2, 6))

    def __getattr__(self, name):
        if name in Model._COEFF and (Model._COEFF[name][0] == 0 or self.__dict__.get("passive_scalar")):
            eq, which = Model._COEFF[name]
            v = self._ctx.coeff(eq, which if which < 6 else 0)
            return v * v if which == 6 else v
        if name in Model._LAZY:
            d = self.__dict__
            d["x"], d["y"] = np.meshgrid(np.arange(0.5, self.nx, 1.) / self.nx * self.L,
                                         np.arange(0.5, self.ny, 1.) / self.ny * self.W)
            d["k"], d["l"] = np.meshgrid(self.kk, self.ll)
            d["ik"], d["il"] = 1j * d["k"], 1j * d["l"]
            d["wv2"] = d["k"] ** 2 + d["l"] ** 2
            d["wv"] = np.sqrt(d["wv2"])
            d["wv4"] = d["wv2"] ** 2
            nz = d["wv2"] != 0.
            d["wv2i"] = np.zeros_like(d["wv2"])
            d["wv2i"][nz] = d["wv2"][nz] ** -1
            return d[name]
        if name == "lapc" and self.__dict__.get("passive_scalar"):      # the array the reference leaves behind after a tick
            return self.ifft(-self.wv2 * self.ch)
        if name in ("C2", "gradC2", "Gamma_c") and self.__dict__.get("passive_scalar") and "_ctx" in self.__dict__:
            # every stage of the reference's step ends in _calc_derived_fields (QGModel.py:351, :365, :378, :391): after a step the
            # three are those of the NEW c-hat (Gamma_c with the fourth stage's u, v, as at a tick), tick or no tick
            self._calc_derived_fields()
            return self.__dict__[name]
        fields = {"q": _lib.F_Q, "qh": _lib.F_QH, "p": _lib.F_P, "ph": _lib.F_PH, "u": _lib.F_U, "v": _lib.F_V}
        if self.__dict__.get("passive_scalar"):
            fields.update(c=_lib.F_C, ch=_lib.F_CH)
        if name in fields:
            if name in self._user:
                return self._user[name]
            if name not in self._cache:
                if name in ("u", "v") and self.__dict__.get("_uv_stage4"):
                    # after a step the reference's u, v are those of its fourth stage (QGModel.py:375 vs :396-397)
                    ph4 = -self.wv2i * self._ctx.field(_lib.F_QH_STAGE4)
                    self._cache["u"], self._cache["v"] = self.ifft(-self.il * ph4), self.ifft(self.ik * ph4)
                else:
                    self._cache[name] = self._ctx.field(fields[name])
            return self._cache[name]
        raise AttributeError(name)

    def _initialize_filter(self):
        """ref: niwqg/QGModel.py:283-300"""
        k, l = self.kk[None, :], self.ll[:, None]
        if self.use_filter:
            cphi = 0.65 * pi
            wvx = np.sqrt((k * self.dx) ** 2. + (l * self.dy) ** 2.)
            self.filtr = np.exp(-23.6 * (wvx - cphi) ** 4.)
            self.filtr[wvx <= cphi] = 1.
            self.logger.info(' Using filter')
        else:
            self.filtr = np.ones((self.nl, self.nk))
            self.logger.info(' No dealiasing; no filter')

    def _dirty(self):
        self._cache.clear()
        self._user.clear()

    # --- reference API
    def fft(self, x):
        """numpy.fft.rfft2 semantics.  ref: niwqg/QGModel.py:551"""
        return self._ctx.rfft2(x)

    def ifft(self, x):
        """numpy.fft.irfft2 semantics.  ref: niwqg/QGModel.py:552"""
        return self._ctx.irfft2(x)

    def set_q(self, q):
        """ref: niwqg/QGModel.py:507-520.  The reference's set_q does not touch self.u, self.v: they stay whatever the last
        jacobian_psi_q, _calc_cfl or step left behind, i.e. those of the OLD psi, until the next of these calls.  They are read
        out before the new q goes in and kept as they are."""
        keep = (self.u, self.v) if self.__dict__.get("_uv_defined") else None
        self._ctx.set_q(q)
        self._dirty()
        if keep is not None:
            self._uv_stage4 = False
            self._user["u"], self._user["v"] = keep
        self._user["q"] = q
        if self._ctx.budgets_enabled:
            self._ctx.scalar(_lib.S_KE)             # drop increments that belong to the old state
        self.Ke = self._calc_ke_qg()

    def set_c(self, c):
        """ref: niwqg/QGModel.py:522-534"""
        if not self.passive_scalar:
            raise RuntimeError("set_c: the model was built with passive_scalar=False")
        self._ctx.set_c(c)
        self._dirty()
        self._user["c"] = c
        if self._ctx.budgets_enabled:
            self._ctx.scalar(_lib.S_PW)             # drop increments that belong to the old state
        self.cvar = self.spec_var(self.ch)

    def jacobian_psi_c(self):
        """ik F[u c] + il F[v c] (ref: niwqg/QGModel.py:483-495); diagnostics ticks only -- inside a step the row
        kernel forms these products next to those of q."""
        if "u" in self._user:                          # u, v older than the current q (set_q leaves them alone: QGModel.py:507-520)
            u, v = self._user["u"], self._user["v"]
        elif self.__dict__.get("_uv_stage4"):
            # the reference's u, v at a tick are those of the last jacobian_psi_q call, i.e. of the state at which
            # the step evaluated its fourth stage, not of the new state (QGModel.py:375 vs :396)
            ph4 = -self.wv2i * self._ctx.field(_lib.F_QH_STAGE4)
            u, v = self.ifft(-self.il * ph4), self.ifft(self.ik * ph4)
        elif hasattr(self._ctx, "jacobian_psi_c"):
            return self._ctx.jacobian_psi_c()          # u, v of the current psi: the row kernel's own products, as in a step
        else:
            u, v = self.u, self.v
        return self.ik * self.fft(u * self.c) + self.il * self.fft(v * self.c)

    def _invert(self):
        self._ctx.invert()
        self._cache.pop("ph", None)
        self._cache.pop("p", None)

    # hooks the reference declares and never calls (ref: niwqg/QGModel.py:271-281, :303-304)
    def _initialize_background(self):
        raise NotImplementedError('needs to be implemented by Model subclass')

    def _initialize_inversion_matrix(self):
        raise NotImplementedError('needs to be implemented by Model subclass')

    def _initialize_forcing(self):
        raise NotImplementedError('needs to be implemented by Model subclass')

    def _do_external_forcing(self):
        pass

    def jacobian_psi_q(self):
        """ik F[u q] + il F[v q] on the half spectrum, [0,0] NOT zeroed.  ref: niwqg/QGModel.py:469-481"""
        self._uv_current()
        return self._ctx.jacobian_psi_q()

    def _uv_current(self):
        """jacobian_psi_q and _calc_cfl leave the u, v of the CURRENT psi behind (QGModel.py:473-474, :626-627)"""
        self._uv_defined = True
        self._user.pop("u", None)
        self._user.pop("v", None)
        if self.__dict__.get("_uv_stage4"):
            self._uv_stage4 = False
            self._cache.pop("u", None)
            self._cache.pop("v", None)

    def spec_var(self, ph):
        """ref: niwqg/QGModel.py:611-619"""
        var_dens = 2. * np.abs(ph) ** 2 / self.M ** 2
        var_dens[:, 0] *= 0.5
        var_dens[:, -1] *= 0.5
        var_dens[0, 0] = 0
        return var_dens.sum()

    def _step_etdrk4(self):
        self._ctx.step(1)
        self._after_steps()

    def _after_steps(self):
        self._dirty()
        self._stepped = True
        self._uv_stage4 = self._uv_defined = True
        if self._ctx.budgets_enabled:
            self.Ke += self._ctx.scalar(_lib.S_KE)
            if self.passive_scalar:
                self.cvar += self._ctx.scalar(_lib.S_PW)        # ref: niwqg/QGModel.py:394
        if self.passive_scalar:
            for k in ("C2", "gradC2", "Gamma_c"):               # recomputed by the step itself in the reference: lazily here
                self.__dict__.pop(k, None)

    def _snapshot_fields(self):
        """the reference always asks for 't', 'q', 'c' (niwqg/QGModel.py:221); c exists only with the passive scalar"""
        return ['t', 'q', 'c'] if self.passive_scalar else ['t', 'q']

    def _step_forward(self):
        self._step_etdrk4()
        increment_diagnostics(self)
        self._print_status()
        save_snapshots(self, fields=self._snapshot_fields())

    def _quiet_steps(self, n_left):
        for j in range(n_left):
            tcb = self.tc + j
            if (tcb % self.tdiags) == 0 or ((tcb + 1) % self.twrite) == 0:
                return j
            if self.save_to_disk and ((tcb + 1) % self.tsnaps) == 0:
                return j
        return n_left - 1

    def _steps_left(self, cap):
        t, n = self.t, 0
        while t < self.tmax and n < cap:
            t += self.dt
            n += 1
        return n

    def run(self):
        """ref: niwqg/QGModel.py:184-207"""
        self._defer_snapshots = True              # snapshots are written while the next batch of steps runs
        try:
            if self.save_to_disk:
                save_snapshots(self, fields=self._snapshot_fields())
            while self.t < self.tmax:
                quiet = self._quiet_steps(self._steps_left(4096))
                if quiet > 0:
                    self._ctx.step(quiet)
                    flush_snapshots(self)
                    for _ in range(quiet):
                        self.tc += 1
                        self.t += self.dt
                    self._after_steps()
                self._step_forward()
            flush_snapshots(self)
            if self.save_to_disk:
                save_diagnostics(self)
        finally:
            self._defer_snapshots = False
            flush_pending_quietly(self)      # (a failure in here must not mask the exception that is already on its way)

    def run_with_snapshots(self, tsnapstart=0., tsnapint=432000.):
        tsnapints = np.ceil(tsnapint / self.dt)
        while self.t < self.tmax:
            self._step_forward()
            if self.t >= tsnapstart and (self.tc % tsnapints) == 0:
                yield self.t
        return

    def _print_status(self):
        """ref: niwqg/QGModel.py:554-578"""
        self.tc += 1
        self.t += self.dt
        if (self.tc % self.twrite) == 0:
            self.ke = self._calc_ke_qg()
            self.cfl = self._calc_cfl()
            self.logger.info('Step: %i, Time: %4.3e, P: %4.3e , Ke: %4.3e, CFL: %4.3f',
                             self.tc, self.t, self.t / self.tmax, self.ke, self.cfl)
            assert self.cfl < self.cflmax, self.logger.error('CFL condition violated')

    def _calc_ke_qg(self):
        return self._ctx.scalar(_lib.S_KE_QG)

    # The purely spectral diagnostics come from the half-spectrum sums of ONE device pass (nq_diagnostics, entries
    # [6..14], include/niwqg_amd.h): no plane is downloaded at a tick.
    def _dsums(self):
        if "_dsums" not in self._cache:
            self._cache["_dsums"] = self._ctx.diagnostic_sums()
        return self._cache["_dsums"]

    def _calc_ens(self):
        """0.5 mean(q^2) (ref: niwqg/QGModel.py:657) by Parseval"""
        return 0.5 * self._dsums()[6] / float(self.M) ** 2

    def _calc_ep_psi(self):
        """ref: niwqg/QGModel.py:588-593 by Parseval"""
        s = self._dsums()
        return (self.nu4 * s[12] + self.nu * s[13] + self.mu * s[14]) / float(self.M) ** 2

    def _calc_chi_q(self):
        """ref: niwqg/QGModel.py:606-609"""
        return -self.nu4 * self._dsums()[7] / float(self.M) ** 2

    def _calc_cfl(self):
        self._uv_current()
        return self._ctx.scalar(_lib.S_CFL) * self.dt / self.dx      # max reduction on the device

    # The passive scalar's tick entries come from five more device sums (nq_diagnostics [16..20]): nothing of c is
    # downloaded at a tick.  M2 = (nx ny)^2; mean(lap c ^2) = s[18]/M2, mean(lap^2 c lap c) = -s[19]/M2 by Parseval.
    def _calc_ep_c(self):
        """ref: niwqg/QGModel.py:595-598 (nu, not nuc, multiplies gradC2 there)"""
        if not self.passive_scalar:
            return -2 * self.nu4c * 0. - 2 * self.nu * self.gradC2 - 2 * self.muc * self.C2
        return -2 * self.nu4c * self._dsums()[18] / float(self.M) ** 2 - 2 * self.nu * self.gradC2 - 2 * self.muc * self.C2

    def _calc_chi_c(self):
        """ref: niwqg/QGModel.py:600-604"""
        if not self.passive_scalar:
            return 0.0
        s, M2 = self._dsums(), float(self.M) ** 2
        return -2 * self.nu4c * s[19] / M2 - 2 * self.nu * s[18] / M2 - 2 * self.muc * self.gradC2

    def _initialize_diagnostics(self):
        """ref: niwqg/QGModel.py:632-722 (the passive-scalar entries report zeros, as the reference
        does when passive_scalar=False, QGModel.py:734-737)"""
        self.diagnostics = dict()
        self.C2, self.gradC2, self.cvar, self.Gamma_c = 0., 0., 0., 0.
        if not self.passive_scalar:
            self.lapc = np.array([0.])
        table = [
            ('time', 'Time', 'seconds', lambda s: s.t),
            ('ke_qg', 'Quasigeostrophic Kinetic Energy', r'm^2 s^{-2}', lambda s: s._calc_ke_qg()),
            ('Ke', 'Quasigeostrophic Kinetic Energy, from energy equation', r'm^2 s^{-2}', lambda s: s.Ke),
            ('ens', 'Quasigeostrophic Potential Enstrophy', r's^{-2}', lambda s: s._calc_ens()),
            ('ep_psi', 'The hyperviscous dissipation of QG kinetic energy', r'$m^2 s^{-3}$',
             lambda s: s._calc_ep_psi()),
            ('chi_q', 'The hyperviscous dissipation of QG kinetic energy', r'$s^{-3}$', lambda s: s._calc_chi_q()),
            ('C2', 'Passive tracer variance', r'[scalar]^2', lambda s: s.C2),
            ('cvar', 'Passive tracer variance, from variance equation', r'[scalar]^2', lambda s: s.cvar),
            ('gradC2', 'Gradient of Passive tracer variance', r'[scalar]^2 / m^2', lambda s: s.gradC2),
            ('Gamma_c', 'Rate of generation of passive tracer gradient variance', r'[scalar]^2 / (m^2 s)',
             lambda s: s.Gamma_c),
            ('ep_c', 'The dissipation of tracer variance', r'$s^{-3}$', lambda s: s._calc_ep_c()),
            ('chi_c', 'The dissipation of tracer gradient variance', r'$s^{-3}$', lambda s: s._calc_chi_c()),
        ]
        for name, desc, units, fn in table:
            add_diagnostic(self, name, description=desc, units=units, types='scalar', function=fn)

    def _calc_derived_fields(self):
        """ref: niwqg/QGModel.py:724-737 from the device sums (Parseval); ``lapc`` is left as an on-demand attribute"""
        if self.passive_scalar:
            s, M2 = self._dsums(), float(self.M) ** 2
            self.C2 = s[16] / M2
            self.gradC2 = s[17] / M2
            self.Gamma_c = 2 * s[20] / M2
            self.__dict__.pop("lapc", None)
        else:
            self.C2, self.gradC2, self.cvar, self.Gamma_c = 0., 0., 0., 0.
            self.lapc = np.array([0.])
