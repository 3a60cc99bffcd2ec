"""Host-side mirror of ``niwqg.Kernel.Kernel`` for the MI355X stepper.

Same constructor keywords, methods and attributes as the reference (ref: niwqg/Kernel.py:70-152 and
SURVEY.md section 8b); every time step runs in HIP kernels behind the C ABI (include/niwqg_amd.h).
Field attributes (``q``, ``phi``, ``qh`` ...) are fetched from the device on access and cached until
the state changes.  There is no CPU path: without the HIP library and a GPU the constructor raises.
"""
import logging

import numpy as np
from numpy import pi

from . import _lib
from .Diagnostics import add_diagnostic, increment_diagnostics
from .Saving import (initialize_save_snapshots, save_setup, save_snapshots, save_diagnostics, flush_snapshots, flush_pending_quietly)

_DEVICE_FIELDS = {"q": _lib.F_Q, "p": _lib.F_P, "phi": _lib.F_PHI, "phih": _lib.F_PHIH, "u": _lib.F_U,
                  "v": _lib.F_V, "phix": _lib.F_PHIX, "phiy": _lib.F_PHIY}


def hermitian_full(half):
    """(ny, nx/2+1) half spectrum of a real field -> the reference's (ny, nx) layout."""
    n = half.shape[0]
    full = np.empty((n, n), complex)
    full[:, :n // 2 + 1] = half
    inner = half[:, 1:n // 2]
    full[:, n // 2 + 1:] = np.conj(np.roll(inner[::-1, :], 1, axis=0))[:, ::-1]
    return full


def project_self_mirrored_columns(half):
    """Hermitian part (in l) of the k=0 and k=N/2 columns: what fft(real field) has there."""
    out = half.copy()
    for c in (0, half.shape[1] - 1):
        col = half[:, c]
        out[:, c] = 0.5 * (col + np.conj(np.roll(col[::-1], 1)))
    return out


class Kernel(object):
    """Pseudo-spectral NIW-QG kernel; subclasses fix the inversion (``model_id``)."""

    model_id = None

    def __new__(cls, *args, **kwargs):
        """Grids the fused kernels have no plan for (anything but a power of two in [64, 8192]) get the any-size mix-in in front
        of the class: same class name, same surface, every device operation restated on whole planes (niwqg_amd/_anysize.py)."""
        nx = kwargs.get("nx", args[0] if args else 128)
        if not _lib.has_fused_plan(nx) and not getattr(cls, "_any_size", False):
            from . import _anysize
            if not _anysize.supported(nx):
                raise RuntimeError("nx = %r: the fused kernels take powers of two in [64, 8192], the any-size path even nx in "
                                   "[4, %d] and 16384" % (nx, _anysize.NX_MAX))
            cls = _anysize.specialise(cls, _anysize.KernelFamily)
        return object.__new__(cls)

    def __init__(self, nx=128, ny=None, L=5e5, dt=10000., twrite=1000., tmax=250000., use_filter=True,
                 cflmax=0.8, U=.0, f=1.e-4, N=0.01, m=0.025, g=9.81, nu4=0, nu4w=0, nu=20, nuw=50., mu=0,
                 muw=0, dealias=False, save_to_disk=False, overwrite=True, tsave_snapshots=10, tdiags=10,
                 path='output/', use_mkl=False, nthreads=1, device=0, budgets=True, exact_qh=False, slab=None,
                 nchunks=2):
        # ref: niwqg/Kernel.py:100-137 -- note ny is ignored there too (quirk Q3)
        self.nx = nx
        self.ny = nx
        self.L = L
        self.W = L
        self.dt = dt
        self.twrite = twrite
        self.tmax = tmax
        self.dealias = dealias
        self.U, self.g = U, g
        self.nu4, self.nu4w, self.nu, self.nuw, self.mu, self.muw = nu4, nu4w, nu, nuw, mu, muw
        self.f, self.N, self.m = f, N, m
        self.kappa = self.m * self.f / self.N
        self.kappa2 = self.kappa ** 2
        self.cflmax = cflmax
        self.hslash = self.f / self.kappa2
        self.save_to_disk = save_to_disk
        self.overwrite = overwrite
        self.tsnaps = tsave_snapshots
        self.tdiags = tdiags
        self.path = path
        self.use_filter = use_filter
        self.use_mkl, self.nthreads = use_mkl, nthreads
        # The reference's 2/3 mask (Kernel.py:277-281) is not mirror-symmetric, so its q-hat is not Hermitian:
        # the device then keeps a second half-spectrum copy ("dual copy").  With symmetric filters the
        # reference's qh is reproduced on the whole plane WITHOUT it (the Nyquist-row passenger is carried as one
        # extra row: nq_get_qh_passenger); exact_qh=True still selects the dual copy explicitly.
        self._dual = bool(exact_qh) or (bool(dealias) and not use_filter)
        # ref: niwqg/CoupledModel.py:38-42 / UnCoupledModel.py / YBJModel.py (_allocate_variables): array types and shapes
        self.dtype_real, self.dtype_cplx = np.dtype('float64'), np.dtype('complex128')
        self.shape_real = self.shape_cplx = (self.ny, self.nx)

        self._initialize_logger()
        self.logger.info(self.model)
        self._initialize_grid()
        self._initialize_filter()
        # One simulation over several GPUs: under torch.distributed.run (WORLD_SIZE > 1) the model
        # is slab-decomposed over the ranks, every rank constructing the same Model(...); slab=P puts P peer ranks into
        # this process on one GPU (the test double); slab=False forces a single-GPU model inside a multi-process job
        # (ensemble members, replicas).
        import os
        if slab is None:
            slab = int(os.environ.get("WORLD_SIZE", "1")) > 1 and _lib.has_fused_plan(nx)
        phys = dict(U=U, f=f, kappa2=self.kappa2, nu=nu, nu4=nu4, mu=mu, nuw=nuw, nu4w=nu4w, muw=muw)
        self._ctx = self._create_context(phys, budgets, device, slab, nchunks)
        self._cache = {}
        self._user = {}
        self._initialize_time()
        # ref: niwqg/Kernel.py:145-148 (set-up file, snapshot directory); raises if no writer exists (Saving.py)
        initialize_save_snapshots(self, self.path)
        save_setup(self)
        self._initialize_diagnostics()
        self.Ke = self.ke = 0.0
        self.Pw = self.Kw = 0.0

    # ------------------------------------------------------------------ setup
    def _create_context(self, phys, budgets, device, slab, nchunks):
        """the device side of the model: one fused-kernel context, or a slab-decomposed one (the any-size mix-in overrides this)"""
        if slab:
            from .slab import SlabContext
            return SlabContext(self.model_id, self.nx, self.kk, self.ll, self.filtr, self.dt, peers=(slab if slab is not True else None),
                               nchunks=nchunks, device=(device if slab is not True else None), budgets=budgets,
                               dual_q=self._dual, **phys)
        return _lib.Context(self.model_id, self.nx, self.kk, self.ll, self.filtr, self.dt, budgets=budgets, device=device,
                            dual_q=self._dual, **phys)

    def _initialize_logger(self):
        """ref: niwqg/Kernel.py:286-304"""
        self.logger = logging.getLogger(__name__)
        if not self.logger.handlers:
            h = logging.StreamHandler()
            h.setFormatter(logging.Formatter('%(levelname)s: %(message)s'))
            self.logger.addHandler(h)
        self.logger.setLevel(10)
        self.logger.propagate = False
        self.logger.info(' Logger initialized')

    def _initialize_time(self):
        self.t = 0
        self.tc = 0

    def _initialize_grid(self):
        """1-D wavenumbers eagerly, 2-D planes lazily.  ref: niwqg/Kernel.py:227-265"""
        self.dk = 2. * pi / self.L
        self.dl = 2. * pi / self.L
        self.nl = self.ny
        self.nk = self.nl
        self.ll = self.dl * np.append(np.arange(0., self.nx / 2), np.arange(-self.nx / 2, 0.))
        self.kk = self.ll.copy()
        self.dx = self.L / self.nx
        self.dy = self.W / self.ny
        self.M = self.nx * self.ny

    _LAZY = ("x", "y", "k", "l", "ik", "il", "wv2", "wv", "wv4", "wv2i")
    _COEFF = dict(expch=(0, 0), expch_h=(0, 1), Qh=(0, 2), f0=(0, 3), fab=(0, 4), fc=(0, 5), expch2=(0, 6),
                  expchw=(1, 0), expch_hw=(1, 1), Qhw=(1, 2), f0w=(1, 3), fabw=(1, 4), fcw=(1, 5), expch2w=(1, 6))

    def __getattr__(self, name):
        if name in Kernel._LAZY:
            self._build_planes()
            return self.__dict__[name]
        if name in _DEVICE_FIELDS or name in ("qh", "ph", "qwh", "q_psi", "qw"):
            return self._field(name)
        if name in Kernel._COEFF:         # ETDRK4 coefficient planes (ref: niwqg/Kernel.py:417-454), from the device on demand
            eq, which = Kernel._COEFF[name]
            v = self._ctx.coeff(eq, which if which < 6 else 0)
            if eq == 0:                   # device keeps k = 0..nx/2; c(l,-k) = conj c(l,k), and so is every plane derived from it
                v = np.concatenate([v, np.conj(v[:, 1:self.nx // 2][:, ::-1])], axis=1)
            return v * v if which == 6 else v          # expch2 = exp(2 c dt)
        if name == "c":                  # the linear operator the reference leaves behind: the wave equation's (Kernel.py:440-442)
            advect = np.zeros((self.nl, self.nk), complex) - 1j * self.k * self.U
            return advect + (-self.nu4w * self.wv4 - 0.5j * self.f * (self.wv2 / self.kappa2) - self.nuw * self.wv2 - self.muw)
        if name in Kernel._EC_NAMES and self.__dict__.get("_ec_stage4") and "_ctx" in self.__dict__:
            self._energy_conversion_of_stage4()
            return self.__dict__[name]
        if name == "lapphi":             # what the last _calc_energy_conversion left behind: a step's fourth stage's, or a tick's
            if self.__dict__.get("_ec_stage4"):
                if "lapphi" not in self._cache:
                    self._energy_conversion_of_stage4()
                return self._cache["lapphi"]
            return self.ifft(-self.wv2 * (self._tick_field("phih") if self.__dict__.get("_tick_stale") else self.phih))
        if name == "upsilon":
            a2 = np.abs(self._tick_field("phi") if self.__dict__.get("_tick_stale") else self.phi) ** 2
            return a2 - a2.mean()
        if self.__dict__.get("model_id", type(self).model_id) == _lib.COUPLED:
            # what CoupledModel._invert leaves behind (ref: niwqg/CoupledModel.py:83-92), on demand through the FFT seam
            if name == "phi2":
                return np.abs(self.phi) ** 2
            if name == "gphi2h":
                return -self.wv2 * self.fft(self.phi2)
            if name == "pw":
                return self.ifft(self.wv2i * self.qwh).real
            if name == "pv":
                return self.ifft(-(self.wv2i * self.qh)).real
        raise AttributeError(name)

    def _build_planes(self):
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

    def _initialize_filter(self):
        """ref: niwqg/Kernel.py:267-284 (same arithmetic, built from the 1-D wavenumbers)"""
        k, l = self.kk[None, :], self.ll[:, None]
        if self.use_filter:
            cphi = 0.65 * pi
            wvx = np.sqrt((k * self.dx) ** 2. + (l * self.dy) ** 2.)
            self.filtr = np.exp(-23.6 * (wvx - cphi) ** 4.)
            self.filtr[wvx <= cphi] = 1.
            self.logger.info(' Using filter')
        elif self.dealias:
            self.filtr = np.ones((self.nl, self.nk))
            self.filtr[self.nx // 3:2 * self.nx // 3, :] = 0.
            self.filtr[:, self.ny // 3:2 * self.ny // 3] = 0.
            self.logger.info(' Dealiasing with 2/3 rule')
        else:
            self.filtr = np.ones((self.nl, self.nk))
            self.logger.info(' No dealiasing; no filter')

    # ------------------------------------------------------------------ device state views
    def _dirty(self):
        self._cache.clear()
        self._user.clear()

    def _full_qh(self, half, minus, passenger):
        """the reference's (ny, nx) qh from the device's half spectrum"""
        n = self.nx
        v = hermitian_full(half)
        if minus is not None:           # k < 0 side from the second copy: qh(-l,-k) = conj(X-(l,k))
            inner = minus[:, 1:n // 2]
            v[:, n // 2 + 1:] = np.conj(np.roll(inner[::-1, :], 1, axis=0))[:, ::-1]
        else:                           # the anti-Hermitian passenger of row ny/2 (ref Kernel.py:471-486, :327), one device row
            a = passenger[1:n // 2]
            v[n // 2, 1:n // 2] += a
            v[n // 2, n // 2 + 1:] -= np.conj(a)[::-1]
        return v

    # What only a diagnostics tick refreshes on the reference's instance -- upsilon (Kernel.py:618, inside the 'skew' diagnostic),
    # CoupledModel's phq, phw, uq, vq, uw, vw (CoupledModel.py:99-113), YBJModel's lapphi (its step never calls
    # _calc_energy_conversion) -- stays the TICK's until the next tick, however many steps follow (golden g19).  The tick keeps the
    # spectra they derive from on the device (nq_tick_snapshot); the arrays are rebuilt from those on demand.
    _tick_stale = _tick_taken = False

    def _tick_snapshot(self):
        self._ctx.tick_snapshot()
        self._tick_passenger = None if self._dual else self._ctx.qh_passenger()
        self._tick_taken, self._tick_stale = True, False
        self._tick_cache = {}

    def _tick_field(self, name):
        """qh, qwh (full planes), phih, phi of the last tick"""
        t, c = self._tick_cache, self._ctx
        if name not in t:
            if name == "qh":
                t[name] = self._full_qh(c.field(_lib.F_QH_TICK), c.field(_lib.F_QH_MINUS_TICK) if self._dual else None, self._tick_passenger)
            elif name == "qwh":
                t[name] = hermitian_full(c.field(_lib.F_QWH_TICK))
            elif name == "phih":
                t[name] = c.field(_lib.F_PHIH_TICK)
            elif name == "phi":
                t[name] = self.ifft(self._tick_field("phih"))
        return t[name]

    def _field(self, name):
        if name in self._user:
            return self._user[name]
        if name not in self._cache:
            c = self._ctx
            if name == "qh":
                v = self._full_qh(c.field(_lib.F_QH), c.field(_lib.F_QH_MINUS) if self._dual else None,
                                  None if self._dual else c.qh_passenger())
            elif name == "ph":
                v = hermitian_full(project_self_mirrored_columns(c.field(_lib.F_PH)))
            elif name == "qwh":
                v = hermitian_full(c.field(_lib.F_QWH)) if self.model_id == _lib.COUPLED else None
            elif name == "qw":
                v = c.field(_lib.F_QW)
            elif name == "q_psi":
                v = self._field("q") - self._field("qw") if self.model_id == _lib.COUPLED else self._field("q")
            elif name in ("u", "v") and self._uv_stage4:
                self._cache["u"], self._cache["v"] = self._uv_of_stage4()
                return self._cache[name]
            else:
                v = c.field(_DEVICE_FIELDS[name])
            self._cache[name] = v
        return self._cache[name]

    # After a step the reference's self.u, self.v are NOT those of the new state: the last jacobian_psi_q call of
    # _step_etdrk4 is the fourth stage's (Kernel.py:364-368), the final update and its _invert come after it (:381-397).
    # They stay like that until the next jacobian_psi_q, set_q or diagnostics tick (Kernel.py:681).  The step keeps that
    # stage's qh and phih in its rotating buffers until the next step, so u, v are rebuilt from them on demand, on the host
    # side of the FFT seam, with the reference's own expressions (CoupledModel.py:75-97, UnCoupledModel.py:54-64).
    _uv_stage4 = False

    def _uv_of_stage4(self):
        return self._fields_of_stage4()[:2]

    def _fields_of_stage4(self):
        """u, v, q_psi, phi, phih, phix, phiy as the fourth stage of the last step saw them (ref Kernel.py:355-368: the third
        update, phi = ifft(phih), _invert, _calc_rel_vorticity)"""
        c, n = self._ctx, self.nx
        qh4 = c.field(_lib.F_QH_STAGE4)
        if self._dual:                   # physical space sees the Hermitian part: the mean of the two copies
            qh4[:, 1:n // 2] = 0.5 * (qh4[:, 1:n // 2] + c.field(_lib.F_QH_MINUS_STAGE4)[:, 1:n // 2])
        qh4 = hermitian_full(qh4)
        pv = self.ifft(-(self.wv2i * qh4)).real
        q4 = self.ifft(qh4).real
        phih4 = c.field(_lib.F_PHIH_STAGE4)
        phi4 = self.ifft(phih4)
        if self.model_id == _lib.COUPLED:
            phix, phiy = self.ifft(self.ik * phih4), self.ifft(self.il * phih4)
            jh = self.fft((1j * (np.conj(phix) * phiy - np.conj(phiy) * phix)).real)
            jh[0, 0] = 0
            qwh = 0.5 * (0.5 * (-self.wv2 * self.fft(np.abs(phi4) ** 2)) + jh) / self.f
            qwh *= self.filtr
            pv = pv + self.ifft(self.wv2i * qwh).real
            q4 = q4 - self.ifft(qwh).real                    # q_psi (CoupledModel.py:145-152)
        else:                                                # as last refreshed BEFORE the stage (quirk Q1: UnCoupledModel._invert
            phix, phiy = self.__dict__.get("_ec_grad") or (self.phix, self.phiy)    # leaves them; see _keep_stage4_grad_phi)
        ph4 = self.fft(pv)
        return self.ifft(-self.il * ph4).real, self.ifft(self.ik * ph4).real, q4, phi4, phih4, phix, phiy

    # The same holds for what _calc_energy_conversion leaves behind: _step_etdrk4 calls it at the start of every stage (Kernel.py:
    # 319, :338, :355, :370), so after a step gamma1, gamma2, xi1, xi2, pi and lapphi are the FOURTH stage's until a diagnostics
    # tick recomputes them from the new state (golden g19).  Rebuilt on demand with the reference's expressions (Kernel.py:682-701).
    _EC_NAMES = ("gamma1", "gamma2", "xi1", "xi2", "pi")
    _ec_stage4 = False

    def _keep_stage4_grad_phi(self):
        """UnCoupledModel, before anything refreshes phix, phiy (a status line's or set_phi's _calc_pe_niw, Kernel.py:610): the fourth
        stage's conversions were formed with the gradients as they are NOW.  If nobody has read them yet, keep those gradients (two
        planes to the host: only on a status line or a set_phi that follows a step without a diagnostics tick)."""
        d = self.__dict__
        if (self.model_id == _lib.UNCOUPLED and d.get("_ec_stage4") and "_ec_grad" not in d
                and not all(k in d for k in Kernel._EC_NAMES)):
            d["_ec_grad"] = (self.phix, self.phiy)

    def _energy_conversion_of_stage4(self):
        u, v, q_psi, phi, phih, phix, phiy = self._fields_of_stage4()
        adv = u * phix + v * phiy
        lapphi = self.ifft(-self.wv2 * phih)
        diss = -self.nu4w * self.ifft(self.wv4 * phih) + self.nuw * lapphi - self.muw * phi
        div_fw = 0.5 * self.hslash * (np.conj(phi) * lapphi).imag
        d = self.__dict__
        d["gamma1"] = (0.5 * q_psi * div_fw).mean() / self.f
        d["gamma2"] = 0.5 * self.hslash * ((np.conj(lapphi) * adv).real).mean() / self.f
        d["xi1"] = (-(diss * np.conj(adv)).imag).mean() / self.f
        d["xi2"] = (0.5 * (diss * np.conj(phi)).real * q_psi).mean() / self.f
        d["pi"] = (0.5 * phi.mean() * (q_psi * np.conj(phi)).mean()).imag
        self._cache["lapphi"] = lapphi

    # ------------------------------------------------------------------ public API of the reference
    def fft(self, x):
        """ref: niwqg/Kernel.py:562-566 (numpy.fft.fft2 semantics)"""
        return self._ctx.fft2(x)

    def ifft(self, x):
        return self._ctx.ifft2(x)

    def set_q(self, q):
        """ref: niwqg/Kernel.py:520-535 -- inverts with the current phi (quirk Q2)"""
        self._ctx.set_q(q)
        self._dirty()
        self._uv_stage4 = False                     # u, v of the new psi (Kernel.py:533-534)
        self._user["q"] = q
        if self._ctx.budgets_enabled:
            self._ctx.take_budget_increments()      # drop increments that belong to the old state
        self.Ke = self.ke = self._calc_ke_qg()

    def set_phi(self, phi):
        """ref: niwqg/Kernel.py:538-551 -- does NOT re-invert (quirk Q2)"""
        self._keep_stage4_grad_phi()
        self._ctx.set_phi(phi)
        keep = {k: v for k, v in self._cache.items() if k not in ("phi", "phih", "phix", "phiy", "_dsums")}
        keepu = {k: v for k, v in self._user.items() if k != "phi"}
        self._cache, self._user = keep, keepu
        self._user["phi"] = phi
        self.Pw = self._calc_pe_niw()
        self.Kw = self._calc_ke_niw()

    def _invert(self):
        self._ctx.invert()
        self._dirty()

    def _calc_rel_vorticity(self):
        """q_psi is formed on the device inside the product kernel; nothing to do on the host."""

    def _calc_strain(self):
        """Geostrophic rate of strain, 4 psi_xy^2 + (psi_xx - psi_yy)^2, left in ``qg_strain`` (ref: niwqg/Kernel.py:503-509);
        three inverse transforms through the device FFT seam."""
        ph = self.ph
        pxx, pyy = self.ifft(-self.k * self.k * ph).real, self.ifft(-self.l * self.l * ph).real
        pxy = self.ifft(-self.k * self.l * ph).real
        self.qg_strain = 4 * (pxy ** 2) + (pxx - pyy) ** 2

    def _calc_OW(self):
        """Okubo-Weiss parameter exactly as the reference forms it, qg_strain**2 - q_psi**2 with qg_strain already a
        squared rate (ref: niwqg/Kernel.py:511-518)."""
        self._calc_rel_vorticity()
        self._calc_strain()
        return self.qg_strain ** 2 - self.q_psi ** 2

    def jacobian_psi_q(self):
        """ik F[u q] + il F[v q], [0,0] = 0.  ref: niwqg/Kernel.py:471-486"""
        if self._uv_stage4:                         # leaves the u, v of the CURRENT psi behind
            self._uv_stage4 = False
            self._cache.pop("u", None)
            self._cache.pop("v", None)
        return self._ctx.jacobian_psi_q()

    def jacobian_psi_phi(self):
        """F[u phix + v phiy], [0,0] = 0.  ref: niwqg/Kernel.py:457-469"""
        if self._uv_stage4:                         # the u, v a step left behind (see _uv_of_stage4), phix, phiy as last refreshed
            jh = self.fft(self.u * self.phix + self.v * self.phiy)
            if self.model_id != _lib.YBJ:
                jh[0, 0] = 0
            return jh
        return self._ctx.jacobian_psi_phi()

    def spec_var(self, ph):
        """ref: niwqg/Kernel.py:654-658"""
        var_dens = np.abs(ph) ** 2 / self.M ** 2
        var_dens[0, 0] = 0.
        return var_dens.sum()

    # ------------------------------------------------------------------ stepping
    def _step_etdrk4(self):
        """One ETDRK4 step on the device.  ref: niwqg/Kernel.py:307-397"""
        # a status line follows this step (Kernel.py:587-590) and no diagnostics tick refreshes u, v before it (Diagnostics.py:43):
        # its CFL comes from the FOURTH stage's u, v (Kernel.py:594 with :364-368), so that stage's maxima are recorded on the way
        self._cfl_recorded = (self.model_id != _lib.YBJ and ((self.tc + 1) % self.twrite) == 0 and (self.tc % self.tdiags) != 0)
        if self._cfl_recorded:
            self._ctx.request_stage4_max()
        self._ctx.step(1)
        self._after_steps()

    def _after_steps(self):
        self._dirty()
        self._uv_stage4 = self.model_id != _lib.YBJ          # (YBJModel: psi, u, v are steady)
        self._ec_stage4 = self._uv_stage4                    # (and its step never calls _calc_energy_conversion)
        self.__dict__.pop("_ec_grad", None)
        self._tick_stale = self._tick_taken
        if self._ec_stage4:
            for k in Kernel._EC_NAMES:
                self.__dict__.pop(k, None)
        if self._ctx.budgets_enabled:
            dKe, dPw, dKw = self._ctx.take_budget_increments()
            self.Ke += dKe
            self.Pw += dPw
            self.Kw += dKw

    def _step_forward(self):
        """ref: niwqg/Kernel.py:205-217"""
        self._step_etdrk4()
        increment_diagnostics(self)
        self._print_status()
        save_snapshots(self, fields=['t', 'q', 'phi'])

    def _quiet_steps(self, n_left):
        """How many of the next ``n_left`` steps need no host attention before the first one that
        does: a diagnostics tick fires after a step when tc_before % tdiags == 0 (Diagnostics.py:43),
        a status line when (tc_before + 1) % twrite == 0 (Kernel.py:587-590)."""
        for j in range(n_left):
            tcb = self.tc + j
            if (tcb % self.tdiags) == 0 or ((tcb + 1) % self.twrite) == 0:
                return j
            if self.save_to_disk and ((tcb + 1) % self.tsnaps) == 0:      # a snapshot after this step (Saving.py:70)
                return j
        return n_left - 1

    def _steps_left(self, cap=1 << 30):
        """Replays the reference's float clock ``while t < tmax: t += dt`` (Kernel.py:198,:588)."""
        t, n = self.t, 0
        while t < self.tmax and n < cap:
            t += self.dt
            n += 1
        return n

    def run(self):
        """ref: niwqg/Kernel.py:183-203.  Steps between host-visible events are batched into one
        nq_step call; the sequence of diagnostics ticks and status lines is the reference's."""
        self._defer_snapshots = True              # snapshots are written while the next batch of steps runs
        try:
            if self.save_to_disk:                     # the initial condition (Kernel.py:194-195)
                save_snapshots(self, fields=['t', 'q', 'phi'])
            while self.t < self.tmax:
                quiet = self._quiet_steps(self._steps_left(4096))
                if quiet > 0:
                    self._ctx.step(quiet)             # asynchronous: a pending snapshot is written while these steps run
                    flush_snapshots(self)
                    for _ in range(quiet):
                        self.tc += 1
                        self.t += self.dt
                    self._after_steps()
                self._step_forward()
            flush_snapshots(self)
            if self.save_to_disk:                     # Kernel.py:202-203
                save_diagnostics(self)
        finally:
            self._defer_snapshots = False
            flush_pending_quietly(self)      # (a failure in here must not mask the exception that is already on its way)

    def run_with_snapshots(self, tsnapstart=0., tsnapint=432000.):
        """ref: niwqg/Kernel.py:161-181"""
        tsnapints = np.ceil(tsnapint / self.dt)
        while self.t < self.tmax:
            self._step_forward()
            if self.t >= tsnapstart and (self.tc % tsnapints) == 0:
                yield self.t
        return

    def _print_status(self):
        """ref: niwqg/Kernel.py:568-598"""
        self.tc += 1
        self.t += self.dt
        if (self.tc % self.twrite) == 0:
            self.ke = self._calc_ke_qg()
            self.kew = self._calc_ke_niw()
            self.pew = self._calc_pe_niw()
            self.cfl = self._status_cfl()
            self.logger.info('Step: %4i, Time: %2.1e, P: %2.1e, Ke: %4.3e, Kw: %4.3e, Pw: %4.3e, CFL: %3.2f',
                             self.tc, self.t, self.t / self.tmax, self.ke, self.kew, self.pew, self.cfl)
            assert self.cfl < self.cflmax, self.logger.error('CFL condition violated')

    # ------------------------------------------------------------------ scalar integrals
    def _calc_ke_qg(self):
        """ref: niwqg/Kernel.py:600-602 (device reduction over the half spectrum)"""
        return self._ctx.scalar(_lib.S_KE_QG)

    def _calc_ke_niw(self):
        """ref: niwqg/Kernel.py:604-606 (Parseval form, device reduction)"""
        return self._ctx.scalar(_lib.S_KE_NIW)

    def _calc_pe_niw(self):
        """ref: niwqg/Kernel.py:608-611 -- including its side effect on phix/phiy (quirk Q1)"""
        self._keep_stage4_grad_phi()
        self._ctx.refresh_grad_phi()
        self._cache.pop("phix", None)
        self._cache.pop("phiy", None)
        pe = self._ctx.scalar(_lib.S_PE_NIW)
        self._grad2_mean = 4. * self.kappa2 * pe         # mean(|phix|^2 + |phiy|^2) of the refreshed gradients
        return pe

    _cfl_recorded = False

    def _status_cfl(self):
        """CFL of the status line (ref: niwqg/Kernel.py:594, :660-662).  After a step without a tick the reference's u, v are the
        fourth stage's: _step_etdrk4 had that stage's max |u|, max |v| recorded on the device (nq_request_stage4_max), max |phi| is
        the new state's."""
        if self._uv_stage4 and self._cfl_recorded:
            return self._ctx.status_cfl_max() * self.dt / self.dx
        return self._calc_cfl()

    def _calc_cfl(self):
        """ref: niwqg/Kernel.py:660-662"""
        if self._uv_stage4:              # u, v as a step left them (the fourth stage's): rebuilt on the host side, on demand
            return np.abs(np.hstack([self.u, self.v, np.abs(self.phi)])).max() * self.dt / self.dx
        return self._ctx.scalar(_lib.S_CFL) * self.dt / self.dx      # max reduction on the device

    # Everything below is evaluated from the 32 raw sums of ONE device pass (nq_diagnostics): no plane is
    # downloaded at a diagnostics tick.  Index map: include/niwqg_amd.h.
    def _dsums(self):
        if "_dsums" not in self._cache:
            self._cache["_dsums"] = self._ctx.diagnostic_sums()
        return self._cache["_dsums"]

    def _grad2(self, s):
        """mean(|phix|^2 + |phiy|^2) of phix, phiy AS LAST REFRESHED (UnCoupled keeps stale ones: quirk Q1)"""
        if self.model_id in (_lib.UNCOUPLED, _lib.YBJ) and hasattr(self, "_grad2_mean"):
            return self._grad2_mean
        return s[1] / self._M2

    @property
    def _M2(self):
        return (float(self.nx) * self.ny) ** 2

    def _calc_ens(self):
        """ref: niwqg/Kernel.py:625-627 by Parseval"""
        return 0.5 * self._dsums()[6] / self._M2

    def _calc_conc(self):
        """ref: niwqg/Kernel.py:613-619"""
        s, M = self._dsums(), float(self.nx) * self.ny
        with np.errstate(invalid="ignore", divide="ignore"):
            return (s[21] / M) / np.sqrt(s[20] / M) / np.sqrt(s[19] / M)

    def _calc_skewness(self):
        """ref: niwqg/Kernel.py:621-623"""
        s, M = self._dsums(), float(self.nx) * self.ny
        with np.errstate(invalid="ignore", divide="ignore"):
            return (s[18] / M) / ((s[17] / M) ** 1.5)

    def _calc_ep_phi(self):
        """ref: niwqg/Kernel.py:629-633"""
        s = self._dsums()
        return (-self.nu4w * s[2] - self.muw * s[0]) / self._M2 - self.nuw * self._grad2(s)

    def _calc_ep_psi(self):
        """ref: niwqg/Kernel.py:635-640"""
        s = self._dsums()
        return (self.nu4 * s[12] + self.nu * s[13] + self.mu * s[14]) / self._M2

    def _calc_chi_q(self):
        """ref: niwqg/Kernel.py:642-644"""
        return -self.nu4 * self._dsums()[7] / self._M2

    def _calc_chi_phi(self):
        """ref: niwqg/Kernel.py:646-652"""
        s = self._dsums()
        return ((-0.5 * self.nu4w * s[3] - 0.5 * self.nuw * s[2]) / self._M2
                - 0.5 * self.muw * self._grad2(s)) / self.kappa2

    def _calc_energy_conversion(self):
        """Diagnostic-tick version of ref niwqg/Kernel.py:664-701: gamma1, gamma2, xi1, xi2 as projections of
        F[u phix + v phiy] and F[phi q_psi] on lap_h and diss_h, pi from two domain means."""
        s, M = self._dsums(), float(self.nx) * self.ny
        M2f = self._M2 * self.f
        self._ec_stage4 = False
        self.__dict__.pop("_ec_grad", None)
        self.gamma2 = 0.5 * self.hslash * s[24] / M2f
        self.xi1 = s[27] / M2f
        self.gamma1 = 0.25 * self.hslash * s[28] / M2f
        self.xi2 = 0.5 * s[31] / M2f
        self.pi = (0.5 * (complex(s[4], s[5]) / M) * (complex(s[22], -s[23]) / M)).imag

    def _calc_icke_niw(self):
        s = self._dsums()
        self.ke_niw = 0.5 * s[0] / self._M2
        self.cke_niw = 0.5 * (s[4] ** 2 + s[5] ** 2) / self._M2
        self.ike_niw = self.ke_niw - self.cke_niw

    # ------------------------------------------------------------------ diagnostics registry
    def _initialize_diagnostics(self):
        self.diagnostics = dict()
        self._initialize_kernel_diagnostics()
        self._initialize_class_diagnostics()

    def _initialize_kernel_diagnostics(self):
        """Names, order, units as ref niwqg/Kernel.py:718-868."""
        table = [
            ('time', 'Time', 'seconds', lambda s: s.t),
            ('Ke', 'Quasigeostrophic Kinetic Energy, from energy equation', r'm^2 s^{-2}', lambda s: s.Ke),
            ('Pw', 'NIW Potential Energy, from energy equation', r'm^2 s^{-2}', lambda s: s.Pw),
            ('Kw', 'NIW Kinetic Energy, from energy equation', r'm^2 s^{-2}', lambda s: s.Kw),
            ('ke_qg', 'Quasigeostrophic Kinetic Energy', r'm^2 s^{-2}', lambda s: s._calc_ke_qg()),
            ('ens', 'Quasigeostrophic Potential Enstrophy', r's^{-2}', lambda s: s._calc_ens()),
            ('ke_niw', 'Near-inertial Kinetic Energy', r'm^2 s^{-2}', lambda s: s.ke_niw),
            ('cke_niw', 'Kinetic Energy of Laterally Coherent Near-Inertial Waves', r'm^2 s^{-2}',
             lambda s: s.cke_niw),
            ('ike_niw', 'Kinetic Energy of Laterally Incoherent Near-Inertial Waves', r'm^2 s^{-2}',
             lambda s: s.ike_niw),
            ('pe_niw', 'Near-inertial Potential Energy', r'm^2 s^{-2}', lambda s: s._calc_pe_niw()),
            ('conc_niw', 'Correlation between relative vorticity and near-inertial KE', r'unitless',
             lambda s: s._calc_conc()),
            ('skew', 'Skewness', r'unitless', lambda s: s._calc_skewness()),
            ('gamma_r', 'The energy conversion due to refraction', r'$m^2 s^{-3}$', lambda s: s.gamma1),
            ('gamma_a', 'The energy conversion due to advection', r'$m^2 s^{-3}$', lambda s: s.gamma2),
            ('xi_r', 'The QG energy generation due to wave dissipation, vorticity', r'$m^2 s^{-3}$',
             lambda s: s.xi1),
            ('xi_a', 'The QG energy generation due to wave dissipation, advection', r'$m^2 s^{-3}$',
             lambda s: s.xi2),
            ('pi', 'The NIW kinetic energy conversion from coherent to incoherent', r'$m^2 s^{-3}$',
             lambda s: s.pi),
            ('ep_phi', 'The hyperviscous dissipation of NIW kinetic energy', r'$m^2 s^{-3}$',
             lambda s: s._calc_ep_phi()),
            ('ep_psi', 'The hyperviscous dissipation of QG kinetic energy', r'$m^2 s^{-3}$',
             lambda s: s._calc_ep_psi()),
            ('chi_q', 'The hyperviscous dissipation of QG kinetic energy', r'$s^{-3}$', lambda s: s._calc_chi_q()),
            ('chi_phi', 'The hyperviscous dissipation of NIW potential energy', r'$s^{-3}$',
             lambda s: s._calc_chi_phi()),
        ]
        for name, desc, units, fn in table:
            add_diagnostic(self, name, description=desc, units=units, types='scalar', function=fn)

    def _initialize_class_diagnostics(self):
        pass

    def _calc_derived_fields(self):
        self._calc_kernel_derived_fields()
        self._calc_class_derived_fields()

    def _calc_kernel_derived_fields(self):
        if self._uv_stage4:              # a tick recomputes u, v from the current psi (Kernel.py:681)
            self._uv_stage4 = False
            self._cache.pop("u", None)
            self._cache.pop("v", None)
        self._calc_energy_conversion()
        self._calc_icke_niw()

    def _calc_class_derived_fields(self):
        pass
