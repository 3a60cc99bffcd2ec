"""CPU oracle for the niwqg ETDRK4 hot path  --  TEST INFRASTRUCTURE ONLY.

This file is a plain-numpy restatement of the algorithm of the reference
(cesar-rocha/niwqg, read at /root/reference) for the path named in
BASELINE.json: ``Kernel._step_forward`` / ``_step_etdrk4`` and its QGModel twin.
It exists so that the HIP implementation in ``niwqg_amd/`` can be checked on a
machine where the reference itself is not available (the GPU box).

Rules (see the task statement, section 3):
  * only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
    ``cpu_baseline`` leg may import this module -- never the product path;
  * it is pinned against the reference by ``tests/golden/*.npz`` (generated in
    the build container by ``tests/golden/make_golden.py``, which imports the
    real reference) -- ``tests/test_oracle_golden.py`` is that check.

``workers`` (default 1 = the reference's arithmetic and cost exactly: numpy.fft, one thread) lets the parity tests at
2048^2 and above finish in minutes: the coefficient tables are then built chunk-by-chunk on a thread pool (same
arithmetic per element) and the FFT seam goes through ``scipy.fft`` with that many workers, which moves q and phi by
~3e-16 relative ("noise floor").  ``table_workers`` threads only the (one-off) coefficient tables.  The CPU baseline of bench.py times
steps with workers=1 (numpy.fft, one thread) and uses table_workers only to shorten the untimed constructor.

It deliberately keeps the reference's transform count (104 / 72 / 33 full 2-D
transforms per step for Coupled / UnCoupled / QG) so that it can stand in for the
reference's CPU cost; ``fft_calls`` counts them.

All ``ref:`` citations are file:line in /root/reference.
"""
from __future__ import annotations

import numpy as np

TWO_PI = 2.0 * np.pi


# --------------------------------------------------------------------------
# grid, filter, ETDRK4 coefficient tables
# --------------------------------------------------------------------------
class SpectralGrid:
    """Square doubly-periodic grid and its wavenumbers.

    ref: niwqg/Kernel.py:227-265 (c2c layout, ``half=False``) and
         niwqg/QGModel.py:232-269 (r2c layout, ``half=True``).
    The reference ignores ``ny`` (Kernel.py:100-101) so only ``nx`` exists here.
    """

    def __init__(self, nx: int, L: float, half: bool = False):
        self.nx = self.ny = int(nx)
        self.L = self.W = L
        self.half = half
        cell = (np.arange(nx) + 0.5) / nx * L            # cell-centred coordinates
        self.x, self.y = np.meshgrid(cell, cell)
        self.dk = self.dl = TWO_PI / L
        signed = np.concatenate([np.arange(0.0, nx / 2), np.arange(-nx / 2, 0.0)])
        self.ll = self.dl * signed
        if half:
            self.nk = nx // 2 + 1
            self.kk = self.dk * np.arange(0.0, self.nk)   # +N/2 at the end (QGModel.py:249)
        else:
            self.nk = nx
            self.kk = self.ll.copy()                      # -N/2 at index N/2 (Kernel.py:242-244)
        self.nl = nx
        self.k, self.l = np.meshgrid(self.kk, self.ll)
        self.ik, self.il = 1j * self.k, 1j * self.l
        self.dx = self.dy = L / nx
        self.M = nx * nx
        self.wv2 = self.k ** 2 + self.l ** 2
        self.wv = np.sqrt(self.wv2)
        self.wv4 = self.wv2 ** 2
        self.wv2i = np.zeros_like(self.wv2)
        nz = self.wv2 != 0.0
        self.wv2i[nz] = self.wv2[nz] ** -1


def spectral_filter(g: SpectralGrid, use_filter: bool, dealias: bool) -> np.ndarray:
    """``filtr``: exponential filter, or 2/3 mask, or ones.

    ref: niwqg/Kernel.py:267-284.  (QGModel.py:287-300 is the same except that its
    2/3 branch is broken in the reference -- float slice indices -- so only the
    Kernel form is restated; we use integer thirds for both.)
    """
    if use_filter:
        cphi = 0.65 * np.pi
        wvx = np.sqrt((g.k * g.dx) ** 2.0 + (g.l * g.dy) ** 2.0)
        f = np.exp(-23.6 * (wvx - cphi) ** 4.0)
        f[wvx <= cphi] = 1.0
        return f
    f = np.ones_like(g.wv2)
    if dealias:
        n = g.nx
        f[n // 3:2 * n // 3, :] = 0.0
        f[:, n // 3:2 * n // 3] = 0.0
    return f


_CONTOUR_M = 32


def etdrk4_tables(c: np.ndarray, dt: float, rows_per_chunk: int = 64, workers: int = 1) -> dict:
    """ETDRK4 (Cox-Matthews / Kassam-Trefethen) coefficient planes for exponent ``c``.

    ref: niwqg/Kernel.py:419-433 (and :443-454, QGModel.py:429-443).  The reference
    builds (ny,nx,32) temporaries; here the same 32-point unit-circle mean is taken
    row-chunk by row-chunk, which is arithmetically identical per element.
    Returns E=exp(c dt), Eh=exp(c dt/2), Q, f0, fab, fc.
    """
    ch = c * dt
    r = np.exp(2j * np.pi * (np.arange(1.0, _CONTOUR_M + 1) / _CONTOUR_M))
    out = {k: np.empty_like(ch) for k in ("Q", "f0", "fab", "fc")}
    def chunk(r0):
        sl = slice(r0, r0 + rows_per_chunk)
        LR = ch[sl, :, None] + r[None, None, :]
        LR2 = LR * LR
        LR3 = LR2 * LR
        # np.exp(LR) is written out in every line, as the reference writes it: with the exponential held in a variable numpy
        # picks a different multiply loop (operand aliasing), the products differ in the last bit, and next to the contour
        # (|LR| ~ 1e-5, where these means are eps / |LR|^3 rounding noise) that is a 1e-7 relative difference in f0
        out["Q"][sl] = dt * (((np.exp(LR / 2.) - 1.) / LR).mean(axis=-1))
        out["f0"][sl] = dt * (((-4. - LR + (np.exp(LR) * (4. - 3. * LR + LR2))) / LR3).mean(axis=-1))
        out["fab"][sl] = dt * (((2. + LR + np.exp(LR) * (-2. + LR)) / LR3).mean(axis=-1))
        out["fc"][sl] = dt * (((-4. - 3. * LR - LR2 + np.exp(LR) * (4. - LR)) / LR3).mean(axis=-1))

    starts = range(0, ch.shape[0], rows_per_chunk)
    if workers > 1:                 # numpy ufuncs release the GIL: row chunks in parallel, same arithmetic per element
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(workers) as ex:
            list(ex.map(chunk, starts))
    else:
        for r0 in starts:
            chunk(r0)
    out["E"] = np.exp(ch)
    out["Eh"] = np.exp(ch / 2.0)
    return out


def lamb_dipole(g: SpectralGrid, U: float = 0.01, R: float = 1.0) -> np.ndarray:
    """Lamb dipole vorticity, vectorised.  ref: niwqg/InitialConditions.py:77-114
    (the reference uses an O(N^2) Python loop for sin(theta); same values)."""
    from scipy import special
    n = g.nx
    x0, y0 = g.x[n // 2, n // 2], g.y[n // 2, n // 2]
    r = np.sqrt((g.x - x0) ** 2 + (g.y - y0) ** 2)
    s = np.zeros_like(r)
    nzr = r != 0.0
    s[nzr] = (g.y[nzr] - y0) / r[nzr]
    lam = 3.8317 / R
    C = -(2.0 * U * lam) / special.j0(lam * R)
    q = np.zeros_like(r)
    inside = r <= R
    q[inside] = C * special.j1(lam * r[inside]) * s[inside]
    return q


def wave_packet(g: SpectralGrid, k=10, l=0, R=1.0, x0=0.0, y0=0.0) -> np.ndarray:
    """Gaussian wave packet.  ref: niwqg/InitialConditions.py:117-145."""
    r = np.sqrt((g.x - x0) ** 2 + (g.y - y0) ** 2)
    return np.exp(1j * (k * (g.x - x0) + l * (g.y - y0))) * np.exp(-((r / R) ** 2))


# --------------------------------------------------------------------------
# NIW-QG kernel family (CoupledModel / UnCoupledModel)
# --------------------------------------------------------------------------
class NIWQGOracle:
    """Restatement of ``niwqg.Kernel.Kernel`` + ``CoupledModel`` / ``UnCoupledModel``.

    ``kind`` is "coupled", "uncoupled" or "ybj" (``niwqg.YBJModel``: steady psi, only phi is stepped).
    Attribute names follow the reference so that parity tests read like the reference's own tests.
    """

    def __init__(self, kind="coupled", nx=128, ny=None, L=5e5, dt=10000.0, twrite=1000.0,
                 tmax=250000.0, use_filter=True, cflmax=0.8, U=0.0, f=1e-4, N=0.01,
                 m=0.025, g=9.81, nu4=0, nu4w=0, nu=20, nuw=50.0, mu=0, muw=0,
                 dealias=False, tdiags=10, coeff_chunk=64, workers=1, table_workers=None):
        assert kind in ("coupled", "uncoupled", "ybj")
        self.kind = kind
        self.workers = int(workers)
        table_workers = self.workers if table_workers is None else int(table_workers)
        # ref: niwqg/Kernel.py:100-137 (parameter bookkeeping; ny ignored)
        self.nx = self.ny = nx
        self.L = self.W = L
        self.dt, self.twrite, self.tmax = dt, twrite, tmax
        self.U, self.g = U, g
        self.nu4, self.nu4w, self.nu, self.nuw, self.mu, self.muw = nu4, nu4w, nu, nuw, mu, muw
        self.f, self.N, self.m = f, N, m
        self.kappa = m * f / N
        self.kappa2 = self.kappa ** 2
        self.hslash = f / self.kappa2
        self.cflmax = cflmax
        self.tdiags = tdiags
        self.use_filter, self.dealias = use_filter, dealias

        G = self.grid = SpectralGrid(nx, L, half=False)
        for name in ("x", "y", "kk", "ll", "k", "l", "ik", "il", "dx", "dy", "M",
                     "wv", "wv2", "wv4", "wv2i", "dk", "dl"):
            setattr(self, name, getattr(G, name))

        # ref: niwqg/CoupledModel.py:33-55 / UnCoupledModel.py:29-51
        shp = (nx, nx)
        self.q = np.zeros(shp)
        self.qh = np.zeros(shp, complex)
        self.p = np.zeros(shp)
        self.ph = np.zeros(shp, complex)
        self.phi = np.zeros(shp, complex)
        self.phih = np.zeros(shp, complex)

        self.filtr = spectral_filter(G, use_filter, dealias)

        # ref: niwqg/Kernel.py:417-418, :440-442  (linear operators of the two equations)
        cq = np.zeros(shp, complex) - 1j * self.k * U
        cq += -nu4 * self.wv4 - nu * self.wv2 - mu
        cw = np.zeros(shp, complex) - 1j * self.k * U
        cw += -nu4w * self.wv4 - 0.5j * f * (self.wv2 / self.kappa2) - nuw * self.wv2 - muw
        self.coef_q = etdrk4_tables(cq, dt, coeff_chunk, table_workers)
        self.coef_w = etdrk4_tables(cw, dt, coeff_chunk, table_workers)

        self.t = 0          # ref: niwqg/Kernel.py:219-225
        self.tc = 0
        self.fft_calls = [0, 0]   # [forward, inverse]
        self.diagnostics = {}

    # ---- FFT seam (ref: niwqg/Kernel.py:553-566) --------------------------
    def fft(self, a):
        self.fft_calls[0] += 1
        if self.workers > 1:
            import scipy.fft
            return scipy.fft.fft2(a, workers=self.workers)
        return np.fft.fft2(a)

    def ifft(self, a):
        self.fft_calls[1] += 1
        if self.workers > 1:
            import scipy.fft
            return scipy.fft.ifft2(a, workers=self.workers)
        return np.fft.ifft2(a)

    # ---- model closures ----------------------------------------------------
    def jacobian_phic_phi(self):
        """ref: niwqg/CoupledModel.py:59-73."""
        self.phix, self.phiy = self.ifft(self.ik * self.phih), self.ifft(self.il * self.phih)
        jh = self.fft((1j * (np.conj(self.phix) * self.phiy - np.conj(self.phiy) * self.phix)).real)
        jh[0, 0] = 0
        return jh

    def _invert(self):
        if self.kind == "coupled":
            # ref: niwqg/CoupledModel.py:75-97
            self.phi2 = np.abs(self.phi) ** 2
            self.gphi2h = -self.wv2 * self.fft(self.phi2)
            self.qwh = 0.5 * (0.5 * self.gphi2h + self.jacobian_phic_phi()) / self.f
            self.qwh *= self.filtr
            self.pw = self.ifft(self.wv2i * self.qwh).real
            self.pv = self.ifft(-(self.wv2i * self.qh)).real
            self.p = self.pv + self.pw
            self.ph = self.fft(self.p)
            self.q = self.ifft(self.qh).real
        elif self.kind == "ybj":
            # ref: niwqg/YBJModel.py:141-146 (purely spectral; p and q are left alone)
            self.ph = -self.wv2i * self.qh
        else:
            # ref: niwqg/UnCoupledModel.py:54-64  (phix/phiy NOT refreshed: quirk Q1)
            self.p = self.ifft(-(self.wv2i * self.qh)).real
            self.ph = self.fft(self.p)
            self.q = self.ifft(self.qh).real

    def _calc_rel_vorticity(self):
        if self.kind == "coupled":
            # ref: niwqg/CoupledModel.py:145-152
            self.qw = self.ifft(self.qwh).real
            self.q_psi = self.q - self.qw
        else:
            # ref: niwqg/Kernel.py:492-501
            self.q_psi = self.q

    def jacobian_psi_q(self):
        """ref: niwqg/Kernel.py:471-486."""
        self.u = self.ifft(-self.il * self.ph).real
        self.v = self.ifft(self.ik * self.ph).real
        q = self.ifft(self.qh).real
        jh = self.ik * self.fft(self.u * q) + self.il * self.fft(self.v * q)
        jh[0, 0] = 0
        return jh

    def jacobian_psi_phi(self):
        """ref: niwqg/Kernel.py:457-469."""
        jh = self.fft(self.u * self.phix + self.v * self.phiy)
        if self.kind != "ybj":          # niwqg/YBJModel.py:123-133 does NOT zero [0,0]
            jh[0, 0] = 0
        return jh

    # ---- initial state (ref: niwqg/Kernel.py:520-551) -------------------------
    def set_q(self, q):
        self.q = q
        self.qh = self.fft(self.q)
        self._invert()
        self._calc_rel_vorticity()
        self.u = self.ifft(-self.il * self.ph).real
        self.v = self.ifft(self.ik * self.ph).real
        self.Ke = self.ke = self._calc_ke_qg()

    def set_phi(self, phi):
        self.phi = phi
        self.phih = self.fft(self.phi)
        self.Pw = self._calc_pe_niw()
        self.Kw = self._calc_ke_niw()

    # ---- scalar integrals ---------------------------------------------------
    def spec_var(self, ah):
        """ref: niwqg/Kernel.py:654-658."""
        d = np.abs(ah) ** 2 / self.M ** 2
        d[0, 0] = 0.0
        return d.sum()

    def _calc_ke_qg(self):          # ref: niwqg/Kernel.py:600-602
        return 0.5 * self.spec_var(self.wv * self.ph)

    def _calc_ke_niw(self):         # ref: niwqg/Kernel.py:604-606
        return 0.5 * (np.abs(self.phi) ** 2).mean()

    def _calc_pe_niw(self):         # ref: niwqg/Kernel.py:608-611  (side effect: phix, phiy)
        self.phix, self.phiy = self.ifft(self.ik * self.phih), self.ifft(self.il * self.phih)
        return 0.25 * (np.abs(self.phix) ** 2 + np.abs(self.phiy) ** 2).mean() / self.kappa2

    def _calc_cfl(self):            # ref: niwqg/Kernel.py:660-662
        return np.abs(np.hstack([self.u, self.v, np.abs(self.phi)])).max() * self.dt / self.dx

    def _calc_ep_phi(self):         # ref: niwqg/Kernel.py:629-633
        return (-self.nu4w * (np.abs(self.lapphi) ** 2).mean()
                - self.nuw * (np.abs(self.phix) ** 2 + np.abs(self.phiy) ** 2).mean()
                - self.muw * (np.abs(self.phi) ** 2).mean())

    def _calc_ep_psi(self):         # ref: niwqg/Kernel.py:635-640
        lap2psi = self.ifft(self.wv4 * self.ph).real
        lapq = self.ifft(-self.wv2 * self.qh).real
        return (self.nu4 * (self.q * lap2psi).mean() - self.nu * (self.p * lapq).mean()
                + self.mu * (self.p * self.q).mean())

    def _calc_chi_q(self):          # ref: niwqg/Kernel.py:642-644
        return -self.nu4 * self.spec_var(self.wv2 * self.qh)

    def _calc_chi_phi(self):        # ref: niwqg/Kernel.py:646-652
        lphix = self.ifft(-self.ik * self.wv2 * self.phih)
        lphiy = self.ifft(-self.il * self.wv2 * self.phih)
        return (-0.5 * self.nu4w * (np.abs(lphix) ** 2 + np.abs(lphiy) ** 2).mean() / self.kappa2
                - 0.5 * self.nuw * (np.abs(self.lapphi) ** 2).mean() / self.kappa2
                - 0.5 * self.muw * (np.abs(self.phix) ** 2 + np.abs(self.phiy) ** 2).mean() / self.kappa2)

    def _calc_energy_conversion(self):
        """ref: niwqg/Kernel.py:664-701 (lapphi bypasses the seam: quirk Q5, :685)."""
        self.u = self.ifft(-self.il * self.ph).real
        self.v = self.ifft(self.ik * self.ph).real
        self._calc_rel_vorticity()
        J = self.u * self.phix + self.v * self.phiy
        self.fft_calls[1] += 1
        self.lapphi = np.fft.ifft2(-self.wv2 * self.phih) if self.workers <= 1 else self.ifft(-self.wv2 * self.phih)
        lap2phi = self.ifft(self.wv4 * self.phih)
        diss = -self.nu4w * lap2phi + self.nuw * self.lapphi - self.muw * self.phi
        J_diss = -(diss * np.conj(J)).imag
        L_diss = 0.5 * (diss * np.conj(self.phi)).real * self.q_psi
        divFw = 0.5 * self.hslash * (np.conj(self.phi) * self.lapphi).imag
        self.gamma1 = (0.5 * self.q_psi * divFw).mean() / self.f
        self.gamma2 = 0.5 * self.hslash * ((np.conj(self.lapphi) * J).real).mean() / self.f
        self.xi1 = J_diss.mean() / self.f
        self.xi2 = L_diss.mean() / self.f
        self.pi = (0.5 * self.phi.mean() * (self.q_psi * np.conj(self.phi)).mean()).imag

    # ---- one ETDRK4 step (ref: niwqg/Kernel.py:307-397) ------------------------
    def _budget_rates(self):
        self._calc_energy_conversion()
        k = -(self.gamma1 + self.gamma2) + (self.xi1 + self.xi2) + self._calc_ep_psi()
        p = self.gamma1 + self.gamma2 + self._calc_chi_phi()
        a = self._calc_ep_phi()
        return k, p, a

    def _nonlinear_q(self):
        return -self.jacobian_psi_q()

    def _nonlinear_w(self):
        return -self.jacobian_psi_phi() - 0.5j * self.fft(self.phi * self.q_psi)

    def _to_physical(self):
        self.phi = self.ifft(self.phih)
        self._invert()
        self._calc_rel_vorticity()

    def _step_etdrk4_ybj(self):
        """ref: niwqg/YBJModel.py:52-87.  Only phi is stepped; phix/phiy are refreshed from the current phih before
        every stage but ``self.phi`` (the refraction factor) only after the step, and u, v, q_psi stay those of
        set_q (or of the last diagnostics tick, which recomputes the same values); no budget accumulation."""
        cw, F = self.coef_w, self.filtr

        def grad():
            self.phix, self.phiy = self.ifft(self.ik * self.phih), self.ifft(self.il * self.phih)

        self.phih0 = self.phih.copy()
        grad()
        N0w = self._nonlinear_w()
        self.phih = (cw["Eh"] * self.phih0 + N0w * cw["Q"]) * F
        self.phih1 = self.phih.copy()
        grad()
        Naw = self._nonlinear_w()
        self.phih = (cw["Eh"] * self.phih0 + Naw * cw["Q"]) * F
        grad()
        Nbw = self._nonlinear_w()
        self.phih = (cw["Eh"] * self.phih1 + (2.0 * Nbw - N0w) * cw["Q"]) * F
        self._calc_rel_vorticity()
        grad()
        Ncw = self._nonlinear_w()
        self.phih = (cw["E"] * self.phih0 + N0w * cw["f0"] + 2.0 * (Naw + Nbw) * cw["fab"]
                     + Ncw * cw["fc"]) * F
        self.phi = self.ifft(self.phih)

    def _step_etdrk4(self):
        if self.kind == "ybj":
            return self._step_etdrk4_ybj()
        cq, cw, F = self.coef_q, self.coef_w, self.filtr
        rates = []
        # stage 1 (ref :319-339)
        rates.append(self._budget_rates())
        self.qh0 = self.qh.copy()
        N0 = self._nonlinear_q()
        self.qh = (cq["Eh"] * self.qh0 + N0 * cq["Q"]) * F
        self.qh1 = self.qh.copy()
        self.phih0 = self.phih.copy()
        N0w = self._nonlinear_w()
        self.phih = (cw["Eh"] * self.phih0 + N0w * cw["Q"]) * F
        self.phih1 = self.phih.copy()
        self._to_physical()
        # stage 2 (ref :341-356)
        rates.append(self._budget_rates())
        Na = self._nonlinear_q()
        self.qh = (cq["Eh"] * self.qh0 + Na * cq["Q"]) * F
        Naw = self._nonlinear_w()
        self.phih = (cw["Eh"] * self.phih0 + Naw * cw["Q"]) * F
        self._to_physical()
        # stage 3 (ref :358-373)
        rates.append(self._budget_rates())
        Nb = self._nonlinear_q()
        self.qh = (cq["Eh"] * self.qh1 + (2.0 * Nb - N0) * cq["Q"]) * F
        Nbw = self._nonlinear_w()
        self.phih = (cw["Eh"] * self.phih1 + (2.0 * Nbw - N0w) * cw["Q"]) * F
        self._to_physical()
        # stage 4 (ref :375-387)
        rates.append(self._budget_rates())
        Nc = self._nonlinear_q()
        self.qh = (cq["E"] * self.qh0 + N0 * cq["f0"] + 2.0 * (Na + Nb) * cq["fab"]
                   + Nc * cq["fc"]) * F
        Ncw = self._nonlinear_w()
        self.phih = (cw["E"] * self.phih0 + N0w * cw["f0"] + 2.0 * (Naw + Nbw) * cw["fab"]
                     + Ncw * cw["fc"]) * F
        # budget accumulators (ref :390-392)
        (k1, p1, a1), (k2, p2, a2), (k3, p3, a3), (k4, p4, a4) = rates
        self.Ke += self.dt * (k1 + 2 * (k2 + k3) + k4) / 6.0
        self.Pw += self.dt * (p1 + 2 * (p2 + p3) + p4) / 6.0
        self.Kw += self.dt * (a1 + 2 * (a2 + a3) + a4) / 6.0
        self._to_physical()       # ref :395-397

    # ---- diagnostics tick (ref: niwqg/Diagnostics.py:41-58, Kernel.py:718-878) --
    def _calc_ke_qg_decomp(self):   # ref: niwqg/CoupledModel.py:99-113
        self.phq = -self.wv2i * self.qh
        self.ke_qg_q = 0.5 * self.spec_var(self.wv * self.phq)
        self.phw = self.wv2i * self.qwh
        self.ke_qg_w = 0.5 * self.spec_var(self.wv * self.phw)
        uq, vq = self.ifft(-self.il * self.phq).real, self.ifft(self.ik * self.phq).real
        uw, vw = self.ifft(-self.il * self.phw).real, self.ifft(self.ik * self.phw).real
        self.ke_qg_qw = (uq * uw).mean() + (vq * vw).mean()

    def _calc_conc(self):           # ref: niwqg/Kernel.py:613-619
        ups = np.abs(self.phi) ** 2 - (np.abs(self.phi) ** 2).mean()
        with np.errstate(invalid="ignore", divide="ignore"):
            return (ups * self.q_psi).mean() / ups.std() / self.q_psi.std()

    def _calc_skewness(self):       # ref: niwqg/Kernel.py:621-623
        return (self.q_psi ** 3).mean() / (((self.q_psi ** 2).mean()) ** 1.5)

    def _diagnostic_values(self):
        """Evaluation order of the registered lambdas (ref: niwqg/Kernel.py:718-868,
        CoupledModel.py:115-136).  ``pe_niw`` refreshes phix/phiy (quirk Q1)."""
        self._calc_energy_conversion()                       # Kernel.py:875-878
        self.ke_niw = self._calc_ke_niw()
        self.cke_niw = 0.5 * (np.abs(self.phi.mean()) ** 2)
        self.ike_niw = self.ke_niw - self.cke_niw
        if self.kind == "coupled":
            self._calc_ke_qg_decomp()
        vals = [("time", self.t), ("Ke", self.Ke), ("Pw", self.Pw), ("Kw", self.Kw),
                ("ke_qg", self._calc_ke_qg()), ("ens", 0.5 * (self.q ** 2).mean()),
                ("ke_niw", self.ke_niw), ("cke_niw", self.cke_niw), ("ike_niw", self.ike_niw),
                ("pe_niw", self._calc_pe_niw()), ("conc_niw", self._calc_conc()),
                ("skew", self._calc_skewness()), ("gamma_r", self.gamma1),
                ("gamma_a", self.gamma2), ("xi_r", self.xi1), ("xi_a", self.xi2),
                ("pi", self.pi), ("ep_phi", self._calc_ep_phi()), ("ep_psi", self._calc_ep_psi()),
                ("chi_q", self._calc_chi_q()), ("chi_phi", self._calc_chi_phi())]
        if self.kind == "coupled":
            vals += [("ke_qg_q", self.ke_qg_q), ("ke_qg_w", self.ke_qg_w),
                     ("ke_qg_qw", self.ke_qg_qw)]
        return vals

    def _increment_diagnostics(self):
        if not (self.tc % self.tdiags):
            for name, val in self._diagnostic_values():
                self.diagnostics.setdefault(name, []).append(val)

    # ---- stepping (ref: niwqg/Kernel.py:183-217, :568-598) ----------------------
    def _print_status(self):
        self.tc += 1
        self.t += self.dt
        if (self.tc % self.twrite) == 0:
            self.ke = self._calc_ke_qg()
            self.kew = self._calc_ke_niw()
            self.pew = self._calc_pe_niw()
            self.cfl = self._calc_cfl()
            assert self.cfl < self.cflmax, "CFL condition violated"

    def _step_forward(self):
        self._step_etdrk4()
        self._increment_diagnostics()
        self._print_status()

    def run(self):
        while self.t < self.tmax:
            self._step_forward()

    def diag(self, name):
        return np.array(self.diagnostics[name])


# --------------------------------------------------------------------------
# barotropic QG model on the half spectrum (rfft2)
# --------------------------------------------------------------------------
class QGOracle:
    """Restatement of ``niwqg.QGModel.Model``, with or without its passive scalar.

    ref: niwqg/QGModel.py:65-140 (constructor), :328-407 (step), :469-505, :522-534 (set_c), :595-604, :724-737.
    """

    def __init__(self, nx=128, ny=None, L=5e5, dt=10000.0, twrite=1000, tmax=250000.0,
                 use_filter=True, U=0.0, nu4=5e9, nu=0, mu=0, beta=0, dealias=False,
                 tdiags=10, coeff_chunk=64, passive_scalar=False, nu4c=5e9, nuc=0, muc=0, workers=1,
                 table_workers=None):
        self.workers = int(workers)
        table_workers = self.workers if table_workers is None else int(table_workers)
        self.nx = self.ny = nx
        self.L = self.W = L
        self.dt, self.twrite, self.tmax, self.tdiags = dt, twrite, tmax, tdiags
        self.U, self.beta, self.nu4, self.nu, self.mu = U, beta, nu4, nu, mu
        self.use_filter, self.dealias = use_filter, dealias
        G = self.grid = SpectralGrid(nx, L, half=True)
        for name in ("x", "y", "kk", "ll", "k", "l", "ik", "il", "dx", "dy", "M",
                     "wv", "wv2", "wv4", "wv2i"):
            setattr(self, name, getattr(G, name))
        self.q = np.zeros((nx, nx))
        self.p = np.zeros((nx, nx))
        self.qh = np.zeros((nx, nx // 2 + 1), complex)
        self.ph = np.zeros((nx, nx // 2 + 1), complex)
        self.filtr = spectral_filter(G, use_filter, dealias)
        # ref: niwqg/QGModel.py:426-429
        c = np.zeros(self.qh.shape, complex)
        c += -nu4 * self.wv4 - nu * self.wv2 - mu - 1j * self.k * U
        c += beta * self.ik * self.wv2i
        self.coef_q = etdrk4_tables(c, dt, coeff_chunk, table_workers)
        self.passive_scalar, self.nu4c, self.nuc, self.muc = passive_scalar, nu4c, nuc, muc
        if passive_scalar:          # ref: niwqg/QGModel.py:446-466 (no mean-flow or beta term in this operator)
            cc = np.zeros(self.qh.shape, complex)
            cc += -nu4c * self.wv4 - nuc * self.wv2 - muc
            self.coef_c = etdrk4_tables(cc, dt, coeff_chunk, table_workers)
        self.C2, self.gradC2, self.cvar, self.Gamma_c = 0.0, 0.0, 0.0, 0.0
        self.t = 0
        self.tc = 0
        self.cflmax = 0.5           # ref: niwqg/QGModel.py:135
        self.fft_calls = [0, 0]
        self.diagnostics = {}

    def fft(self, a):               # ref: niwqg/QGModel.py:551
        self.fft_calls[0] += 1
        if self.workers > 1:
            import scipy.fft
            return scipy.fft.rfft2(a, workers=self.workers)
        return np.fft.rfft2(a)

    def ifft(self, a):              # ref: niwqg/QGModel.py:552
        self.fft_calls[1] += 1
        if self.workers > 1:
            import scipy.fft
            return scipy.fft.irfft2(a, workers=self.workers)
        return np.fft.irfft2(a)

    def spec_var(self, ah):         # ref: niwqg/QGModel.py:611-619
        d = 2.0 * np.abs(ah) ** 2 / self.M ** 2
        d[:, 0] *= 0.5
        d[:, -1] *= 0.5
        d[0, 0] = 0
        return d.sum()

    def jacobian_psi_q(self):       # ref: niwqg/QGModel.py:469-481 ([0,0] NOT zeroed)
        self.u, self.v = self.ifft(-self.il * self.ph), self.ifft(self.ik * self.ph)
        q = self.ifft(self.qh)
        return self.ik * self.fft(self.u * q) + self.il * self.fft(self.v * q)

    def jacobian_psi_c(self):       # ref: niwqg/QGModel.py:483-495 (u, v of the last jacobian_psi_q)
        self.c = self.ifft(self.ch)
        return self.ik * self.fft(self.u * self.c) + self.il * self.fft(self.v * self.c)

    def set_c(self, c):             # ref: niwqg/QGModel.py:522-534
        self.c = c
        self.ch = self.fft(self.c)
        self.cvar = self.spec_var(self.ch)

    def _calc_derived_fields(self):     # ref: niwqg/QGModel.py:724-737
        if self.passive_scalar:
            self.C2 = self.spec_var(self.ch)
            self.gradC2 = self.spec_var(self.wv * self.ch)
            self.lapc = self.ifft(-self.wv2 * self.ch)
            self.Gamma_c = 2 * (self.lapc * self.ifft(self.jacobian_psi_c())).mean()

    def _calc_ep_c(self):           # ref: niwqg/QGModel.py:595-598 (nu, not nuc, multiplies gradC2 there)
        return -2 * self.nu4c * (self.lapc ** 2).mean() - 2 * self.nu * self.gradC2 - 2 * self.muc * self.C2

    def _calc_chi_c(self):          # ref: niwqg/QGModel.py:600-604
        lap2c = self.ifft(self.wv4 * self.ch)
        return (2 * self.nu4c * (lap2c * self.lapc).mean() - 2 * self.nu * (self.lapc ** 2).mean()
                - 2 * self.muc * self.gradC2)

    def _invert(self):              # ref: niwqg/QGModel.py:497-505
        self.ph = -self.wv2i * self.qh
        self.p = self.ifft(self.ph)

    def set_q(self, q):             # ref: niwqg/QGModel.py:507-520
        self.q = q
        self.qh = self.fft(self.q)
        self._invert()
        self.Ke = self._calc_ke_qg()

    def _calc_ke_qg(self):          # ref: niwqg/QGModel.py:580-582
        return 0.5 * self.spec_var(self.wv * self.ph)

    def _calc_ep_psi(self):         # ref: niwqg/QGModel.py:588-593 (self.q is start-of-step q
        lap2psi = self.ifft(self.wv4 * self.ph)   # for k1..k3: it is only refreshed at :401)
        lapq = self.ifft(-self.wv2 * self.qh)
        return (self.nu4 * (self.q * lap2psi).mean() - self.nu * (self.p * lapq).mean()
                + self.mu * (self.p * self.q).mean())

    def _calc_chi_q(self):          # ref: niwqg/QGModel.py:606-609
        return -self.nu4 * self.spec_var(self.wv2 * self.qh)

    def _calc_cfl(self):            # ref: niwqg/QGModel.py:621-629
        self.u = self.ifft(-self.il * self.ph)
        self.v = self.ifft(self.ik * self.ph)
        return np.abs(np.hstack([self.u, self.v])).max() * self.dt / self.dx

    def _step_etdrk4(self):         # ref: niwqg/QGModel.py:328-407
        c, F = self.coef_q, self.filtr
        ps = self.passive_scalar
        cc = self.coef_c if ps else None
        self.qh0 = self.qh.copy()
        N0 = -self.jacobian_psi_q()
        self.qh = (c["Eh"] * self.qh0 + N0 * c["Q"]) * F
        self.qh1 = self.qh.copy()
        if ps:                      # the scalar is advected by the u, v that jacobian_psi_q just computed
            self.ch0 = self.ch.copy()
            M0 = -self.jacobian_psi_c()
            self.ch = (cc["Eh"] * self.ch0 + M0 * cc["Q"]) * F
            self.ch1 = self.ch.copy()
            self._calc_derived_fields()
            c1 = self._calc_ep_c()
        self._invert()
        k1 = self._calc_ep_psi()
        Na = -self.jacobian_psi_q()
        self.qh = (c["Eh"] * self.qh0 + Na * c["Q"]) * F
        if ps:
            Ma = -self.jacobian_psi_c()
            self.ch = (cc["Eh"] * self.ch0 + Ma * cc["Q"]) * F
            self._calc_derived_fields()
            c2 = self._calc_ep_c()
        self._invert()
        k2 = self._calc_ep_psi()
        Nb = -self.jacobian_psi_q()
        self.qh = (c["Eh"] * self.qh1 + (2.0 * Nb - N0) * c["Q"]) * F
        if ps:
            Mb = -self.jacobian_psi_c()
            self.ch = (cc["Eh"] * self.ch1 + (2.0 * Mb - M0) * cc["Q"]) * F
            self._calc_derived_fields()
            c3 = self._calc_ep_c()
        self._invert()
        k3 = self._calc_ep_psi()
        Nc = -self.jacobian_psi_q()
        self.qh = (c["E"] * self.qh0 + N0 * c["f0"] + 2.0 * (Na + Nb) * c["fab"]
                   + Nc * c["fc"]) * F
        if ps:
            Mc = -self.jacobian_psi_c()
            self.ch = (cc["E"] * self.ch0 + M0 * cc["f0"] + 2.0 * (Ma + Mb) * cc["fab"]
                       + Mc * cc["fc"]) * F
            self._calc_derived_fields()
            c4 = self._calc_ep_c()
            self.cvar += self.dt * (c1 + 2 * (c2 + c3) + c4) / 6.0
        self._invert()
        self.q = self.ifft(self.qh)
        if ps:
            self.c = self.ifft(self.ch)
        k4 = self._calc_ep_psi()
        self.Ke += self.dt * (k1 + 2 * (k2 + k3) + k4) / 6.0

    def _diagnostic_values(self):   # ref: niwqg/QGModel.py:632-737
        self._calc_derived_fields()
        vals = [("time", self.t), ("ke_qg", self._calc_ke_qg()), ("Ke", self.Ke),
                ("ens", 0.5 * (self.q ** 2).mean()), ("ep_psi", self._calc_ep_psi()),
                ("chi_q", self._calc_chi_q())]
        if self.passive_scalar:
            vals += [("C2", self.C2), ("cvar", self.cvar), ("gradC2", self.gradC2), ("Gamma_c", self.Gamma_c),
                     ("ep_c", self._calc_ep_c()), ("chi_c", self._calc_chi_c())]
        else:
            # the reference registers the scalar's entries whatever passive_scalar says (QGModel.py:690-722) and, without the
            # scalar, evaluates them on C2 = gradC2 = cvar = Gamma_c = 0, c = ch = 0, lapc = [0.] (QGModel.py:734-737): zeros
            vals += [("C2", 0.0), ("cvar", 0.0), ("gradC2", 0.0), ("Gamma_c", 0.0), ("ep_c", 0.0), ("chi_c", 0.0)]
        return vals

    def _increment_diagnostics(self):
        if not (self.tc % self.tdiags):
            for name, val in self._diagnostic_values():
                self.diagnostics.setdefault(name, []).append(val)

    def _print_status(self):        # ref: niwqg/QGModel.py:554-578
        self.tc += 1
        self.t += self.dt
        if (self.tc % self.twrite) == 0:
            self.ke = self._calc_ke_qg()
            self.cfl = self._calc_cfl()
            assert self.cfl < self.cflmax, "CFL condition violated"

    def _step_forward(self):
        self._step_etdrk4()
        self._increment_diagnostics()
        self._print_status()

    def run(self):
        while self.t < self.tmax:
            self._step_forward()

    def diag(self, name):
        return np.array(self.diagnostics[name])
