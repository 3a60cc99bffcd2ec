"""Numpy model of the REDUCED pipeline that the HIP kernels execute -- TEST INFRASTRUCTURE ONLY.

The reference does 104 full c2c transforms per CoupledModel step; the device code
does 36 (9 per stage), keeps every real field on a half spectrum, replaces the
``fft(ifft(.).real)`` round trip of the psi-inversion by an algebraic Hermitian
projection, and evaluates most budget integrals with Parseval sums.  This module
states that algorithm in numpy, 2-D transform by 2-D transform, so that

  * tests/test_reduced_pipeline.py can prove on the CPU that it reproduces the
    faithful oracle (oracle/niwqg_oracle.py, itself pinned to the reference) to
    roundoff, including the Nyquist-line subtleties listed below;
  * the GPU tests can compare individual kernels with the matching function here.

Like everything under oracle/, it is never imported by the product path.

Half-spectrum ("HS") convention for spectra of REAL fields, Kernel family
-----------------------------------------------------------------------
An HS array has shape (N, N/2+1): all l, k-index 0..N/2, *full complex values*.
Wavenumber of column N/2 is kk[N/2] (= -N/2*dk for the Kernel family, +N/2*dk for
QGModel).  The reference's full-plane array X_full is recovered as
    X_full[l, k]   = X[l, k]                         0 <= k <= N/2
    X_full[-l,-k]  = conj(X[l, k])                   0 <  k <  N/2
plus, on row l=N/2 only, an anti-Hermitian "passenger" (see ``NyquistPassenger``)
that the reference carries in q-hat but that never reaches physical space.
Columns 0 and N/2 are self-mirrored in k, hold arbitrary complex data, and are
Hermitian-projected in l whenever a real field is synthesised (numpy's irfft2
does exactly that by dropping the imaginary part after the y-transform).
"""
from __future__ import annotations

import numpy as np

from .niwqg_oracle import SpectralGrid, spectral_filter, etdrk4_tables


def herm_cols(X):
    """Hermitian projection in l of the two self-mirrored columns (k=0 and k=N/2)."""
    Y = X.copy()
    for c in (0, X.shape[1] - 1):
        col = X[:, c]
        Y[:, c] = 0.5 * (col + np.conj(np.roll(col[::-1], 1)))
    return Y


def hs_to_full(X, passenger=None):
    """Rebuild the reference's (N,N) layout from an HS array (see module docstring)."""
    n = X.shape[0]
    full = np.zeros((n, n), complex)
    full[:, :n // 2 + 1] = X
    inner = X[:, 1:n // 2]                                    # k = 1 .. N/2-1
    mirrored = np.conj(np.roll(inner[::-1, :], 1, axis=0))    # row -l
    full[:, n // 2 + 1:] = mirrored[:, ::-1]                  # column -k
    if passenger is not None:
        full[n // 2, 1:n // 2] += passenger
        full[n // 2, n // 2 + 1:] -= np.conj(passenger)[::-1]
    return full


def full_to_hs(F):
    """HS part of a full-plane spectrum of a REAL field (drops the passenger row content
    by Hermitian-projecting row l=N/2 of the interior columns)."""
    n = F.shape[0]
    X = F[:, :n // 2 + 1].copy()
    k = np.arange(1, n // 2)
    X[n // 2, 1:n // 2] = 0.5 * (F[n // 2, k] + np.conj(F[n // 2, n - k]))
    return X


def hs_mean_product(A, B):
    """mean(a*b) for real fields a=irfft2(A), b=irfft2(B) by Parseval on the half spectrum."""
    n = A.shape[0]
    A, B = herm_cols(A), herm_cols(B)
    w = np.full(A.shape[1], 2.0)
    w[0] = w[-1] = 1.0
    return float((w * (A * np.conj(B)).real).sum()) / n ** 4


class ReducedNIWQG:
    """Reduced-transform model of CoupledModel / UnCoupledModel (see module docstring).

    State between stages is what the device keeps: ``qh`` (HS), ``phih`` (full plane)
    and the products of the last inversion (``ph``, ``qwh`` in HS; ``phi``; and the
    gradient source ``phih_grad`` from which phix/phiy derive -- for UnCoupled it goes
    stale exactly like the reference's phix/phiy, quirk Q1).
    """

    def __init__(self, kind="coupled", nx=128, L=5e5, dt=10000.0, tmax=250000.0, twrite=1000.0,
                 use_filter=True, cflmax=0.8, U=0.0, f=1e-4, N=0.01, m=0.025, nu4=0, nu4w=0, nu=20,
                 nuw=50.0, mu=0, muw=0, dealias=False, tdiags=10, budgets=True, **_ignored):
        self.kind, self.nx, self.L, self.dt = kind, nx, L, dt
        self.tmax, self.twrite, self.tdiags, self.cflmax = tmax, twrite, tdiags, cflmax
        self.U, self.f = U, f
        self.nu4, self.nu4w, self.nu, self.nuw, self.mu, self.muw = nu4, nu4w, nu, nuw, mu, muw
        self.kappa2 = (m * f / N) ** 2
        self.hslash = f / self.kappa2
        self.budgets = budgets
        n, h = nx, nx // 2 + 1
        G = self.grid = SpectralGrid(nx, L, half=False)
        self.kk, self.ll = G.kk, G.ll
        # full-plane (phi equation) operators
        self.ik, self.il, self.wv2 = G.ik, G.il, G.wv2
        self.filtr = spectral_filter(G, use_filter, dealias)
        cw = (-1j * G.k * U - nu4w * G.wv4 - 0.5j * f * (G.wv2 / self.kappa2) - nuw * G.wv2 - muw) + 0j
        self.cw = etdrk4_tables(cw, dt)
        # half-spectrum (q equation) operators: first N/2+1 columns of the Kernel-convention planes
        self.k_h, self.l_h = G.k[:, :h], G.l[:, :h]
        self.wv2_h, self.wv2i_h, self.wv4_h = G.wv2[:, :h], G.wv2i[:, :h], G.wv4[:, :h]
        self.filtr_h = self.filtr[:, :h]
        cq = (-1j * G.k * U - nu4 * G.wv4 - nu * G.wv2 - mu) + 0j
        self.cq = {key: val[:, :h] for key, val in etdrk4_tables(cq, dt).items()}
        # "il" with the Nyquist row removed on interior columns (emulates .real of the c2c reference)
        lz = self.l_h.copy()
        lz[n // 2, 1:n // 2] = 0.0
        self.ilz_h = 1j * lz
        self.ik_h = 1j * self.k_h
        self.qh = np.zeros((n, h), complex)
        self.phih = np.zeros((n, n), complex)
        self.phih_grad = np.zeros((n, n), complex)
        self.phi = np.zeros((n, n), complex)
        self.ph = np.zeros((n, h), complex)
        self.qwh = np.zeros((n, h), complex)
        self.passenger = np.zeros(n // 2 - 1, complex)
        self.t, self.tc = 0, 0
        self.n2d = 0            # number of 2-D transforms executed (c2c-equivalents)

    # -- 2-D transforms: each call is ONE c2c-equivalent on the device.
    def _c2c_inv(self, X):
        self.n2d += 1
        return np.fft.ifft2(X)

    def _c2c_fwd(self, x):
        self.n2d += 1
        return np.fft.fft2(x)

    def _pair_inv(self, A, B):
        """two real fields from two HS spectra = one packed c2c on the device"""
        self.n2d += 1
        return np.fft.irfft2(A), np.fft.irfft2(B)

    def _pair_fwd(self, a, b):
        self.n2d += 1
        return np.fft.rfft2(a), np.fft.rfft2(b)

    # -- pieces -------------------------------------------------------------
    def _grad_phi(self):
        return self._c2c_inv(self.ik * self.phih_grad), self._c2c_inv(self.il * self.phih_grad)

    def _invert(self, refresh_phi=True):
        """CoupledModel._invert + _calc_rel_vorticity (ref CoupledModel.py:75-97, :145-152) or the
        UnCoupled ones (UnCoupledModel.py:54-64).  3 c2c + 1 packed forward for Coupled."""
        if refresh_phi:
            self.phi = self._c2c_inv(self.phih)
        if self.kind == "coupled":
            self.phih_grad = self.phih.copy()
            phix, phiy = self._grad_phi()
            a = np.abs(self.phi) ** 2
            b = -2.0 * (np.conj(phix) * phiy).imag
            A, B = self._pair_fwd(a, b)
            B[0, 0] = 0.0
            self.qwh = self.filtr_h * (0.5 * (0.5 * (-self.wv2_h) * A + B) / self.f)
            self.ph = self.wv2i_h * (self.qwh - herm_cols(self.qh))
        else:
            self.ph = -self.wv2i_h * herm_cols(self.qh)

    def refresh_grad_phi(self):
        """what _calc_pe_niw does to phix/phiy (ref Kernel.py:610): quirk Q1."""
        self.phih_grad = self.phih.copy()

    def set_q(self, q):
        self.qh = np.fft.rfft2(q)
        self.passenger[:] = 0.0
        self._invert(refresh_phi=False)
        self.Ke = 0.5 * hs_mean_product(np.sqrt(self.wv2_h) * self.ph, np.sqrt(self.wv2_h) * self.ph)

    def set_phi(self, phi):
        self.phi = np.array(phi, complex)
        self.phih = np.fft.fft2(self.phi)
        self.refresh_grad_phi()
        M2 = float(self.nx) ** 4
        self.Pw = 0.25 * (self.wv2 * np.abs(self.phih) ** 2).sum() / M2 / self.kappa2
        self.Kw = 0.5 * (np.abs(self.phih) ** 2).sum() / M2

    def _stage_rhs(self, want_budget):
        """Nonlinear terms of both equations from the current inversion products.
        Coupled: 2 packed inverse + 2 c2c inverse (phix, phiy) + 1 packed forward + 2 c2c forward;
        phi itself is carried from the inversion."""
        n = self.nx
        u, v = self._pair_inv(-self.ilz_h * self.ph, self.ik_h * self.ph)
        if self.kind == "coupled":
            q, qw = self._pair_inv(self.qh, self.qwh)
            qpsi = q - qw
        else:
            q, _ = self._pair_inv(self.qh, self.qh * 0)
            qpsi = q
        phix, phiy = self._grad_phi()
        F1, F2 = self._pair_fwd(u * q, v * q)
        Nq = -(self.ik_h * F1 + self.ilz_h * F2)
        Nq[0, 0] = 0.0
        Npass = self.il[n // 2, 1:n // 2] * F2[n // 2, 1:n // 2] * -1.0     # passenger source, row l=N/2
        Jphys = u * phix + v * phiy
        # ONE transform for the whole phi tendency: N_phi = F[-J - (i/2) phi q_psi].  The reference zeroes
        # jach[0,0] but not the refraction term (Kernel.py:468 vs :332): the domain sum of J is added back there.
        Wn = self._c2c_fwd(-Jphys - 0.5j * self.phi * qpsi)
        Nw = Wn.copy()
        Nw[0, 0] += Jphys.sum()
        rates = None
        if want_budget:
            rates = self._budget_rates(Wn)
        return Nq, Nw, Npass, rates

    def _budget_rates(self, Wn):
        """k, p, a of ref Kernel.py:319-322, every integral a Parseval sum.  gamma1 and xi2 are triple products
        in physical space, but q_psi*phi is the refraction source, and the budgets only ever use gamma1+gamma2 and
        xi1+xi2, which are projections of the WHOLE phi tendency Wn = F[-J - (i/2) phi q_psi] (un-zeroed at [0,0]):
            gamma1 + gamma2 = -hslash/2 sum Re(conj(lap_h) Wn) / (M^2 f),   xi1 + xi2 = -sum Im(conj(diss_h) Wn) / (M^2 f)
        so neither lap(phi), diss(phi), J nor R is needed on its own."""
        n = self.nx
        M2 = float(n) ** 4
        g = self.phih_grad
        lapphi_h = -self.wv2 * self.phih
        lap2phi_h = self.wv2 ** 2 * self.phih
        diss_h = -self.nu4w * lap2phi_h + self.nuw * lapphi_h - self.muw * self.phih
        gamma12 = -0.5 * self.hslash * (np.conj(lapphi_h) * Wn).real.sum() / M2 / self.f
        xi12 = -(np.conj(diss_h) * Wn).imag.sum() / M2 / self.f
        # ep_psi (ref Kernel.py:635-640): all Parseval on the half spectrum
        ep_psi = (self.nu4 * hs_mean_product(self.qh, self.wv4_h * self.ph)
                  - self.nu * hs_mean_product(self.ph, -self.wv2_h * self.qh)
                  + self.mu * hs_mean_product(self.ph, self.qh))
        grad2 = (self.wv2 * np.abs(g) ** 2).sum() / M2          # mean(|phix|^2+|phiy|^2), stale-aware
        lap2 = (self.wv2 ** 2 * np.abs(self.phih) ** 2).sum() / M2
        glap2 = (self.wv2 ** 3 * np.abs(self.phih) ** 2).sum() / M2
        phi2 = (np.abs(self.phih) ** 2).sum() / M2
        chi_phi = (-0.5 * self.nu4w * glap2 - 0.5 * self.nuw * lap2 - 0.5 * self.muw * grad2) / self.kappa2
        ep_phi = -self.nu4w * lap2 - self.nuw * grad2 - self.muw * phi2
        k = -gamma12 + xi12 + ep_psi
        p = gamma12 + chi_phi
        return k, p, ep_phi

    def _etd(self, c, F, y0, y1, N, stage):
        N0, Na, Nb, Nc = N
        if stage == 0:
            return (c["Eh"] * y0 + N0 * c["Q"]) * F
        if stage == 1:
            return (c["Eh"] * y0 + Na * c["Q"]) * F
        if stage == 2:
            return (c["Eh"] * y1 + (2.0 * Nb - N0) * c["Q"]) * F
        return (c["E"] * y0 + N0 * c["f0"] + 2.0 * (Na + Nb) * c["fab"] + Nc * c["fc"]) * F

    def step(self):
        n = self.nx
        row = (n // 2, slice(1, n // 2))
        cp = {key: val[row] for key, val in self.cq.items()}
        Fp = self.filtr_h[row]
        q0, w0, p0 = self.qh.copy(), self.phih.copy(), self.passenger.copy()
        q1 = w1 = p1 = None
        Nq, Nw, Np, rates = [None] * 4, [None] * 4, [None] * 4, []
        for s in range(4):
            Nq[s], Nw[s], Np[s], r = self._stage_rhs(self.budgets)
            rates.append(r)
            self.qh = self._etd(self.cq, self.filtr_h, q0, q1, Nq, s)
            self.phih = self._etd(self.cw, self.filtr, w0, w1, Nw, s)
            self.passenger = self._etd(cp, Fp, p0, p1, Np, s)
            if s == 0:
                q1, w1, p1 = self.qh.copy(), self.phih.copy(), self.passenger.copy()
            self._invert()
        if self.budgets:
            (k1, a1, e1), (k2, a2, e2), (k3, a3, e3), (k4, a4, e4) = rates
            self.Ke += self.dt * (k1 + 2 * (k2 + k3) + k4) / 6.0
            self.Pw += self.dt * (a1 + 2 * (a2 + a3) + a4) / 6.0
            self.Kw += self.dt * (e1 + 2 * (e2 + e3) + e4) / 6.0
        self.tc += 1
        self.t += self.dt

    # -- views in the reference's layout ---------------------------------------
    @property
    def q(self):
        return np.fft.irfft2(self.qh)

    @property
    def qh_full(self):
        return hs_to_full(self.qh, self.passenger)

    @property
    def ph_full(self):
        return hs_to_full(self.ph)

    @property
    def qwh_full(self):
        return hs_to_full(self.qwh)


class ReducedQG:
    """Reduced model of QGModel (ref QGModel.py:328-407): 2 packed inverse + 1 packed forward per
    stage... here u,v,q need 3 real inverses = 2 packed c2c (one slot idle) and uq,vq = 1 packed."""

    def __init__(self, nx=128, L=5e5, dt=10000.0, use_filter=True, U=0.0, nu4=5e9, nu=0, mu=0, beta=0,
                 dealias=False, **_ignored):
        self.nx, self.L, self.dt = nx, L, dt
        self.nu4, self.nu, self.mu = nu4, nu, mu
        G = self.grid = SpectralGrid(nx, L, half=True)
        self.ik, self.il = G.ik, G.il
        self.wv2, self.wv2i, self.wv4 = G.wv2, G.wv2i, G.wv4
        self.filtr = spectral_filter(G, use_filter, dealias)
        c = (-nu4 * G.wv4 - nu * G.wv2 - mu - 1j * G.k * U) + 0j
        c = c + beta * G.ik * G.wv2i
        self.c = etdrk4_tables(c, dt)
        self.qh = np.zeros((nx, nx // 2 + 1), complex)
        self.ph = self.qh.copy()
        self.tc = 0

    def set_q(self, q):
        self.qh = np.fft.rfft2(q)
        self.ph = -self.wv2i * self.qh
        self.Ke = 0.5 * self._spec_var(np.sqrt(self.wv2) * self.ph)

    def _spec_var(self, ah):
        d = 2.0 * np.abs(ah) ** 2 / float(self.nx) ** 4
        d[:, 0] *= 0.5
        d[:, -1] *= 0.5
        d[0, 0] = 0
        return d.sum()

    def _rhs(self):
        u = np.fft.irfft2(-self.il * self.ph)
        v = np.fft.irfft2(self.ik * self.ph)
        q = np.fft.irfft2(self.qh)
        return -(self.ik * np.fft.rfft2(u * q) + self.il * np.fft.rfft2(v * q))

    def _ep_psi(self, qh_for_q):
        """ref QGModel.py:588-593 with self.q = irfft2(qh_for_q) (stale within the step)."""
        return (self.nu4 * hs_mean_product(qh_for_q, self.wv4 * self.ph)
                - self.nu * hs_mean_product(self.ph, -self.wv2 * self.qh)
                + self.mu * hs_mean_product(self.ph, qh_for_q))

    def step(self):
        c, F = self.c, self.filtr
        q0 = self.qh.copy()
        N0 = self._rhs()
        self.qh = (c["Eh"] * q0 + N0 * c["Q"]) * F
        q1 = self.qh.copy()
        self.ph = -self.wv2i * self.qh
        k1 = self._ep_psi(q0)
        Na = self._rhs()
        self.qh = (c["Eh"] * q0 + Na * c["Q"]) * F
        self.ph = -self.wv2i * self.qh
        k2 = self._ep_psi(q0)
        Nb = self._rhs()
        self.qh = (c["Eh"] * q1 + (2.0 * Nb - N0) * c["Q"]) * F
        self.ph = -self.wv2i * self.qh
        k3 = self._ep_psi(q0)
        Nc = self._rhs()
        self.qh = (c["E"] * q0 + N0 * c["f0"] + 2.0 * (Na + Nb) * c["fab"] + Nc * c["fc"]) * F
        self.ph = -self.wv2i * self.qh
        k4 = self._ep_psi(self.qh)
        self.Ke += self.dt * (k1 + 2 * (k2 + k3) + k4) / 6.0
        self.tc += 1

    @property
    def q(self):
        return np.fft.irfft2(self.qh)
