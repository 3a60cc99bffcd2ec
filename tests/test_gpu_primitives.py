"""GPU parity of the building blocks: FFT seam, ETDRK4 coefficient planes, initial inversion.
All calls go through the C ABI (niwqg_amd._lib.Context)."""
import numpy as np
import pytest

from oracle import niwqg_oracle as O
from test_oracle_golden import notebook_kwargs, rel, L, K0, U0

pytestmark = pytest.mark.gpu


def make_ctx(kind, nx, use_filter=True, budgets=False, **over):
    from niwqg_amd import _lib
    kw = notebook_kwargs(nx, use_filter)
    kw.update(over)
    orc = O.NIWQGOracle(kind, **kw) if kind != "qg" else None
    model = {"coupled": _lib.COUPLED, "uncoupled": _lib.UNCOUPLED}[kind]
    ctx = _lib.Context(model, nx, orc.kk, orc.ll, orc.filtr, kw["dt"], U=kw["U"], f=kw["f"], kappa2=orc.kappa2,
                       nu=kw["nu"], nu4=kw["nu4"], mu=kw["mu"], nuw=kw["nuw"], nu4w=kw["nu4w"], muw=kw["muw"],
                       budgets=budgets)
    return ctx, orc


@pytest.mark.parametrize("nx", [64, 128, 256, 512, 1024])
def test_fft_seam_matches_numpy(nx):
    ctx, _ = make_ctx("coupled", nx)
    rng = np.random.default_rng(nx)
    a = rng.standard_normal((nx, nx)) + 1j * rng.standard_normal((nx, nx))
    r = rng.standard_normal((nx, nx))
    assert rel(ctx.fft2(a), np.fft.fft2(a)) < 2e-15
    assert rel(ctx.ifft2(a), np.fft.ifft2(a)) < 2e-15
    assert rel(ctx.rfft2(r), np.fft.rfft2(r)) < 2e-15
    h = np.fft.rfft2(r) * (1 + 0.3j)          # not Hermitian on the self-mirrored columns
    assert rel(ctx.irfft2(h), np.fft.irfft2(h)) < 2e-15
    assert rel(ctx.ifft2(ctx.fft2(a)), a) < 2e-15


def test_etdrk4_coefficient_planes():
    ctx, orc = make_ctx("coupled", 64, use_filter=True)
    names = ["E", "Eh", "Q", "f0", "fab", "fc"]
    for i, nm in enumerate(names):
        mine = ctx.coeff(0, i)
        ref = orc.coef_q[nm][:, :33]
        assert np.all(np.abs(mine - ref) <= 1e-11 * np.abs(ref)), ("q", nm, np.abs(mine - ref).max())
        mine = ctx.coeff(1, i)
        ref = orc.coef_w[nm]
        assert np.all(np.abs(mine - ref) <= 1e-11 * np.abs(ref)), ("w", nm)


def test_initial_inversion_and_fields():
    nx = 64
    ctx, orc = make_ctx("coupled", nx)
    from niwqg_amd import _lib
    q0 = O.lamb_dipole(orc.grid, U=U0, R=2 * np.pi / K0)
    phi0 = 0.2 * O.wave_packet(orc.grid, k=3 * K0, l=K0, R=L / 6, x0=L / 2, y0=L / 2)
    # order phi then q so that psi contains the wave part
    orc.set_phi(phi0); orc.set_q(q0)
    ctx.set_phi(phi0); ctx.set_q(q0)
    h = nx // 2 + 1
    assert rel(ctx.field(_lib.F_QH), orc.qh[:, :h]) < 1e-14
    assert rel(ctx.field(_lib.F_PHIH), orc.phih) < 1e-14
    assert rel(ctx.field(_lib.F_PHI), orc.phi) < 1e-14
    assert rel(ctx.field(_lib.F_QWH), orc.qwh[:, :h]) < 1e-13
    assert rel(ctx.field(_lib.F_Q), orc.q) < 1e-14
    assert rel(ctx.field(_lib.F_P), orc.p) < 1e-13
    assert rel(ctx.field(_lib.F_U), orc.u) < 1e-13
    assert rel(ctx.field(_lib.F_V), orc.v) < 1e-13
    assert rel(ctx.field(_lib.F_QW), orc.qw) < 1e-13
    assert rel(ctx.field(_lib.F_PHIX), orc.phix) < 1e-13
    assert rel(ctx.field(_lib.F_PHIY), orc.phiy) < 1e-13
    assert abs(ctx.scalar(_lib.S_CFL) * orc.dt / orc.dx - orc._calc_cfl()) < 1e-13 * orc._calc_cfl()
    assert abs(ctx.scalar(_lib.S_KE_QG) - orc._calc_ke_qg()) < 1e-13 * orc._calc_ke_qg()
    assert abs(ctx.scalar(_lib.S_KE_NIW) - orc._calc_ke_niw()) < 1e-13 * orc._calc_ke_niw()
    assert abs(ctx.scalar(_lib.S_PE_NIW) - orc._calc_pe_niw()) < 1e-13 * orc._calc_pe_niw()
    f1, f2 = ctx.products_uq_vq()
    assert rel(f1, np.fft.rfft2(orc.u * orc.q)) < 1e-13
    assert rel(f2, np.fft.rfft2(orc.v * orc.q)) < 1e-13
    assert rel(ctx.advection_phi(), np.fft.fft2(orc.u * orc.phix + orc.v * orc.phiy)) < 1e-13
    wj = np.fft.rfft2((1j * (np.conj(orc.phix) * orc.phiy - np.conj(orc.phiy) * orc.phix)).real)
    assert rel(ctx.wave_jacobian(), wj) < 1e-13


@pytest.mark.parametrize("use_filter", [False, True])
def test_coupled_steps_match_oracle(use_filter):
    nx = 64
    ctx, orc = make_ctx("coupled", nx, use_filter=use_filter)
    from niwqg_amd import _lib
    q0 = O.lamb_dipole(orc.grid, U=U0, R=2 * np.pi / K0)
    phi0 = 0.2 * O.wave_packet(orc.grid, k=3 * K0, l=K0, R=L / 6, x0=L / 2, y0=L / 2)
    orc.set_q(q0); orc.set_phi(phi0)
    ctx.set_q(q0); ctx.set_phi(phi0)
    for n in (1, 9):
        for _ in range(n):
            orc._step_forward()
        ctx.step(n)
        print("steps", orc.tc, rel(ctx.field(_lib.F_Q), orc.q), rel(ctx.field(_lib.F_PHI), orc.phi))
        assert rel(ctx.field(_lib.F_Q), orc.q) < 1e-12
        assert rel(ctx.field(_lib.F_PHI), orc.phi) < 1e-12
        assert rel(ctx.field(_lib.F_PHIH), orc.phih) < 1e-12


def test_c_abi_fails_loudly_with_error_text():
    """Every misuse returns a negative code and leaves a message in nq_last_error: nothing is silently ignored."""
    import ctypes
    from niwqg_amd import _lib
    lib = _lib.lib()
    kk = np.zeros(96)
    p = _lib.Params(model=0, nx=96, budgets=1, dual_q=0, dt=1.0, U=0, f=1e-4, kappa2=1, nu=0, nu4=0, mu=0, nuw=0,
                    nu4w=0, muw=0, beta=0, passive_scalar=0, nu4c=0, nuc=0, muc=0)
    h = ctypes.c_void_p()
    r = np.exp(2j * np.pi * (np.arange(1.0, 33.0) / 32.0)).view(np.float64)
    rc = lib.nq_create(ctypes.byref(p), _lib._dptr(kk), _lib._dptr(kk), _lib._dptr(np.ones((96, 96))), _lib._dptr(r), 0,
                     ctypes.byref(h))
    assert rc < 0 and b"nx=96" in lib.nq_last_error(None)               # not a power of two
    p.nx, p.model = 64, 7
    rc = lib.nq_create(ctypes.byref(p), _lib._dptr(kk), _lib._dptr(kk), _lib._dptr(np.ones((64, 64))), _lib._dptr(r), 0,
                     ctypes.byref(h))
    assert rc < 0 and b"model" in lib.nq_last_error(None)
    dk = 2 * np.pi / L
    ll = dk * np.append(np.arange(0., 32), np.arange(-32., 0.))
    ctx = _lib.Context(_lib.QG, 64, dk * np.arange(0., 33), ll, np.ones((64, 33)), 100.0, nu4=1e9)
    with pytest.raises(RuntimeError, match="wave field"):
        ctx.set_phi(np.zeros((64, 64), complex))                      # QGModel has no phi
    with pytest.raises(RuntimeError, match="passive scalar"):
        ctx.set_c(np.zeros((64, 64)))
    with pytest.raises(RuntimeError):
        ctx.field(99)
    cw, _ = make_ctx("coupled", 64)
    with pytest.raises(RuntimeError, match="set_phi"):
        cw.diagnostic_sums()                                          # no phi yet
    out = np.zeros(4)
    assert lib.nq_phase(cw.h, 0, 9) < 0                                 # stage out of range
    assert lib.nq_destroy(None) == 0
