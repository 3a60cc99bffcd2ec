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
    if kind == "qg":
        dk = 2 * np.pi / L
        ll = dk * np.append(np.arange(0., nx / 2), np.arange(-nx / 2, 0.))
        ctx = _lib.Context(_lib.QG, nx, dk * np.arange(0., nx // 2 + 1), ll, np.ones((nx, nx // 2 + 1)), kw["dt"],
                           U=kw["U"], nu4=kw["nu4"], budgets=budgets)
        return ctx, None
    orc = O.NIWQGOracle(kind, **kw)
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
    jq = orc.ik * np.fft.fft2(orc.u * orc.q) + orc.il * np.fft.fft2(orc.v * orc.q)
    jq[0, 0] = 0
    assert rel(ctx.jacobian_psi_q(), jq) < 1e-13
    jp = np.fft.fft2(orc.u * orc.phix + orc.v * orc.phiy)
    jp[0, 0] = 0
    assert rel(ctx.jacobian_psi_phi(), jp) < 1e-13
    wj = np.fft.fft2((1j * (np.conj(orc.phix) * orc.phiy - np.conj(orc.phiy) * orc.phix)).real)
    wj[0, 0] = 0
    assert rel(ctx.jacobian_phic_phi(), wj) < 1e-13
    assert rel(ctx.refraction(), np.fft.fft2(orc.phi * orc.q_psi)) < 1e-13
    assert rel(ctx.field(_lib.F_QPSI), orc.q_psi) < 1e-13


def test_device_functions_against_the_reference_vectors(golden):
    """The reference's own per-function outputs (g1_functions_64.npz, written by make_golden.py from the reference)
    compared with the DEVICE directly, not through the oracle: the three Jacobians (ref Kernel.py:457-486,
    CoupledModel.py:59-73), the refraction source fft(phi*q_psi) (Kernel.py:332), the inversion outputs and the
    budget scalars of Kernel.py:664-701."""
    import niwqg_amd
    from test_oracle_golden import notebook_kwargs
    g = golden("g1_functions_64.npz")
    m = niwqg_amd.CoupledModel.Model(**notebook_kwargs(64, True))
    m.set_q(g["q0"])
    m.set_phi(g["phi0"])
    m._invert()
    m._calc_rel_vorticity()
    for name in ("ph", "qwh", "q_psi", "phix", "phiy", "phih", "u", "v"):
        assert rel(getattr(m, name), g[name]) < 1e-13, name
    assert rel(m.qh, g["qh"]) < 1e-13
    assert rel(m.jacobian_psi_q(), g["jac_psi_q"]) < 1e-13
    assert rel(m.jacobian_psi_phi(), g["jac_psi_phi"]) < 1e-13
    assert rel(m.jacobian_phic_phi(), g["jac_phic_phi"]) < 1e-13
    assert m.jacobian_psi_q()[0, 0] == 0 and m.jacobian_psi_phi()[0, 0] == 0 and m.jacobian_phic_phi()[0, 0] == 0
    assert rel(m._ctx.refraction(), g["refraction"]) < 1e-13
    assert rel(m._ctx.field(niwqg_amd._lib.F_QPSI), g["q_psi"]) < 1e-13
    m._calc_energy_conversion()
    b = np.array([m.gamma1, m.gamma2, m.xi1, m.xi2, m.pi, m._calc_ep_psi(), m._calc_chi_phi(), m._calc_ep_phi()])
    assert np.allclose(b, g["budget"], rtol=1e-9, atol=1e-25), (b, g["budget"])
    e = np.array([m._calc_ke_qg(), m._calc_ke_niw(), m._calc_pe_niw(), m._calc_cfl()])
    assert np.allclose(e, g["energies"], rtol=1e-12)


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


def test_every_export_survives_a_null_context():
    """A NULL ctx is an error code, never a crash (the four exports that used to dereference it first included)."""
    import ctypes
    from niwqg_amd import _lib
    lib = _lib.lib()
    buf = np.zeros(64)
    d = _lib._dptr(buf)
    n, f = ctypes.c_int(), ctypes.c_float()
    vp, ll_ = ctypes.c_void_p(), ctypes.c_longlong()
    assert lib.nq_destroy(None) == 0
    assert lib.nq_stream(None) is None
    assert lib.nq_device_bytes(None) == 0
    assert lib.nq_field_doubles(None, 0) == -1
    calls = [lambda: lib.nq_set_q(None, d), lambda: lib.nq_set_c(None, d), lambda: lib.nq_set_phi(None, d),
             lambda: lib.nq_invert(None), lambda: lib.nq_refresh_grad_phi(None), lambda: lib.nq_step(None, 1),
             lambda: lib.nq_sync(None), lambda: lib.nq_get_field(None, 0, d),
             lambda: lib.nq_get_scalar(None, 0, ctypes.byref(ctypes.c_double())),
             lambda: lib.nq_fft2(None, d, d), lambda: lib.nq_ifft2(None, d, d), lambda: lib.nq_rfft2(None, d, d),
             lambda: lib.nq_irfft2(None, d, d), lambda: lib.nq_jacobian_psi_q(None, d),
             lambda: lib.nq_jacobian_psi_phi(None, d), lambda: lib.nq_jacobian_phic_phi(None, d),
             lambda: lib.nq_products_uq_vq(None, d), lambda: lib.nq_refraction(None, d),
             lambda: lib.nq_diagnostics(None, d), lambda: lib.nq_get_coeff(None, 0, 0, d),
             lambda: lib.nq_timer_start(None), lambda: lib.nq_timer_stop(None, ctypes.byref(f)),
             lambda: lib.nq_profile_enable(None, 0), lambda: lib.nq_profile_read(None, ctypes.byref(n), ctypes.byref(f)),
             lambda: lib.nq_profile_read_all(None, (ctypes.c_int * 6)(), (ctypes.c_float * 6)()),
             lambda: lib.nq_slab_info(None, (ctypes.c_int * 8)()),
             lambda: lib.nq_group_buffers(None, 0, ctypes.byref(vp), ctypes.byref(vp), ctypes.byref(ll_)),
             lambda: lib.nq_upload_spectral(None, 0, d), lambda: lib.nq_download_spectral(None, 0, d),
             lambda: lib.nq_phase(None, 0, 0), lambda: lib.nq_reduce_buffer(None, 0, ctypes.byref(vp), ctypes.byref(n)),
             lambda: lib.nq_tick_snapshot(None), lambda: lib.nq_request_stage4_max(None), lambda: lib.nq_get_stage4_max(None, d)]
    for i, call in enumerate(calls):
        assert call() < 0, i
    # and a live context with NULL output pointers
    cw, _ = make_ctx("coupled", 64)
    for fn in (lib.nq_jacobian_psi_q, lib.nq_jacobian_psi_phi, lib.nq_jacobian_phic_phi, lib.nq_products_uq_vq,
               lib.nq_refraction, lib.nq_diagnostics):
        assert fn(cw.h, None) < 0
    # the tick's spectra do not exist before the first nq_tick_snapshot: an error with text, not a read of nothing
    big = np.zeros(2 * 64 * 64)
    for fid in (_lib.F_QH_TICK, _lib.F_PHIH_TICK, _lib.F_QWH_TICK):
        assert lib.nq_get_field(cw.h, fid, _lib._dptr(big)) == -4 and b"nq_tick_snapshot" in lib.nq_last_error(cw.h)
    assert lib.nq_get_field(cw.h, _lib.F_QH_MINUS_TICK, _lib._dptr(big)) == -4              # (and never in a context without dual_q)
    assert lib.nq_tick_snapshot(cw.h) == 0 and lib.nq_sync(cw.h) == 0
    assert lib.nq_get_field(cw.h, _lib.F_PHIH_TICK, _lib._dptr(big)) == 0
    assert lib.nq_get_field(cw.h, _lib.F_QH_MINUS_TICK, _lib._dptr(big)) == -4


@pytest.mark.parametrize("kind", ["coupled", "qg"])
def test_documented_buffer_sizes_are_exact(kind):
    """A caller that allocates exactly what include/niwqg_amd.h states is never overrun: every output export writes into a
    buffer of the documented size followed by a canary."""
    import ctypes
    from niwqg_amd import _lib
    lib = _lib.lib()
    nx, h = 64, 33
    ctx, orc = make_ctx(kind, nx)
    rng = np.random.default_rng(5)
    ctx.set_q(1e-5 * rng.standard_normal((nx, nx)))
    if kind == "coupled":
        ctx.set_phi(rng.standard_normal((nx, nx)) + 1j * rng.standard_normal((nx, nx)))
    CANARY = 12345.678

    def guarded(ndoubles, call):
        buf = np.full(ndoubles + 64, CANARY)
        rc = call(_lib._dptr(buf))
        assert rc == 0, lib.nq_last_error(ctx.h)
        assert np.all(buf[ndoubles:] == CANARY)
        assert not np.any(buf[:ndoubles] == CANARY)          # and the documented extent is fully written
        return buf[:ndoubles]

    full, half = 2 * nx * nx, 2 * nx * h
    guarded(full if kind == "coupled" else half, lambda p: lib.nq_jacobian_psi_q(ctx.h, p))
    guarded(2 * half, lambda p: lib.nq_products_uq_vq(ctx.h, p))
    if kind == "coupled":
        guarded(full, lambda p: lib.nq_jacobian_psi_phi(ctx.h, p))
        guarded(full, lambda p: lib.nq_jacobian_phic_phi(ctx.h, p))
        guarded(full, lambda p: lib.nq_refraction(ctx.h, p))
        guarded(32, lambda p: lib.nq_diagnostics(ctx.h, p))
        for which in range(6):
            guarded(full, lambda p: lib.nq_get_coeff(ctx.h, 1, which, p))
    for which in range(6):
        guarded(half, lambda p: lib.nq_get_coeff(ctx.h, 0, which, p))
    fields = [_lib.F_Q, _lib.F_QH, _lib.F_P, _lib.F_PH, _lib.F_U, _lib.F_V, _lib.F_QPSI]
    assert lib.nq_tick_snapshot(ctx.h) == 0
    fields += [_lib.F_QH_TICK]
    if kind == "coupled":
        fields += [_lib.F_PHI, _lib.F_PHIH, _lib.F_QW, _lib.F_QWH, _lib.F_PHIX, _lib.F_PHIY, _lib.F_PHIH_TICK, _lib.F_QWH_TICK]
    for fid in fields:
        nd = lib.nq_field_doubles(ctx.h, fid)
        assert nd > 0
        guarded(nd, lambda p: lib.nq_get_field(ctx.h, fid, p))
    x = rng.standard_normal((nx, nx))
    xin = np.ascontiguousarray(x)
    if kind == "qg":
        guarded(half, lambda p: lib.nq_rfft2(ctx.h, _lib._dptr(xin), p))
        spec = np.ascontiguousarray(np.fft.rfft2(x)).view(np.float64)
        guarded(nx * nx, lambda p: lib.nq_irfft2(ctx.h, _lib._dptr(spec), p))
    else:
        z = np.ascontiguousarray(x + 0j).view(np.float64)
        guarded(full, lambda p: lib.nq_fft2(ctx.h, _lib._dptr(z), p))
        guarded(full, lambda p: lib.nq_ifft2(ctx.h, _lib._dptr(z), p))
