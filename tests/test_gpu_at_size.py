"""Parity AT SIZE: the BASELINE configurations on their own workloads, and states that fill the whole spectrum.

The small-grid tests pin the algorithm; these pin the size-dependent machinery -- the per-size row plans (2048: 256
threads x 8 points, 4096: persistent 512 x 8 with two transforms in flight, 8192: even/odd split), high-index twiddles,
the filter band, the Nyquist row/column rules -- against the CPU oracle (oracle/niwqg_oracle.py, pinned to the reference
by tests/test_oracle_golden.py) on white-noise states, where every mode of the grid is populated.

The oracle runs with ``workers=NW`` here (thread pool for its coefficient tables, scipy.fft for its FFT seam: 3e-16
away from the numpy.fft path) so that one step at 4096^2 costs a few minutes instead of half an hour;
tests/test_oracle_golden.py::test_oracle_workers_option_is_arithmetic_neutral pins that option on the CPU.

Tolerances: relative L2; BASELINE bar 1e-10 over 100 steps.
"""
import os

import numpy as np
import pytest

from oracle import niwqg_oracle as O
from test_oracle_golden import notebook_kwargs, rel, L, K0

pytestmark = pytest.mark.gpu

NW = max(1, min(16, (os.cpu_count() or 2) - 1))


def steps(m, n):
    while m.tc < n:
        m._step_forward()


# ---- BASELINE config 2: QGModel 2048^2, random q ---------------------------------------------------------------------
def test_config2_qgmodel_2048_random_q_against_the_oracle():
    """ref: niwqg/QGModel.py:328-407 on BASELINE.json configs[1] exactly as bench.py --model qg --nx 2048 builds it."""
    import niwqg_amd
    import bench
    nx = 2048
    kw = bench.c3_kwargs(nx, "qg")
    q0 = 1e-5 * np.random.default_rng(0).standard_normal((nx, nx))
    m = niwqg_amd.QGModel.Model(**kw)
    o = O.QGOracle(coeff_chunk=8, workers=NW, **kw)
    for x in (m, o):
        x.set_q(q0)
    for _ in range(2):
        o._step_forward()
    steps(m, 2)
    eq, eh, ep = rel(m.q, o.q), rel(m.qh, o.qh), rel(m.ph, o.ph)
    print("QG 2048^2 randn, 2 steps: rel q %.2e qh %.2e ph %.2e; Ke %.6e vs %.6e" % (eq, eh, ep, m.Ke, o.Ke))
    assert eq < 1e-11 and eh < 1e-11 and ep < 1e-11
    assert abs(m.Ke - o.Ke) < 1e-10 * abs(o.Ke)
    assert abs(m._calc_ke_qg() - o._calc_ke_qg()) < 1e-11 * o._calc_ke_qg()
    # the spectrum really is full: the last retained column / the Nyquist row carry energy before the filter acts
    h0 = np.abs(np.fft.rfft2(q0))
    assert h0[nx // 2, 5] > 0 and h0[7, nx // 2] > 0


# ---- BASELINE config 5: UnCoupledModel 1024^2 ensemble members --------------------------------------------------------
@pytest.mark.parametrize("member,tdiags", [(0, 1), (0, 10 ** 9), (7, 1), (7, 10 ** 9)])
def test_config5_uncoupled_1024_member_against_the_oracle(member, tdiags):
    """ref: niwqg/UnCoupledModel.py:54-64 with quirk Q1 (stale phix, phiy: the trajectory depends on tdiags) at the size and
    on the initial condition of BASELINE.json configs[4] (niwqg_amd.ensemble.config5_member)."""
    from niwqg_amd import ensemble
    nx = 1024
    m = ensemble.config5_member(member, nx=nx, tdiags=tdiags)
    kw = dict(L=m.L, nx=nx, tmax=1e30, dt=m.dt, m=m.m, N=m.N, f=m.f, twrite=10 ** 9, tdiags=tdiags, nu4=m.nu4, nu4w=m.nu4w,
              nu=m.nu, nuw=m.nuw, mu=m.mu, muw=m.muw, use_filter=True, U=m.U)
    o = O.NIWQGOracle("uncoupled", coeff_chunk=8, workers=NW, **kw)
    o.set_q(1e-5 * np.random.default_rng(member).standard_normal((nx, nx)))
    o.set_phi(0.1 * O.wave_packet(o.grid, k=3 * K0, l=0, R=L / 6, x0=L / 2, y0=L / 2))
    for _ in range(2):
        o._step_forward()
    steps(m, 2)
    eq, ep, eh = rel(m.q, o.q), rel(m.phi, o.phi), rel(m.phih, o.phih)
    print("UnCoupled 1024^2 member %d tdiags=%g, 2 steps: rel q %.2e phi %.2e phih %.2e" % (member, tdiags, eq, ep, eh))
    assert eq < 1e-11 and ep < 1e-11 and eh < 1e-11
    assert rel(m.phix, o.phix) < 1e-11 and rel(m.phiy, o.phiy) < 1e-11      # stale or fresh, as the reference leaves them
    assert np.allclose([m.Ke, m.Pw, m.Kw], [o.Ke, o.Pw, o.Kw], rtol=1e-8)


def test_config5_q1_is_visible_at_size():
    """tdiags=1 and tdiags=inf must DIFFER (quirk Q1 acts at 1024^2 as it does in golden g4 at 64^2)."""
    from niwqg_amd import ensemble
    a = ensemble.config5_member(3, nx=1024, tdiags=1)
    b = ensemble.config5_member(3, nx=1024, tdiags=10 ** 9)
    steps(a, 3)
    steps(b, 3)
    assert rel(a.phi, b.phi) > 1e-6


# ---- rough fields at size: CoupledModel, every dissipation parameter non-zero ----------------------------------------
def rough_kwargs(nx):
    kw = notebook_kwargs(nx, True)
    kw.update(nu4w=kw["nu4"] * 0.1, mu=1e-8, muw=2e-8)
    return kw


@pytest.mark.parametrize("nx", [2048, 4096])
def test_rough_field_coupled_step_against_the_oracle(nx):
    """One full ETDRK4 step (ref: niwqg/Kernel.py:307-397, CoupledModel.py:59-97) of white-noise q and phi: the filter
    band, the Nyquist lines and every high-index twiddle of the size's own kernels are exercised and compared."""
    import niwqg_amd
    kw = rough_kwargs(nx)
    rng = np.random.default_rng(11)
    q0 = 1e-5 * rng.standard_normal((nx, nx))
    phi0 = 0.05 * (rng.standard_normal((nx, nx)) + 1j * rng.standard_normal((nx, nx)))
    m = niwqg_amd.CoupledModel.Model(**kw)
    m.set_q(q0)
    m.set_phi(phi0)
    steps(m, 1)
    got = dict(q=m.q, phi=m.phi, phih=m.phih, qh=m.qh, ph=m.ph, b=[m.Ke, m.Pw, m.Kw])
    del m                                            # free the device before the oracle's host arrays grow
    o = O.NIWQGOracle("coupled", coeff_chunk=4, workers=NW, **kw)
    o.set_q(q0)
    o.set_phi(phi0)
    o._step_forward()
    errs = {k: rel(got[k], getattr(o, k)) for k in ("q", "phi", "phih", "ph")}
    errs["qh"] = rel(got["qh"], o.qh)
    print("Coupled %d^2 rough field, 1 step:" % nx, {k: "%.2e" % v for k, v in errs.items()})
    for k, v in errs.items():
        assert v < 1e-11, (k, v)
    assert np.allclose(got["b"], [o.Ke, o.Pw, o.Kw], rtol=1e-8)
    # what was compared is rough: the filter-band modes are populated in the oracle's result
    k65 = int(0.65 * nx / 2) + 3
    assert np.abs(o.phih[k65, k65]) > 0 and np.abs(o.qh[k65, 3]) > 0


# ---- 8192^2: the seam and the even/odd row kernels on a full spectrum --------------------------------------------------
def test_fft_seam_8192_against_numpy():
    """nq_fft2 / nq_ifft2 (ref: niwqg/Kernel.py:562-566) of a random 8192^2 plane against scipy/numpy pocketfft."""
    import scipy.fft
    import niwqg_amd
    nx = 8192
    m = niwqg_amd.CoupledModel.Model(**notebook_kwargs(nx, True))
    rng = np.random.default_rng(3)
    a = rng.standard_normal((nx, nx)) + 1j * rng.standard_normal((nx, nx))
    fa = m.fft(a)
    ref = scipy.fft.fft2(a, workers=NW)
    e = rel(fa, ref)
    print("8192^2 nq_fft2 vs pocketfft: %.2e" % e)
    assert e < 2e-15
    del ref
    e = rel(m.ifft(fa), a)
    assert e < 2e-15


def test_row_kernels_8192_on_a_full_spectrum_against_numpy():
    """k_x_products_eo / k_x_wavepv_eo (the 8192-point rows as two 4096-point problems) on white-noise q and phi, through the
    C-ABI Jacobian exports, against the reference's formulas evaluated with pocketfft on the host:
    jacobian_psi_q (ref Kernel.py:471-486), jacobian_psi_phi (:457-469), refraction (:332), jacobian_phic_phi and the
    wave-PV inversion (CoupledModel.py:59-97)."""
    import scipy.fft
    import niwqg_amd
    nx = 8192
    kw = rough_kwargs(nx)
    m = niwqg_amd.CoupledModel.Model(**kw)
    rng = np.random.default_rng(21)
    q0 = 1e-5 * rng.standard_normal((nx, nx))
    phi0 = 0.05 * (rng.standard_normal((nx, nx)) + 1j * rng.standard_normal((nx, nx)))
    m.set_phi(phi0)            # phi first: the inversion of set_q then contains the wave part (quirk Q2)
    m.set_q(q0)

    def F(a):
        return scipy.fft.fft2(a, workers=NW)

    def Fi(a):
        return scipy.fft.ifft2(a, workers=NW)

    kk = m.kk
    ik, il = 1j * kk[None, :], 1j * m.ll[:, None]
    wv2 = kk[None, :] ** 2 + m.ll[:, None] ** 2
    wv2i = np.zeros_like(wv2)
    wv2i[wv2 != 0] = 1.0 / wv2[wv2 != 0]
    phih = F(phi0)
    phix, phiy = Fi(ik * phih), Fi(il * phih)
    jw = F((1j * (np.conj(phix) * phiy - np.conj(phiy) * phix)).real)
    jw[0, 0] = 0
    e = rel(m.jacobian_phic_phi(), jw)
    print("8192^2 jacobian_phic_phi %.2e" % e)
    assert e < 1e-12
    qwh = 0.5 * (0.5 * (-wv2 * F(np.abs(phi0) ** 2)) + jw) / m.f * m.filtr
    del jw
    qh = F(q0)
    pw = Fi(wv2i * qwh).real
    pv = Fi(-(wv2i * qh)).real
    ph = F(pv + pw)
    del pw, pv
    assert rel(m.ph, ph) < 1e-12
    u, v = Fi(-il * ph).real, Fi(ik * ph).real
    del ph
    q = Fi(qh).real
    jq = ik * F(u * q) + il * F(v * q)
    jq[0, 0] = 0
    e = rel(m.jacobian_psi_q(), jq)
    print("8192^2 jacobian_psi_q %.2e" % e)
    assert e < 1e-12
    del jq
    jp = F(u * phix + v * phiy)
    jp[0, 0] = 0
    e = rel(m.jacobian_psi_phi(), jp)
    print("8192^2 jacobian_psi_phi %.2e" % e)
    assert e < 1e-12
    del jp, u, v, phix, phiy
    q_psi = q - Fi(qwh).real
    e = rel(m._ctx.refraction(), F(phi0 * q_psi))
    print("8192^2 refraction %.2e" % e)
    assert e < 1e-12


# ---- the REAL reference at 2048^2 (golden g11: projections, sub-samples, norms; make_golden.py g11) -------------------------
def seeded_projections(field, seed, n=256):
    rng = np.random.default_rng(seed)
    ny, nx = field.shape
    out = np.empty(n, field.dtype)
    for i in range(n):
        sy = rng.integers(0, 2, ny) * 2.0 - 1.0
        sx = rng.integers(0, 2, nx) * 2.0 - 1.0
        out[i] = sy @ field @ sx
    return out


def l2_error_estimate(field, ref_proj, ref_norm, seed):
    """random-sign projections of the error have standard deviation = its l2 norm: rms |delta proj| / ||ref||"""
    d = seeded_projections(field, seed) - ref_proj
    return float(np.sqrt(np.mean(np.abs(d) ** 2)) / ref_norm)


def test_config2_against_the_reference_itself_at_2048(golden):
    """BASELINE config 2 after two steps against numbers produced by RUNNING THE REFERENCE at 2048^2 (golden g11): 256 random
    projections of q and of qh (their error estimates the relative l2 error of the whole field), a 64x64 sub-sample, Ke."""
    import niwqg_amd
    import bench
    g = golden("g11_at_size_2048.npz")
    nx = 2048
    kw = bench.c3_kwargs(nx, "qg")
    assert np.allclose([kw["dt"], kw["nu4"], kw["L"], kw["U"]], g["qg_params"], rtol=1e-15)
    m = niwqg_amd.QGModel.Model(**kw)
    m.set_q(1e-5 * np.random.default_rng(0).standard_normal((nx, nx)))
    steps(m, 2)
    norm = float(g["qg_q_norm"])
    e = l2_error_estimate(m.q, g["qg_q_proj"], norm, 101)
    es = rel(m.q[::32, ::32], g["qg_q_sub"])
    print("config 2 vs the reference at 2048^2: l2 estimate %.2e, sub-sample %.2e" % (e, es))
    assert e < 2e-11 and es < 2e-11            # the |c dt| ~ 1 shell of the contour-mean planes
    assert abs(m.Ke - float(g["qg_Ke"])) < 1e-10 * abs(float(g["qg_Ke"]))


def test_rough_coupled_step_against_the_reference_itself_at_2048(golden):
    import niwqg_amd
    g = golden("g11_at_size_2048.npz")
    nx = 2048
    m = niwqg_amd.CoupledModel.Model(**rough_kwargs(nx))
    rng = np.random.default_rng(11)
    m.set_q(1e-5 * rng.standard_normal((nx, nx)))
    m.set_phi(0.05 * (rng.standard_normal((nx, nx)) + 1j * rng.standard_normal((nx, nx))))
    steps(m, 1)
    eq = l2_error_estimate(m.q, g["cpl_q_proj"], float(g["cpl_norms"][0]), 201)
    ep = l2_error_estimate(m.phi, g["cpl_phi_proj"], float(g["cpl_norms"][1]), 202)
    print("rough Coupled 2048^2 vs the reference: l2 estimates q %.2e phi %.2e" % (eq, ep))
    assert eq < 1e-11 and ep < 1e-11
    assert rel(m.q[::32, ::32], g["cpl_q_sub"]) < 1e-11 and rel(m.phi[::32, ::32], g["cpl_phi_sub"]) < 1e-11
    assert np.allclose([m.Ke, m.Pw, m.Kw], g["cpl_budgets"], rtol=1e-8)


def test_headline_grid_on_eight_slab_ranks_equals_the_single_context():
    """CoupledModel 4096^2 (BASELINE's headline grid) decomposed over 8 peer ranks on the one GPU, white-noise state with
    every dissipation term on: three steps through the class API give the single-context model's fields and budgets --
    which `test_rough_field_coupled_step_against_the_oracle[4096]` pins to the oracle."""
    import niwqg_amd
    rng = np.random.default_rng(7)
    kw = notebook_kwargs(4096, True)
    kw.update(nu4w=1e10, mu=1e-8, muw=2e-8)
    q0 = 1e-5 * rng.standard_normal((4096, 4096))
    phi0 = 0.05 * (rng.standard_normal((4096, 4096)) + 1j * rng.standard_normal((4096, 4096)))
    res = {}
    for tag, slab in (("one", False), ("slab8", 8)):
        m = niwqg_amd.CoupledModel.Model(slab=slab, nchunks=2, **kw)
        m.set_q(q0)
        m.set_phi(phi0)
        steps(m, 3)
        res[tag] = (m.q.copy(), m.phi.copy(), m.phih.copy(), [m.Ke, m.Pw, m.Kw])
        m._ctx.close()
        del m
    a, b = res["one"], res["slab8"]
    assert rel(b[0], a[0]) < 1e-13 and rel(b[1], a[1]) < 1e-13 and rel(b[2], a[2]) < 1e-13
    assert np.allclose(b[3], a[3], rtol=1e-11)


def test_config4_grid_on_eight_slab_ranks_equals_the_single_context_8192():
    """BASELINE config 4's grid and partition on a FULL spectrum: CoupledModel 8192^2, white-noise q and phi, every
    dissipation term on, one step on 8 peer ranks with 4 row chunks against the single context.  This is what runs the
    slab instantiations of the even/odd row kernels (k_x_products_eo<8192,0,true>, k_x_wavepv_eo<8192,true>) on
    high-index twiddles, the filter band and the Nyquist lines; the single-context kernels are pinned to the reference's
    formulas by test_row_kernels_8192_on_a_full_spectrum_against_numpy.  The first model is freed before the second is built."""
    import niwqg_amd
    nx = 8192
    rng = np.random.default_rng(17)
    kw = rough_kwargs(nx)
    q0 = 1e-5 * rng.standard_normal((nx, nx))
    phi0 = 0.05 * (rng.standard_normal((nx, nx)) + 1j * rng.standard_normal((nx, nx)))
    res = {}
    # (two ranks: 2049 half-spectrum columns per rank, a width that is not a power of two and the largest the block-index
    # division of the row kernels ever sees -- MArr::magic, checked column by column in nq_create)
    for tag, slab, nch in (("one", False, 4), ("slab8", 8, 4), ("slab2", 2, 1)):
        m = niwqg_amd.CoupledModel.Model(slab=slab, nchunks=nch, **kw)
        m.set_q(q0)
        m.set_phi(phi0)
        steps(m, 1)
        res[tag] = (m.q.copy(), m.phi.copy(), m.phih.copy(), m.qh.copy(), [m.Ke, m.Pw, m.Kw])
        m._ctx.close()
        del m
    a = res["one"]
    for tag in ("slab8", "slab2"):
        b = res[tag]
        errs = [rel(b[i], a[i]) for i in range(4)]
        print("8192^2 white noise, %s vs one context: q %.2e phi %.2e phih %.2e qh %.2e" % ((tag,) + tuple(errs)))
        assert max(errs) < 1e-13, tag
        assert np.allclose(b[4], a[4], rtol=1e-11), tag
    # the state compared is rough: the filter band and the Nyquist row carry data
    k65 = int(0.65 * nx / 2) + 3
    assert np.abs(a[2][k65, k65]) > 0 and np.abs(a[3][nx // 2, 5]) > 0


def test_rough_coupled_ten_steps_against_the_reference_itself_at_2048(golden):
    """The REAL reference at 2048^2 over 5 and 10 steps (golden g12, make_golden.py g12: CoupledModel, the white-noise state
    and parameters of g11's coupled case): 256 random projections of q, phi, qh (full plane: the Nyquist-row passenger
    included) and phih, a 64x64 sub-sample, norms, budgets."""
    import niwqg_amd
    g = golden("g12_coupled_2048_10steps.npz")
    nx = 2048
    m = niwqg_amd.CoupledModel.Model(**rough_kwargs(nx))
    rng = np.random.default_rng(11)
    m.set_q(1e-5 * rng.standard_normal((nx, nx)))
    m.set_phi(0.05 * (rng.standard_normal((nx, nx)) + 1j * rng.standard_normal((nx, nx))))
    for n in (5, 10):
        steps(m, n)
        t = "s%d_" % n
        nq, nphi = float(g[t + "norms"][0]), float(g[t + "norms"][1])
        e = dict(q=l2_error_estimate(m.q, g[t + "q_proj"], nq, 301), phi=l2_error_estimate(m.phi, g[t + "phi_proj"], nphi, 302),
                 qh=l2_error_estimate(m.qh, g[t + "qh_proj"], nq * nx, 303),
                 phih=l2_error_estimate(m.phih, g[t + "phih_proj"], nphi * nx, 304),
                 q_sub=rel(m.q[::32, ::32], g[t + "q_sub"]), phi_sub=rel(m.phi[::32, ::32], g[t + "phi_sub"]))
        print("rough Coupled 2048^2 vs the reference after %d steps:" % n, {k: "%.2e" % v for k, v in e.items()})
        for k, v in e.items():
            assert v < 1e-10, (n, k, v)
        assert np.allclose([m.Ke, m.Pw, m.Kw], g[t + "budgets"], rtol=1e-8)


@pytest.mark.parametrize("nx", [1024, 2048])
def test_lamb_dipole_100_steps_against_the_reference_itself(golden, nx):
    """BASELINE.json's literal criterion at size (golden g16, make_golden.py g16): the REAL reference, CoupledModel, LambDipole q +
    uniform phi (ref examples/LambDipole.py:45-58, niwqg/InitialConditions.py:77-114), notebook parameters scaled to the grid,
    FILTER ON, after 50 and 100 steps (ref niwqg/Kernel.py:307-397).  The dipole's vorticity has a kink at r = R: its spectrum
    reaches the filter band from the first step and the nonlinear cascade keeps feeding it, so the size-specific transform plans
    (1024: 32 x 32 columns; 2048: 32 x 64) are compared over the whole horizon on a state that is not band-limited.  Compared:
    256 seeded random projections of q, phi, qh (full plane), phih (their error estimates the relative l2 error of the whole
    field), a 64 x 64 sub-sample, the low 32 x 32 corner of both spectra and a strip inside the filter band element by element,
    the in-step budgets.  Bar: BASELINE's relative RMS < 1e-10; achieved figures printed."""
    import niwqg_amd
    from niwqg_amd import InitialConditions as ic
    import os
    name = "g16_coupled_lamb_%d_100steps.npz" % nx
    if not os.path.exists(os.path.join(os.path.dirname(__file__), "golden", name)):
        pytest.skip("golden %s not generated (make_golden.py g16, G16_SIZES=%d)" % (name, nx))
    g = golden(name)
    kw = notebook_kwargs(nx, True)
    assert np.allclose([nx, kw["dt"], kw["nu4"], kw["nu"], kw["nuw"], kw["U"]], g["params"], rtol=1e-15)
    m = niwqg_amd.CoupledModel.Model(**kw)
    q0 = ic.LambDipole(m, U=0.1, R=2 * np.pi / K0)
    assert np.array_equal(q0[::nx // 64, ::nx // 64], g["q0_sub"]) and np.linalg.norm(q0) == float(g["q0_norm"])   # same input, bit for bit
    m.set_q(q0)
    m.set_phi((np.ones((nx, nx)) + 1j) * 0.2 / np.sqrt(2))
    st = nx // 64
    b = int(0.36 * nx)
    for n in (50, 100):
        steps(m, n)
        t = "s%d_" % n
        nq, nphi, nqh, nphih = [float(v) for v in g[t + "norms"]]
        qh, phih = m.qh, m.phih
        e = dict(q=l2_error_estimate(m.q, g[t + "q_proj"], nq, 401), phi=l2_error_estimate(m.phi, g[t + "phi_proj"], nphi, 402),
                 qh=l2_error_estimate(qh, g[t + "qh_proj"], nqh, 403), phih=l2_error_estimate(phih, g[t + "phih_proj"], nphih, 404),
                 q_sub=rel(m.q[::st, ::st], g[t + "q_sub"]), phi_sub=rel(m.phi[::st, ::st], g[t + "phi_sub"]),
                 qh_low=rel(qh[:32, :32], g[t + "qh_low"]), phih_low=rel(phih[:32, :32], g[t + "phih_low"]))
        # across the filter's cut-off (k index 0.65 nx / 2, l small) the spectra fall from the cascade's level to nothing, and deep
        # inside the band they are 1e-35 of the peak: compare both strips against the size of the field they belong to
        c0 = int(0.65 * nx / 2) - 32
        band = dict(qh_edge=np.linalg.norm(qh[:8, c0:c0 + 64] - g[t + "qh_edge"]) / nqh,
                    phih_edge=np.linalg.norm(phih[:8, c0:c0 + 64] - g[t + "phih_edge"]) / nphih,
                    # just below the cut-off ON ITS OWN SCALE: these modes are 1e-13 (1024^2) ... 1e-17 (2048^2) of the spectrum's norm,
                    # at or barely above the rounding floor of a field whose peak they are not -- printed for the record, not asserted
                    qh_edge_own=rel(qh[:8, c0:c0 + 24], g[t + "qh_edge"][:, :24]),
                    qh_band=np.linalg.norm(qh[b:b + 8, b:b + 64] - g[t + "qh_band"]) / nqh,
                    phih_band=np.linalg.norm(phih[b:b + 8, b:b + 64] - g[t + "phih_band"]) / nphih)
        print("LambDipole Coupled %d^2, filter on, vs the reference after %d steps:" % (nx, n), {k: "%.2e" % v for k, v in {**e, **band}.items()})
        for k, v in {**e, **band}.items():
            if k != "qh_edge_own":            # (informational: at 2048^2 these modes sit AT the rounding floor of the spectrum, 5e-18 of its norm)
                assert v < 1e-10, (n, k, v)
        assert np.allclose([m.Ke, m.Pw, m.Kw], g[t + "budgets"], rtol=1e-8)


@pytest.mark.parametrize("family", ["unc", "ybj", "qgc"])
def test_other_families_100_steps_against_the_reference_itself_at_1024(golden, family):
    """The three model families golden g16 does not cover, at 1024^2 against numbers produced by RUNNING THE REFERENCE (golden
    g18, make_golden.py g18): UnCoupledModel on BASELINE config 5's member 3 (ref niwqg/UnCoupledModel.py:54-64; quirk Q1 acts:
    tdiags = 10, so phix, phiy are refreshed by the tenth steps' diagnostics ticks only, niwqg/Kernel.py:608-611), YBJModel (ref
    niwqg/YBJModel.py:52-87) and QGModel with beta and its passive scalar (ref niwqg/QGModel.py:328-407, :483-495), after 50 and
    100 steps: 256 seeded random projections of every state field (their error estimates the relative l2 error of the whole
    field), a 64 x 64 sub-sample, the budget / variance scalars and every diagnostics series (ten ticks).  Bar: BASELINE's
    relative RMS < 1e-10; achieved figures printed."""
    import niwqg_amd
    from niwqg_amd import InitialConditions as ic
    name = "g18_families_1024_100steps.npz"
    if not os.path.exists(os.path.join(os.path.dirname(__file__), "golden", name)):
        pytest.skip("golden %s not generated (make_golden.py g18)" % name)
    g = golden(name)
    nx = 1024
    U0 = 0.1
    TE = 1.0 / (U0 * K0)
    if family == "unc":
        from niwqg_amd import ensemble
        m = ensemble.config5_member(3, nx=nx, tdiags=10)
        fields, scalars = ["q", "phi", "qh", "phih", "phix", "phiy"], ["Ke", "Pw", "Kw"]
    elif family == "ybj":
        kw = notebook_kwargs(nx, True)
        kw.update(tdiags=10, nu4w=3e9 * (64.0 / nx) ** 4, muw=1e-7)
        m = niwqg_amd.YBJModel.Model(**kw)
        q0 = ic.LambDipole(m, U=U0, R=2 * np.pi / K0)
        assert np.array_equal(q0[::nx // 64, ::nx // 64], g["ybj_q0_sub"]) and np.linalg.norm(q0) == float(g["ybj_q0_norm"])
        m.set_q(q0)
        m.set_phi(0.2 * ic.WavePacket(m, k=2 * K0, l=K0, R=L / 6, x0=L / 2, y0=L / 2) + 0.05)
        fields, scalars = ["phi", "phih", "phix", "phiy"], ["Ke", "Pw", "Kw"]
    else:
        dt = 0.05 * TE * 128 / nx / 2
        m = niwqg_amd.QGModel.Model(L=L, nx=nx, tmax=1e30, dt=dt, twrite=10 ** 9, nu4=7.5e8 * (256.0 / nx) ** 4, nu=5.0,
                                    mu=1e-8, use_filter=True, U=-U0, tdiags=10, beta=2e-11, passive_scalar=True,
                                    nu4c=3e9 * (64.0 / nx) ** 4, nuc=2.0, muc=1e-8)
        assert np.allclose([dt, m.nu4, m.nu4c], g["qgc_params"], rtol=1e-15)
        m.set_q(ic.LambDipole(m, U=U0, R=2 * np.pi / K0))
        m.set_c(np.sin(2 * np.pi * 3 * m.x / L) * np.cos(2 * np.pi * 2 * m.y / L) + 0.3)
        fields, scalars = ["q", "c", "qh", "ch"], ["Ke", "cvar", "C2", "gradC2"]
    for n in (50, 100):
        steps(m, n)
        t = "%s_s%d_" % (family, n)
        norms = g[t + "norms"]
        e = {}
        for i, f in enumerate(fields):
            a = getattr(m, f)
            e[f] = l2_error_estimate(a, g[t + f + "_proj"], float(norms[i]), 500 + i)
            e[f + "_sub"] = rel(a[::nx // 64, ::a.shape[1] // 64], g[t + f + "_sub"])
        print("%s 1024^2 vs the reference after %d steps:" % (family, n), {k: "%.2e" % v for k, v in e.items()})
        for k, v in e.items():
            assert v < 1e-10, (family, n, k, v)
        # (C2, gradC2 are the values the last diagnostics tick left, ref niwqg/QGModel.py:724-737: compared as they stand)
        assert np.allclose([getattr(m, k) for k in scalars], g[t + "scalars"], rtol=1e-8), (family, n)
    for dname in m.diagnostics:
        ref = g["%s_diag_%s" % (family, dname)]
        got = np.asarray(m.diagnostics[dname]['value'], dtype=float)
        assert got.shape == ref.shape, (dname, got.shape, ref.shape)
        scale = np.abs(ref).max()
        floor = 1e-13 if dname in ("skew", "conc_niw") else 1e-300      # (vanishing moments of order-one fields: 7e-16 here)
        assert np.allclose(got, ref, rtol=1e-7, atol=1e-9 * scale + floor), (family, dname, np.abs(got - ref).max(), scale)


@pytest.mark.parametrize("nx", [1024] + ([2048] if os.environ.get("NQ_DEALIAS_2048") else []))
def test_dealias_roundoff_growth_at_size_is_the_oracles_own(nx):
    """Why the at-size fuzz draws only 10 steps under the 2/3 mask (tests/test_gpu_models.py,
    test_randomly_drawn_configurations_at_size_through_resolution_independence): with dealias=True (ref niwqg/Kernel.py:277-281),
    the size's dt (advective CFL ~0.4) and weak dissipation, differences at rounding level are AMPLIFIED every step by the
    reference's own arithmetic.  Round 3 showed it at 256^2 only (gpurun_out/dg256.log); here it is measured at 1024^2 and 2048^2:
      (1) device and oracle side by side, 20 steps: while their states still agree to 1e-9, the per-step growth factors of
          max |qh| outside the initial band agree to 1e-6 (the deterministic cascade is the same);
      (2) the oracle against ITSELF, the second copy started from q0 perturbed at 1e-16 relative: the difference grows by a
          factor r_o > 2 per step -- the instability is the oracle's, i.e. the reference's algorithm's;
      (3) the device-oracle difference grows at that same rate (within a factor 1.5 per step) and stays within 1e3 of the
          oracle-oracle difference at every step: the device adds nothing of its own.
    The 2048^2 case (three minutes of oracle time) runs with NQ_DEALIAS_2048=1; its log of round 4 is kept in
    profiles/r04_dealias_growth_at_size.txt (x18 per step in the oracle against itself, x18 device against oracle)."""
    import copy
    import niwqg_amd
    import test_gpu_models as T
    from test_oracle_golden import TE, U0, MZ, NB, F0
    nsteps = 20
    kw = dict(L=L, nx=nx, tmax=1e30, dt=0.025 * TE * 128 / nx, twrite=10 ** 9, tdiags=10 ** 9, use_filter=False, dealias=True, U=0.05,
              nu4=5e11 * (128.0 / nx) ** 4 * 0.35, nu=20.0, mu=0.0, m=0.5 * MZ, N=NB, f=F0, nuw=0.0,
              nu4w=0.035 * 5e11 * (128.0 / nx) ** 4, muw=2e-8)
    o = O.NIWQGOracle("uncoupled", coeff_chunk=8, workers=NW, **kw)
    o2 = copy.deepcopy(o)
    m = niwqg_amd.UnCoupledModel.Model(**kw)
    q1, phi1 = T._random_band_limited_state(o.grid.x, o.grid.y, 18106, True)
    for x in (m, o):
        x.set_q(q1)
        x.set_phi(phi1)
    o2.set_q(q1 * (1.0 + 1e-16 * np.random.default_rng(5).standard_normal(q1.shape)))
    o2.set_phi(phi1)
    band = np.zeros(nx, bool)
    band[np.r_[0:13, nx - 12:nx]] = True
    out = ~(band[:, None] & band[None, :])
    a, b, d_dev, d_orc, peak = [], [], [], [], []
    for n in range(nsteps):
        m._step_forward()
        o._step_forward()
        o2._step_forward()
        mq, oq = m.qh, o.qh
        a.append(float(np.abs(mq[out]).max()))
        b.append(float(np.abs(oq[out]).max()))
        peak.append(float(np.abs(oq).max()))
        d_dev.append(rel(mq, oq))
        d_orc.append(rel(o2.qh, oq))
        print("%d^2 step %2d  outside the band: device %.6e oracle %.6e   device-oracle %.1e   oracle-oracle(perturbed 1e-16) %.1e"
              % (nx, n + 1, a[-1], b[-1], d_dev[-1], d_orc[-1]), flush=True)
        if not (np.isfinite(a[-1]) and np.isfinite(b[-1])) or max(a[-1], b[-1]) > 1e100:
            break
    a, b, d_dev, d_orc, peak = (np.array(v) for v in (a, b, d_dev, d_orc, peak))
    ga, gb = a[1:] / a[:-1], b[1:] / b[:-1]
    clean = d_dev < 1e-9
    assert clean.sum() >= 4, d_dev
    # the maxima themselves to 1e-6 (plus a rounding floor: the spectrum's peak is 1e8 .. 1e12 times larger), hence their ratios
    assert (np.abs(a - b)[clean] <= 1e-6 * b[clean] + 1e-15 * peak[clean]).all(), (a, b)
    both = clean[1:] & clean[:-1] & (b[:-1] > 1e-11 * peak[:-1])
    assert both.sum() >= 2, (b, peak)
    assert np.allclose(ga[both], gb[both], rtol=1e-5), (ga, gb)
    # amplification per step, over the steps where the differences are above rounding and below saturation
    def rate(d):
        w = (d > 1e-14) & (d < 1e-2)
        i = np.nonzero(w)[0]
        assert len(i) >= 4, d
        return (d[i[-1]] / d[i[0]]) ** (1.0 / (i[-1] - i[0]))
    r_o, r_d = rate(d_orc), rate(d_dev)
    print("%d^2: per-step amplification of a rounding-level difference: oracle vs itself x%.2f, device vs oracle x%.2f" % (nx, r_o, r_d))
    assert r_o > 2.0, r_o
    assert 1 / 1.5 < r_d / r_o < 1.5, (r_d, r_o)
    live = np.isfinite(d_dev) & np.isfinite(d_orc) & (d_orc < 0.1)       # (once the two oracles differ by O(1) there is nothing left to compare)
    assert (d_dev[live] <= 1e3 * d_orc[live] + 1e-13).all(), (d_dev, d_orc)


def test_qg_passive_scalar_row_kernel_8192_on_a_full_spectrum_against_numpy():
    """QGModel with its passive scalar at 8192^2 (k_x_products_eo<8192, MODE_QGC>: q and c as ONE packed transform per parity, c
    rescaled per row by a power of two, the products u c, v c leaving as a second packed pair): white-noise q (1e-5) and c (O(1)),
    jacobian_psi_q and jacobian_psi_c (ref QGModel.py:469-495) against the reference's formulas evaluated with pocketfft."""
    import scipy.fft
    import niwqg_amd
    import bench
    nx = 8192
    kw = bench.c3_kwargs(nx, "qg")
    m = niwqg_amd.QGModel.Model(passive_scalar=True, nu4c=kw["nu4"], **kw)
    rng = np.random.default_rng(31)
    q0 = 1e-5 * rng.standard_normal((nx, nx))
    c0 = 1.0 + rng.standard_normal((nx, nx))
    m.set_q(q0)
    m.set_c(c0)
    kk, ll = np.asarray(m.kk).ravel(), np.asarray(m.ll).ravel()
    ik, il = 1j * kk[None, :], 1j * ll[:, None]
    wv2 = kk[None, :] ** 2 + ll[:, None] ** 2
    wv2i = np.zeros_like(wv2)
    wv2i[wv2 != 0] = 1.0 / wv2[wv2 != 0]

    def F(a):
        return scipy.fft.rfft2(a, workers=NW)

    def Fi(a):
        return scipy.fft.irfft2(a, s=(nx, nx), workers=NW)

    ph = -wv2i * F(q0)
    u, v = Fi(-il * ph), Fi(ik * ph)
    jq = ik * F(u * q0) + il * F(v * q0)
    e = rel(m.jacobian_psi_q(), jq)
    print("8192^2 QG jacobian_psi_q %.2e" % e)
    assert e < 1e-12
    del jq
    jc = ik * F(u * c0) + il * F(v * c0)
    e = rel(m.jacobian_psi_c(), jc)
    print("8192^2 QG jacobian_psi_c %.2e" % e)
    assert e < 1e-12


def test_uncoupled_and_qg_row_kernels_8192_on_a_full_spectrum_against_numpy():
    """The other instantiations of the even/odd row kernel at 8192^2 on white noise, against the reference's formulas evaluated with
    pocketfft: UnCoupledModel (k_x_products_eo<8192, MODE_UNCOUPLED>: q alone in its transform, phix / phiy from the rows as last
    refreshed, phi's row read separately; ref Kernel.py:457-486, :332, UnCoupledModel.py:54-64) and QGModel without its passive scalar
    (MODE_QG; ref QGModel.py:469-481)."""
    import scipy.fft
    import niwqg_amd
    import bench
    nx = 8192
    rng = np.random.default_rng(41)
    q0 = 1e-5 * rng.standard_normal((nx, nx))
    # ---- QGModel, rfft2 semantics
    m = niwqg_amd.QGModel.Model(**bench.c3_kwargs(nx, "qg"))
    m.set_q(q0)
    kk, ll = np.asarray(m.kk).ravel(), np.asarray(m.ll).ravel()
    ik, il = 1j * kk[None, :], 1j * ll[:, None]
    wv2 = kk[None, :] ** 2 + ll[:, None] ** 2
    wv2i = np.zeros_like(wv2)
    wv2i[wv2 != 0] = 1.0 / wv2[wv2 != 0]
    ph = -wv2i * scipy.fft.rfft2(q0, workers=NW)
    u = scipy.fft.irfft2(-il * ph, s=(nx, nx), workers=NW)
    v = scipy.fft.irfft2(ik * ph, s=(nx, nx), workers=NW)
    jq = ik * scipy.fft.rfft2(u * q0, workers=NW) + il * scipy.fft.rfft2(v * q0, workers=NW)
    e = rel(m.jacobian_psi_q(), jq)
    print("8192^2 QGModel (no scalar) jacobian_psi_q %.2e" % e)
    assert e < 1e-12
    m._ctx.close()
    del m, jq, ph, u, v, wv2, wv2i
    # ---- UnCoupledModel, c2c semantics with .real projections
    kw = rough_kwargs(nx)
    m = niwqg_amd.UnCoupledModel.Model(**kw)
    phi0 = 0.05 * (rng.standard_normal((nx, nx)) + 1j * rng.standard_normal((nx, nx)))
    m.set_q(q0)
    m.set_phi(phi0)            # refreshes phix, phiy (Kernel.py:548-551 -> _calc_pe_niw)

    def F(a):
        return scipy.fft.fft2(a, workers=NW)

    def Fi(a):
        return scipy.fft.ifft2(a, workers=NW)

    kk = np.asarray(m.kk).ravel()
    ik, il = 1j * kk[None, :], 1j * np.asarray(m.ll).ravel()[:, None]
    wv2 = kk[None, :] ** 2 + np.asarray(m.ll).ravel()[:, None] ** 2
    wv2i = np.zeros_like(wv2)
    wv2i[wv2 != 0] = 1.0 / wv2[wv2 != 0]
    ph = F(Fi(-(wv2i * F(q0))).real)
    del wv2, wv2i
    u, v = Fi(-il * ph).real, Fi(ik * ph).real
    del ph
    jq = ik * F(u * q0) + il * F(v * q0)
    jq[0, 0] = 0
    e = rel(m.jacobian_psi_q(), jq)
    print("8192^2 UnCoupledModel jacobian_psi_q %.2e" % e)
    assert e < 1e-12
    del jq
    phih = F(phi0)
    jp = F(u * Fi(ik * phih) + v * Fi(il * phih))
    jp[0, 0] = 0
    del phih, u, v
    e = rel(m.jacobian_psi_phi(), jp)
    print("8192^2 UnCoupledModel jacobian_psi_phi %.2e" % e)
    assert e < 1e-12
    del jp
    e = rel(m._ctx.refraction(), F(phi0 * q0))         # q_psi = q without wave feedback (Kernel.py:492-501)
    print("8192^2 UnCoupledModel refraction %.2e" % e)
    assert e < 1e-12
