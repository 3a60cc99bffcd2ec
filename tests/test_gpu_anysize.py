"""GPU parity of the any-size path (niwqg_amd/_anysize.py, csrc/nq_anysize.hpp): grids the fused kernels have no plan for.

The reference takes any nx (ref niwqg/Kernel.py:100-103; numpy.fft transforms any length, :562-566).  Expected values: numpy.fft
for the transforms, golden g17 (the REAL reference on 96^2 and 192^2 grids, make_golden.py g17) for trajectories, the
reference-pinned oracle for everything else (the oracle is plain numpy: any nx).
"""
import numpy as np
import pytest

from oracle import niwqg_oracle as O
from test_oracle_golden import notebook_kwargs, rel, L, K0, U0, TE
import test_gpu_models as T

pytestmark = pytest.mark.gpu


def models():
    import niwqg_amd
    return niwqg_amd


@pytest.mark.parametrize("nx", [4, 6, 10, 16, 30, 32, 48, 96, 100, 192, 250, 320, 384, 640, 1000, 1536, 2560, 3000, 3072])
def test_fft_seam_of_any_length_against_numpy(nx):
    """Kernel.fft / ifft (numpy.fft.fft2 / ifft2 semantics, ref niwqg/Kernel.py:562-566) and QGModel.fft / ifft (rfft2 / irfft2,
    QGModel.py:551-552) on the device: Bluestein for lengths with factors 3, 5, 7 ..., powers of two below the fused range, the
    largest in-register work rows (3000 -> 8192); the radix-3 / radix-5 split for 3 m and 5 m with m a power of two >= 64 (192, 320,
    384, 640, 1536, 2560, 3072)."""
    rng = np.random.default_rng(nx)
    m = models().UnCoupledModel.Model(nx=nx)
    assert getattr(m, "_any_size", False) and type(m).__name__ == "Model"
    a = rng.standard_normal((nx, nx)) + 1j * rng.standard_normal((nx, nx))
    f = m.fft(a)
    e1, e2 = rel(f, np.fft.fft2(a)), rel(m.ifft(f), a)
    qg = models().QGModel.Model(nx=nx)
    r = rng.standard_normal((nx, nx))
    h = qg.fft(r)
    assert h.shape == (nx, nx // 2 + 1)
    e3 = rel(h, np.fft.rfft2(r))
    # irfft2 semantics: whatever sits in the imaginary parts of the self-mirrored columns is ignored like numpy ignores it
    g = np.fft.rfft2(r) * (1 + 0.3j)
    e4 = rel(qg.ifft(g), np.fft.irfft2(g, s=(nx, nx)))
    print("nx %d: fft2 %.1e  ifft2(fft2) %.1e  rfft2 %.1e  irfft2 %.1e" % (nx, e1, e2, e3, e4))
    assert max(e1, e2, e3, e4) < 5e-15 * max(1.0, np.log2(nx) / 4)


@pytest.mark.parametrize("nx", [96, 192])
def test_reference_trajectories_on_grids_that_are_not_powers_of_two(golden, nx):
    """golden g17: the REAL reference, CoupledModel and QGModel, LambDipole, filter on, after 1, 10 and 100 steps"""
    g = golden("g17_non_power_of_two.npz")
    M = models()
    m = M.CoupledModel.Model(**notebook_kwargs(nx, True))
    q0 = g["c%d_q0" % nx]
    m.set_q(q0)
    m.set_phi((np.ones((nx, nx)) + 1j) * (2 * U0) / np.sqrt(2))
    for n in (1, 10, 100):
        T.steps(m, n)
        t = "c%d_s%d_" % (nx, n)
        e = {k: rel(getattr(m, k), g[t + k]) for k in ("q", "phi", "qh", "phih")}
        print("CoupledModel %d^2 vs the reference after %d steps:" % (nx, n), {k: "%.1e" % v for k, v in e.items()})
        assert max(e.values()) < 1e-10, (n, e)
        assert np.allclose([m.Ke, m.Pw, m.Kw], g[t + "budgets"], rtol=1e-9)
    dt, nu4 = [float(v) for v in g["qg%d_params" % nx]]
    qg = M.QGModel.Model(L=L, nx=nx, tmax=1e30, dt=dt, twrite=10 ** 9, nu4=nu4, use_filter=True, U=-U0, tdiags=10 ** 9)
    qg.set_q(q0)
    for n in (1, 10, 100):
        T.steps(qg, n)
        t = "qg%d_s%d_" % (nx, n)
        e = {k: rel(getattr(qg, k), g[t + k]) for k in ("q", "qh")}
        print("QGModel %d^2 vs the reference after %d steps:" % (nx, n), {k: "%.1e" % v for k, v in e.items()})
        assert max(e.values()) < 1e-10, (n, e)
        assert abs(qg.Ke - float(g[t + "Ke"])) <= 1e-9 * abs(float(g[t + "Ke"]))


@pytest.mark.parametrize("seed,nx", [(s, n) for s in range(6) for n in (96, 100)] + [(s, 192) for s in range(6, 10)] + [(s, 384) for s in (10, 11)]
                         + [(12, 16), (13, 32), (14, 48), (15, 16), (16, 32), (17, 12)])
def test_randomly_drawn_configurations_on_any_grid_against_the_oracle(seed, nx):
    """The option fuzz of test_gpu_models.py (model class, filter / 2-3 mask / none, mean flow, every dissipation term, beta,
    passive scalar, tick cadence; fields, spectra, budgets, every diagnostics series) on grids with factors 3 and 5 and on powers
    of two below the fused range."""
    T.random_configuration_against_the_oracle(seed, nx_force=nx)


@pytest.mark.parametrize("seed,nx", [(s, n) for s in range(6) for n in (96,)] + [(6, 100), (7, 48), (8, 192)])
def test_randomly_drawn_call_sequences_on_any_grid_against_the_oracle(seed, nx):
    """The class surface as a state machine (set_q / set_phi in either order, steps with status lines, the Jacobians, energies, CFL,
    attribute reads) on the any-size path"""
    T.random_call_sequence_against_the_oracle(seed, nx_force=nx)


def test_the_references_own_assertions_on_a_96_grid():
    """niwqg/tests/test_fft.py, test_advection.py, test_diffusion.py (ref) with nx = 96 instead of the default 128"""
    M = models()
    nx = 96
    m = M.CoupledModel.Model(use_filter=False, nx=nx)
    rng = np.random.default_rng(3)
    q = rng.standard_normal((nx, nx))
    assert np.allclose(q, m.ifft(m.fft(q)).real, rtol=1e-15)
    qh = m.fft(q)
    assert abs(m.spec_var(qh) - q.var()) / q.var() < 1e-13
    # advection of a slanted plane wave vanishes (test_advection.py:12-32)
    k, l = 5 * 2 * np.pi / m.L, 9 * 2 * np.pi / m.L
    p = np.sin(k * m.x + l * m.y)
    m.set_q(-(k ** 2 + l ** 2) * p)
    m.set_phi(p + 0j)
    m._invert()
    assert np.abs(m.ifft(m.jacobian_psi_q())).std() < 1e-12           # (the reference's own bound, absolute)
    assert np.abs(m.ifft(m.jacobian_phic_phi())).std() < 1e-12
    assert np.abs(m.ifft(m.jacobian_psi_phi())).std() < 1e-12
    # linear decay is exact (test_diffusion.py:12-27)
    m = M.CoupledModel.Model(use_filter=False, nx=nx, nu4=1e14, nu=0, dt=1000., tdiags=10 ** 9, twrite=10 ** 9)
    m.tmax = 10 * m.dt
    qi = np.sin(k * m.x + l * m.y)
    m.set_q(qi)
    m.set_phi(np.zeros((nx, nx), complex))
    m.run()
    assert np.allclose(m.qh, m.fft(qi) * np.exp(-m.nu4 * m.wv4 * m.tmax), rtol=1e-12, atol=1e-12 * nx * nx)


BIG = bool(__import__("os").environ.get("NQ_ANYSIZE_BIG"))       # the 16384^2 cases: two minutes and 30 GB of host memory (round-4 log in profiles/)


@pytest.mark.parametrize("nx", [5000, 6144] + ([16384] if BIG else []))
def test_fft_seam_with_four_step_work_rows_against_pocketfft(nx):
    """Transform lengths whose work rows are 16384 points long -- Bluestein for 4096 < n <= 8192 (5000 = 2^3 5^4, 6144 = 3 2^11) and
    16384 itself without a chirp -- run as a four-step 128 x 128 transform on the 128-point row engine (csrc/nq_anysize.hpp)."""
    import scipy.fft
    import os
    nw = max(1, min(16, (os.cpu_count() or 2) - 1))
    rng = np.random.default_rng(nx)
    m = models().QGModel.Model(nx=nx)
    assert getattr(m, "_any_size", False)
    r = rng.standard_normal((nx, nx))
    h = m.fft(r)
    ref = scipy.fft.rfft2(r, workers=nw)
    e1 = rel(h, ref)
    e2 = rel(m.ifft(ref), r)
    print("nx %d: rfft2 %.1e  irfft2 %.1e" % (nx, e1, e2))
    assert max(e1, e2) < 5e-15


@pytest.mark.skipif(not BIG, reason="NQ_ANYSIZE_BIG=1 runs the 16384^2 cases")
def test_qgmodel_16384_through_resolution_independence():
    """QGModel at 16384^2 on the any-size path (the reference cannot construct this size; a fused plan does not exist): a
    band-limited state stepped 5 times with the size's own dt and hyperviscosity against the reference-pinned oracle at 128^2
    with the same coefficients -- the pseudo-spectral step is exact at any resolution that holds the band."""
    nx, nsteps = 16384, 5
    kw = dict(L=L, nx=128, tmax=1e30, dt=0.05 * TE * 128 / nx, twrite=10 ** 9, tdiags=10 ** 9, use_filter=True, U=-U0,
              nu4=5e11 * (128.0 / nx) ** 4, nu=20.0, mu=1e-8, beta=2e-11)
    o = O.QGOracle(**kw)
    q0, _ = T._random_band_limited_state(o.grid.x, o.grid.y, 777, False)
    o.set_q(q0)
    for _ in range(nsteps):
        o._step_forward()
    m = models().QGModel.Model(**dict(kw, nx=nx))
    q1, _ = T._random_band_limited_state(m.x, m.y, 777, False)
    m.set_q(q1)
    del q1
    T.steps(m, nsteps)
    ref = T._low_modes_half(o.qh, o.kk, o.ll, o.grid.x.ravel()[0], o.grid.y.ravel()[0], 128)
    got = T._low_modes_half(m.qh, np.asarray(m.kk).ravel(), np.asarray(m.ll).ravel(), m.x.ravel()[0], m.y.ravel()[0], nx)
    err = np.abs(got - ref).max() / np.abs(ref).max()
    print("QGModel 16384^2, %d steps, low modes against the oracle: %.1e" % (nsteps, err))
    assert err < 1e-10
    assert abs(m.Ke - o.Ke) <= 1e-9 * abs(o.Ke)


def test_grids_nobody_can_run_fail_loudly():
    M = models()
    for nx in (97, 5, 8194, 32768, 2):
        with pytest.raises(RuntimeError):
            M.CoupledModel.Model(nx=nx)
        with pytest.raises(RuntimeError):
            M.QGModel.Model(nx=nx)
    with pytest.raises(NotImplementedError):
        M.CoupledModel.Model(nx=96, slab=2)
