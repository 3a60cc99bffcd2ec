"""CPU proof that the reduced pipeline the HIP kernels implement (oracle/reduced_pipeline.py)
reproduces the faithful oracle (pinned to the reference by test_oracle_golden.py) to roundoff,
including full-plane q-hat with its Nyquist-row passenger, and the budget accumulators."""
import numpy as np
import pytest

from oracle import niwqg_oracle as O
from oracle import reduced_pipeline as R
from test_oracle_golden import notebook_kwargs, rel, L, K0, U0


def make_pair(kind, nx, use_filter, **extra):
    kw = notebook_kwargs(nx, use_filter)
    kw.update(extra)
    return O.NIWQGOracle(kind, **kw), R.ReducedNIWQG(kind, **kw)


@pytest.mark.parametrize("use_filter", [False, True])
def test_coupled_reduced_equals_faithful(use_filter):
    nx = 64
    a, b = make_pair("coupled", nx, use_filter)
    q0 = O.lamb_dipole(a.grid, U=U0, R=2 * np.pi / K0)
    phi0 = 0.2 * O.wave_packet(a.grid, k=3 * K0, l=K0, R=L / 6, x0=L / 2, y0=L / 2)
    for m in (a, b):
        m.set_q(q0)
        m.set_phi(phi0)
    assert rel(b.ph_full, a.ph) < 1e-14          # Q2: wave-free psi after set_q; set_phi
    for n in range(30):
        a._step_forward()
        b.step()
    assert rel(b.q, a.q) < 1e-12
    assert rel(b.phi, a.phi) < 1e-12
    assert rel(b.phih, a.phih) < 1e-12
    assert rel(b.qh_full, a.qh) < 1e-12           # includes Nyquist column and passenger row
    assert rel(b.ph_full, a.ph) < 1e-12
    assert rel(b.qwh_full, a.qwh) < 1e-12
    assert np.allclose([b.Ke, b.Pw, b.Kw], [a.Ke, a.Pw, a.Kw], rtol=1e-9)
    # 10 c2c-equivalents per stage in this model (the device shares phi, phix, phiy between the
    # inversion and the next stage: 8; the budget integrals need none) instead of the reference's 26
    assert b.n2d == 30 * 4 * 10 + 3


def test_passenger_is_needed_for_exact_qh_without_filter():
    a, b = make_pair("coupled", 64, False)
    q0 = O.lamb_dipole(a.grid, U=U0, R=2 * np.pi / K0)
    phi0 = (np.ones_like(q0) + 1j) * 2 * U0 / np.sqrt(2)
    for m in (a, b):
        m.set_q(q0)
        m.set_phi(phi0)
    for n in range(10):
        a._step_forward()
        b.step()
    assert rel(b.qh_full, a.qh) < 1e-12
    without = R.hs_to_full(b.qh)
    assert rel(without, a.qh) > 1e-12             # dropping it is visible in q-hat ...
    assert rel(np.fft.ifft2(without).real, a.q) < 1e-12   # ... but never in physical q


def test_rough_field_all_budget_terms():
    # white-noise fields put energy on every Nyquist line; nu4w, mu, muw switch on every budget term
    a, b = make_pair("coupled", 64, False, nu4w=1e10, mu=1e-8, muw=2e-8)
    rng = np.random.default_rng(1)
    q0 = 1e-5 * rng.standard_normal((64, 64))
    phi0 = 0.05 * (rng.standard_normal((64, 64)) + 1j * rng.standard_normal((64, 64)))
    for m in (a, b):
        m.set_q(q0)
        m.set_phi(phi0)
    for n in range(5):
        a._step_forward()
        b.step()
    assert rel(b.q, a.q) < 1e-11 and rel(b.phi, a.phi) < 1e-11
    assert np.allclose([b.Ke, b.Pw, b.Kw], [a.Ke, a.Pw, a.Kw], rtol=1e-8)


@pytest.mark.parametrize("tdiags", [1, 10 ** 9])
def test_uncoupled_reduced_with_stale_gradient(tdiags):
    a, b = make_pair("uncoupled", 64, True, tdiags=tdiags)
    q0 = O.lamb_dipole(a.grid, U=U0, R=2 * np.pi / K0)
    phi0 = 0.1 * O.wave_packet(a.grid, k=3 * K0, l=0, R=L / 6, x0=L / 2, y0=L / 2)
    for m in (a, b):
        m.set_q(q0)
        m.set_phi(phi0)
    for n in range(20):
        a._step_forward()
        b.step()
        if not ((b.tc - 1) % tdiags):     # the diagnostics tick is evaluated before tc advances
            b.refresh_grad_phi()
    assert rel(b.q, a.q) < 1e-12 and rel(b.phi, a.phi) < 1e-12
    assert np.allclose([b.Ke, b.Pw, b.Kw], [a.Ke, a.Pw, a.Kw], rtol=1e-9)


def test_qg_reduced_equals_faithful():
    kw = dict(L=L, nx=64, dt=0.1 * (1.0 / (U0 * K0)), nu4=7.5e8, nu=5.0, mu=1e-8, use_filter=True,
              U=-U0, beta=2e-11)
    a = O.QGOracle(tmax=1e30, twrite=10 ** 9, tdiags=10 ** 9, **kw)
    b = R.ReducedQG(**kw)
    q0 = O.lamb_dipole(a.grid, U=U0, R=2 * np.pi / K0)
    a.set_q(q0)
    b.set_q(q0)
    for n in range(20):
        a._step_forward()
        b.step()
    assert rel(b.qh, a.qh) < 1e-12 and rel(b.q, a.q) < 1e-12
    assert np.isclose(b.Ke, a.Ke, rtol=1e-10)


def test_two_thirds_mask_of_the_reference_is_not_mirror_symmetric():
    """Why dealias=True is a separate device path: the reference zeroes indices [N//3, 2N//3)
    (ref Kernel.py:277-281), which is not invariant under k -> -k for any power-of-two N, so the
    reference's q-hat stops being Hermitian and a single half-spectrum copy cannot represent it."""
    for n in (64, 128, 256, 4096):
        z = np.zeros(n, bool)
        z[n // 3:2 * n // 3] = True
        assert not np.array_equal(z, np.roll(z[::-1], 1))
