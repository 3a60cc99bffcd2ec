"""Initial conditions with the reference's call signatures (ref: niwqg/InitialConditions.py).

``model`` only needs ``x, y, nx, wv, wv2, fft, ifft, spec_var``.  LambDipole is vectorised (the
reference loops over N^2 points in Python, InitialConditions.py:102-107) with identical values.
"""
import numpy as np
from scipy import special


def _random_phase_psi(model, ckappa, E):
    phase = np.random.rand(*model.wv2.shape) * 2 * np.pi
    ph = ckappa * np.cos(phase) + 1j * ckappa * np.sin(phase)
    ph = model.fft(model.ifft(ph).real)
    Eaux = 0.5 * model.spec_var(model.wv * ph)
    pih = np.sqrt(E / Eaux) * ph
    return model.ifft(-model.wv2 * pih).real


def McWilliams1984(model, k0=6, E=0.5):
    """Random vorticity with McWilliams' (1984) red spectrum.  ref: InitialConditions.py:4-41"""
    ckappa = np.zeros_like(model.wv2)
    fk = model.wv != 0
    ckappa[fk] = np.sqrt(model.wv2[fk] * (1. + (model.wv2[fk] / k0 ** 2) ** 2)) ** -1
    return _random_phase_psi(model, ckappa, E)


def Danioux2015(model, k0=6, E=0.5):
    """Single-wavenumber-band random vorticity.  ref: InitialConditions.py:43-75"""
    ckappa = np.zeros_like(model.wv2)
    fk = model.wv != 0
    ckappa[fk] = np.sqrt(model.wv[fk] * np.exp(-(model.wv2[fk] / k0 ** 2)))
    return _random_phase_psi(model, ckappa, E)


def LambDipole(model, U=.01, R=1.):
    """Lamb dipole of radius R translating at U.  ref: InitialConditions.py:77-114"""
    N = model.nx
    x, y = model.x, model.y
    x0, y0 = x[N // 2, N // 2], y[N // 2, N // 2]
    r = np.sqrt((x - x0) ** 2 + (y - y0) ** 2)
    s = np.zeros_like(r)
    away = r != 0.
    s[away] = (y[away] - y0) / r[away]
    lam = 3.8317 / R
    C = -(2. * U * lam) / special.j0(lam * R)
    q = np.zeros_like(r)
    inside = r <= R
    q[inside] = C * special.j1(lam * r[inside]) * s[inside]
    return q


def WavePacket(model, k=10, l=0, R=1, x0=0., y0=0.):
    """Gaussian wave packet.  ref: InitialConditions.py:117-145"""
    x, y = model.x, model.y
    r = np.sqrt((x - x0) ** 2 + (y - y0) ** 2)
    phi = np.exp(1j * (k * (x - x0) + l * (y - y0)))
    phi *= np.exp(-((r / R) ** 2))
    return phi


def PlaneWave(model, k=10, l=0, phase=0.):
    """Plane wave (note: ``phase`` is added outside the imaginary unit, as in the reference).
    ref: InitialConditions.py:147-169"""
    return np.exp(1j * (k * model.x + l * model.y) + phase)
