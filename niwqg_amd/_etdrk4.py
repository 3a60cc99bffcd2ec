"""The entries of the ETDRK4 planes that only numpy can reproduce.

The reference evaluates Qh, f0, fab and fc as the mean over 32 points of the unit circle around ``c dt`` (Kernel.py:420-433,
:443-454; QGModel.py:434-443, :456-465).  Where ``c dt`` lies close to MINUS one of those points, one term of the mean is the
removable singularity of ``(e^z - 1 - z - ...)/z^3`` evaluated by cancellation at ``|z| = distance``: what the reference then
holds is the rounding error of its libm's ``exp``, amplified by ``eps / distance^3``.  With ``U = 0`` the q operator is real, so
on every grid some ``c dt`` comes within ~1e-5 of ``-1`` and the reference's f0 there is off by O(1) -- reproducibly, because
numpy is deterministic, but out of reach of the device's own ``exp``.  The library therefore lists those entries
(``nq_coeff_near_contour``) and this module recomputes them on the host with the reference's numpy expression, operation for
operation, and hands them back (``nq_coeff_patch``).  A few hundred entries at 512^2, ~1e5 at 8192^2; everything else stays on
the device, where the two evaluations agree to ~1e-13.

The Kernel family keeps q on the half spectrum while the reference evolves the full plane with ``c(l, k)`` and ``c(-l, -k)``
side by side; the physical (Hermitian) part of its q then advances with the mean of ``F(l, k)`` and ``conj F(-l, -k)``, which
is what is patched in -- or, when the 2/3-rule mask keeps only one of the two modes (dual-copy contexts), that mode's own
coefficient.
"""
import os

import numpy as np

#: distance from a contour point below which an entry is recomputed here: eps / (32 delta^3) ~ 3e-14 is what is left outside
DELTA = 0.05

QG = 2      # _lib.QG (not imported: _lib imports this module)


def contour_tables(ch, dt):
    """Qh, f0, fab, fc for a 1-D array of ``c dt`` values: the reference's expression (Kernel.py:424-433) on an (n, 32)
    array instead of (ny, nx, 32) -- elementwise operations and a mean over the last, contiguous axis, so every entry gets the
    bits the reference computes for it."""
    M = 32
    rho = 1.
    r = rho * np.exp(2j * np.pi * ((np.arange(1., M + 1)) / M))
    LR = ch[..., np.newaxis] + r[np.newaxis, ...]
    LR2 = LR * LR
    LR3 = LR2 * LR
    with np.errstate(all="ignore"):      # c dt exactly on the contour is 0/0 in the reference as well
        Qh = dt * (((np.exp(LR / 2.) - 1.) / LR).mean(axis=-1))
        f0 = dt * (((-4. - LR + (np.exp(LR) * (4. - 3. * LR + LR2))) / LR3).mean(axis=-1))
        fab = dt * (((2. + LR + np.exp(LR) * (-2. + LR)) / LR3).mean(axis=-1))
        fc = dt * (((-4. - 3. * LR - LR2 + np.exp(LR) * (4. - LR)) / LR3).mean(axis=-1))
    return np.stack([Qh, f0, fab, fc], axis=-1)


def linear_operator(model, eq, k, l, prm):
    """``c`` at the given wavenumbers, in the reference's order of operations (Kernel.py:417-418, :440-442;
    QGModel.py:426-428, :452-453)."""
    wv2 = k ** 2 + l ** 2
    wv4 = wv2 ** 2
    if model == QG and eq == 0:
        wv2i = np.zeros_like(wv2)
        nz = wv2 != 0.
        wv2i[nz] = wv2[nz] ** -1
        c = np.zeros(k.shape, complex)
        c += -prm["nu4"] * wv4 - prm["nu"] * wv2 - prm["mu"] - 1j * k * prm["U"]
        c += prm["beta"] * (1j * k) * wv2i
    elif model == QG:
        c = np.zeros(k.shape, complex)
        c += -prm["nu4c"] * wv4 - prm["nuc"] * wv2 - prm["muc"]
    elif eq == 0:
        c = np.zeros(k.shape, complex) - 1j * k * prm["U"]
        c += -prm["nu4"] * wv4 - prm["nu"] * wv2 - prm["mu"]
    else:
        c = np.zeros(k.shape, complex) - 1j * k * prm["U"]
        c += -prm["nu4w"] * wv4 - 0.5j * prm["f"] * (wv2 / prm["kappa2"]) - prm["nuw"] * wv2 - prm["muw"]
    return c


def patch_near_contour(near, patch, model, nx, kk, ll, filtr, dt, prm, equations, delta=DELTA):
    """For every public equation id in ``equations`` (0: q, 1: phi, 2: QGModel's passive scalar): ask the device which entries
    sit within ``delta`` of the contour (``near(eq, delta) -> (l, k)`` index arrays, k global), recompute them as the reference
    does and hand them to ``patch(eq, l, k, vals)``.  Returns {eq: number of entries}."""
    counts = {}
    if os.environ.get("NIWQG_AMD_CONTOUR_PATCH", "1") == "0":      # for measuring what the patch is worth
        return counts
    for eq in equations:
        li, ki = near(eq, delta)
        counts[eq] = len(li)
        if len(li) == 0:
            continue
        order = np.lexsort((ki, li))
        li, ki = li[order], ki[order]
        vals = contour_tables(linear_operator(model, eq, kk[ki], ll[li], prm) * dt, dt)
        if model != QG and eq == 0:
            lm, km = (nx - li) % nx, (nx - ki) % nx
            mirror = np.conj(contour_tables(linear_operator(model, eq, kk[km], ll[lm], prm) * dt, dt))
            fp, fm = filtr[li, ki][:, None], filtr[lm, km][:, None]
            w = fp + fm
            one_sided = (fp != fm) & (w != 0.)
            mean = 0.5 * (vals + mirror)
            with np.errstate(all="ignore"):
                vals = np.where(one_sided, (fp * vals + fm * mirror) / np.where(w != 0., w, 1.), mean)
        patch(eq, li, ki, np.ascontiguousarray(vals, dtype=np.complex128))
    return counts
