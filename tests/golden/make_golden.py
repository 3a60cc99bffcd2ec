#!/usr/bin/env python
"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE.

Run in the build container only (needs /root/reference):

    python tests/golden/make_golden.py

The reference imports ``h5py`` at module top but only uses it when
``save_to_disk=True``; h5py is not installed here, so an empty stub module is put
in ``sys.modules`` first.  Nothing of the reference is
copied: only its *outputs* (arrays, scalars) on seeded inputs are stored.

Files written (all numpy ``.npz``, loadable with ``allow_pickle=False``):
  g1_functions_64.npz     per-function vectors, CoupledModel 64^2
  g2_coupled_*.npz        CoupledModel LambDipole trajectories (notebook parameters)
  g3_qg_*.npz             QGModel LambDipole trajectories (examples/LambDipole_qg.py)
  g4_quirks_64.npz        UnCoupledModel tdiags dependence (Q1), set order (Q2)
  g6_notebook_diags.npz   diagnostics time series of the 128^2 notebook run (400 steps)
  g7_checksums.npz        scalar checksums at 256^2 / 512^2
  g8_ybj_64.npz           YBJModel (steady psi): trajectory, diagnostics series, stale phix/phiy
  g9_initial_conditions_64.npz   all five generators of niwqg/InitialConditions.py (seeded)
  g10_qg_passive_64.npz   QGModel with passive_scalar=True: trajectory of q and c, cvar, diagnostics series
  g11_at_size_2048.npz    the REAL reference at 2048^2 on white-noise states (BASELINE config 2: QGModel random q, 2 steps;
                          CoupledModel rough q and phi, 1 step): not the fields (32-64 MB each) but what pins them -- 256
                          seeded random projections, a 64x64 sub-sample, norms, budgets; the tests regenerate the same
                          projection vectors from the seed (g11 takes ~6 minutes and 16 GB here; not in the default list)
  g12_coupled_2048_10steps.npz  the REAL reference, CoupledModel 2048^2, the same white-noise state and parameters as g11's
                          coupled case, after 5 and 10 steps: projections, sub-samples, norms, budgets (round 3; ~8 minutes
                          and 16 GB here; not in the default list)
  g13_contour_entries.npz the REAL reference where its ETDRK4 planes are ill-conditioned: two configurations with U = 0 (real q
                          operator) whose c dt comes within 3e-5 of the contour point -1, found by the randomized parity
                          test.  Qh, f0, fab, fc at every entry within 0.05 of the contour (there the reference's value is
                          numpy's rounding error times eps / distance^3), and for the 256^2 CoupledModel (2/3-rule
                          dealiasing, inviscid waves) q and phi after 6 steps from seeded white noise, which feeds them
  g16_coupled_lamb_<nx>_100steps.npz  the REAL reference, CoupledModel LambDipole (filter on) at 1024^2 / 2048^2 after 50 and 100
                          steps: projections, sub-samples, spectral corners, norms, budgets (round 4; not in the default list)
  g17_non_power_of_two.npz  the REAL reference on 96^2 and 192^2 grids, CoupledModel and QGModel, 1 / 10 / 100 steps (round 4)
  g18_families_1024_100steps.npz  the REAL reference at 1024^2 for the other three model families, 50 and 100 steps with the
                          diagnostics ticking every 10 steps: UnCoupledModel (BASELINE config 5, member 3), YBJModel (dipole + wave
                          packet), QGModel with beta and its passive scalar: projections, sub-samples, norms, budgets, the
                          diagnostics series (round 4; ~15 minutes here; not in the default list)
  g19_attributes_after_steps_without_ticks.npz  the g15 inventory with tdiags = 3 and no status lines: what the reference's
                          instances carry after two steps without a diagnostics tick (round 4)
  g14_instance_attributes.npz  what a freshly constructed instance of each of the four model classes carries (nx = 64, every other
                          argument at its default): names and values of the scalar attributes, names, shapes, dtypes and two
                          checksums of the array attributes
"""
import os
import sys
import types
import logging

import numpy as np

sys.modules.setdefault("h5py", types.ModuleType("h5py"))
sys.path.insert(0, "/root/reference")

from niwqg import CoupledModel, UnCoupledModel, QGModel, YBJModel      # noqa: E402
from niwqg import InitialConditions as ic                    # noqa: E402

logging.getLogger("niwqg.Kernel").setLevel(logging.ERROR)
logging.getLogger("niwqg.QGModel").setLevel(logging.ERROR)

HERE = os.path.dirname(os.path.abspath(__file__))


def save(name, **arrays):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrays)
    print("wrote", name, "%.1f KiB" % (os.path.getsize(path) / 1024.0))


# ---- shared physical parameters (examples/LambDipole_CoupledModel.ipynb cells 3-7) ----
F0, NB, L = 1e-4, 0.01, 2 * np.pi * 200e3
MZ = 2 * np.pi / 280.0
K0 = 10 * (2 * np.pi / L)
U0 = 0.1
TE = 1.0 / (U0 * K0)


def notebook_kwargs(nx, use_filter, nsteps, tdiags=1):
    dt = 0.025 * TE * 128 / nx
    return dict(L=L, nx=nx, tmax=(nsteps - 0.5) * dt, dt=dt, m=MZ, N=NB, f=F0,
                twrite=10 ** 9, nu4=5e11 * (128.0 / nx) ** 4, nu4w=0.0, nu=20, nuw=50.0,
                mu=0.0, muw=0.0, use_filter=use_filter, U=-U0, tdiags=tdiags,
                save_to_disk=False, dealias=False)


def lamb_and_uniform(m):
    q = ic.LambDipole(m, U=U0, R=2 * np.pi / K0)
    phi = (np.ones_like(q) + 1j) * (2 * U0) / np.sqrt(2)
    return q, phi


def step_to(m, n):
    while m.tc < n:
        m._step_forward()


# ---------------------------------------------------------------- G1
def g1():
    nx = 64
    out = {}
    for mode, kw in (("exp", dict(use_filter=True)), ("twothirds", dict(use_filter=False, dealias=True)),
                     ("none", dict(use_filter=False))):
        m = CoupledModel.Model(nx=nx, **kw)
        out["filtr_" + mode] = m.filtr
    m = CoupledModel.Model(**notebook_kwargs(nx, True, 1))
    q, _ = lamb_and_uniform(m)
    phi = 0.2 * ic.WavePacket(m, k=3 * K0, l=K0, R=L / 6, x0=L / 2, y0=L / 2)
    m.set_q(q)
    m.set_phi(phi)
    m._invert()                      # make ph/qwh consistent with phi (undo quirk Q2 for this vector)
    m._calc_rel_vorticity()
    out.update(q0=q, phi0=phi, kk=m.kk, ll=m.ll,
               expch=m.expch, expch_h=m.expch_h, Qh=m.Qh, f0=m.f0, fab=m.fab, fc=m.fc,
               expchw=m.expchw, expch_hw=m.expch_hw, Qhw=m.Qhw, f0w=m.f0w, fabw=m.fabw, fcw=m.fcw,
               ph=m.ph, qwh=m.qwh, q_psi=m.q_psi, phix=m.phix, phiy=m.phiy, qh=m.qh, phih=m.phih)
    out["jac_psi_q"] = m.jacobian_psi_q()
    out["u"], out["v"] = m.u, m.v
    out["jac_psi_phi"] = m.jacobian_psi_phi()
    out["jac_phic_phi"] = m.jacobian_phic_phi()
    out["refraction"] = m.fft(m.phi * m.q_psi)
    m._calc_energy_conversion()
    out["budget"] = np.array([m.gamma1, m.gamma2, m.xi1, m.xi2, m.pi,
                              m._calc_ep_psi(), m._calc_chi_phi(), m._calc_ep_phi()])
    out["energies"] = np.array([m._calc_ke_qg(), m._calc_ke_niw(), m._calc_pe_niw(), m._calc_cfl()])
    out["params"] = np.array([m.dt, m.nu4, m.L, m.U, m.f, m.kappa2])
    save("g1_functions_64.npz", **out)


# ---------------------------------------------------------------- G2
def g2():
    for nx, snaps, full in ((64, (1, 10, 100), True), (128, (100,), False)):
        for use_filter in (False, True):
            m = CoupledModel.Model(**notebook_kwargs(nx, use_filter, max(snaps), tdiags=10 ** 9))
            q, phi = lamb_and_uniform(m)
            m.set_q(q)
            m.set_phi(phi)
            out = dict(q0=q, phi0=phi, snaps=np.array(snaps))
            for n in snaps:
                step_to(m, n)
                out["q_%d" % n] = m.q.copy()
                out["phi_%d" % n] = m.phi.copy()
                out["budgets_%d" % n] = np.array([m.Ke, m.Pw, m.Kw])
                if full or n == max(snaps):
                    out["qh_%d" % n] = m.qh.copy()
                    out["phih_%d" % n] = m.phih.copy()
                    out["ph_%d" % n] = m.ph.copy()
            save("g2_coupled_%d_%s.npz" % (nx, "filter" if use_filter else "nofilter"), **out)


# ---------------------------------------------------------------- G3
def g3():
    for nx, snaps in ((64, (1, 10, 200)), (256, (200,))):
        dt = 0.05 * TE * 128 / nx if nx < 128 else 0.05 * TE
        m = QGModel.Model(L=L, nx=nx, tmax=1e30, dt=dt, twrite=10 ** 9, nu4=7.5e8, use_filter=False,
                          save_to_disk=False, U=-U0, tdiags=10 ** 9, beta=0.0, passive_scalar=False)
        q = ic.LambDipole(m, U=U0, R=2 * np.pi / K0)
        m.set_q(q)
        out = dict(q0=q, snaps=np.array(snaps), dt=np.array(dt))
        for n in snaps:
            step_to(m, n)
            out["q_%d" % n] = m.q.copy()
            out["qh_%d" % n] = m.qh.copy()
            out["Ke_%d" % n] = np.array(m.Ke)
        save("g3_qg_%d.npz" % nx, **out)
    # beta-plane + filter variant (exercises beta term and exponential filter on the half spectrum)
    nx = 64
    m = QGModel.Model(L=L, nx=nx, tmax=1e30, dt=0.05 * TE * 2, twrite=10 ** 9, nu4=7.5e8, nu=5.0, mu=1e-8,
                      use_filter=True, save_to_disk=False, U=-U0, tdiags=10 ** 9, beta=2e-11,
                      passive_scalar=False)
    q = ic.LambDipole(m, U=U0, R=2 * np.pi / K0)
    m.set_q(q)
    step_to(m, 20)
    save("g3_qg_64_beta.npz", q0=q, q_20=m.q.copy(), qh_20=m.qh.copy(), Ke_20=np.array(m.Ke),
         dt=np.array(m.dt))


# ---------------------------------------------------------------- G4
def g4():
    nx, out = 64, {}
    rng = np.random.default_rng(0)
    for tag, tdiags in (("td1", 1), ("tdinf", 10 ** 9)):
        kw = notebook_kwargs(nx, True, 20, tdiags=tdiags)
        m = UnCoupledModel.Model(**kw)
        q = ic.LambDipole(m, U=U0, R=2 * np.pi / K0)
        phi = 0.1 * ic.WavePacket(m, k=3 * K0, l=0, R=L / 6, x0=L / 2, y0=L / 2)
        m.set_q(q)
        m.set_phi(phi)
        step_to(m, 20)
        out.update({"unc_q0": q, "unc_phi0": phi, "unc_q_" + tag: m.q.copy(),
                    "unc_phi_" + tag: m.phi.copy(), "unc_qh_" + tag: m.qh.copy(),
                    "unc_phih_" + tag: m.phih.copy(),
                    "unc_budgets_" + tag: np.array([m.Ke, m.Pw, m.Kw])})
    for tag in ("q_then_phi", "phi_then_q"):
        m = CoupledModel.Model(**notebook_kwargs(nx, True, 1, tdiags=10 ** 9))
        q, _ = lamb_and_uniform(m)
        phi = 0.2 * ic.WavePacket(m, k=3 * K0, l=K0, R=L / 6, x0=L / 2, y0=L / 2)
        if tag == "q_then_phi":
            m.set_q(q); m.set_phi(phi)
        else:
            m.set_phi(phi); m.set_q(q)
        out["order_ph0_" + tag] = m.ph.copy()
        step_to(m, 1)
        out["order_q_" + tag] = m.q.copy()
        out["order_phi_" + tag] = m.phi.copy()
    out["order_q0"], out["order_phi0"] = q, phi
    # random-phase rough field, two-thirds dealiasing, nonzero nu4w/mu/muw: exercises every budget term
    kw = notebook_kwargs(nx, False, 5, tdiags=10 ** 9)
    kw.update(dealias=True, nu4w=1e10, mu=1e-8, muw=2e-8)
    m = CoupledModel.Model(**kw)
    q = 1e-5 * rng.standard_normal((nx, nx))
    phi = 0.05 * (rng.standard_normal((nx, nx)) + 1j * rng.standard_normal((nx, nx)))
    m.set_q(q)
    m.set_phi(phi)
    step_to(m, 5)
    out.update(rough_q0=q, rough_phi0=phi, rough_q=m.q.copy(), rough_phi=m.phi.copy(),
               rough_qh=m.qh.copy(), rough_phih=m.phih.copy(),
               rough_budgets=np.array([m.Ke, m.Pw, m.Kw]))
    save("g4_quirks_64.npz", **out)


# ---------------------------------------------------------------- G6
def g6():
    nx = 128
    dt = 0.025 * TE
    m = CoupledModel.Model(L=L, nx=nx, tmax=10 * TE, dt=dt, m=MZ, N=NB, f=F0,
                           twrite=int(1 * (2 * np.pi / F0) / dt), nu4=5e11, nu4w=0e10, nu=20, nuw=50e0,
                           mu=0.0, muw=0.0, use_filter=False, U=-U0, tdiags=1, save_to_disk=False,
                           dealias=False)
    q, phi = lamb_and_uniform(m)
    m.set_q(q)
    m.set_phi(phi)
    status = []
    while m.t < m.tmax:
        m._step_forward()
        if (m.tc % m.twrite) == 0:
            status.append([m.tc, m.t, m.ke, m.kew, m.pew, m.cfl])
    out = {k: np.asarray(v["value"]) for k, v in m.diagnostics.items()}
    out["status"] = np.array(status)
    out["final_q"] = m.q
    out["final_phi"] = m.phi
    save("g6_notebook_diags.npz", **out)


# ---------------------------------------------------------------- G7
def g8():
    """YBJModel 64^2 (niwqg/YBJModel.py): dipole + wave packet on a uniform wave, 20 steps, tdiags=1 and tdiags=inf."""
    nx, nsteps = 64, 20
    out = {}
    for tag, td in (("td1", 1), ("tdinf", 10 ** 9)):
        for use_filter in (True, False):
            kw = notebook_kwargs(nx, use_filter, nsteps, tdiags=td)
            kw.update(nu4w=3e9, muw=1e-7)
            m = YBJModel.Model(**kw)
            q0 = ic.LambDipole(m, U=U0, R=2 * np.pi / K0)
            phi0 = 0.2 * ic.WavePacket(m, k=2 * K0, l=K0, R=L / 6, x0=L / 2, y0=L / 2) + 0.05
            m.set_q(q0)
            m.set_phi(phi0)
            m.run()
            assert m.tc == nsteps
            key = "%s_%s" % (tag, "filter" if use_filter else "nofilter")
            out["phi_" + key] = m.phi
            out["phih_" + key] = m.phih
            out["phix_" + key] = m.phix
            out["phiy_" + key] = m.phiy
            out["scalars_" + key] = np.array([m.Ke, m.Pw, m.Kw, m._calc_ke_niw(), m._calc_ke_qg()])
            if td == 1:
                for name, d in m.diagnostics.items():
                    out["diag_%s_%s" % (name, key)] = np.asarray(d["value"], dtype=float)
            out["q0"], out["phi0"] = q0, phi0
    save("g8_ybj_64.npz", **out)


def g7():
    out = {}
    for nx in (256, 512):
        m = CoupledModel.Model(**notebook_kwargs(nx, True, 3, tdiags=10 ** 9))
        q, phi = lamb_and_uniform(m)
        m.set_q(q)
        m.set_phi(phi)
        step_to(m, 3)
        out["c%d" % nx] = np.array([m.spec_var(m.qh), m.spec_var(m.phih), m.q.mean(), np.abs(m.q).max(),
                                    np.abs(m.phi).max(), m.phi.mean().real, m.phi.mean().imag,
                                    m.Ke, m.Pw, m.Kw, (m.q ** 2).sum(), (np.abs(m.phi) ** 2).sum()])
    save("g7_checksums.npz", **out)


def g9():
    """niwqg/InitialConditions.py on a 64^2 CoupledModel grid; numpy's global RNG seeded before each random field."""
    m = CoupledModel.Model(**notebook_kwargs(64, True, 1))
    out = {}
    np.random.seed(11)
    out["mcwilliams"] = ic.McWilliams1984(m, k0=6 * 2 * np.pi / L, E=0.5 * U0 ** 2)
    np.random.seed(12)
    out["danioux"] = ic.Danioux2015(m, k0=8 * 2 * np.pi / L, E=0.5 * U0 ** 2)
    out["lamb"] = ic.LambDipole(m, U=U0, R=2 * np.pi / K0)
    out["packet"] = ic.WavePacket(m, k=3 * K0, l=K0, R=L / 6, x0=L / 3, y0=L / 2)
    out["plane"] = ic.PlaneWave(m, k=3 * K0, l=2 * K0, phase=0.3)
    save("g9_initial_conditions_64.npz", **out)


def g10():
    """QGModel 64^2 with its passive scalar (niwqg/QGModel.py:345-404, :483-495, :522-534, :595-604, :724-737)."""
    nx, nsteps = 64, 20
    dt = 0.05 * TE * 128 / nx / 2
    out = {}
    for use_filter in (True, False):
        m = QGModel.Model(L=L, nx=nx, tmax=(nsteps - 0.5) * dt, dt=dt, twrite=10 ** 9, nu4=7.5e8 * 16, nu=5.0,
                          mu=1e-8, use_filter=use_filter, U=-U0, tdiags=1, beta=2e-11, passive_scalar=True,
                          nu4c=3e9, nuc=2.0, muc=1e-8, save_to_disk=False)
        q0 = ic.LambDipole(m, U=U0, R=2 * np.pi / K0)
        c0 = np.sin(2 * np.pi * 3 * m.x / L) * np.cos(2 * np.pi * 2 * m.y / L) + 0.3
        m.set_q(q0)
        m.set_c(c0)
        m.run()
        assert m.tc == nsteps
        key = "filter" if use_filter else "nofilter"
        out["dt"] = dt
        out["q0"], out["c0"] = q0, c0
        out["q_" + key], out["qh_" + key] = m.q, m.qh
        out["c_" + key], out["ch_" + key] = m.c, m.ch
        out["scalars_" + key] = np.array([m.Ke, m.cvar, m.C2, m.gradC2])
        for name, d in m.diagnostics.items():
            out["diag_%s_%s" % (name, key)] = np.asarray(d["value"], dtype=float)
    save("g10_qg_passive_64.npz", **out)


# ---------------------------------------------------------------- G11: the reference itself at size
def projections(field, seed, n=256):
    """n inner products of the field with seeded random-sign vectors (one row of +-1 per projection would be 32 MB each at
    2048^2: use separable signs sx[i] * sy[j], regenerated from the seed by the tests)"""
    rng = np.random.default_rng(seed)
    ny, nx = field.shape
    out = np.empty(n, field.dtype)
    for i in range(n):
        sy = rng.integers(0, 2, ny) * 2.0 - 1.0
        sx = rng.integers(0, 2, nx) * 2.0 - 1.0
        out[i] = sy @ field @ sx
    return out


def g11():
    nx = 2048
    out = {}
    # BASELINE config 2 exactly as bench.py --model qg --nx 2048 builds it (dt = 0.05 Te / 8, nu4 = 7.5e8 / 64;
    # round 4 -- rounds 2 and 3 ran it with half that dt)
    dt, nu4 = 0.05 * TE / 8, 7.5e8 / 64
    m = QGModel.Model(L=L, nx=nx, tmax=1e30, dt=dt, twrite=10 ** 9, nu4=nu4, use_filter=True, save_to_disk=False,
                      U=-U0, tdiags=10 ** 9)
    q0 = 1e-5 * np.random.default_rng(0).standard_normal((nx, nx))
    m.set_q(q0)
    step_to(m, 2)
    out["qg_params"] = np.array([dt, nu4, L, -U0])
    out["qg_q_proj"] = projections(m.q, 101)
    out["qg_q_sub"] = m.q[::32, ::32].copy()
    out["qg_q_norm"] = np.array(np.linalg.norm(m.q))
    out["qg_qh_proj"] = projections(m.qh, 102)
    out["qg_Ke"] = np.array(m.Ke)
    # the noise-amplified entries of the contour-mean planes (|c dt| ~ 1: DESIGN.md section 6): the reference's own values
    ch = (-m.nu4 * m.wv4 - 1j * m.k * m.U) * m.dt
    near = np.argsort(np.abs(np.abs(ch) - 1.0).ravel())[:64]
    out["qg_coef_idx"] = near
    for nm in ("Qh", "f0", "fab", "fc"):
        out["qg_coef_" + nm] = getattr(m, nm).ravel()[near]
    del m
    # CoupledModel, white-noise q and phi, every dissipation term on: one step (tests/test_gpu_at_size.py: rough_kwargs)
    kw = notebook_kwargs(nx, True, 1, tdiags=10 ** 9)
    kw.update(nu4w=kw["nu4"] * 0.1, mu=1e-8, muw=2e-8)
    m = CoupledModel.Model(**kw)
    rng = np.random.default_rng(11)
    q0 = 1e-5 * rng.standard_normal((nx, nx))
    phi0 = 0.05 * (rng.standard_normal((nx, nx)) + 1j * rng.standard_normal((nx, nx)))
    m.set_q(q0)
    m.set_phi(phi0)
    step_to(m, 1)
    out["cpl_q_proj"] = projections(m.q, 201)
    out["cpl_phi_proj"] = projections(m.phi, 202)
    out["cpl_q_sub"], out["cpl_phi_sub"] = m.q[::32, ::32].copy(), m.phi[::32, ::32].copy()
    out["cpl_norms"] = np.array([np.linalg.norm(m.q), np.linalg.norm(m.phi)])
    out["cpl_budgets"] = np.array([m.Ke, m.Pw, m.Kw])
    save("g11_at_size_2048.npz", **out)


def g12():
    nx = 2048
    out = {}
    kw = notebook_kwargs(nx, True, 10, tdiags=10 ** 9)
    kw.update(nu4w=kw["nu4"] * 0.1, mu=1e-8, muw=2e-8)
    m = CoupledModel.Model(**kw)
    rng = np.random.default_rng(11)
    q0 = 1e-5 * rng.standard_normal((nx, nx))
    phi0 = 0.05 * (rng.standard_normal((nx, nx)) + 1j * rng.standard_normal((nx, nx)))
    m.set_q(q0)
    m.set_phi(phi0)
    for n in (5, 10):
        step_to(m, n)
        tag = "s%d_" % n
        out[tag + "q_proj"] = projections(m.q, 301)
        out[tag + "phi_proj"] = projections(m.phi, 302)
        out[tag + "qh_proj"] = projections(m.qh, 303)
        out[tag + "phih_proj"] = projections(m.phih, 304)
        out[tag + "q_sub"], out[tag + "phi_sub"] = m.q[::32, ::32].copy(), m.phi[::32, ::32].copy()
        out[tag + "norms"] = np.array([np.linalg.norm(m.q), np.linalg.norm(m.phi)])
        out[tag + "budgets"] = np.array([m.Ke, m.Pw, m.Kw])
        print("g12: step", n, "done", flush=True)
    save("g12_coupled_2048_10steps.npz", **out)


def contour_entries(model, names, ch, tag, out, delta=0.05):
    r = np.exp(2j * np.pi * (np.arange(1., 33.) / 32.))
    dist = np.abs(ch[..., None] + r).min(axis=-1)
    li, ki = np.nonzero(dist < delta)
    out[tag + "l"], out[tag + "k"], out[tag + "dist"] = li.astype(np.int32), ki.astype(np.int32), dist[li, ki]
    for nm in names:
        out[tag + nm] = getattr(model, nm)[li, ki]
    print("g13:", tag, len(li), "entries within", delta, "closest", dist.min(), flush=True)


def g13():
    out = {}
    # (a) QGModel 512^2, exponential filter, U = 0, beta, nu = mu = 0 (seed 140 of the randomized parity test)
    kw = dict(L=L, nx=512, tmax=1e30, dt=625.0, twrite=10 ** 9, tdiags=10 ** 9, use_filter=True, dealias=False, U=0.0,
              nu4=1887323331.1493955, nu=0.0, mu=0.0, beta=2e-11, save_to_disk=False)
    m = QGModel.Model(**kw)
    c = np.zeros((m.nl, m.nk), complex)
    c += -m.nu4 * m.wv4 - m.nu * m.wv2 - m.mu - 1j * m.k * m.U
    c += m.beta * m.ik * m.wv2i
    contour_entries(m, ("Qh", "f0", "fab", "fc"), c * m.dt, "qg_", out)
    rng = np.random.default_rng(13)
    m.set_q(1e-5 * rng.standard_normal((512, 512)))
    step_to(m, 6)
    out["qg_q6_proj"], out["qg_q6_sub"], out["qg_q6_norm"] = projections(m.q, 401), m.q[::8, ::8].copy(), np.linalg.norm(m.q)
    out["qg_qh6_proj"] = projections(m.qh, 402)
    # (b) CoupledModel 256^2, 2/3-rule mask, U = 0, no wave dissipation but muw (seed 42 of the same test)
    kw = dict(L=L, nx=256, tmax=1e30, dt=1250.0, twrite=10 ** 9, tdiags=10 ** 9, use_filter=False, dealias=True, U=0.0,
              nu4=48273918940.97328, nu=20.0, mu=1e-8, nuw=0.0, nu4w=0.0, muw=2e-8, m=MZ, N=NB, f=F0, save_to_disk=False)
    m = CoupledModel.Model(**kw)
    cq = np.zeros((m.nl, m.nk), complex) - 1j * m.k * m.U
    cq += -m.nu4 * m.wv4 - m.nu * m.wv2 - m.mu
    contour_entries(m, ("Qh", "f0", "fab", "fc"), cq * m.dt, "cq_", out)
    contour_entries(m, ("Qhw", "f0w", "fabw", "fcw"), m.c * m.dt, "cw_", out)      # the reference leaves the wave operator in m.c
    rng = np.random.default_rng(14)
    m.set_q(1e-5 * rng.standard_normal((256, 256)))
    m.set_phi(0.05 * (rng.standard_normal((256, 256)) + 1j * rng.standard_normal((256, 256))))
    step_to(m, 6)
    out["c_q6"], out["c_phi6"] = m.q.copy(), m.phi.copy()
    out["c_budgets"] = np.array([m.Ke, m.Pw, m.Kw])
    # (c) a mean flow: c dt off the real axis, the entries next to the contour points on either side of -1 (QGModel 256^2,
    #     U = -U0, beta; CoupledModel 128^2 with U = U0/2: its phi operator has the dispersion term as well)
    kw = dict(L=L, nx=256, tmax=1e30, dt=0.05 * TE, twrite=10 ** 9, tdiags=10 ** 9, use_filter=True, dealias=False, U=-U0,
              nu4=3.1e10, nu=5.0, mu=1e-8, beta=2e-11, save_to_disk=False)
    m = QGModel.Model(**kw)
    c = np.zeros((m.nl, m.nk), complex)
    c += -m.nu4 * m.wv4 - m.nu * m.wv2 - m.mu - 1j * m.k * m.U
    c += m.beta * m.ik * m.wv2i
    contour_entries(m, ("Qh", "f0", "fab", "fc"), c * m.dt, "qgu_", out)
    out["qgu_params"] = np.array([kw["dt"], kw["nu4"], kw["nu"], kw["mu"], kw["U"], kw["beta"]])
    kw = dict(L=L, nx=128, tmax=1e30, dt=0.025 * TE, twrite=10 ** 9, tdiags=10 ** 9, use_filter=True, dealias=False, U=0.5 * U0,
              nu4=5e11, nu=20.0, mu=0.0, nuw=50.0, nu4w=5e10, muw=2e-8, m=MZ, N=NB, f=F0, save_to_disk=False)
    m = CoupledModel.Model(**kw)
    cq = np.zeros((m.nl, m.nk), complex) - 1j * m.k * m.U
    cq += -m.nu4 * m.wv4 - m.nu * m.wv2 - m.mu
    contour_entries(m, ("Qh", "f0", "fab", "fc"), cq * m.dt, "cuq_", out)
    contour_entries(m, ("Qhw", "f0w", "fabw", "fcw"), m.c * m.dt, "cuw_", out)
    out["cu_params"] = np.array([kw["dt"], kw["nu4"], kw["nu"], kw["nuw"], kw["nu4w"], kw["muw"], kw["U"]])
    save("g13_contour_entries.npz", **out)


def g14():
    out = {}
    attribute_inventory(out, "", lambda mod: mod.Model(nx=64))
    save("g14_instance_attributes.npz", **out)


def g15():
    """the same inventory after set_q, set_phi (set_c) and three steps with a diagnostics tick and a status line at every step"""
    out = {}

    def stepped(mod):
        if mod is QGModel:
            m = mod.Model(L=L, nx=64, tmax=1e30, dt=0.05 * TE * 2, twrite=1, nu4=7.5e8 * 16, nu=5.0, mu=1e-8, use_filter=True, U=-U0,
                          tdiags=1, beta=2e-11, passive_scalar=True, nu4c=3e9, nuc=2.0, muc=1e-8, save_to_disk=False)
        else:
            kw = notebook_kwargs(64, True, 10, tdiags=1)
            kw.update(nu4w=1e10, mu=1e-8, muw=2e-8, twrite=1)
            m = mod.Model(**kw)
        rng = np.random.default_rng(15)
        m.set_q(ic.LambDipole(m, U=U0, R=2 * np.pi / K0) + 1e-6 * rng.standard_normal((64, 64)))
        if mod is QGModel:
            m.set_c(1.0 + 0.3 * rng.standard_normal((64, 64)))
        else:
            m.set_phi(ic.WavePacket(m, k=2 * K0, l=K0, R=L / 6, x0=L / 2, y0=L / 2) * 0.1
                      + 0.01 * (rng.standard_normal((64, 64)) + 1j * rng.standard_normal((64, 64))))
        step_to(m, 3)
        return m
    attribute_inventory(out, "", stepped)
    save("g15_attributes_after_three_steps.npz", **out)


def g19():
    """round 4: the g15 inventory after steps WITHOUT a diagnostics tick or a status line -- tdiags = 3, twrite = 10**9, three
    steps: the only tick is the first step's (tc = 0), the second and third steps leave behind what the step itself refreshes
    (found by golden g18: QGModel's step recomputes C2, gradC2, lapc, Gamma_c in every stage, niwqg/QGModel.py:351-391)."""
    out = {}

    def stepped(mod):
        if mod is QGModel:
            m = mod.Model(L=L, nx=64, tmax=1e30, dt=0.05 * TE * 2, twrite=10 ** 9, nu4=7.5e8 * 16, nu=5.0, mu=1e-8, use_filter=True,
                          U=-U0, tdiags=3, beta=2e-11, passive_scalar=True, nu4c=3e9, nuc=2.0, muc=1e-8, save_to_disk=False)
        else:
            kw = notebook_kwargs(64, True, 10, tdiags=3)
            kw.update(nu4w=1e10, mu=1e-8, muw=2e-8, twrite=10 ** 9)
            m = mod.Model(**kw)
        rng = np.random.default_rng(19)
        m.set_q(ic.LambDipole(m, U=U0, R=2 * np.pi / K0) + 1e-6 * rng.standard_normal((64, 64)))
        if mod is QGModel:
            m.set_c(1.0 + 0.3 * rng.standard_normal((64, 64)))
        else:
            m.set_phi(ic.WavePacket(m, k=2 * K0, l=K0, R=L / 6, x0=L / 2, y0=L / 2) * 0.1
                      + 0.01 * (rng.standard_normal((64, 64)) + 1j * rng.standard_normal((64, 64))))
        step_to(m, 3)
        return m
    attribute_inventory(out, "", stepped)
    save("g19_attributes_after_steps_without_ticks.npz", **out)


def g16(sizes=None):
    """round 4: BASELINE's literal criterion at size -- the REAL reference, CoupledModel, LambDipole q + uniform phi, notebook
    parameters scaled to the grid (examples/LambDipole.py:45-58), FILTER ON, after 50 and 100 steps.  The dipole's vorticity has a
    kink at r = R, so its spectrum reaches the filter band from step 0 and the cascade feeds it over the horizon.  One file per
    size: `python make_golden.py g16` writes 1024^2 (~7 min, 4 GB here); `G16_SIZES=2048 python make_golden.py g16` the 2048^2 one
    (~50 min, 16 GB)."""
    sizes = sizes or [int(s) for s in os.environ.get("G16_SIZES", "1024").split(",")]
    for nx in sizes:
        out = {}
        kw = notebook_kwargs(nx, True, 100, tdiags=10 ** 9)
        m = CoupledModel.Model(**kw)
        q0, phi0 = lamb_and_uniform(m)
        m.set_q(q0)
        m.set_phi(phi0)
        out["params"] = np.array([nx, kw["dt"], kw["nu4"], kw["nu"], kw["nuw"], kw["U"]])
        out["q0_sub"] = q0[::nx // 64, ::nx // 64].copy()
        out["q0_norm"] = np.array(np.linalg.norm(q0))
        for n in (50, 100):
            step_to(m, n)
            tag = "s%d_" % n
            out[tag + "q_proj"] = projections(m.q, 401)
            out[tag + "phi_proj"] = projections(m.phi, 402)
            out[tag + "qh_proj"] = projections(m.qh, 403)
            out[tag + "phih_proj"] = projections(m.phih, 404)
            out[tag + "q_sub"], out[tag + "phi_sub"] = m.q[::nx // 64, ::nx // 64].copy(), m.phi[::nx // 64, ::nx // 64].copy()
            # the low corner of the spectra (the energetic modes) and a strip inside the filter band, element by element
            out[tag + "qh_low"], out[tag + "phih_low"] = m.qh[:32, :32].copy(), m.phih[:32, :32].copy()
            b = int(0.36 * nx)
            out[tag + "qh_band"], out[tag + "phih_band"] = m.qh[b:b + 8, b:b + 64].copy(), m.phih[b:b + 8, b:b + 64].copy()
            # ... and a strip across the filter's cut-off (index 0.65 nx / 2 along k, l small), where the filtered cascade lives
            c0 = int(0.65 * nx / 2) - 32
            out[tag + "qh_edge"], out[tag + "phih_edge"] = m.qh[:8, c0:c0 + 64].copy(), m.phih[:8, c0:c0 + 64].copy()
            out[tag + "norms"] = np.array([np.linalg.norm(m.q), np.linalg.norm(m.phi), np.linalg.norm(m.qh), np.linalg.norm(m.phih)])
            out[tag + "budgets"] = np.array([m.Ke, m.Pw, m.Kw])
            print("g16: nx", nx, "step", n, "done", flush=True)
        save("g16_coupled_lamb_%d_100steps.npz" % nx, **out)
        del m


def g18():
    """round 4: the REAL reference at 1024^2 for the three model families golden g16 does not cover, 50 and 100 steps,
    tdiags = 10 (so the diagnostics tick ten times inside the horizon and UnCoupledModel's quirk Q1 -- phix, phiy refreshed only
    by the tick, niwqg/Kernel.py:608-611 -- acts at size)."""
    nx = 1024
    out = {}

    def record(tag, m, fields, scalars):
        for n in (50, 100):
            step_to(m, n)
            t = "%s_s%d_" % (tag, n)
            norms = []
            for i, name in enumerate(fields):
                a = getattr(m, name)
                out[t + name + "_proj"] = projections(a, 500 + i)
                out[t + name + "_sub"] = a[::nx // 64, ::a.shape[1] // 64 if a.shape[1] >= 64 else 1].copy()
                norms.append(np.linalg.norm(a))
            out[t + "norms"] = np.array(norms)
            out[t + "scalars"] = np.array([getattr(m, k) for k in scalars], dtype=float)
            print("g18:", tag, "step", n, "done", flush=True)
        for name, d in m.diagnostics.items():
            out["%s_diag_%s" % (tag, name)] = np.asarray(d["value"], dtype=float)

    # UnCoupledModel: BASELINE config 5, member 3 (niwqg_amd/ensemble.py config5_member)
    kw = notebook_kwargs(nx, True, 100, tdiags=10)
    m = UnCoupledModel.Model(**kw)
    m.set_q(1e-5 * np.random.default_rng(3).standard_normal((nx, nx)))
    m.set_phi(0.1 * ic.WavePacket(m, k=3 * K0, l=0, R=L / 6, x0=L / 2, y0=L / 2))
    record("unc", m, ["q", "phi", "qh", "phih", "phix", "phiy"], ["Ke", "Pw", "Kw"])
    del m
    # YBJModel: steady dipole, wave packet on a uniform wave (g8's state at size)
    kw = notebook_kwargs(nx, True, 100, tdiags=10)
    kw.update(nu4w=3e9 * (64.0 / nx) ** 4, muw=1e-7)
    m = YBJModel.Model(**kw)
    q0 = ic.LambDipole(m, U=U0, R=2 * np.pi / K0)
    out["ybj_q0_sub"], out["ybj_q0_norm"] = q0[::nx // 64, ::nx // 64].copy(), np.array(np.linalg.norm(q0))
    m.set_q(q0)
    m.set_phi(0.2 * ic.WavePacket(m, k=2 * K0, l=K0, R=L / 6, x0=L / 2, y0=L / 2) + 0.05)
    record("ybj", m, ["phi", "phih", "phix", "phiy"], ["Ke", "Pw", "Kw"])
    del m
    # QGModel with beta and the passive scalar (g10's state at size)
    dt = 0.05 * TE * 128 / nx / 2
    m = QGModel.Model(L=L, nx=nx, tmax=(100 - 0.5) * dt, dt=dt, twrite=10 ** 9, nu4=7.5e8 * (256.0 / nx) ** 4, nu=5.0,
                      mu=1e-8, use_filter=True, U=-U0, tdiags=10, beta=2e-11, passive_scalar=True,
                      nu4c=3e9 * (64.0 / nx) ** 4, nuc=2.0, muc=1e-8, save_to_disk=False)
    out["qgc_params"] = np.array([dt, m.nu4, m.nu4c])
    m.set_q(q0)
    m.set_c(np.sin(2 * np.pi * 3 * m.x / L) * np.cos(2 * np.pi * 2 * m.y / L) + 0.3)
    record("qgc", m, ["q", "c", "qh", "ch"], ["Ke", "cvar", "C2", "gradC2"])
    save("g18_families_1024_100steps.npz", **out)


def g17():
    """round 4: grids that are not powers of two (the reference takes any nx: niwqg/Kernel.py:100-103, numpy.fft any length).
    CoupledModel and QGModel, LambDipole, filter on, 96^2 and 192^2, after 1, 10 and 100 steps: whole fields."""
    out = {}
    for nx in (96, 192):
        kw = notebook_kwargs(nx, True, 100, tdiags=10 ** 9)
        m = CoupledModel.Model(**kw)
        q0, phi0 = lamb_and_uniform(m)
        m.set_q(q0)
        m.set_phi(phi0)
        out["c%d_q0" % nx] = q0
        for n in (1, 10, 100):
            step_to(m, n)
            t = "c%d_s%d_" % (nx, n)
            out[t + "q"], out[t + "phi"], out[t + "qh"], out[t + "phih"] = m.q.copy(), m.phi.copy(), m.qh.copy(), m.phih.copy()
            out[t + "budgets"] = np.array([m.Ke, m.Pw, m.Kw])
        dt = 0.05 * TE * 128 / nx
        m = QGModel.Model(L=L, nx=nx, tmax=1e30, dt=dt, twrite=10 ** 9, nu4=7.5e8 * (128.0 / nx) ** 4, use_filter=True, save_to_disk=False,
                          U=-U0, tdiags=10 ** 9)
        m.set_q(q0)
        out["qg%d_params" % nx] = np.array([dt, m.nu4])
        for n in (1, 10, 100):
            step_to(m, n)
            t = "qg%d_s%d_" % (nx, n)
            out[t + "q"], out[t + "qh"], out[t + "Ke"] = m.q.copy(), m.qh.copy(), np.array(m.Ke)
    save("g17_non_power_of_two.npz", **out)


def attribute_inventory(out, prefix, make):
    for tag, mod in (("coupled", CoupledModel), ("uncoupled", UnCoupledModel), ("qg", QGModel), ("ybj", YBJModel)):
        tag = prefix + tag
        m = make(mod)
        num, txt, arr = {}, {}, {}
        for k, v in m.__dict__.items():
            if isinstance(v, (bool, np.bool_)):
                num[k] = float(v)
            elif isinstance(v, (int, float, np.integer, np.floating)):
                num[k] = float(v)
            elif isinstance(v, str):
                txt[k] = v
            elif isinstance(v, np.ndarray):
                arr[k] = v
        out[tag + "_num_names"] = np.array(sorted(num), dtype="U32")
        out[tag + "_num_values"] = np.array([num[k] for k in sorted(num)])
        out[tag + "_txt_names"] = np.array(sorted(txt), dtype="U32")
        out[tag + "_txt_values"] = np.array([txt[k] for k in sorted(txt)], dtype="U64")
        names = sorted(arr)
        out[tag + "_arr_names"] = np.array(names, dtype="U32")
        out[tag + "_arr_shapes"] = np.array([list(arr[k].shape) + [0] * (2 - arr[k].ndim) for k in names])
        out[tag + "_arr_dtypes"] = np.array([str(arr[k].dtype) for k in names], dtype="U16")
        cs = []
        for k in names:
            a = arr[k].astype(complex).ravel()
            w = np.cos(0.37 * np.arange(a.size))
            cs.append([a.sum(), (a * w).sum()])
        out[tag + "_arr_checksums"] = np.array(cs)
        out[tag + "_other_names"] = np.array(sorted(k for k in m.__dict__ if k not in num and k not in txt and k not in arr), dtype="U32")


if __name__ == "__main__":
    which = sys.argv[1:] or ["g1", "g2", "g3", "g4", "g6", "g7", "g8", "g9", "g10"]
    for w in which:
        globals()[w]()
