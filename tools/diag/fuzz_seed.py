"""Re-run one seed of tests/test_gpu_models.py::test_randomly_drawn_configurations_against_the_oracle and show WHERE the device and
the oracle differ: the ETDRK4 planes (the reference's contour means cancel catastrophically where |c dt| ~ 1: DESIGN.md section 6)
and the spectral location of the field error.      python tools/diag/fuzz_seed.py SEED"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle import niwqg_oracle as O
from test_oracle_golden import rel, L, K0, U0, TE, F0, NB, MZ
import niwqg_amd as mods

seed = int(sys.argv[1])
rng = np.random.default_rng(1000 + seed)
kind = ["coupled", "uncoupled", "qg", "ybj", "coupled", "qg"][seed % 6]
nx = int(rng.choice([64, 128, 256, 512] if kind != "coupled" else [64, 128, 256]))
filt = int(rng.integers(0, 3))
if kind == "qg" and filt == 1:
    filt = 0
dt = 0.025 * TE * 128 / nx * float(rng.choice([0.5, 1.0]))
tdiags = int(rng.choice([1, 2, 10 ** 9]))
kw = dict(L=L, nx=nx, tmax=1e30, dt=dt, twrite=10 ** 9, tdiags=tdiags, use_filter=filt == 0, dealias=filt == 1,
          U=float(rng.choice([0.0, -U0, 0.5 * U0])), nu4=5e11 * (128.0 / nx) ** 4 * float(rng.uniform(0.2, 2.0)),
          nu=float(rng.choice([0.0, 20.0])), mu=float(rng.choice([0.0, 1e-8])))
if kind == "qg":
    passive = bool(rng.integers(0, 2))
    kw.update(beta=float(rng.choice([0.0, 2e-11])), passive_scalar=passive, nu4c=kw["nu4"] * 0.5, nuc=2.0, muc=1e-8)
    m, o = mods.QGModel.Model(**kw), O.QGOracle(**kw)
else:
    kw.update(m=MZ, N=NB, f=F0, nuw=float(rng.choice([0.0, 50.0])), nu4w=float(rng.choice([0.0, 0.1])) * kw["nu4"],
              muw=float(rng.choice([0.0, 2e-8])))
    cls = {"coupled": mods.CoupledModel, "uncoupled": mods.UnCoupledModel, "ybj": mods.YBJModel}[kind]
    m, o = cls.Model(**kw), O.NIWQGOracle(kind, **kw)
print("seed", seed, kind, {k: v for k, v in kw.items() if k not in ("L", "tmax", "twrite", "m", "N", "f")})
q0 = O.lamb_dipole(o.grid, U=U0, R=2 * np.pi / K0) + 2e-6 * rng.standard_normal((nx, nx))
for x in (m, o):
    x.set_q(q0)
if kind != "qg":
    phi0 = 0.1 * O.wave_packet(o.grid, k=2 * K0, l=K0, R=L / 6, x0=L / 2, y0=L / 2) + 0.02 * (
        rng.standard_normal((nx, nx)) + 1j * rng.standard_normal((nx, nx)))
    for x in (m, o):
        x.set_phi(phi0)
elif kw["passive_scalar"]:
    c0 = 1.0 + 0.3 * rng.standard_normal((nx, nx))
    for x in (m, o):
        x.set_c(c0)
for name, key in (("expch", "E"), ("expch_h", "Eh"), ("Qh", "Q"), ("f0", "f0"), ("fab", "fab"), ("fc", "fc")):
    a, b = getattr(m, name), o.coef_q[key]
    ncol = min(a.shape[1], b.shape[1])
    a, b = a[:, :ncol], b[:, :ncol]
    d = np.abs(a - b) / np.maximum(np.abs(b), 1e-300)
    i = np.unravel_index(np.argmax(d), d.shape)
    print("  %-8s l2 rel %.2e   worst entry %.2e at (l, k) = %s (|value| %.2e)" % (name, rel(a, b), d[i], i, abs(b[i])))
import copy
o2 = O.QGOracle(**kw) if kind == "qg" else O.NIWQGOracle(kind, **kw)
o2.set_q(q0 * (1.0 + 1e-15 * rng.standard_normal((nx, nx))))
if kind != "qg":
    o2.set_phi(phi0)
elif kw["passive_scalar"]:
    o2.set_c(c0)
for n in range(1, 7):
    o._step_forward()
    o2._step_forward()
    m._step_forward()
    print("   oracle vs oracle with q0 perturbed by 1e-15 relative: rel q %.2e; max|q| %.3e" % (rel(o2.q, o.q), np.abs(o.q).max()))
    e = np.abs(m.qh - o.qh)
    i = np.unravel_index(np.argmax(e), e.shape)
    print("step %d: rel q %.2e qh %.2e; worst |dqh| %.2e at (l, k) = %s where |qh| = %.2e" % (n, rel(m.q, o.q), rel(m.qh, o.qh), e[i], i, abs(o.qh[i])))
