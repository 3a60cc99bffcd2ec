"""Replay seeds of tests/test_gpu_models.py::test_randomly_drawn_call_sequences_against_the_oracle without stopping at the first
mismatch: per action, the relative difference of what the call returned and of the state it leaves (u, v, phix, phiy, q, ph, phi).
    python tools/diag/call_sequence.py SEED [SEED ...]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import logging
import numpy as np
logging.disable(logging.CRITICAL)
import test_gpu_models as T
from test_gpu_models import rel


def state(m, o, kind):
    out = {}
    names = ["u", "v", "q", "ph"] + (["phi", "phix", "phiy"] if kind != "qg" else [])
    for nm in names:
        try:
            out[nm] = rel(getattr(m, nm), getattr(o, nm))
        except Exception as e:       # noqa
            out[nm] = float("nan")
    return " ".join("%s %.0e" % (k, v) for k, v in out.items() if not (v < 1e-10))


for seed in [int(a) for a in sys.argv[1:]]:
    arng = np.random.default_rng(9000 + seed)
    m, o, kind, kw, rng, tag = T.draw_configuration(seed, order_rng=arng)
    nx = kw["nx"]
    wave = kind in ("coupled", "uncoupled")
    actions = ["step", "step", "step", "read", "energies", "cfl", "set_q"]
    if kind != "ybj":
        actions += ["jq", "jq"]
    if wave:
        actions += ["jphi", "set_phi", "pe"]
    if kind == "coupled":
        actions += ["jcc"]
    print("seed", seed, tag, "| initial state:", state(m, o, kind) or "equal")
    for n in range(12):
        a = str(arng.choice(actions))
        r = ""
        if a == "step":
            o._step_forward(); m._step_forward()
        elif a == "jq":
            r = "%.1e" % rel(m.jacobian_psi_q(), o.jacobian_psi_q())
        elif a == "jphi":
            r = "%.1e" % rel(m.jacobian_psi_phi(), o.jacobian_psi_phi())
        elif a == "jcc":
            if kind == "coupled":
                r = "%.1e" % rel(m.jacobian_phic_phi(), o.jacobian_phic_phi())
        elif a == "energies":
            r = "%.1e" % (abs(m._calc_ke_qg() - o._calc_ke_qg()) / abs(o._calc_ke_qg()))
            if kind != "qg":
                r += " %.1e" % (abs(m._calc_ke_niw() - o._calc_ke_niw()) / abs(o._calc_ke_niw()))
        elif a == "pe":
            r = "%.1e" % (abs(m._calc_pe_niw() - o._calc_pe_niw()) / abs(o._calc_pe_niw()))
        elif a == "cfl":
            r = "%.1e" % (abs(m._calc_cfl() - o._calc_cfl()) / abs(o._calc_cfl()))
        elif a == "set_q":
            q1 = 1e-6 * arng.standard_normal((nx, nx)) + 0.5 * np.asarray(o.q)
            for x in (m, o):
                x.set_q(q1)
        elif a == "set_phi":
            p1 = 0.05 * (arng.standard_normal((nx, nx)) + 1j * arng.standard_normal((nx, nx))) + 0.5 * np.asarray(o.phi)
            for x in (m, o):
                x.set_phi(p1)
        elif a == "read":
            names = ["q", "qh", "ph"] + (["phi", "phih", "u", "v"] if kind != "qg" else []) + (["p"] if kind != "ybj" else [])
            nm = str(arng.choice(names))
            r = "%s %.1e" % (nm, rel(getattr(m, nm), getattr(o, nm)))
        print("   %2d %-8s %-18s state differs: %s" % (n, a, r, state(m, o, kind) or "-"))
