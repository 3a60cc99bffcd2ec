"""How far apart are device and oracle on the diagnostics the randomized test skips (nearly vanishing or cancellation-prone
integrals)?  For seeds 0..N-1 of test_randomly_drawn_configurations_against_the_oracle: worst |a - b| / max|b| per name.
    python tools/diag/fuzz_skipped_diags.py N"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import logging
import numpy as np
logging.disable(logging.CRITICAL)
import test_gpu_models as T

worst = {}
for seed in range(int(sys.argv[1])):
    m, o, kind, kw, rng, tag = T.draw_configuration(seed)
    for _ in range(6):
        o._step_forward()
    T.steps(m, 6)
    for name in ("skew", "conc_niw", "Gamma_c", "pi", "gamma_r", "gamma_a", "xi_r", "xi_a"):
        if name not in o.diagnostics or name not in m.diagnostics:
            continue
        a = np.atleast_1d(np.asarray(m.diagnostics[name]["value"], float))
        b = np.atleast_1d(np.asarray(o.diag(name), float))
        if a.shape != b.shape or b.size == 0:
            print("SHAPE", tag, name, a.shape, b.shape)
            continue
        e = float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))
        if e > worst.get(name, (0, ""))[0]:
            worst[name] = (e, "%s: device %s oracle %s" % (tag, a[-1], b[-1]))
for k, v in worst.items():
    print("%-10s worst relative difference %.2e   %s" % (k, v[0], v[1]))
