"""More seeds of the option fuzz and the call-sequence fuzz on grids without a fused plan (tests/test_gpu_anysize.py runs a fixed set):
    python tools/diag/anysize_fuzz.py [first_seed] [count]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import logging
import numpy as np
logging.disable(logging.CRITICAL)
import test_gpu_models as T

first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
count = int(sys.argv[2]) if len(sys.argv) > 2 else 40
sizes = [96, 100, 150, 200, 384, 48, 20, 36, 250, 12]
bad = 0
for s in range(first, first + count):
    nx = sizes[s % len(sizes)]
    for name, fn in (("config", T.random_configuration_against_the_oracle), ("calls", T.random_call_sequence_against_the_oracle)):
        try:
            with np.errstate(all="ignore"):
                fn(s, nx_force=nx)
            print("seed %d nx %d %s ok" % (s, nx, name), flush=True)
        except AssertionError as e:
            bad += 1
            print("seed %d nx %d %s FAILED: %s" % (s, nx, name, str(e)[:400]), flush=True)
print("%d failures" % bad)
sys.exit(1 if bad else 0)
