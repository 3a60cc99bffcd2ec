"""Which library calls one any-size CoupledModel step makes (counts per kind): python tools/diag/anysize_opmix.py [nx]"""
import collections
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import logging
import numpy as np
logging.disable(logging.CRITICAL)
import niwqg_amd
from niwqg_amd import _anysize as A
import bench

nx = int(sys.argv[1]) if len(sys.argv) > 1 else 1536
m = niwqg_amd.CoupledModel.Model(**bench.c3_kwargs(nx, "coupled"))
rng = np.random.default_rng(0)
m.set_q(1e-5 * rng.standard_normal((nx, nx)))
m.set_phi(0.05 * (rng.standard_normal((nx, nx)) + 1j * rng.standard_normal((nx, nx))))
m._step_etdrk4()
names = {v: k for k, v in vars(A).items() if k.startswith("EW_")}
rnames = {v: k for k, v in vars(A).items() if k.startswith("RD_")}
cnt = collections.Counter()
ew0, red0, fft0 = A.Plane._ew, A.Plane._reduce, m._fft


def ew(self, op, a, b=None, c=None, **kw):
    cnt[names[op]] += 1
    return ew0(self, op, a, b, c, **kw)


def red(self, op, other=None):
    cnt[rnames[op]] += 1
    return red0(self, op, other)


def fft(a, inverse=False):
    cnt["fft2" if not inverse else "ifft2"] += 1
    return fft0(a, inverse)


A.Plane._ew, A.Plane._reduce, m._fft = ew, red, fft
m._step_etdrk4()
for k, v in cnt.most_common():
    print("%-14s %4d" % (k, v))
print("total element-wise %d, reductions %d, transforms %d" % (sum(v for k, v in cnt.items() if k.startswith("EW_")),
                                                                sum(v for k, v in cnt.items() if k.startswith("RD_")), cnt["fft2"] + cnt["ifft2"]))
