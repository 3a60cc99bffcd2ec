import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np
from oracle import niwqg_oracle as O
from test_oracle_golden import notebook_kwargs, rel, L, K0, U0
from test_gpu_primitives import make_ctx
from niwqg_amd import _lib
nx=64
ctx, orc = make_ctx("coupled", nx)
q0 = O.lamb_dipole(orc.grid, U=U0, R=2 * np.pi / K0)
phi0 = 0.2 * O.wave_packet(orc.grid, k=3 * K0, l=K0, R=L / 6, x0=L / 2, y0=L / 2)
orc.set_phi(phi0); orc.set_q(q0)
ctx.set_phi(phi0); ctx.set_q(q0)
h = nx//2+1
for name,fid,ref in [("qh",_lib.F_QH,orc.qh[:,:h]),("phih",_lib.F_PHIH,orc.phih),("phi",_lib.F_PHI,orc.phi),("qwh",_lib.F_QWH,orc.qwh[:,:h]),
    ("ph",_lib.F_PH,orc.ph[:,:h]),("q",_lib.F_Q,orc.q),("p",_lib.F_P,orc.p),("u",_lib.F_U,orc.u),("v",_lib.F_V,orc.v),("qw",_lib.F_QW,orc.qw),("phix",_lib.F_PHIX,orc.phix),("phiy",_lib.F_PHIY,orc.phiy)]:
    a=ctx.field(fid); d=np.abs(a-ref); i=np.unravel_index(d.argmax(), d.shape)
    print(name, rel(a,ref), 'worst at', i, a[i], ref[i])
wj = np.fft.rfft2((1j * (np.conj(orc.phix) * orc.phiy - np.conj(orc.phiy) * orc.phix)).real)
a=ctx.wave_jacobian(); d=np.abs(a-wj); i=np.unravel_index(d.argmax(), d.shape)
print('wavejac', rel(a,wj), i, a[i], wj[i])
f1, f2 = ctx.products_uq_vq()
print('uq', rel(f1, np.fft.rfft2(orc.u * orc.q)), 'vq', rel(f2, np.fft.rfft2(orc.v * orc.q)))
print('adv', rel(ctx.advection_phi(), np.fft.fft2(orc.u * orc.phix + orc.v * orc.phiy)))
