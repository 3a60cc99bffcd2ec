"""Independent ensemble members on one GPU (BASELINE config 5: 64 UnCoupledModel 1024^2 members, 8 per GPU, no
collective in the data path; SURVEY section 8e "replicas only").

Every member is a full model with its own device context, i.e. its own HIP stream: ``Ensemble.step(n)`` queues n
steps on every member before it waits for any of them, so the (short, latency-bound at 1024^2) kernels of different
members overlap on the device.  Members are assigned to ranks with ``distributed.shard_members``.
"""
import numpy as np

from .distributed import shard_members


class Ensemble(object):
    def __init__(self, make_member, n_members, rank=0, world=1):
        """make_member(j) -> model with q and phi set (member id j decides its seed)."""
        self.ids = shard_members(n_members, rank, world)
        self.members = [make_member(j) for j in self.ids]

    def step(self, nsteps):
        """nsteps steps on every member.  Stretches without host-visible events (status line, diagnostics tick --
        Kernel._quiet_steps) are queued on all members before any of them is waited for; a step that ends in an
        event is taken through the member's own _step_forward, exactly as Kernel.run() does (the tick at tc == 0
        refreshes phix/phiy and so belongs to the trajectory: quirk Q1)."""
        left = [nsteps] * len(self.members)
        while any(left):
            quiet = [min(m._quiet_steps(n), n) if n else 0 for m, n in zip(self.members, left)]
            for m, q in zip(self.members, quiet):         # asynchronous: one stream per member
                if q:
                    m._ctx.step(q)
            for i, (m, q) in enumerate(zip(self.members, quiet)):
                if q:
                    m._ctx.sync()
                    for _ in range(q):                    # the reference's float clock (Kernel.py:198,:588)
                        m.tc += 1
                        m.t += m.dt
                    m._after_steps()
                    left[i] -= q
            for i, m in enumerate(self.members):
                if left[i] and m._quiet_steps(left[i]) == 0:
                    m._step_forward()
                    left[i] -= 1

    def run(self):
        """Every member to its own tmax; diagnostics ticks / status lines fire per member as in Kernel.run()."""
        for m in self.members:
            m.run()

    def member_steps(self):
        return sum(m.tc for m in self.members)


def config5_member(j, nx=1024, device=0, **overrides):
    """Member j of BASELINE config 5: UnCoupledModel 1024^2, q = 1e-5 randn(default_rng(j)),
    phi = 0.1 * WavePacket(k=3 k0, R=L/6, x0=y0=L/2)."""
    from . import UnCoupledModel, InitialConditions as ic
    L = 2 * np.pi * 200e3
    k0 = 10 * (2 * np.pi / L)
    U0 = 0.1
    Te = 1.0 / (U0 * k0)
    kw = dict(L=L, nx=nx, tmax=1e30, dt=0.025 * Te * 128 / nx, m=2 * np.pi / 280.0, N=0.01, f=1e-4, twrite=10 ** 9,
              tdiags=10 ** 9, nu4=5e11 * (128.0 / nx) ** 4, nu4w=0.0, nu=20, nuw=50.0, mu=0.0, muw=0.0,
              use_filter=True, U=-U0, device=device, slab=False)      # members never share a simulation
    kw.update(overrides)
    m = UnCoupledModel.Model(**kw)
    m.set_q(1e-5 * np.random.default_rng(j).standard_normal((nx, nx)))
    m.set_phi(0.1 * ic.WavePacket(m, k=3 * k0, l=0, R=L / 6, x0=L / 2, y0=L / 2))
    return m
