"""niwqg_amd: MI355X-native drop-in for the time-stepping path of cesar-rocha/niwqg.

    from niwqg_amd import CoupledModel, UnCoupledModel, QGModel, InitialConditions
    m = CoupledModel.Model(nx=4096, ...); m.set_q(q); m.set_phi(phi); m.run()

mirrors ``from niwqg import ...`` (ref: niwqg/__init__.py:1-5).  The HIP library is built in-tree by
``niwqg_amd.build()``; there is no CPU fallback.
"""
__version__ = '0.1'

from ._lib import build   # noqa: F401
from . import Diagnostics   # noqa: F401
from . import InitialConditions   # noqa: F401
from . import Saving   # noqa: F401
from . import Kernel   # noqa: F401
from . import CoupledModel   # noqa: F401
from . import UnCoupledModel   # noqa: F401
from . import QGModel   # noqa: F401
from . import YBJModel   # noqa: F401
