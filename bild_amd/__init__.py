"""
bild_amd -- MI355X-native Rouse Kalman-filter likelihood for BILD
(Bayesian Inference of Looping Dynamics).

One hot path, behind the reference's own interfaces:

* `models.MultiStateRouse.logL(profile, traj)`      (reference bild/models.py:265-278)
* `amis.FixedkSampler.logL(ss, thetas)`             (reference bild/amis.py:717-739)

both evaluated by hand-written HIP kernels for gfx950 through the C ABI declared in
``include/bild_amd.h`` (``bild_amd/libbild_amd.so``).  There is no CPU fallback.
"""
from . import rouse, profiles, util, trajectory, models, amis, choicesampler, core, postproc  # noqa: F401
from .core import sample, sample_many, SamplingResults  # noqa: F401
from .models import MultiStateModel, MultiStateRouse, FactorizedModel  # noqa: F401
from .amis import FixedkSampler, Dirichlet, CFC  # noqa: F401
from .profiles import Loopingprofile  # noqa: F401
from .trajectory import Trajectory  # noqa: F401

__version__ = '0.1.0'
