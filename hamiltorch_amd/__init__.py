"""hamiltorch_amd -- MI355X-native HMC / RMHMC sampling engine with the hamiltorch API.

Exports the names of ``hamiltorch/__init__.py`` (reference lines 3-4) so that
``import hamiltorch_amd as hamiltorch`` is a drop-in for the sampling path.
"""
#: what the reference exports (hamiltorch/__init__.py:1): scripts that gate on ``hamiltorch.__version__`` see the API level this package
#: mirrors; the package's own release number is ``__amd_version__``
__version__ = '0.4.1'
__reference_version__ = __version__
__amd_version__ = '0.5.0'

from . import util  # noqa: F401
from .samplers import (sample, sample_model, predict_model, sample_split_model,  # noqa: F401
                       Sampler, Integrator, Metric)
from .util import set_random_seed  # noqa: F401
from .models import GaussianTarget  # noqa: F401
from . import samplers  # noqa: F401
