"""amplisolve_amd -- MI355X-native AmpliSolve error-estimation + Poisson-calling hot path.

The product is native: HIP kernels behind include/amplisolve_hip.h and a C++ host
behind include/amplisolve_host.h (plus the two drop-in executables under bin/).
This Python package is the thin test / bench harness over those C ABIs.
"""
from ._lib import AmpliError, AmpliNoDevice, hip_lib, host_lib  # noqa: F401
from .api import Context, ErrorTable, NT  # noqa: F401
