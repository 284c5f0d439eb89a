"""gmrfx -- host-side mirror (Python, over ctypes) of the Julia plug-in that attaches
libgmrfx.so to GaussianMarkovRandomFields.jl's solver seams. See INTEGRATION.md."""
from .backend import MI355XBackend, SymbolicInfo  # noqa: F401
from .ordering import PinDenseColumns, ordering_permutation  # noqa: F401
from ._lib import GmrfxError, NoDeviceError, PosDefException  # noqa: F401
from .kron import KroneckerWorkspace  # noqa: F401
from .spacetime import spacetime_coords, spacetime_precision  # noqa: F401
