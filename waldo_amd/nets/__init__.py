from .lvd import Warper, compute_occ, reduce_comp, gather_time, scale  # noqa: F401
from .wif import WIF  # noqa: F401
