from .lvd import (Warper, compute_occ, reduce_comp, gather_time, scale,  # noqa: F401
                  estimate_alpha_grid_occ, decode_output)
from .wif import WIF  # noqa: F401
