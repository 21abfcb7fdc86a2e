"""waldo_amd -- MI355X (gfx950) native WIF warp/composite hot path of 16lemoing/waldo.

Only what the path needs: ``csrc/`` (HIP kernels + C ABI, see include/waldo_hip.h), the ctypes
binding (``_lib``), autograd wrappers (``functional``) and host-side mirrors of the reference's
operator modules (``modules.warp``, ``nets.lvd``, ``nets.wif``).
"""
from . import functional  # noqa: F401
from .modules.warp import TPSWarp, InverseWarp  # noqa: F401

__version__ = "0.1.0"
