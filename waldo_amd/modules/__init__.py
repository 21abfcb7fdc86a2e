from .warp import TPSWarp, InverseWarp, kernel_distance  # noqa: F401
