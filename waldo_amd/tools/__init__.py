from .utils import get_grid, get_gaussian_kernel  # noqa: F401
