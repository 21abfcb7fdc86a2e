"""Drop-in replacements for the reference's models/modules/warp.py operator modules.

Same constructor / forward signatures and the same persistent buffer names (so a reference
``state_dict`` loads unchanged), but ``forward`` launches hand-written gfx950 kernels through
the C ABI (include/waldo_hip.h).  There is no CPU path: calling ``forward`` on CPU tensors or
without the built library raises.
"""
import torch
import torch.nn as nn

from .. import functional as WF
from ..tools.utils import get_gaussian_kernel, get_grid


def kernel_distance(pts_1, pts_2, eps=1e-8):
    """TPS radial basis phi = 0.5 * d * log(d + eps) on the expanded squared distance
    d = |a|^2 + |b|^2 - 2 a.b  (reference: models/modules/warp.py:15-18; the expanded form is
    kept because it fixes the fp32 rounding of the init-time buffers)."""
    sq1 = (pts_1 * pts_1).sum(dim=-1, keepdim=True)          # (N, 1)
    sq2 = (pts_2 * pts_2).sum(dim=-1, keepdim=True).t()      # (1, M)
    d = sq1 + sq2 - 2 * pts_1 @ pts_2.t()
    return 0.5 * d * d.add(eps).log()


class TPSWarp(nn.Module):
    """Thin-plate-spline sampling-grid synthesis (reference: models/modules/warp.py:21-55).

    ``forward(src_pts)``: (B, N, 2) control points -> (B, H, W, 2) sampling grid, computed as
    ``tgt_grid_repr @ (inverse_kernel @ [src_pts; 0])`` -- the reference's association -- by
    ``waldo_tps_mapping_fwd`` + ``waldo_tps_grid_fwd``.

    Buffers ``inverse_kernel`` (N+3, N+3), ``pad`` (3, 2), ``tgt_grid_repr`` (HW, N+3) are
    persistent with the reference's names/shapes; ``basis_t`` (N+3, HW) is the transposed copy
    the kernels read (coalesced over pixels), non-persistent and rebuilt on load.
    """

    def __init__(self, tgt_height, tgt_width, tgt_pts):
        super().__init__()
        self.tgt_shape = [tgt_height, tgt_width]
        tgt_pts = tgt_pts.float()
        n = tgt_pts.size(0)
        fwd = torch.zeros(n + 3, n + 3)
        fwd[:n, :n] = kernel_distance(tgt_pts, tgt_pts)
        fwd[:n, n] = 1
        fwd[n, :n] = 1
        fwd[:n, n + 1:] = tgt_pts
        fwd[n + 1:, :n] = tgt_pts.t()
        # torch.inverse returns a column-major result: stored row-major (same values, same state-dict
        # entry), or every call of the kernels would first copy it into a contiguous temporary
        inverse_kernel = torch.inverse(fwd).contiguous()
        raster = get_grid(tgt_height, tgt_width).view(-1, 2)
        rep = torch.cat([kernel_distance(raster, tgt_pts),
                         torch.ones(tgt_height * tgt_width, 1), raster], dim=1)
        self.register_buffer("inverse_kernel", inverse_kernel)
        self.register_buffer("pad", torch.zeros(3, 2))
        self.register_buffer("tgt_grid_repr", rep)
        self.register_buffer("basis_t", rep.t().contiguous(), persistent=False)
        self.register_load_state_dict_post_hook(TPSWarp._refresh_basis)

    @staticmethod
    def _refresh_basis(module, incompatible_keys):
        module.basis_t = module.tgt_grid_repr.t().contiguous()

    def forward(self, src_pts):
        h, w = self.tgt_shape
        return WF.tps_grid(self.inverse_kernel, self.basis_t, src_pts, h, w)


class InverseWarp(nn.Module):
    """Grid inversion by forward splat + hole filling (reference: models/modules/warp.py:58-174).

    ``forward(src_grid, niter=5, pad=True, erode=True)``: (B, Hs, Ws, 2) layer->image grid ->
    (B, H, W, 2) image->layer grid, by ``waldo_inverse_warp_fwd`` (winner election with an integer
    atomicMin instead of the reference's two sorts; Jacobi fill passes; erosion; crop), with an
    exact backward w.r.t. the displacement values.  Buffers keep the reference's names.

    Differences from the reference, on purpose: among colliding samples the first in tie-break
    order wins (sample index for ``num_perm == 1``, position in ``perm[p]`` for ``num_perm > 1``;
    the reference's result under a stable sort -- its own depends on torch.sort's implementation);
    ``pad=False`` and an even ``kernel_size`` (which fail with shape errors in the reference) raise; every odd
    ``kernel_size`` works (3, the default every script uses, in one launch each way).  ``num_perm > 1``
    (warp.py:91-111, unused by every script) runs one inversion per permutation and averages the
    results, which equals averaging the elected fields first (see ``WF.inverse_warp``)."""

    def __init__(self, src_height, src_width, tgt_height, tgt_width, kernel_size=3, num_perm=1):
        super().__init__()
        self.kernel_size = kernel_size
        self.src_shape = [src_height, src_width]
        self.tgt_shape = [tgt_height, tgt_width]
        self.num_perm = num_perm
        self.register_buffer("kernel", get_gaussian_kernel(kernel_size).view(1, 1, kernel_size, kernel_size))
        self.register_buffer("src_grid", get_grid(src_height, src_width))
        self.register_buffer("tgt_grid", get_grid(tgt_height, tgt_width))
        self.register_buffer("x_grid", torch.arange(tgt_width).view(1, -1).repeat(tgt_height, 1).view(1, -1).float())
        self.register_buffer("y_grid", torch.arange(tgt_height).view(-1, 1).repeat(1, tgt_width).view(1, -1).float())
        self.register_buffer("perm", torch.stack([torch.randperm(tgt_height * tgt_width) for _ in range(num_perm)]))

    def forward(self, src_grid, niter=5, pad=True, erode=True):
        if self.kernel_size % 2 == 0:
            raise ValueError("InverseWarp: an even kernel_size changes the raster under conv2d(padding=k // 2) and "
                             "fails in the reference (warp.py:140-146); not supported")
        if not pad:
            raise ValueError("InverseWarp: pad=False fails with a shape error in the reference "
                             "(warp.py:169-173); not supported")
        return WF.inverse_warp(src_grid, self.src_grid[0], self.tgt_grid[0], self.kernel.reshape(-1),
                               niter=niter, erode=erode, perm=self.perm if self.num_perm > 1 else None)
