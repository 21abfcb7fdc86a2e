"""CPU oracle for the WIF warp/composite hot path -- TEST INFRASTRUCTURE, NOT PRODUCT.

A plain PyTorch-CPU restatement of the reference's algorithm for the path named in
BASELINE.json (TPS grid -> bilinear backward warp -> occlusion/soft-alpha composite and the
Warper / WIF glue around it).  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this module; the product (``waldo_amd``) never does and fails
loudly when its HIP library is missing.

Parity status: PINNED.  The reference ships no tests for this path, so
the oracle is pinned against outputs of the reference itself, generated in the build container
by importing the reference's modules (``oracle/make_golden.py`` -> ``tests/golden/*.npz``) and,
when /root/reference is present, by live differential tests (``tests/test_oracle_live.py``).

Every function cites the reference file:line it restates.  The arithmetic is written from the
formulas (loops over layers, explicit 4-corner bilinear taps) rather than by re-using the
reference's broadcasting expressions, so that an agreement between the two is evidence.
``torch.nn.functional.grid_sample`` / ``interpolate`` are the platform (PyTorch), not the
reference; ``bilinear_sample`` below restates grid_sample from its documented formula and the
tests check the two against each other.
"""
import math

import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------------------
# grids and small helpers
# --------------------------------------------------------------------------------------


def get_grid(height, width, dtype=torch.float32):
    """Pixel-centre normalised coordinates, last dim (x, y).  tools/utils.py:293-297.

    x_j = -1 + (2j+1)/W, y_i = -1 + (2i+1)/H -- the texel centres of
    grid_sample(align_corners=False).  The reference builds them with torch.linspace in fp32;
    we do the same so that the buffers are bit-identical.
    """
    xs = torch.linspace(-1.0 + 1.0 / width, 1.0 - 1.0 / width, width, dtype=dtype)
    ys = torch.linspace(-1.0 + 1.0 / height, 1.0 - 1.0 / height, height, dtype=dtype)
    g = torch.empty(1, height, width, 2, dtype=dtype)
    g[0, :, :, 0] = xs.view(1, width)
    g[0, :, :, 1] = ys.view(height, 1)
    return g


def get_gaussian_kernel(k, sigma_div=6):
    """Normalised k x k Gaussian, sigma = k / sigma_div.  tools/utils.py:273-291."""
    c = torch.arange(k)
    xg = c.repeat(k).view(k, k)
    yg = xg.t()
    mean = (k - 1) / 2.0
    var = (k / sigma_div) ** 2.0
    sq = (xg - mean) ** 2.0 + (yg - mean) ** 2.0
    ker = (1.0 / (2.0 * math.pi * var)) * torch.exp(-sq / (2 * var))
    return ker / ker.sum()


def kernel_distance(p1, p2, eps=1e-8):
    """TPS radial basis 0.5 * d * log(d + eps), d = expanded squared distance.
    models/modules/warp.py:15-18 (the expanded |a|^2+|b|^2-2ab form is kept on purpose:
    it is what fixes the fp32 rounding of the precomputed buffers)."""
    n, m = p1.shape[0], p2.shape[0]
    d = (p1 ** 2).sum(-1).view(n, 1) + (p2 ** 2).sum(-1).view(1, m) - 2 * p1 @ p2.t()
    return 0.5 * d * torch.log(d + eps)


# --------------------------------------------------------------------------------------
# A2: thin-plate-spline grid synthesis
# --------------------------------------------------------------------------------------


def tps_init(height, width, tgt_pts):
    """Buffers of TPSWarp.__init__ (models/modules/warp.py:21-47).

    returns inverse_kernel (N+3, N+3) and tgt_grid_repr (H*W, N+3)."""
    tgt_pts = tgt_pts.float()
    n = tgt_pts.shape[0]
    fk = torch.zeros(n + 3, n + 3)
    fk[:n, :n] = kernel_distance(tgt_pts, tgt_pts)
    fk[:n, n] = 1
    fk[n, :n] = 1
    fk[:n, n + 1:] = tgt_pts
    fk[n + 1:, :n] = tgt_pts.t()
    inverse_kernel = torch.inverse(fk)
    g = get_grid(height, width).view(-1, 2)
    rep = torch.cat([kernel_distance(g, tgt_pts), torch.ones(height * width, 1), g], dim=1)
    return inverse_kernel, rep


def tps_mapping(inverse_kernel, src_pts):
    """mapping = K^-1 [src_pts; 0_{3x2}]  (models/modules/warp.py:52-53)."""
    b = src_pts.shape[0]
    src_pts = src_pts.to(inverse_kernel.dtype)
    x = torch.cat([src_pts, src_pts.new_zeros(b, 3, 2)], dim=1)
    return torch.matmul(inverse_kernel, x)


def tps_grid(inverse_kernel, tgt_grid_repr, src_pts, height, width):
    """TPSWarp.forward (models/modules/warp.py:49-55), keeping the reference's association
    repr @ (K^-1 @ pts)."""
    mapping = tps_mapping(inverse_kernel, src_pts)
    return torch.matmul(tgt_grid_repr, mapping).view(src_pts.shape[0], height, width, 2)


# --------------------------------------------------------------------------------------
# A4/A5: bilinear backward warp (grid_sample, bilinear / zeros / align_corners=False)
# --------------------------------------------------------------------------------------


def bilinear_sample(inp, grid):
    """Restatement of F.grid_sample defaults from the documented formula
    ix = ((x + 1) * W - 1) / 2, 4 taps, out-of-range taps contribute 0.
    inp (N, C, Hi, Wi), grid (N, Ho, Wo, 2) -> (N, C, Ho, Wo).  Differentiable w.r.t. both.
    All F.grid_sample call sites on the path use the defaults, e.g. models/nets/lvd.py:518,548."""
    n, c, hi, wi = inp.shape
    _, ho, wo, _ = grid.shape
    ix = ((grid[..., 0] + 1) * wi - 1) / 2
    iy = ((grid[..., 1] + 1) * hi - 1) / 2
    x0 = torch.floor(ix)
    y0 = torch.floor(iy)
    fx = ix - x0
    fy = iy - y0
    flat = inp.reshape(n, c, hi * wi)
    out = inp.new_zeros(n, c, ho, wo)
    for dy, wy in ((0, 1 - fy), (1, fy)):
        for dx, wx in ((0, 1 - fx), (1, fx)):
            xi = x0 + dx
            yi = y0 + dy
            ok = (xi >= 0) & (xi <= wi - 1) & (yi >= 0) & (yi <= hi - 1)
            lin = (yi.clamp(0, hi - 1) * wi + xi.clamp(0, wi - 1)).long().view(n, 1, ho * wo)
            tap = torch.gather(flat, 2, lin.expand(n, c, ho * wo)).view(n, c, ho, wo)
            out = out + tap * (wx * wy * ok.to(inp.dtype)).unsqueeze(1)
    return out


def grid_sample(inp, grid, explicit=False):
    """The platform op (or its explicit restatement) -- what the reference calls."""
    if explicit:
        return bilinear_sample(inp, grid)
    return F.grid_sample(inp, grid, mode="bilinear", padding_mode="zeros", align_corners=False)


def grid_sample_delta(inp, grid, delta, explicit=False):
    """grid_sample(x + delta, grid) - delta: zero padding becomes "-delta" padding.
    Warper.obj_to_output / bg_to_output, models/nets/lvd.py:548,559."""
    return grid_sample(inp + delta, grid, explicit) - delta


# --------------------------------------------------------------------------------------
# A6: occlusion matrix and soft-alpha composite
# --------------------------------------------------------------------------------------


def compute_occ(occ_score, eps=1e-6):
    """LVD.compute_occ, models/nets/lvd.py:59-68.  occ_score (B, T, No) -> (B, T, No+1, No+1).
    occ[i, j] = s_i / (s_i + s_j) - 0.5 * [i == j], s = exp(-score^2) + eps, then a leading
    column of ones (every object occludes the background) and a leading row of zeros."""
    b, t, no = occ_score.shape
    s = torch.exp(-occ_score ** 2) + eps
    occ = occ_score.new_zeros(b, t, no + 1, no + 1)
    for i in range(no):
        occ[:, :, i + 1, 0] = 1.0
        for j in range(no):
            v = s[:, :, i] / (s[:, :, i] + s[:, :, j])
            if i == j:
                v = v - 0.5
            occ[:, :, i + 1, j + 1] = v
    return occ


def occlusion_product(alpha, occ):
    """alpha'_j = alpha_j * prod_i (1 - alpha_i * occ[i, j]).
    alpha (..., L, h, w) in [0, 1]; occ (..., L, L).  The spec form is LVD.reduce_comp
    (models/nets/lvd.py:109-111); the live uses are lvd.py:651-652, 686, 764-765, 809."""
    nl = alpha.shape[-3]
    outs = []
    for j in range(nl):
        p = torch.ones_like(alpha[..., 0, :, :])
        for i in range(nl):
            p = p * (1 - alpha[..., i, :, :] * occ[..., i, j, None, None])
        outs.append(alpha[..., j, :, :] * p)
    return torch.stack(outs, dim=-3)


def reduce_comp(vid, occ, flow=None):
    """LVD.reduce_comp, models/nets/lvd.py:100-114.
    vid (B, T, L, C+1, H, W) in [-1, 1] (last channel = alpha), occ (B, T, L, L).
    returns composited vid (B, T, C, H, W) in [-1, 1], alpha' (B, T, L, H, W) in [-1, 1]
    and the composited flow (or None)."""
    v = (vid + 1) / 2
    alpha = v[:, :, :, -1].clone()
    alpha[:, :, 0] = 1.0  # background alpha forced to one (lvd.py:105)
    alpha = occlusion_product(alpha, occ)  # B T L H W
    out = (alpha.unsqueeze(3) * v[:, :, :, :-1]).sum(dim=2)
    fl = None
    if flow is not None:
        fl = (alpha[:, :-1].unsqueeze(3) * flow).sum(dim=2)
    return 2 * out - 1, 2 * alpha - 1, fl


# --------------------------------------------------------------------------------------
# the fused synthetic hot path of BASELINE.md section 3 / SURVEY.md 8(d)
# --------------------------------------------------------------------------------------


def warp_composite(layers, src_pts, occ, inverse_kernel, tgt_grid_repr, explicit=False, delta=0.0):
    """TPS grid (A2) -> grid_sample of each 4-channel layer (A4) -> reduce_comp (A6).
    delta: grid_sample(x + delta) - delta as Warper.obj_to_output / bg_to_output (lvd.py:548,559).

    layers (F, L, 4, H, W) in [-1, 1]; src_pts (F*L, K, 2); occ (F, L, L);
    returns rgb (F, 3, H, W), alpha' (F, L, H, W) -- both in [-1, 1]."""
    f, nl, c, h, w = layers.shape
    grid = tps_grid(inverse_kernel, tgt_grid_repr, src_pts, h, w)
    warped = grid_sample_delta(layers.reshape(f * nl, c, h, w), grid, delta, explicit).view(f, 1, nl, c, h, w)
    rgb, alpha, _ = reduce_comp(warped, occ.view(f, 1, nl, nl))
    return rgb[:, 0], alpha[:, 0]


def make_synthetic(frames, nl, h, w, k_side=4, seed=0, sigma=0.05, smooth=0):
    """Synthetic workload of SURVEY.md 8(d): identical inputs for the CPU and GPU legs.
    smooth > 0 replaces the white-noise layers by noise drawn at 1/smooth resolution and
    upsampled bilinearly (natural-image-like spectra: the bilinear interpolant then has no
    O(1) jumps in its derivative from texel to texel, which white noise has)."""
    g = torch.Generator().manual_seed(seed)
    if smooth > 0:
        lo = torch.rand(frames * nl, 4, max(h // smooth, 2), max(w // smooth, 2), generator=g) * 2 - 1
        layers = F.interpolate(lo, size=(h, w), mode="bilinear", align_corners=False)
        layers = layers.view(frames, nl, 4, h, w).contiguous()
    else:
        layers = torch.rand(frames, nl, 4, h, w, generator=g) * 2 - 1
    ctrl = get_grid(k_side, k_side).view(1, k_side * k_side, 2)
    pts = ctrl + sigma * torch.randn(frames * nl, k_side * k_side, 2, generator=g)
    occ = compute_occ(torch.randn(frames, 1, nl - 1, generator=g))[:, 0]
    inv, rep = tps_init(h, w, ctrl[0])
    return layers, pts, occ, inv, rep


# --------------------------------------------------------------------------------------
# A3: grid inversion (forward splat + hole filling)
# --------------------------------------------------------------------------------------


def inverse_warp(src_grid, tgt_shape, niter=5, pad=True, erode=True, kernel_size=3, perm=None):
    """InverseWarp.forward, models/modules/warp.py:71-174 (perm None: num_perm == 1).

    src_grid (B, Hs, Ws, 2) maps layer space -> image space; the result (B, H, W, 2) maps image
    space -> layer space.  Restated as: (1) displacement, bilinearly resized to the target
    resolution; (2) round-half-even splat of the negated displacement, where among several
    sources landing on one cell the one with the LOWEST source index wins (that is what the
    reference's stable sort + first-of-run mask does, warp.py:113-123); (3) niter Jacobi
    passes: the 4-neighbour ring of the filled set takes the Gaussian-weighted mean of the
    filled 3x3 neighbours (warp.py:135-151); (4) optional erosion (warp.py:153-162);
    (5) unfilled cells get the offset (2W, 2H) px, i.e. sample out of range (warp.py:164-167).
    With perm (P, H*W), P > 1 (warp.py:91-111): step (2) runs once per row with the samples taken
    in that order -- the one standing first in perm[p] wins -- and the P elected fields are
    averaged; the set of occupied cells is the same for every row (the reference takes row 0's).
    Differentiable w.r.t. the displacement values (not the integer cell indices)."""
    b, hs, ws, _ = src_grid.shape
    h, w = tgt_shape
    n = niter
    d = src_grid - get_grid(hs, ws)
    d = F.interpolate(d.permute(0, 3, 1, 2), size=(h, w), mode="bilinear", align_corners=False)
    dx = d[:, 0].reshape(b, -1) * w / 2
    dy = d[:, 1].reshape(b, -1) * h / 2
    xs = torch.arange(w, dtype=torch.float32).repeat(h).view(1, -1)
    ys = torch.arange(h, dtype=torch.float32).repeat_interleave(w).view(1, -1)
    tx = torch.round(xs + dx).long()
    ty = torch.round(ys + dy).long()
    inside = (tx >= 0) & (ty >= 0) & (tx <= w - 1) & (ty <= h - 1)
    cell = torch.where(inside, ty * w + tx, torch.full_like(tx, h * w))  # h*w = dump slot
    # lowest source index wins: process sources in DEcreasing index order so that the last
    # write (lowest index) survives.  index_put_ with accumulate=False is not ordered, so do
    # the winner selection explicitly with a scatter-min over source indices.
    orders = [None] if perm is None else [row.long() for row in perm]
    inv_dx = inv_dy = 0
    for order in orders:
        # position of each sample in the tie-break order (its own index for num_perm == 1)
        pos = torch.arange(h * w)
        if order is not None:
            pos = torch.empty(h * w, dtype=torch.long).scatter_(0, order, torch.arange(h * w))
        first = torch.full((b, h * w + 1), h * w, dtype=torch.long)
        first = first.scatter_reduce(1, cell, pos.view(1, -1).expand(b, -1), reduce="amin",
                                     include_self=True)[:, :h * w]
        mask = first < h * w
        wsafe = first.clamp(max=h * w - 1)
        if order is not None:
            wsafe = order[wsafe]
        inv_dx = inv_dx + torch.where(mask, -torch.gather(dx, 1, wsafe), torch.zeros_like(dx))
        inv_dy = inv_dy + torch.where(mask, -torch.gather(dy, 1, wsafe), torch.zeros_like(dy))
    inv_dx = (inv_dx / len(orders)).view(b, h, w)
    inv_dy = (inv_dy / len(orders)).view(b, h, w)
    mask = mask.view(b, h, w)
    if pad:
        p = n + 1
        inv_dx = F.pad(inv_dx, (p, p, p, p))
        inv_dy = F.pad(inv_dy, (p, p, p, p))
        mask = F.pad(mask, (p, p, p, p))
    hp, wp = inv_dx.shape[-2:]
    ker = get_gaussian_kernel(kernel_size).view(1, 1, kernel_size, kernel_size)
    r = kernel_size // 2

    def shift_or(m):
        """cells having at least one 4-neighbour in m"""
        o = torch.zeros_like(m)
        o[:, 1:] |= m[:, :-1]
        o[:, :-1] |= m[:, 1:]
        o[:, :, 1:] |= m[:, :, :-1]
        o[:, :, :-1] |= m[:, :, 1:]
        return o

    for _ in range(niter):
        ring = shift_or(mask) & ~mask
        sx = F.conv2d(inv_dx.view(b, 1, hp, wp), ker, padding=r).view(b, hp, wp)
        sy = F.conv2d(inv_dy.view(b, 1, hp, wp), ker, padding=r).view(b, hp, wp)
        sm = F.conv2d(mask.float().view(b, 1, hp, wp), ker, padding=r).view(b, hp, wp)
        den = torch.where(ring, sm, torch.ones_like(sm))
        inv_dx = torch.where(ring, sx / den, inv_dx)
        inv_dy = torch.where(ring, sy / den, inv_dy)
        mask = mask | ring
    if erode:
        for _ in range(niter):
            mask = mask & ~(shift_or(~mask) & mask)
    inv_dx = torch.where(mask, inv_dx, torch.full_like(inv_dx, 2.0 * w))
    inv_dy = torch.where(mask, inv_dy, torch.full_like(inv_dy, 2.0 * h))
    # the reference crops by N+1 unconditionally (warp.py:169-170)
    inv_dx = inv_dx[:, n + 1:-(n + 1), n + 1:-(n + 1)]
    inv_dy = inv_dy[:, n + 1:-(n + 1), n + 1:-(n + 1)]
    dt = torch.stack([inv_dx * 2 / w, inv_dy * 2 / h], dim=3)
    if not pad:
        # the reference adds a cropped field to the full-size grid (warp.py:172-173) and fails
        # with a shape error for pad=False; no call site uses it
        raise ValueError("inverse_warp: pad=False is unusable in the reference (shape mismatch)")
    return get_grid(h, w) + dt
