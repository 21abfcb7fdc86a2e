"""Autograd wrappers over the C-ABI HIP kernels (include/waldo_hip.h).

Every function here launches hand-written gfx950 kernels on the caller's current HIP stream; no
function has a CPU or eager-PyTorch fallback (``_lib.check_cuda`` raises on CPU tensors).
PyTorch is used for memory (output allocation), streams and autograd bookkeeping only.
"""
import math

import torch

from . import _lib


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


_MERGE_FILL_BYTES = 1 << 20


def _zeros_like_each(*tensors):
    """Zero-filled tensors shaped like each argument (None -> None), carved out of ONE buffer and filled by ONE launch:
    the gradients a backward kernel accumulates into with atomics -- at the LVD recipe a fill is a 4 us launch whatever
    its size, and a backward call that zeroed two or three small tensors paid for each.  Every view starts on a
    256-byte boundary."""
    # (only SMALL tensors share a buffer: a view keeps the whole buffer alive until every view is consumed -- a
    # full-resolution gradient tied to a tiny one that waits for another op's backward, or that AccumulateGrad keeps
    # as a leaf's .grad, would stay allocated that long)
    want = [t for t in tensors if t is not None and t.numel() * t.element_size() < _MERGE_FILL_BYTES]
    if len(want) < 2:
        return [torch.zeros_like(t) if t is not None else None for t in tensors]
    sizes = [(t.numel() + 63) // 64 * 64 for t in want]
    flat = torch.zeros(sum(sizes), dtype=want[0].dtype, device=want[0].device)
    out, at = [], 0
    it = iter(sizes)
    for t in tensors:
        if t is None:
            out.append(None)
        elif t.numel() * t.element_size() >= _MERGE_FILL_BYTES:
            out.append(torch.zeros_like(t))
        else:
            n = next(it)
            out.append(flat[at:at + t.numel()].view(t.shape))
            at += n
    return out


class ArangeIndex(torch.Tensor):
    """A frame index KNOWN on the host to be 0, 1, ..., n - 1 (``arange_index``): ``time_gather`` may then hand out
    the clip itself instead of a gathered copy.  Plain tensors never take that short cut -- telling would need a
    device -> host read."""
    __torch_function__ = torch._C._disabled_torch_function_impl  # (results of ops on it are plain tensors)

    def __deepcopy__(self, memo):
        if _is_arange_index(self):
            return arange_index(self.numel(), self.device)
        return self.as_subclass(torch.Tensor).clone()


def arange_index(n, device=None):
    """``torch.arange(n)`` (int64) for ``pred_ts`` when every frame is predicted in order (the LVD recipe's
    ``ctx_mode "prev"``, models/synthesizer.py:833-835), marked as such: ``time_gather(x, None, pred_ts, num_ctx=1)``
    returns a VIEW of ``x`` for it (no copy forward, autograd's own sum backward) -- an alias of the caller's clip,
    unlike the reference's ``x[:, pred_ts]``; do not write to the result in place.  The mark holds for this very
    tensor while nobody writes to it (version counter).  Under HIP-graph capture the choice is frozen into the
    graph like every host-side decision."""
    t = torch.arange(int(n), device=device, dtype=torch.int64)
    if t.is_inference():  # (no version counter to vouch for it: a plain index, gathered by the kernel)
        return t
    t = t.as_subclass(ArangeIndex)
    t._waldo_arange = (int(n), t._version)
    return t


def _is_arange_index(ts):
    mark = getattr(ts, "_waldo_arange", None)
    return mark is not None and not ts.is_inference() and mark == (ts.numel(), ts._version)


def normalise_time_index(ts):
    """``ctx_ts`` / ``pred_ts`` as the kernels take them (int64, contiguous), made ONCE per decode by the
    caller (``Warper``) and handed to every op of the chain instead of converting an expanded view
    (synthesizer.py:438) into a fresh temporary per op.  No read-back: the kernels validate the indices on the
    device (``_lib.IndexStatus``)."""
    return _c(ts.long())


def _status(status):
    """(IndexStatus to hand to the kernels, whether this call checks it itself): a caller that passes its own status
    words checks them when it chooses (``Warper``: lazily, without a synchronisation); a stand-alone call uses the
    module's and is checked -- with a synchronisation -- before it returns, as the reference's ``gather`` raises."""
    return (status, False) if status is not None else (_lib.default_index_status(), True)


# --------------------------------------------------------------------------------------
# A2: TPS
# --------------------------------------------------------------------------------------
class _TpsMapping(torch.autograd.Function):
    @staticmethod
    def forward(ctx, inverse_kernel, src_pts):
        _lib.check_cuda(inverse_kernel, src_pts)
        src_pts = _c(src_pts)
        inverse_kernel = _c(inverse_kernel)
        b, n, _ = src_pts.shape
        mapping = src_pts.new_empty(b, n + 3, 2)
        with _lib.on_device(src_pts.device):
            _lib.call("waldo_tps_mapping_fwd", _lib.ptr(inverse_kernel), _lib.ptr(src_pts),
                      _lib.ptr(mapping), b, n, _lib.current_stream(src_pts.device))
        ctx.save_for_backward(inverse_kernel)
        ctx.n = n
        return mapping

    @staticmethod
    def backward(ctx, grad_mapping):
        (inverse_kernel,) = ctx.saved_tensors
        grad_mapping = _c(grad_mapping)
        b = grad_mapping.shape[0]
        grad_pts = grad_mapping.new_empty(b, ctx.n, 2)
        with _lib.on_device(grad_mapping.device):
            _lib.call("waldo_tps_mapping_bwd", _lib.ptr(inverse_kernel), _lib.ptr(grad_mapping),
                      _lib.ptr(grad_pts), b, ctx.n, _lib.current_stream(grad_mapping.device))
        return None, grad_pts


class _TpsGrid(torch.autograd.Function):
    @staticmethod
    def forward(ctx, basis_t, mapping):
        _lib.check_cuda(basis_t, mapping)
        mapping = _c(mapping)
        basis_t = _c(basis_t)
        b, k3, _ = mapping.shape
        hw = basis_t.shape[1]
        grid = mapping.new_empty(b, hw, 2)
        with _lib.on_device(mapping.device):
            _lib.call("waldo_tps_grid_fwd", _lib.ptr(basis_t), _lib.ptr(mapping), _lib.ptr(grid),
                      b, hw, k3, _lib.current_stream(mapping.device))
        ctx.save_for_backward(basis_t)
        ctx.k3 = k3
        return grid

    @staticmethod
    def backward(ctx, grad_grid):
        (basis_t,) = ctx.saved_tensors
        grad_grid = _c(grad_grid)
        b, hw, _ = grad_grid.shape
        grad_mapping = grad_grid.new_empty(b, ctx.k3, 2)  # zero-filled by the launcher
        with _lib.on_device(grad_grid.device):
            _lib.call("waldo_tps_grid_bwd", _lib.ptr(basis_t), _lib.ptr(grad_grid),
                      _lib.ptr(grad_mapping), b, hw, ctx.k3,
                      _lib.current_stream(grad_grid.device))
        return None, grad_mapping


def tps_mapping(inverse_kernel, src_pts):
    """mapping (B, N+3, 2) = K^-1 @ [src_pts; 0]  (models/modules/warp.py:52-53)."""
    return _TpsMapping.apply(inverse_kernel, src_pts.float())


def tps_grid(inverse_kernel, basis_t, src_pts, height, width):
    """TPSWarp.forward (models/modules/warp.py:49-55): (B, N, 2) -> (B, H, W, 2)."""
    mapping = tps_mapping(inverse_kernel, src_pts)
    return _TpsGrid.apply(basis_t, mapping).view(src_pts.shape[0], height, width, 2)


# --------------------------------------------------------------------------------------
# A3: grid inversion
# --------------------------------------------------------------------------------------
class _InverseWarp(torch.autograd.Function):
    @staticmethod
    def forward(ctx, src_grid, src_id, tgt_id, gauss, niter, erode, rank, order):
        _lib.check_cuda(src_grid, src_id, tgt_id, gauss)
        src_grid, src_id, tgt_id, gauss = _c(src_grid), _c(src_id), _c(tgt_id), _c(gauss)
        b, hs, ws, _ = src_grid.shape
        h, w = tgt_id.shape[-3], tgt_id.shape[-2]
        ksize = int(round(gauss.numel() ** 0.5))
        if ksize * ksize != gauss.numel():
            raise _lib.WaldoHipError(f"inverse_warp: a Gaussian kernel of {gauss.numel()} elements (K x K)")
        pad = niter + 1
        hwp = (h + 2 * pad) * (w + 2 * pad)
        dev = src_grid.device
        out = src_grid.new_empty(b, h, w, 2)
        dxy = src_grid.new_empty(b, 2, h * w)
        cell = torch.empty(b, h * w, dtype=torch.int32, device=dev)
        winner = torch.empty(b, h * w, dtype=torch.int32, device=dev)
        field_a = src_grid.new_empty(b, 2, hwp)
        field_b = src_grid.new_empty(b, 2, hwp)
        fill_iter = torch.empty(b, hwp, dtype=torch.uint8, device=dev)
        denom = src_grid.new_empty(b, hwp)
        mask_a = torch.empty(b, hwp, dtype=torch.uint8, device=dev)
        mask_b = torch.empty(b, hwp, dtype=torch.uint8, device=dev)
        work = (_lib.ptr(out), _lib.ptr(dxy), _lib.ptr(cell), _lib.ptr(winner), _lib.ptr(field_a),
                _lib.ptr(field_b), _lib.ptr(fill_iter), _lib.ptr(denom), _lib.ptr(mask_a),
                _lib.ptr(mask_b), b, hs, ws, h, w, niter, int(bool(erode)), ksize)
        with _lib.on_device(dev):
            if order is None:
                _lib.call("waldo_inverse_warp_fwd", _lib.ptr(src_grid), _lib.ptr(src_id),
                          _lib.ptr(tgt_id), _lib.ptr(gauss), *work, _lib.current_stream(dev))
            else:
                if not (rank.is_cuda and order.is_cuda):
                    raise _lib.WaldoHipError("inverse_warp: rank / order must be on the GPU")
                if (rank.dtype != torch.int32 or order.dtype != torch.int32
                        or rank.numel() != h * w or order.numel() != h * w):
                    raise _lib.WaldoHipError("inverse_warp: rank / order must be int32 of H*W elements")
                rank, order = rank.contiguous(), order.contiguous()
                _lib.call("waldo_inverse_warp_order_fwd", _lib.ptr(src_grid), _lib.ptr(src_id),
                          _lib.ptr(tgt_id), _lib.ptr(gauss), _lib.ptr(rank), _lib.ptr(order), *work,
                          _lib.current_stream(dev))
        ctx.save_for_backward(gauss, cell, winner, fill_iter, denom, mask_a)
        ctx.cfg = (b, hs, ws, h, w, niter, ksize)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        gauss, cell, winner, fill_iter, denom, mask = ctx.saved_tensors
        b, hs, ws, h, w, niter, ksize = ctx.cfg
        grad_out = _c(grad_out)
        gfield = grad_out.new_empty(b, 2, fill_iter.shape[1])
        gsrc = grad_out.new_empty(b, hs, ws, 2)
        with _lib.on_device(grad_out.device):
            _lib.call("waldo_inverse_warp_bwd", _lib.ptr(grad_out), _lib.ptr(gauss), _lib.ptr(cell),
                      _lib.ptr(winner), _lib.ptr(fill_iter), _lib.ptr(denom), _lib.ptr(mask),
                      _lib.ptr(gfield), _lib.ptr(gsrc), b, hs, ws, h, w, niter, ksize,
                      _lib.current_stream(grad_out.device))
        return gsrc, None, None, None, None, None, None, None


def inverse_warp(src_grid, src_id, tgt_id, gauss, niter=5, erode=True, perm=None):
    """InverseWarp.forward (models/modules/warp.py:71-174; pad).
    src_grid (B, Hs, Ws, 2) -> (B, H, W, 2); src_id / tgt_id are the identity grids of the two
    rasters, gauss the normalised K x K Gaussian, flattened (the reference module's buffers; K odd -- 3 in every
    script: one launch each way; other sizes run the fill passes one by one).

    perm None: num_perm == 1 (the lowest sample index wins a contested cell, warp.py:113-123).
    perm (P, H*W) integer, P > 1: the reference's tie-break averaging (warp.py:91-111): for each
    row the sample standing first in that order wins, and the P elected fields are averaged.  The
    fill / erosion / crop are linear in the elected field for a fixed set of occupied cells (which
    does not depend on the order), so the average is taken over the P results instead."""
    if perm is None:
        return _InverseWarp.apply(src_grid, src_id, tgt_id, gauss, int(niter), bool(erode), None,
                                  None)
    order = perm.to(torch.int32)
    rank = torch.empty_like(order)
    pos = torch.arange(order.shape[1], dtype=torch.int32, device=order.device)
    rank.scatter_(1, order.long(), pos.expand_as(order))
    out = None
    for p in range(order.shape[0]):
        o = _InverseWarp.apply(src_grid, src_id, tgt_id, gauss, int(niter), bool(erode), rank[p],
                               order[p])
        out = o if out is None else out + o
    return out / order.shape[0]


# --------------------------------------------------------------------------------------
# A4/A5: bilinear warp
# --------------------------------------------------------------------------------------
class _GridSample(torch.autograd.Function):
    @staticmethod
    def forward(ctx, inp, grid, delta, outer_div, inner, want_mask=False):
        _lib.check_cuda(inp, grid)
        inp = _c(inp)
        grid = _c(grid)
        nin, c, hi, wi = inp.shape
        n, ho, wo, two = grid.shape
        assert two == 2
        if outer_div is None:
            if nin != n:
                raise _lib.WaldoHipError(f"grid_sample: batch mismatch {nin} vs {n}")
            outer_div = inner = max(n, 1)
        out = inp.new_empty(n, c, ho, wo)
        mask = inp.new_empty(n, 1, ho, wo) if want_mask else None
        with _lib.on_device(inp.device):
            if want_mask:
                _lib.call("waldo_grid_sample2d_ex_fwd", _lib.ptr(inp), _lib.ptr(grid), _lib.ptr(out), _lib.ptr(mask),
                          n, c, hi, wi, ho, wo, float(delta), outer_div, inner, max(n, 1), max(n, 1),
                          max(n, 1), max(n, 1), 0, 1.0, 0.0, _lib.current_stream(inp.device))
            else:
                _lib.call("waldo_grid_sample2d_fwd", _lib.ptr(inp), _lib.ptr(grid), _lib.ptr(out), n,
                          c, hi, wi, ho, wo, float(delta), outer_div, inner, max(n, 1), max(n, 1),
                          _lib.current_stream(inp.device))
        ctx.save_for_backward(inp, grid)
        ctx.cfg = (float(delta), outer_div, inner)
        if want_mask:
            ctx.mark_non_differentiable(mask)
            return out, mask
        return out

    @staticmethod
    def backward(ctx, grad_out, _grad_mask=None):
        inp, grid = ctx.saved_tensors
        delta, outer_div, inner = ctx.cfg
        grad_out = _c(grad_out)
        nin, c, hi, wi = inp.shape
        n, ho, wo, _ = grid.shape
        gi = torch.zeros_like(inp) if ctx.needs_input_grad[0] else None
        gg = torch.empty_like(grid) if ctx.needs_input_grad[1] else None
        with _lib.on_device(inp.device):
            _lib.call("waldo_grid_sample2d_bwd", _lib.ptr(inp), _lib.ptr(grid),
                      _lib.ptr(grad_out), _lib.ptr(gi), _lib.ptr(gg), n, c, hi, wi, ho, wo,
                      delta, outer_div, inner, _lib.current_stream(inp.device))
        return gi, gg, None, None, None, None


def grid_sample(inp, grid, delta=0.0, broadcast=None, grid_repeat=None, return_mask=False, out=None):
    """``F.grid_sample(inp + delta, grid) - delta`` with the PyTorch defaults (bilinear, zeros,
    align_corners=False).  inp (Nin, C, Hi, Wi), grid (N, Ho, Wo, 2) -> (N, C, Ho, Wo).

    broadcast=(outer_div, inner): input map used by output n is
    ``(n // outer_div) * inner + n % inner`` -- the reference's ``.expand`` over time
    (models/nets/lvd.py:544,555) without materialising the copies.

    grid_repeat=(n_out, outer_div, inner): the same map for the GRID -- ``grid`` holds (Ng, Ho, Wo, 2) maps and
    output n of ``n_out`` reads map ``(n // outer_div) * inner + n % inner``: the predicted frames' grids
    repeated over the contexts (lvd.py:665-668) without the copies.  Inference only (no gradient).

    return_mask: also return ``grid_sample(ones_like(inp[:, :1]), grid)`` (N, 1, Ho, Wo) -- the warped all-ones image
    of ``Warper.grid_to_flow_ctx``'s ghost test (lvd.py:785-791), a by-product of the same taps (no gradient).

    out=(tensor, group, stride, offset): write output map n into slot ``(n // group) * stride + offset + n % group``
    of ``tensor`` (slots, C, Ho, Wo) instead of a tensor of its own -- the two calls of ``Warper.layer_to_output``
    (lvd.py:533-537) then fill the concatenated tensor directly.  Inference only; returns ``tensor``."""
    if grid_repeat is not None or out is not None:
        if torch.is_grad_enabled() and (inp.requires_grad or grid.requires_grad):
            raise _lib.WaldoHipError("grid_sample: grid_repeat / out are forward only (no gradient flows through them)")
        _lib.check_cuda(inp, grid)
        inp, grid = _c(inp.detach()), _c(grid.detach())
        nin, c, hi, wi = inp.shape
        ng, ho, wo, _ = grid.shape
        if grid_repeat is not None:
            n_out, god, gin = (int(v) for v in grid_repeat)
        else:
            n_out, god, gin = ng, max(ng, 1), max(ng, 1)
        od, inn = broadcast if broadcast is not None else (max(n_out, 1), max(n_out, 1))
        if god < 1 or gin < 1 or (n_out > 0 and ((n_out - 1) // god) * gin + min(gin, n_out) > ng) or \
                (broadcast is None and nin != n_out):
            raise _lib.WaldoHipError(f"grid_sample: grid_repeat {grid_repeat} against {ng} grids / {nin} inputs")
        if out is not None:
            dst, grp, stride, off = out[0], int(out[1]), int(out[2]), int(out[3])
            _lib.check_cuda(dst)
            slots = ((n_out - 1) // grp) * stride + off + min(grp, n_out) if n_out > 0 else 0
            if not dst.is_contiguous() or dst.dtype != inp.dtype or tuple(dst.shape[-3:]) != (c, ho, wo) or \
                    dst.numel() < slots * c * ho * wo or grp < 1 or off < 0 or off + grp > stride:
                raise _lib.WaldoHipError(f"grid_sample: out tensor {tuple(dst.shape)} does not hold slots "
                                         f"(group {grp}, stride {stride}, offset {off}) of {n_out} maps of {(c, ho, wo)}")
            res = dst
        else:
            grp, stride, off = max(n_out, 1), max(n_out, 1), 0
            res = inp.new_empty(n_out, c, ho, wo)
        mask = inp.new_empty(n_out, 1, ho, wo) if return_mask else None
        with _lib.on_device(inp.device):
            if return_mask or out is not None:
                _lib.call("waldo_grid_sample2d_ex_fwd", _lib.ptr(inp), _lib.ptr(grid), _lib.ptr(res), _lib.ptr(mask),
                          n_out, c, hi, wi, ho, wo, float(delta), od, inn, god, gin, grp, stride, off, 1.0, 0.0,
                          _lib.current_stream(inp.device))
            else:
                _lib.call("waldo_grid_sample2d_fwd", _lib.ptr(inp), _lib.ptr(grid), _lib.ptr(res), n_out, c, hi, wi,
                          ho, wo, float(delta), od, inn, god, gin, _lib.current_stream(inp.device))
        return (res, mask) if return_mask else res
    od, inn = broadcast if broadcast is not None else (None, None)
    return _GridSample.apply(inp, grid, delta, od, inn, bool(return_mask))


class _LayersToOutput(torch.autograd.Function):
    @staticmethod
    def forward(ctx, obj, bg, grid_obj, grid_bg, delta_obj, delta_bg, obj_bc, bg_bc, pre, want_mask):
        nf, h, w, _ = grid_bg.shape
        no = grid_obj.shape[0] // max(nf, 1)
        nl = no + 1
        c = obj.shape[1]
        out = obj.new_empty(nf, nl, c, h, w)
        mask = obj.new_empty(nf * no, 1, h, w) if want_mask else None
        calls = ((obj, grid_obj, mask, nf * no, delta_obj, obj_bc, (no, nl, 1)),
                 (bg, grid_bg, None, nf, delta_bg, bg_bc, (1, nl, 0)))
        with _lib.on_device(obj.device):
            for inp, grid, msk, n, delta, bc, slots in calls:
                if n == 0:  # (no frames, or no objects: the background alone)
                    continue
                od, inn = bc if bc is not None else (max(n, 1), max(n, 1))
                _lib.call("waldo_grid_sample2d_ex_fwd", _lib.ptr(inp), _lib.ptr(grid), _lib.ptr(out), _lib.ptr(msk),
                          n, c, inp.shape[2], inp.shape[3], h, w, float(delta), od, inn, max(n, 1), max(n, 1),
                          *slots, float(pre[0]), float(pre[1]), _lib.current_stream(obj.device))
        ctx.save_for_backward(obj, bg, grid_obj, grid_bg)
        ctx.cfg = (float(delta_obj), float(delta_bg), obj_bc, bg_bc, (float(pre[0]), float(pre[1])), no)
        if want_mask:
            ctx.mark_non_differentiable(mask)
            return out, mask
        return out

    @staticmethod
    def backward(ctx, grad_out, _grad_mask=None):
        obj, bg, grid_obj, grid_bg = ctx.saved_tensors
        delta_obj, delta_bg, obj_bc, bg_bc, pre, no = ctx.cfg
        nf, h, w, _ = grid_bg.shape
        nl = no + 1
        c = obj.shape[1]
        grad_out = _c(grad_out)
        need = ctx.needs_input_grad
        res = []
        gis = _zeros_like_each(obj if need[0] else None, bg if need[1] else None)
        calls = ((obj, grid_obj, nf * no, delta_obj, obj_bc, (no, nl, 1), gis[0], need[2]),
                 (bg, grid_bg, nf, delta_bg, bg_bc, (1, nl, 0), gis[1], need[3]))
        with _lib.on_device(obj.device):
            for inp, grid, n, delta, bc, slots, gi, want_g in calls:
                want_i = gi is not None
                gg = torch.empty_like(grid) if want_g else None
                if n > 0 and (want_i or want_g):
                    od, inn = bc if bc is not None else (max(n, 1), max(n, 1))
                    _lib.call("waldo_grid_sample2d_ex_bwd", _lib.ptr(inp), _lib.ptr(grid), _lib.ptr(grad_out),
                              _lib.ptr(gi), _lib.ptr(gg), n, c, inp.shape[2], inp.shape[3], h, w, delta, od, inn,
                              *slots, pre[0], pre[1], _lib.current_stream(obj.device))
                res.append((gi, gg))
        return res[0][0], res[1][0], res[0][1], res[1][1], None, None, None, None, None, None


def layers_to_output(obj, bg, grid_obj, grid_bg, delta_obj=0.0, delta_bg=0.0, obj_broadcast=None, bg_broadcast=None,
                     pre=(1.0, 0.0), return_mask=False):
    """``Warper.layer_to_output`` (models/nets/lvd.py:533-559) as ONE differentiable op: the objects' maps and the
    background's warped straight into the tensor the reference concatenates,

        torch.cat([grid_sample(bg', grid_bg, delta_bg)[:, None], grid_sample(obj', grid_obj, delta_obj).view(F, No, ...)], 1)

    obj (Nin_o, C, Ho, Wo) with grid_obj (F * No, H, W, 2), bg (Nin_b, C, Hb, Wb) with grid_bg (F, H, W, 2) ->
    (F, No + 1, C, H, W), the background at layer 0.  ``*_broadcast`` = (outer_div, inner) as in ``grid_sample`` (the
    reference's ``.expand`` over time).  ``pre`` = (scale, bias): the images warped are ``scale * obj + bias`` and
    ``scale * bg + bias`` -- ``grid_to_flow`` warps ``(alpha + 1) / 2`` (lvd.py:602-606, 716-720) -- without being written
    first; the gradients returned are those of ``obj`` / ``bg``.  ``return_mask``: also the warped all-ones canvas of
    the objects' grids (F * No, 1, H, W) (``grid_sample(..., return_mask=True)``; no gradient).

    Forward: two launches that write disjoint slots of one tensor (no ``cat``); backward: two launches that read their
    slots of the tensor's gradient (no copies of its two slices)."""
    _lib.check_cuda(obj, bg, grid_obj, grid_bg)
    obj, bg, grid_obj, grid_bg = _c(obj), _c(bg), _c(grid_obj), _c(grid_bg)
    nf, h, w, two = grid_bg.shape
    ngo = grid_obj.shape[0]
    if two != 2 or grid_obj.dim() != 4 or tuple(grid_obj.shape[1:]) != (h, w, 2) or obj.dim() != 4 or bg.dim() != 4 or \
            obj.shape[1] != bg.shape[1] or (nf == 0 and ngo != 0) or (nf > 0 and ngo % nf != 0) or obj.dtype != bg.dtype:
        raise _lib.WaldoHipError(f"layers_to_output: obj {tuple(obj.shape)} / grid_obj {tuple(grid_obj.shape)} against "
                                 f"bg {tuple(bg.shape)} / grid_bg {tuple(grid_bg.shape)}")
    for name, inp, n, bc in (("obj", obj, ngo, obj_broadcast), ("bg", bg, nf, bg_broadcast)):
        if bc is None:
            if inp.shape[0] != n:
                raise _lib.WaldoHipError(f"layers_to_output: {inp.shape[0]} {name} images for {n} grids")
        else:
            od, inn = int(bc[0]), int(bc[1])
            if od < 1 or inn < 1 or (n > 0 and ((n - 1) // od) * inn + min(inn, n) > inp.shape[0]):
                raise _lib.WaldoHipError(f"layers_to_output: {name} broadcast {bc} of {n} maps over {inp.shape[0]} images")
    obj_bc = None if obj_broadcast is None else (int(obj_broadcast[0]), int(obj_broadcast[1]))
    bg_bc = None if bg_broadcast is None else (int(bg_broadcast[0]), int(bg_broadcast[1]))
    return _LayersToOutput.apply(obj, bg, grid_obj, grid_bg, float(delta_obj), float(delta_bg), obj_bc, bg_bc,
                                 (float(pre[0]), float(pre[1])), bool(return_mask))


# --------------------------------------------------------------------------------------
# A6: occlusion product
# --------------------------------------------------------------------------------------
class _OccComposite(torch.autograd.Function):
    @staticmethod
    def forward(ctx, alpha, occ, occ_div):
        _lib.check_cuda(alpha, occ)
        alpha = _c(alpha)
        occ = _c(occ)
        m, nl, hw = alpha.shape
        out = torch.empty_like(alpha)
        with _lib.on_device(alpha.device):
            _lib.call("waldo_occ_composite_fwd", _lib.ptr(alpha), _lib.ptr(occ), _lib.ptr(out), m,
                      nl, hw, occ_div, _lib.current_stream(alpha.device))
        ctx.save_for_backward(alpha, occ)
        ctx.occ_div = occ_div
        return out

    @staticmethod
    def backward(ctx, grad_out):
        alpha, occ = ctx.saved_tensors
        grad_out = _c(grad_out)
        m, nl, hw = alpha.shape
        ga = torch.empty_like(alpha)
        go = torch.zeros_like(occ) if ctx.needs_input_grad[1] else None
        with _lib.on_device(alpha.device):
            _lib.call("waldo_occ_composite_bwd", _lib.ptr(alpha), _lib.ptr(occ),
                      _lib.ptr(grad_out), _lib.ptr(ga), _lib.ptr(go), m, nl, hw, ctx.occ_div,
                      _lib.current_stream(alpha.device))
        return ga, go, None


def occ_composite(alpha, occ, occ_div=1):
    """out[m, j] = alpha[m, j] * prod_i (1 - alpha[m, i] * occ[m // occ_div, i, j]).
    alpha (M, L, h, w) or (M, L, HW) in [0, 1]; occ (M // occ_div, L, L).
    The (1 - alpha * occ).prod(dim) * alpha pattern of models/nets/lvd.py:651-652,764-765,809."""
    shape = alpha.shape
    a3 = alpha.reshape(shape[0], shape[1], -1)
    return _OccComposite.apply(a3, occ, int(occ_div)).view(shape)


# --------------------------------------------------------------------------------------
# f2: producers of the path's inputs (csrc/producers.hip)
# --------------------------------------------------------------------------------------
class _ComputeOcc(torch.autograd.Function):
    @staticmethod
    def forward(ctx, score, eps):
        _lib.check_cuda(score)
        score = _c(score)
        m, no = score.shape
        occ = score.new_empty(m, no + 1, no + 1)
        with _lib.on_device(score.device):
            _lib.call("waldo_compute_occ_fwd", _lib.ptr(score), _lib.ptr(occ), m, no, float(eps),
                      _lib.current_stream(score.device))
        ctx.save_for_backward(score)
        ctx.eps = float(eps)
        return occ

    @staticmethod
    def backward(ctx, grad_occ):
        (score,) = ctx.saved_tensors
        grad_occ = _c(grad_occ)
        m, no = score.shape
        gs = torch.empty_like(score)
        with _lib.on_device(score.device):
            _lib.call("waldo_compute_occ_bwd", _lib.ptr(score), _lib.ptr(grad_occ), _lib.ptr(gs), m, no,
                      ctx.eps, _lib.current_stream(score.device))
        return gs, None


def compute_occ(occ_score, eps=1e-6):
    """LVD.compute_occ (models/nets/lvd.py:59-68): occ_score (..., No) -> (..., No+1, No+1)."""
    lead = occ_score.shape[:-1]
    no = occ_score.shape[-1]
    occ = _ComputeOcc.apply(occ_score.reshape(-1, no).float(), eps)
    return occ.view(*lead, no + 1, no + 1)


class _AlphaHead(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, prior, mask, scale, bias, has_alpha, mode):
        _lib.check_cuda(x, prior, mask)
        x = _c(x)
        prior = _c(prior) if prior is not None else None
        mask = _c(mask) if mask is not None else None
        n, c, h, w = x.shape
        if prior is not None and prior.numel() != h * w:
            raise _lib.WaldoHipError(f"alpha_head: prior has {prior.numel()} elements, expected {h}x{w}")
        if mask is not None and mask.numel() != h * w * scale * scale:
            raise _lib.WaldoHipError(f"alpha_head: mask has {mask.numel()} elements, expected "
                                     f"{h * scale}x{w * scale}")
        out = x.new_empty(n, c, h * scale, w * scale)
        with _lib.on_device(x.device):
            _lib.call("waldo_alpha_head_fwd", _lib.ptr(x), _lib.ptr(prior), _lib.ptr(mask), _lib.ptr(out), n, c,
                      h, w, scale, float(bias), int(has_alpha), mode, _lib.current_stream(x.device))
        ctx.save_for_backward(x, prior, mask)
        ctx.cfg = (scale, float(bias), int(has_alpha), mode)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        x, prior, mask = ctx.saved_tensors
        scale, bias, has_alpha, mode = ctx.cfg
        grad_out = _c(grad_out)
        n, c, h, w = x.shape
        gx = torch.empty_like(x)
        with _lib.on_device(x.device):
            _lib.call("waldo_alpha_head_bwd", _lib.ptr(x), _lib.ptr(prior), _lib.ptr(mask), _lib.ptr(grad_out),
                      _lib.ptr(gx), n, c, h, w, scale, bias, has_alpha, mode, _lib.current_stream(x.device))
        return gx, None, None, None, None, None, None


def disocc_test(layer_max):
    """Synthesizer.predict's disocclusion test (models/synthesizer.py:447-450): ``layer_max`` (B, Tc, Tp, H, W) =
    ``alpha_ctx.max(dim=3)[0]`` -> ``dmax`` (B, Tp, H, W) with ``dmax[dmax - dmin > 1] = 0``, max / min over the
    contexts (NaN-propagating as torch's).  One pass; inference only."""
    _lib.check_cuda(layer_max)
    if layer_max.ndim != 5:
        raise _lib.WaldoHipError(f"disocc_test: layer_max {tuple(layer_max.shape)} is not (B, Tc, Tp, H, W)")
    layer_max = _c(layer_max.detach())
    b, tc, tp, h, w = layer_max.shape
    out = layer_max.new_empty(b, tp, h, w)
    with _lib.on_device(layer_max.device):
        _lib.call("waldo_disocc_test_fwd", _lib.ptr(layer_max), _lib.ptr(out), b, tc, tp, h * w,
                  _lib.current_stream(layer_max.device))
    return out


def alpha_head(img, prior=None, mask=None, scale=1, bias=0.0, has_alpha=True, remove=False, freeze=False):
    """ImageDecoder.forward's tail (models/nets/lvd.py:245-254: ``+ init_bias``, ``tanh`` and the
    ``circle`` prior on the last channel, ``scale(img, scale_factor)``) fused with the alpha
    arithmetic of ``LVD.forward(mode="estimate_alpha_grid_occ")`` (lvd.py:128-132: ``remove_obj`` /
    ``freeze_obj`` / ``obj_alpha_mask``).  img (N, C, h, w) -> (N, C, h*scale, w*scale)."""
    mode = 2 if freeze else (1 if remove else 0)
    return _AlphaHead.apply(img.float(), prior, mask, int(scale), bias, bool(has_alpha), mode)


class _PoseAffine(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pose, mul6, bias6, base, mul_delta, pts_mul):
        _lib.check_cuda(pose, mul6, bias6, base)
        pose, mul6, bias6, base = _c(pose), _c(mul6), _c(bias6), _c(base)
        r, d = pose.shape
        p = (d - 6) // 2
        if d != 6 + 2 * p or base.numel() != 2 * p or mul6.numel() != 6 or bias6.numel() != 6:
            raise _lib.WaldoHipError(f"pose_affine: inconsistent shapes pose={tuple(pose.shape)} "
                                     f"base={tuple(base.shape)}")
        out = pose.new_empty(r, p, 2)
        with _lib.on_device(pose.device):
            _lib.call("waldo_pose_affine_fwd", _lib.ptr(pose), _lib.ptr(mul6), _lib.ptr(bias6), _lib.ptr(base),
                      _lib.ptr(out), r, p, float(mul_delta), float(pts_mul), _lib.current_stream(pose.device))
        ctx.save_for_backward(pose, mul6, bias6, base)
        ctx.cfg = (float(mul_delta), float(pts_mul))
        return out

    @staticmethod
    def backward(ctx, grad_out):
        pose, mul6, bias6, base = ctx.saved_tensors
        grad_out = _c(grad_out)
        r, d = pose.shape
        gp = torch.empty_like(pose)
        with _lib.on_device(pose.device):
            _lib.call("waldo_pose_affine_bwd", _lib.ptr(pose), _lib.ptr(mul6), _lib.ptr(bias6), _lib.ptr(base),
                      _lib.ptr(grad_out), _lib.ptr(gp), r, (d - 6) // 2, ctx.cfg[0], ctx.cfg[1],
                      _lib.current_stream(pose.device))
        return gp, None, None, None, None, None


def pose_affine(pose, mul6, bias6, base_pts, mul_delta=1.0, pts_mul=1.0):
    """The pose heads' affine (models/nets/flp.py:259-273): pose (..., 6 + 2P) -> control points
    (..., P, 2) = [pts_mul * base_pts + mul_delta * pose[6:], 1] @ (mul6 * pose[:6] + bias6)."""
    lead = pose.shape[:-1]
    d = pose.shape[-1]
    out = _PoseAffine.apply(pose.reshape(-1, d).float(), mul6.reshape(-1).float(), bias6.reshape(-1).float(),
                            base_pts.reshape(-1, 2).float(), mul_delta, pts_mul)
    return out.view(*lead, (d - 6) // 2, 2)


# --------------------------------------------------------------------------------------
# A12: WIF fusion epilogue
# --------------------------------------------------------------------------------------
class _WifFuse(torch.autograd.Function):
    @staticmethod
    def forward(ctx, vid, net, ab):
        _lib.check_cuda(vid, net)
        vid, net = _c(vid), _c(net)
        b, t, tc, c, h, w = vid.shape
        co = net.shape[3]
        if tuple(net.shape) != (b, t, tc, co, h, w):
            raise _lib.WaldoHipError(f"wif_fuse: shapes {tuple(vid.shape)} vs {tuple(net.shape)}")
        out = vid.new_empty(b, t, 3, h, w)
        with _lib.on_device(vid.device):
            _lib.call("waldo_wif_fuse_fwd", _lib.ptr(vid), _lib.ptr(net), _lib.ptr(out), b * t, tc,
                      c, co, h * w, int(bool(ab)), _lib.current_stream(vid.device))
        ctx.save_for_backward(vid, net, out)
        ctx.ab = int(bool(ab))
        return out

    @staticmethod
    def backward(ctx, grad_out):
        vid, net, out = ctx.saved_tensors
        b, t, tc, c, h, w = vid.shape
        co = net.shape[3]
        grad_out = _c(grad_out)
        gv = torch.empty_like(vid) if ctx.needs_input_grad[0] else None
        gn = torch.empty_like(net) if ctx.needs_input_grad[1] else None
        with _lib.on_device(vid.device):
            _lib.call("waldo_wif_fuse_bwd", _lib.ptr(vid), _lib.ptr(net), _lib.ptr(out),
                      _lib.ptr(grad_out), _lib.ptr(gv), _lib.ptr(gn), b * t, tc, c, co, h * w,
                      ctx.ab, _lib.current_stream(vid.device))
        return gv, gn, None


def wif_fuse(vid, net_out, ab=True):
    """Fusion epilogue of WIF.forward with ii_score (models/nets/wif.py:49-54).
    vid (B, T, Tc, C, H, W): the UNet input after the permute; net_out (B, T, Tc, 4|5, H, W)."""
    return _WifFuse.apply(vid, net_out, ab)


# --------------------------------------------------------------------------------------
# A9: the two HD passes of Warper.grid_to_flow_ctx / grid_to_flow (forward only)
# --------------------------------------------------------------------------------------
class _LytDist(torch.autograd.Function):
    @staticmethod
    def forward(ctx, alpha, lyt, cls, min_cls, first_obj):
        b, tw, la, h, w = alpha.shape
        nl = lyt.shape[2]
        no = la - first_obj
        dev = alpha.device
        dist = alpha.new_empty(b, no, nl)
        mean = alpha.new_empty(b, no, nl)
        total = alpha.new_empty(b, no)
        wsb = _lib.load().waldo_lyt_dist_workspace_bytes(b, tw, no, nl, h, w)
        ws = alpha.new_empty(max(wsb, 4) // 4)
        with _lib.on_device(dev):
            _lib.call("waldo_lyt_dist_fwd", _lib.ptr(alpha), _lib.ptr(lyt), lyt.stride(0), lyt.stride(1),
                      _lib.ptr(cls), float(min_cls), _lib.ptr(dist), _lib.ptr(mean), _lib.ptr(total),
                      _lib.ptr(ws), b, tw, la, first_obj, no, nl, h, w, _lib.current_stream(dev))
        ctx.save_for_backward(alpha, lyt, cls, dist, mean, total)
        ctx.cfg = (float(min_cls), first_obj)
        ctx.mark_non_differentiable(mean, total)
        return dist, mean, total

    @staticmethod
    def backward(ctx, g_dist, _g_mean, _g_total):
        alpha, lyt, cls, dist, mean, total = ctx.saved_tensors
        min_cls, first_obj = ctx.cfg
        b, tw, la, h, w = alpha.shape
        nl = lyt.shape[2]
        no = la - first_obj
        dev = alpha.device
        g_dist = _c(g_dist)
        g_alpha = torch.empty_like(alpha)
        g_cls = torch.empty_like(cls) if cls is not None else None
        wsb = _lib.load().waldo_lyt_dist_workspace_bytes(b, tw, no, nl, h, w)
        ws = alpha.new_empty(max(wsb, 4) // 4)
        with _lib.on_device(dev):
            _lib.call("waldo_lyt_dist_bwd", _lib.ptr(g_dist), _lib.ptr(alpha), _lib.ptr(lyt), lyt.stride(0),
                      lyt.stride(1), _lib.ptr(cls), min_cls, _lib.ptr(dist), _lib.ptr(mean),
                      _lib.ptr(total), _lib.ptr(g_alpha), _lib.ptr(g_cls), _lib.ptr(ws), b, tw, la,
                      first_obj, no, nl, h, w, _lib.current_stream(dev))
        return g_alpha, None, g_cls, None, None


def lyt_dist(alpha, lyt, cls=None, min_cls=0.0, first_obj=1):
    """Class distribution of every object for the layout filter (models/nets/lvd.py:624-634 /
    731-746).  alpha (B, Tw, L, H, W) projected alpha in [0, 1], objects = layers first_obj .. L-1;
    lyt (B, Tw, Nl, H, W) layout logits at the same raster (any batch / frame strides: a channel
    slice of the input works without a copy); cls (B, No, Nl) = the reference's ``cls`` when
    ``weight_cls`` is set, else None.  Returns dist (B, No, Nl).  Differentiable w.r.t. alpha and
    cls; the layout is data."""
    _lib.check_cuda(alpha, lyt, cls)
    alpha = _c(alpha)
    lyt = lyt.detach()
    b, tw, la, h, w = alpha.shape
    if lyt.dim() != 5 or tuple(lyt.shape[:2]) != (b, tw) or tuple(lyt.shape[3:]) != (h, w):
        raise _lib.WaldoHipError(f"lyt_dist: lyt {tuple(lyt.shape)} does not match alpha {tuple(alpha.shape)}")
    if lyt.stride(4) != 1 or lyt.stride(3) != w or lyt.stride(2) != h * w:
        lyt = lyt.contiguous()
    if cls is not None:
        cls = _c(cls)
        if tuple(cls.shape) != (b, la - first_obj, lyt.shape[2]):
            raise _lib.WaldoHipError(f"lyt_dist: cls {tuple(cls.shape)} is not (B, No, Nl)")
    return _LytDist.apply(alpha, lyt, cls, float(min_cls), int(first_obj))[0]


class _FlowCtxAlpha(torch.autograd.Function):
    @staticmethod
    def forward(ctx, alpha_lr, input, dist, occ, tw, chan_off, scale):
        n, nl, h, w = alpha_lr.shape
        b, t, c, hd, wd = input.shape
        ncls = dist.shape[2] if dist is not None else 0
        a01 = alpha_lr.new_empty(n, nl, hd, wd)
        out = alpha_lr.new_empty(n, nl, hd, wd)
        with _lib.on_device(alpha_lr.device):
            _lib.call("waldo_flow_ctx_alpha_fwd", _lib.ptr(alpha_lr), _lib.ptr(input), _lib.ptr(dist),
                      _lib.ptr(occ), _lib.ptr(a01), _lib.ptr(out), None, b, t, tw, nl, ncls, c, chan_off, h, w,
                      scale, _lib.current_stream(alpha_lr.device))
        ctx.save_for_backward(alpha_lr, input, dist, occ)
        ctx.cfg = (tw, chan_off, scale)
        ctx.set_materialize_grads(False)  # backward below handles a missing gradient of either output
        return a01, out

    @staticmethod
    def backward(ctx, g_a01, g_out):
        alpha_lr, input, dist, occ = ctx.saved_tensors
        tw, chan_off, scale = ctx.cfg
        n, nl, h, w = alpha_lr.shape
        b, t, c, hd, wd = input.shape
        ncls = dist.shape[2] if dist is not None else 0
        # alpha_out = 2 a01 - 1
        if g_a01 is None and g_out is None:
            return None, None, None, None, None, None, None
        # (the kernel reads g_a01 + 2 g_out itself: no pass over (B*Tw, L, Hd, Wd) to add them first)
        g_a01 = _c(g_a01) if g_a01 is not None else None
        g_out = _c(g_out) if g_out is not None else None
        g_lr = torch.empty_like(alpha_lr)
        g_dist, g_occ = _zeros_like_each(dist if (dist is not None and ctx.needs_input_grad[2]) else None,
                                         occ if ctx.needs_input_grad[3] else None)
        ws = alpha_lr.new_empty(n, nl, hd, wd) if scale > 1 else None
        with _lib.on_device(alpha_lr.device):
            _lib.call("waldo_flow_ctx_alpha_bwd", _lib.ptr(alpha_lr), _lib.ptr(input), _lib.ptr(dist),
                      _lib.ptr(occ), _lib.ptr(g_a01), _lib.ptr(g_out), _lib.ptr(g_lr), _lib.ptr(g_dist), _lib.ptr(g_occ),
                      _lib.ptr(ws), b, t, tw, nl, ncls, c, chan_off, h, w, scale,
                      _lib.current_stream(alpha_lr.device))
        return g_lr, None, g_dist, g_occ, None, None, None


def flow_ctx_alpha(alpha_lr, input, dist, occ, tw, chan_off, scale, want_alpha=True, want_bits=False):
    """Upsampling + layout filter + first occlusion product (models/nets/lvd.py:731-766).
    alpha_lr (B*Tw, L, H, W) in [0, 1]; input (B, T, C, Hd, Wd) with the layout logits in channels
    [chan_off, chan_off + Nl); dist (B, L-1, Nl) or None (no filter); occ (B, T, L, L).
    Returns (a01, alpha) of shape (B*Tw, L, Hd, Wd): the composited alpha in [0, 1] and 2a - 1.
    Differentiable w.r.t. alpha_lr, dist and occ (the frames / layouts in ``input`` are data).
    ``want_alpha=False`` (no autograd): ``alpha`` is not written and comes back as None -- ``Synthesizer.predict``'s
    reconstruction drops it (synthesizer.py:445), and it is as large as ``a01``.
    ``want_bits`` (no autograd): a third result, ``layer_bits`` (B*Tw, Hd, ceil(Wd / 64)) int32 -- bit l of a word: layer
    l of ``a01`` is non-zero somewhere in that 64-pixel row segment -- for ``flow_ctx_warp(..., layer_bits=...)`` on the
    path without a ghost mask (``Warper.grid_to_flow``)."""
    _lib.check_cuda(alpha_lr, input, occ)
    alpha_lr, input, occ = _c(alpha_lr), _c(input.detach()), _c(occ)
    n, nl, h, w = alpha_lr.shape
    b, t, c, hd, wd = input.shape
    if n != b * tw or tuple(occ.shape) != (b, t, nl, nl) or hd != h * scale or wd != w * scale:
        raise _lib.WaldoHipError(
            f"flow_ctx_alpha: inconsistent shapes alpha_lr={tuple(alpha_lr.shape)} input={tuple(input.shape)} "
            f"occ={tuple(occ.shape)} tw={tw} scale={scale}")
    if dist is not None:
        dist = _c(dist)
        if tuple(dist.shape[:2]) != (b, nl - 1):
            raise _lib.WaldoHipError(f"flow_ctx_alpha: dist {tuple(dist.shape)} is not (B, L-1, Nl)")
    no_grad = not (torch.is_grad_enabled() and (alpha_lr.requires_grad or occ.requires_grad or
                                                (dist is not None and dist.requires_grad)))
    if no_grad and (not want_alpha or want_bits):
        a01 = alpha_lr.new_empty(n, nl, hd, wd)
        alpha = alpha_lr.new_empty(n, nl, hd, wd) if want_alpha else None
        bits = torch.empty(n, hd, (wd + 63) // 64, dtype=torch.int32, device=alpha_lr.device) if want_bits else None
        with _lib.on_device(alpha_lr.device):
            _lib.call("waldo_flow_ctx_alpha_fwd", _lib.ptr(alpha_lr), _lib.ptr(input), _lib.ptr(dist), _lib.ptr(occ),
                      _lib.ptr(a01), _lib.ptr(alpha), _lib.ptr(bits), b, t, tw, nl,
                      dist.shape[2] if dist is not None else 0, c, chan_off, h, w, scale,
                      _lib.current_stream(alpha_lr.device))
        return (a01, alpha, bits) if want_bits else (a01, alpha)
    res = _FlowCtxAlpha.apply(alpha_lr, input, dist, occ, tw, chan_off, scale)
    return (*res, None) if want_bits else res


class _FlowCtxWarp(torch.autograd.Function):
    @staticmethod
    def forward(ctx, flow_lr, isobj_lr, a01, ctx_ts, pred_ts, occ, tw, scale, layer_max, status_ptr, bits_ptr=None):
        m, nl, _, h, w = flow_lr.shape
        b, tc, tp = ctx_ts.shape
        t = occ.shape[1]
        hd, wd = a01.shape[-2:]
        flow = flow_lr.new_empty(m, 2, hd, wd)
        alpha_ctx = flow_lr.new_empty(m, nl, hd, wd)
        disocc = flow_lr.new_empty(m, hd, wd)
        amax = flow_lr.new_empty(m, hd, wd) if layer_max else flow_lr.new_empty(0)
        with _lib.on_device(flow_lr.device):
            _lib.call("waldo_flow_ctx_warp_fwd", _lib.ptr(flow_lr), _lib.ptr(isobj_lr), _lib.ptr(a01),
                      _lib.ptr(ctx_ts), _lib.ptr(pred_ts), _lib.ptr(occ), _lib.ptr(flow), _lib.ptr(alpha_ctx),
                      _lib.ptr(disocc), _lib.ptr(amax) if layer_max else None, bits_ptr, status_ptr, b, t, tw, tc, tp, nl,
                      h, w, scale, _lib.current_stream(flow_lr.device))
        ctx.save_for_backward(flow_lr, isobj_lr, a01, ctx_ts, pred_ts, occ)
        ctx.cfg = (tw, scale)
        ctx.mark_non_differentiable(amax)
        ctx.set_materialize_grads(False)  # unused outputs: None, not zero-filled tensors (the kernel takes NULL)
        return flow, alpha_ctx, disocc, amax

    @staticmethod
    def backward(ctx, g_flow, g_actx, g_dis, _g_amax):
        flow_lr, isobj_lr, a01, ctx_ts, pred_ts, occ = ctx.saved_tensors
        tw, scale = ctx.cfg
        m, nl, _, h, w = flow_lr.shape
        b, tc, tp = ctx_ts.shape
        t = occ.shape[1]
        hd, wd = a01.shape[-2:]
        g_flow = _c(g_flow) if g_flow is not None else None
        g_actx = _c(g_actx) if g_actx is not None else None
        g_dis = _c(g_dis) if g_dis is not None else None
        g_lr = torch.empty_like(flow_lr)
        g_a01, g_occ = _zeros_like_each(a01 if ctx.needs_input_grad[2] else None, occ if ctx.needs_input_grad[5] else None)
        ws = flow_lr.new_empty(m, nl, 2, hd, wd) if scale > 1 else None
        with _lib.on_device(flow_lr.device):
            _lib.call("waldo_flow_ctx_warp_bwd", _lib.ptr(flow_lr), _lib.ptr(isobj_lr), _lib.ptr(a01),
                      _lib.ptr(ctx_ts), _lib.ptr(pred_ts), _lib.ptr(occ), _lib.ptr(g_flow), _lib.ptr(g_actx),
                      _lib.ptr(g_dis), _lib.ptr(g_lr), _lib.ptr(g_a01), _lib.ptr(g_occ), _lib.ptr(ws), b, t, tw,
                      tc, tp, nl, h, w, scale, _lib.current_stream(flow_lr.device))
        return g_lr, None, g_a01, None, None, g_occ, None, None, None, None, None


def _flow_ctx_warp_args(fn, flow_lr, isobj_lr, a01, ctx_ts, pred_ts, occ, tw, scale):
    _lib.check_cuda(flow_lr, a01, occ)
    if not (ctx_ts.is_cuda and pred_ts.is_cuda):
        raise _lib.WaldoHipError(f"{fn}: ctx_ts / pred_ts must be on the GPU")
    flow_lr, a01, occ = _c(flow_lr), _c(a01), _c(occ)
    ctx_ts, pred_ts = _c(ctx_ts.long()), _c(pred_ts.long())
    m, nl, _, h, w = flow_lr.shape
    b, tc, tp = ctx_ts.shape
    t = occ.shape[1]
    if m != b * tc * tp or pred_ts.numel() != tp or tuple(a01.shape) != (b * tw, nl, h * scale, w * scale) \
            or tuple(occ.shape) != (b, t, nl, nl):
        raise _lib.WaldoHipError(
            f"{fn}: inconsistent shapes flow_lr={tuple(flow_lr.shape)} a01={tuple(a01.shape)} "
            f"ctx_ts={tuple(ctx_ts.shape)} pred_ts={tuple(pred_ts.shape)} occ={tuple(occ.shape)}")
    if isobj_lr is not None:
        _lib.check_cuda(isobj_lr)
        isobj_lr = _c(isobj_lr.detach())
        if tuple(isobj_lr.shape) != (m, nl - 1, h, w):
            raise _lib.WaldoHipError(f"{fn}: isobj_lr {tuple(isobj_lr.shape)} is not (M, L-1, H, W)")
    return flow_lr, isobj_lr, a01, ctx_ts, pred_ts, occ


def _layer_bits_ptr(fn, layer_bits, a01):
    """``flow_ctx_alpha(..., want_bits=True)``'s map for this very ``a01`` (shape-checked), or NULL."""
    if layer_bits is None:
        return None
    n, _, hd, wd = a01.shape
    if not layer_bits.is_cuda or layer_bits.dtype != torch.int32 or not layer_bits.is_contiguous() or \
            tuple(layer_bits.shape) != (n, hd, (wd + 63) // 64):
        raise _lib.WaldoHipError(f"{fn}: layer_bits {tuple(layer_bits.shape)} {layer_bits.dtype} is not the int32 "
                                 f"(B*Tw, Hd, ceil(Wd / 64)) map of a01 {tuple(a01.shape)}")
    return layer_bits.data_ptr()


def flow_ctx_warp(flow_lr, isobj_lr, a01, ctx_ts, pred_ts, occ, tw, scale, layer_max=False, status=None,
                  layer_bits=None):
    """Context-alpha warp + ghost mask + disocclusion + second occlusion product + flow compositing
    (models/nets/lvd.py:784-818).  flow_lr (B*Tc*Tp, L, 2, H, W); isobj_lr (B*Tc*Tp, L-1, H, W) or None;
    a01 (B*Tw, L, Hd, Wd) from flow_ctx_alpha; ctx_ts (B, Tc, Tp) long; pred_ts (Tp) long;
    occ (B, T, L, L).  Returns flow (M, 2, Hd, Wd), alpha_ctx (M, L, Hd, Wd) in [-1, 1],
    disocc (M, Hd, Wd).  Differentiable w.r.t. flow_lr, a01 and occ (the thresholded ghost mask
    carries no gradient, as in the reference).  ``layer_max``: a fourth result, ``alpha_ctx.amax(dim=1)``
    (M, Hd, Wd) -- what Synthesizer.predict's disocclusion test computes from alpha_ctx (synthesizer.py:447) --
    as a by-product (no gradient).  ``ctx_ts`` must lie in [0, tw), ``pred_ts`` in [0, T): validated on the device
    (``status``: the caller's ``_lib.IndexStatus``, checked when the caller chooses; None: checked before returning).
    ``layer_bits``: ``flow_ctx_alpha(..., want_bits=True)``'s map for this ``a01`` -- without a ghost mask it lets the
    pass skip, per tile, the layers that are absent wherever the tile samples; the values do not depend on it."""
    flow_lr, isobj_lr, a01, ctx_ts, pred_ts, occ = _flow_ctx_warp_args("flow_ctx_warp", flow_lr, isobj_lr, a01, ctx_ts,
                                                                       pred_ts, occ, tw, scale)
    st, strict = _status(status)
    flow, alpha_ctx, disocc, amax = _FlowCtxWarp.apply(flow_lr, isobj_lr, a01, ctx_ts, pred_ts, occ, tw, scale,
                                                       bool(layer_max), st.ptr,
                                                       _layer_bits_ptr("flow_ctx_warp", layer_bits, a01))
    if strict:
        st.check(sync=True)
    return (flow, alpha_ctx, disocc, amax) if layer_max else (flow, alpha_ctx, disocc)


class RawSlots:
    """What ``flow_ctx_warp_into_raw`` leaves for ``frame_warp_fuse_raw``: the ``raw`` tensor of
    Warper.input_to_output, (B, Tp, Tc', C + L, Hd, Wd), with the alpha slots of its Tc contexts filled, and
    ``score`` (B, Tc, Tp, Hd, Wd) = the per-context sums of (alpha + 1) / 2 (lvd.py:841)."""

    def __init__(self, raw, score, channels, include_self):
        self.raw, self.score, self.channels, self.include_self = raw, score, channels, include_self


def flow_ctx_warp_into_raw(flow_lr, isobj_lr, a01, ctx_ts, pred_ts, occ, tw, scale, channels, include_self,
                           layer_max=False, status=None, layer_bits=None):
    """``flow_ctx_warp`` for the caller that runs ``frame_warp_fuse_raw`` on the result next
    (LVD.forward(mode="decode_output"), lvd.py:141-153), WITHOUT autograd: ``alpha_ctx`` is written straight
    into the alpha slots of input_to_output's ``raw`` tensor (lvd.py:846) and returned as a strided
    (B*Tc*Tp -> B, Tc, Tp, L, Hd, Wd) view of it.  ``channels`` = C of the frames that will be warped,
    ``include_self``: whether ``raw`` gets the extra self context.  Returns (flow, alpha_ctx view (B, Tc, Tp, L,
    Hd, Wd), disocc, amax or None, slots): ``slots`` (a ``RawSlots``) goes to ``frame_warp_fuse_raw``."""
    if torch.is_grad_enabled() and any(x is not None and x.requires_grad for x in (flow_lr, a01, occ)):
        raise _lib.WaldoHipError("flow_ctx_warp_into_raw: no gradient flows through the raw-slot path; use flow_ctx_warp")
    flow_lr, isobj_lr, a01, ctx_ts, pred_ts, occ = _flow_ctx_warp_args("flow_ctx_warp_into_raw", flow_lr, isobj_lr, a01,
                                                                       ctx_ts, pred_ts, occ, tw, scale)
    m, nl, _, h, w = flow_lr.shape
    b, tc, tp = ctx_ts.shape
    t = occ.shape[1]
    hd, wd = a01.shape[-2:]
    tcx = tc + (1 if include_self else 0)
    st, strict = _status(status)
    with torch.no_grad():
        flow = flow_lr.new_empty(m, 2, hd, wd)
        raw = flow_lr.new_empty(b, tp, tcx, channels + nl, hd, wd)
        score = flow_lr.new_empty(b, tc, tp, hd, wd)
        disocc = flow_lr.new_empty(m, hd, wd)
        amax = flow_lr.new_empty(m, hd, wd) if layer_max else None
        with _lib.on_device(flow_lr.device):
            _lib.call("waldo_flow_ctx_warp_raw_fwd", _lib.ptr(flow_lr), _lib.ptr(isobj_lr), _lib.ptr(a01),
                      _lib.ptr(ctx_ts), _lib.ptr(pred_ts), _lib.ptr(occ), _lib.ptr(flow), _lib.ptr(raw),
                      _lib.ptr(score), _lib.ptr(disocc), _lib.ptr(amax),
                      _layer_bits_ptr("flow_ctx_warp_into_raw", layer_bits, a01), st.ptr, b, t, tw, tc, tp, nl, h, w,
                      scale, int(channels), tcx, _lib.current_stream(flow_lr.device))
        alpha_ctx = raw[:, :, :tc, channels:].permute(0, 2, 1, 3, 4, 5)  # (B, Tc, Tp, L, Hd, Wd), strided
    if strict:
        st.check(sync=True)
    return flow, alpha_ctx, disocc, amax, RawSlots(raw, score, int(channels), bool(include_self))


MAX_FUSE_CTX = 8


class _FrameWarpFuse(torch.autograd.Function):
    @staticmethod
    def forward(ctx, input, flow, alpha, ctx_ts, include_self, eps, status_ptr):
        b, t, c, hd, wd = input.shape
        _, tc, tp, nl = alpha.shape[:4]
        tcx = tc + (1 if include_self else 0)
        out = input.new_empty(b, tp, c + 1, hd, wd)
        # stored (B, Tp, Tc', ...) and handed out as the reference's (B, Tc', Tp, ...) VIEW: WIF.forward's
        # permute(0, 2, 1, ...).contiguous() (wif.py:39) then is a no-op instead of a copy of the
        # pipeline's largest tensor (C4 recipe: 11.8 GB read + written per predict)
        raw = input.new_empty(b, tp, tcx, c + nl, hd, wd)
        with _lib.on_device(input.device):
            _lib.call("waldo_frame_warp_fuse_fwd", _lib.ptr(input), _lib.ptr(flow), _lib.ptr(alpha),
                      _lib.ptr(ctx_ts), _lib.ptr(out), _lib.ptr(raw), status_ptr, b, t, tc, tp, c, nl, hd, wd,
                      1 if include_self else 0, float(eps), _lib.current_stream(input.device))
        ctx.save_for_backward(input, flow, alpha, ctx_ts)
        ctx.cfg = (bool(include_self), float(eps))
        # an output the loss does not use comes back as None, not as a zero-filled tensor of its size (raw_output is
        # the largest tensor of the chain; the kernel takes NULL for either gradient)
        ctx.set_materialize_grads(False)
        return out, raw.permute(0, 2, 1, 3, 4, 5)

    @staticmethod
    def backward(ctx, g_out, g_raw):
        input, flow, alpha, ctx_ts = ctx.saved_tensors
        include_self, eps = ctx.cfg
        b, t, c, hd, wd = input.shape
        _, tc, tp, nl = alpha.shape[:4]
        g_out = _c(g_out) if g_out is not None else None
        g_raw = _c(g_raw) if g_raw is not None else None
        g_flow = torch.empty_like(flow)
        g_alpha = torch.empty_like(alpha)
        with _lib.on_device(input.device):
            _lib.call("waldo_frame_warp_fuse_bwd", _lib.ptr(input), _lib.ptr(flow), _lib.ptr(alpha),
                      _lib.ptr(ctx_ts), _lib.ptr(g_out), _lib.ptr(g_raw), _lib.ptr(g_flow), _lib.ptr(g_alpha), b, t,
                      tc, tp, c, nl, hd, wd, 1 if include_self else 0, eps, _lib.current_stream(input.device))
        return None, g_flow, g_alpha, None, None, None, None


def frame_warp_fuse(input, flow, alpha, ctx_ts, include_self=False, eps=1e-6, status=None):
    """Warper.input_to_output (models/nets/lvd.py:830-853).  input (B,T,C,Hd,Wd);
    flow (B,Tc,Tp,2,Hd,Wd); alpha (B,Tc,Tp,L,Hd,Wd) in [-1,1]; ctx_ts (B,Tc,Tp) long.
    Returns (out (B,Tp,C+1,Hd,Wd), raw (B,Tc',Tp,C+L,Hd,Wd)).  Differentiable w.r.t. flow and alpha;
    the frames in ``input`` are data (no gradient is produced for them).  ``ctx_ts`` must lie in [0, T): validated
    on the device (``status``: see ``flow_ctx_warp``)."""
    _lib.check_cuda(input, flow, alpha)
    if not ctx_ts.is_cuda:
        raise _lib.WaldoHipError("frame_warp_fuse: ctx_ts must be on the GPU")
    input, flow = _c(input.detach()), _c(flow)
    ctx_ts = _c(ctx_ts.long())
    b, t, c, hd, wd = input.shape
    _, tc, tp, nl = alpha.shape[:4]
    if tuple(flow.shape) != (b, tc, tp, 2, hd, wd) or tuple(alpha.shape) != (b, tc, tp, nl, hd, wd) \
            or tuple(ctx_ts.shape) != (b, tc, tp):
        raise _lib.WaldoHipError(
            f"frame_warp_fuse: inconsistent shapes input={tuple(input.shape)} flow={tuple(flow.shape)} "
            f"alpha={tuple(alpha.shape)} ctx_ts={tuple(ctx_ts.shape)}")
    st, strict = _status(status)
    res = _FrameWarpFuse.apply(input, flow, _c(alpha), ctx_ts, bool(include_self), eps, st.ptr)
    if strict:
        st.check(sync=True)
    return res


def frame_warp_fuse_raw(input, flow, slots, ctx_ts, eps=1e-6, status=None):
    """``frame_warp_fuse`` behind ``flow_ctx_warp_into_raw`` (no autograd): the context alphas already sit in
    ``slots.raw`` and their per-context sums in ``slots.score``, so they are neither read nor copied -- one score
    plane per context instead of L alpha planes.  The same bits as ``frame_warp_fuse`` on the alpha view
    (tests/test_gpu_warper.py::test_alpha_ctx_written_into_raw_slots).  Returns (out, raw as (B, Tc', Tp, ...))."""
    _lib.check_cuda(input, flow)
    if not ctx_ts.is_cuda:
        raise _lib.WaldoHipError("frame_warp_fuse_raw: ctx_ts must be on the GPU")
    if torch.is_grad_enabled() and flow.requires_grad:
        raise _lib.WaldoHipError("frame_warp_fuse_raw: no gradient flows through the raw-slot path; use frame_warp_fuse")
    input, flow = _c(input.detach()), _c(flow.detach())
    ctx_ts = _c(ctx_ts.long())
    b, t, c, hd, wd = input.shape
    raw, score = slots.raw, slots.score
    _, tc, tp = score.shape[:3]
    nl = raw.shape[3] - c
    tcx = tc + (1 if slots.include_self else 0)
    if slots.channels != c or tuple(raw.shape) != (b, tp, tcx, c + nl, hd, wd) or tuple(score.shape) != (b, tc, tp, hd, wd) \
            or tuple(flow.shape) != (b, tc, tp, 2, hd, wd) or tuple(ctx_ts.shape) != (b, tc, tp):
        raise _lib.WaldoHipError(
            f"frame_warp_fuse_raw: inconsistent shapes input={tuple(input.shape)} flow={tuple(flow.shape)} "
            f"raw={tuple(raw.shape)} score={tuple(score.shape)} ctx_ts={tuple(ctx_ts.shape)}")
    st, strict = _status(status)
    out = input.new_empty(b, tp, c + 1, hd, wd)
    with torch.no_grad(), _lib.on_device(input.device):
        _lib.call("waldo_frame_warp_fuse_raw_fwd", _lib.ptr(input), _lib.ptr(flow), _lib.ptr(score),
                  _lib.ptr(ctx_ts), _lib.ptr(out), _lib.ptr(raw), st.ptr, b, t, tc, tp, c, nl, hd, wd,
                  1 if slots.include_self else 0, float(eps), _lib.current_stream(input.device))
    if strict:
        st.check(sync=True)
    return out, raw.permute(0, 2, 1, 3, 4, 5)


# --------------------------------------------------------------------------------------
# A8: gather_time and the frame arithmetic built on it
# --------------------------------------------------------------------------------------
class _TimeGather(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, ctx_ts, pred_ts, tc, hw, subtract, status_ptr):
        b, t = x.shape[:2]
        p = math.prod(x.shape[2:]) // 2
        tp = pred_ts.numel()
        out = x.new_empty(b, tc, tp, *((p // hw, 2, hw) if hw else (p, 2)))
        with _lib.on_device(x.device):
            _lib.call("waldo_time_gather_fwd", _lib.ptr(x), _lib.ptr(ctx_ts), _lib.ptr(pred_ts), _lib.ptr(out),
                      status_ptr, b, t, tc, tp, p, hw, int(subtract), _lib.current_stream(x.device))
        ctx.save_for_backward(ctx_ts, pred_ts)
        ctx.cfg = (tuple(x.shape), tc, hw, subtract)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        ctx_ts, pred_ts = ctx.saved_tensors
        shape, tc, hw, subtract = ctx.cfg
        grad_out = _c(grad_out)
        gx = grad_out.new_empty(shape)
        b, t = shape[:2]
        with _lib.on_device(grad_out.device):
            _lib.call("waldo_time_gather_bwd", _lib.ptr(grad_out), _lib.ptr(ctx_ts), _lib.ptr(pred_ts), _lib.ptr(gx),
                      b, t, tc, pred_ts.numel(), math.prod(shape[2:]) // 2, hw, int(subtract),
                      _lib.current_stream(grad_out.device))
        return gx, None, None, None, None, None, None


def time_gather(x, ctx_ts, pred_ts, num_ctx=None, subtract=False, channel_first=False, status=None):
    """``gather_time`` (models/nets/lvd.py:462-467) and the frame arithmetic of the flow synthesis
    (lvd.py:660-668, 780-787) on a clip's grids ``x`` (B, T, ..., 2):

    - ``subtract``: ``gather_time(x, ctx_ts) - x[:, pred_ts].unsqueeze(1)``            -> (B, Tc, Tp, ..., 2)
    - ``ctx_ts is None``: ``x[:, pred_ts].unsqueeze(1).expand(-1, num_ctx, ...)``     -> (B, num_ctx, Tp, ..., 2)
    - otherwise ``gather_time(x, ctx_ts)``.

    ``channel_first``: x is (B, T, N, H, W, 2) and the result (B, Tc, Tp, N, 2, H, W) -- the
    ``permute(0, 1, 2, 3, 6, 4, 5)`` of lvd.py:662.  Differentiable w.r.t. ``x``.  Frame indices must lie in [0, T):
    validated on the device (``status``: see ``flow_ctx_warp``).  For ONE context and a ``pred_ts`` made by
    ``arange_index(T)`` the result is a VIEW of ``x`` (see there)."""
    _lib.check_cuda(x)
    if x.ndim < 3 or x.shape[-1] != 2 or (channel_first and x.ndim != 6):
        raise _lib.WaldoHipError(f"time_gather: x {tuple(x.shape)} is not a clip of grids (B, T, ..., 2)")
    if not pred_ts.is_cuda or (ctx_ts is not None and not ctx_ts.is_cuda):
        raise _lib.WaldoHipError("time_gather: frame indices must be on the GPU")
    x = _c(x.float())
    pred_ts = _c(pred_ts.long())
    b, t = x.shape[:2]
    if ctx_ts is not None:
        ctx_ts = _c(ctx_ts.long())
        if ctx_ts.ndim != 3 or ctx_ts.shape[0] != b or ctx_ts.shape[2] != pred_ts.numel():
            raise _lib.WaldoHipError(f"time_gather: ctx_ts {tuple(ctx_ts.shape)} against B={b}, Tp={pred_ts.numel()}")
        tc = ctx_ts.shape[1]
    else:
        if subtract or num_ctx is None:
            raise _lib.WaldoHipError("time_gather: without ctx_ts give num_ctx (and no difference)")
        tc = int(num_ctx)
    tp = pred_ts.numel()
    if ctx_ts is None and tc == 1 and tp == t and not channel_first and _is_arange_index(pred_ts):
        # x[:, [0, 1, ..., T - 1]] for one context: the clip itself (a view: no copy, and the backward is autograd's sum
        # instead of a scatter kernel) -- the LVD recipe's `ctx_mode "prev"`, where every frame is predicted
        return x.view(b, 1, t, *x.shape[2:])
    hw = x.shape[3] * x.shape[4] if channel_first else 0
    st, strict = _status(status)
    out = _TimeGather.apply(x, ctx_ts, pred_ts, tc, hw, bool(subtract), st.ptr)
    if strict:
        st.check(sync=True)
    if channel_first:
        return out.view(b, tc, tp, x.shape[2], 2, x.shape[3], x.shape[4])
    return out.view(b, tc, tp, *x.shape[2:])


def downscale_frames(input, num_frames, first_channel, factor):
    """``scale(input[:, :num_frames, first_channel:], 1 / factor)`` of ``Warper.grid_to_flow[_ctx]``
    (models/nets/lvd.py:611 / 716): the low-resolution copy of the layout channels, the same bits as
    ``F.interpolate(..., scale_factor=1 / factor, mode="bilinear")`` on the device for a power-of-two ``factor``.
    The frames are data: the result carries no gradient."""
    _lib.check_cuda(input)
    b, t, c, hd, wd = input.shape
    s = int(factor)
    if s < 2 or s & (s - 1) or hd % s or wd % s:
        raise _lib.WaldoHipError(f"downscale_frames: factor {factor} on {hd} x {wd} frames (a power of two that divides both)")
    x = _c(input.detach().float())
    out = x.new_empty(b, int(num_frames), c - int(first_channel), hd // s, wd // s)
    with _lib.on_device(x.device):
        _lib.call("waldo_downscale_frames_fwd", _lib.ptr(x), _lib.ptr(out), b, t, int(num_frames), c,
                  int(first_channel), hd // s, wd // s, s, _lib.current_stream(x.device))
    return out


def points_in_polygon(pts, corners):
    """``matplotlib.path.Path(corners).contains_points(pts)`` (radius 0, no transform) on the device, as ``WIF.inpaint``
    uses it (models/nets/wif.py:228-235): ``pts`` (..., 2) float32 (x, y) on the GPU, ``corners`` a sequence of 3 ... 16
    (x, y) pairs of host numbers -> a bool tensor of ``pts.shape[:-1]``.  matplotlib's crossings test in double precision,
    operation by operation: points on an edge get matplotlib's answer."""
    import ctypes
    _lib.check_cuda(pts)
    if pts.shape[-1] != 2:
        raise _lib.WaldoHipError(f"points_in_polygon: points of shape {tuple(pts.shape)} (..., 2)")
    flat = [float(v) for c in corners for v in c]
    k = len(flat) // 2
    if len(flat) != 2 * k or any(len(c) != 2 for c in corners) or k > 16:
        raise _lib.WaldoHipError(f"points_in_polygon: {len(corners)} corners (pairs, at most 16)")
    host = (ctypes.c_double * max(len(flat), 1))(*flat)
    x = _c(pts.detach())
    n = x.numel() // 2
    out = x.new_empty(x.shape[:-1])
    with _lib.on_device(x.device):
        _lib.call("waldo_points_in_polygon_fwd", _lib.ptr(x), ctypes.addressof(host), k, _lib.ptr(out), n,
                  _lib.current_stream(x.device))
    return out > 0


def inpaint_propagate(flow, ident, ref_img, ref_mask, shadow, entering, img, todo, obj, soft_shadow=False,
                      fix_mask=False):
    """One frame of the propagation loop of ``WIF.inpaint`` (models/nets/wif.py:179-211) in one launch -- the warps of
    the reference background, its mask and the shadow mask by ``flow + ident``, the entering objects
    (``entering``: up to two ``(region, look, flow_k)``), the fill of the frame's holes and the inpainter's inputs --
    with the bits of the spelled-out composition.  Returns ``(img, todo, inpainter_img, inpainter_mask)``; with
    ``fix_mask`` the inpainter takes ``img`` itself and ``inpainter_mask`` is the undilated ``1 - (1 - todo)(1 - obj)``."""
    import ctypes
    _lib.check_cuda(flow, ident, ref_img, ref_mask, shadow, img, todo, obj)
    b, c, h, w = img.shape
    if c != 3 or len(entering) > 2:
        raise _lib.WaldoHipError(f"inpaint_propagate: {c} channels, {len(entering)} entering objects (3; at most 2)")
    flow, ident, ref_img, ref_mask, img, todo, obj = (_c(x.detach()) for x in (flow, ident, ref_img, ref_mask, img, todo, obj))
    shadow = _c(shadow.detach()) if shadow is not None else None
    for x, shape in ((flow, (b, h, w, 2)), (ref_img, (b, 3, h, w)), (ref_mask, (b, 1, h, w)), (todo, (b, 1, h, w)),
                     (obj, (b, 1, h, w))) + (((shadow, (b, 1, h, w)),) if shadow is not None else ()):
        if tuple(x.shape) != shape:
            raise _lib.WaldoHipError(f"inpaint_propagate: a tensor of shape {tuple(x.shape)} where {shape} is expected")
    if ident.numel() != h * w * 2:
        raise _lib.WaldoHipError(f"inpaint_propagate: an identity grid of shape {tuple(ident.shape)} for {h} x {w} frames")
    keep = []  # (contiguous copies must outlive the launch's enqueue)
    ptrs = [[], [], []]
    for region, look, fk in entering:
        _lib.check_cuda(region, look, fk)
        region, look, fk = _c(region.detach().expand(b, 1, h, w)), _c(look.detach()), _c(fk.detach())
        if tuple(look.shape) != (b, 3, h, w) or tuple(fk.shape) != (b, h, w, 2):
            raise _lib.WaldoHipError("inpaint_propagate: an entering object's look / flow has the wrong shape")
        keep += [region, look, fk]
        for lst, x in zip(ptrs, (region, look, fk)):
            lst.append(x.data_ptr())
    arrs = [(ctypes.c_void_p * 2)(*(lst + [None] * (2 - len(lst)))) for lst in ptrs]
    img_out, todo_out, inp_mask = torch.empty_like(img), torch.empty_like(todo), torch.empty_like(todo)
    inp_img = None if fix_mask else torch.empty_like(img)
    with _lib.on_device(img.device):
        _lib.call("waldo_inpaint_propagate_fwd", _lib.ptr(flow), _lib.ptr(ident), _lib.ptr(ref_img), _lib.ptr(ref_mask),
                  _lib.ptr(shadow), ctypes.addressof(arrs[0]), ctypes.addressof(arrs[1]), ctypes.addressof(arrs[2]),
                  len(entering), _lib.ptr(img), _lib.ptr(todo), _lib.ptr(obj), _lib.ptr(img_out), _lib.ptr(todo_out),
                  _lib.ptr(inp_img), _lib.ptr(inp_mask), b, h, w, int(bool(soft_shadow)), int(bool(fix_mask)),
                  _lib.current_stream(img.device))
    return img_out, todo_out, (img_out if fix_mask else inp_img), inp_mask


def inpaint_holes(alpha_ctx, last_only=False, fix_thresh=True):
    """The hole and object masks of ``WIF.inpaint`` (models/nets/wif.py:60-75) from ``alpha_ctx`` (B, Tc, Tp, L, H, W)
    in [-1, 1] in ONE pass (any strides over the first four dimensions: the raw-slot view of ``decode_output`` is taken
    as it is): ``(mask, obj_mask)``, each (B, Tp, 1, H, W) of 0 / 1 -- before the optional expansion of wif.py:76-77.
    The sums over the layers are taken in the order of the framework's reduction: the same mask pixels."""
    _lib.check_cuda(alpha_ctx)
    if alpha_ctx.dim() != 6:
        raise _lib.WaldoHipError(f"inpaint_holes: alpha_ctx of shape {tuple(alpha_ctx.shape)} (B, Tc, Tp, L, H, W)")
    x = alpha_ctx.detach()
    b, tc, tp, nl, h, w = x.shape
    if x.stride(5) != 1 or x.stride(4) != w or min(x.stride()[:4]) < 0:
        x = x.contiguous()
    mask = x.new_empty(b, tp, 1, h, w)
    obj_mask = x.new_empty(b, tp, 1, h, w)
    with _lib.on_device(x.device):
        _lib.call("waldo_inpaint_holes_fwd", _lib.ptr(x), x.stride(0), x.stride(1), x.stride(2), x.stride(3), _lib.ptr(mask),
                  _lib.ptr(obj_mask), b, tc, tp, nl, h * w, int(bool(last_only)), 0.1 if fix_thresh else 0.9,
                  _lib.current_stream(x.device))
    return mask, obj_mask


def inpaint_blend(img, todo, fill):
    """``(1 - todo) * img + todo * fill`` (wif.py:214) in one launch; img / fill (B, 3, H, W), todo (B, 1, H, W)."""
    _lib.check_cuda(img, todo, fill)
    img, todo, fill = _c(img.detach()), _c(todo.detach()), _c(fill.detach())
    b, c, h, w = img.shape
    if c != 3 or tuple(fill.shape) != tuple(img.shape) or tuple(todo.shape) != (b, 1, h, w):
        raise _lib.WaldoHipError(f"inpaint_blend: shapes {tuple(img.shape)}, {tuple(todo.shape)}, {tuple(fill.shape)}")
    out = torch.empty_like(img)
    with _lib.on_device(img.device):
        _lib.call("waldo_inpaint_blend_fwd", _lib.ptr(img), _lib.ptr(todo), _lib.ptr(fill), _lib.ptr(out), b, h * w,
                  _lib.current_stream(img.device))
    return out


_EXPAND_STEPS = {None: 15, "": 15, "south": 1, "north": 2, "east": 4, "west": 8}


def mask_expand(mask, num=1, dir=None, soft=False, alpha=0.97):
    """The reference's ``expand`` (tools/utils.py:300-323) as ONE launch: ``num`` rounds of one-pixel growth towards
    the south, north, east and west in that order along dims 2 and 3 of ``mask``, each step seeing the one before
    (``dir`` keeps one of the four); hard masks (``soft=False``) are read as ``mask != 0`` and returned as float
    0 / 1, soft ones grow by ``max(pixel, alpha * neighbour)``.  The same bits as the framework's 8 * num launches
    (``waldo_amd.tools.utils.expand`` states them) for every input; never writes its argument.

    ``mask``: (N, C, H, W) -- or, as ``WIF.inpaint`` calls it on its (B, Tp, 1, H, W) hole masks (wif.py:77), five
    dimensions of which dim 2 has size 1: dims 2 and 3 are then (1, H), so "south / north" have nothing to do and "east /
    west" run along H -- the reference's own behaviour, kept."""
    if not mask.is_cuda:
        raise _lib.WaldoHipError("waldo_amd ops need tensors on the GPU (cuda device); there is no CPU fallback")
    if dir not in _EXPAND_STEPS:
        raise ValueError(f"mask_expand: dir {dir!r} (south, north, east, west or None)")
    steps = _EXPAND_STEPS[dir]
    if mask.dim() == 4:
        planes, h, w = mask.shape[0] * mask.shape[1], mask.shape[2], mask.shape[3]
    elif mask.dim() == 5 and mask.shape[2] == 1:
        # dims (2, 3) = (1, H) with W elementwise behind them: east / west of the reference run along H = the kernel's
        # south / north on (H, W) planes; its south / north see a dimension of size 1
        planes, h, w = mask.shape[0] * mask.shape[1], mask.shape[3], mask.shape[4]
        steps = ((steps >> 2) & 3)
    else:
        raise _lib.WaldoHipError(f"mask_expand: a mask of shape {tuple(mask.shape)} (four dimensions, or five with a "
                                 f"dim 2 of size 1)")
    if soft and mask.dtype != torch.float32:
        raise _lib.WaldoHipError(f"mask_expand: a soft mask of dtype {mask.dtype} (float32: the kernel's arithmetic)")
    x = _c(mask.detach().float())
    num = int(num)
    if num == 0 or steps == 0 or planes == 0:
        return x.clone() if soft else (x != 0).float()
    out = torch.empty_like(x)
    scratch = torch.empty_like(x) if num > 30 else None
    with _lib.on_device(x.device):
        _lib.call("waldo_mask_expand_fwd", _lib.ptr(x), _lib.ptr(out), _lib.ptr(scratch), planes, h, w, num, steps,
                  1 if soft else 0, float(alpha), _lib.current_stream(x.device))
    return out


# --------------------------------------------------------------------------------------
# fused hot path
# --------------------------------------------------------------------------------------
class _WarpComposite(torch.autograd.Function):
    @staticmethod
    def forward(ctx, layers, mapping, occ, basis_t, want_alpha, delta):
        _lib.check_cuda(layers, mapping, occ, basis_t)
        layers, mapping, occ, basis_t = _c(layers), _c(mapping), _c(occ), _c(basis_t)
        f, nl, c, h, w = layers.shape
        if c != 4:
            raise _lib.WaldoHipError("warp_composite expects (F, L, 4, H, W) layers (RGB + alpha)")
        k3 = mapping.shape[1]
        if mapping.shape[0] != f * nl or tuple(occ.shape) != (f, nl, nl) or \
                tuple(basis_t.shape) != (k3, h * w):
            raise _lib.WaldoHipError(
                f"warp_composite: inconsistent shapes layers={tuple(layers.shape)} "
                f"mapping={tuple(mapping.shape)} occ={tuple(occ.shape)} basis_t={tuple(basis_t.shape)}")
        rgb = layers.new_empty(f, 3, h, w)
        alpha = layers.new_empty(f, nl, h, w) if want_alpha else None
        with _lib.on_device(layers.device):
            _lib.call("waldo_warp_composite_fwd", _lib.ptr(layers), _lib.ptr(basis_t),
                      _lib.ptr(mapping), _lib.ptr(occ), _lib.ptr(rgb), _lib.ptr(alpha), f, nl, h,
                      w, k3, float(delta), _lib.current_stream(layers.device))
        ctx.save_for_backward(layers, mapping, occ, basis_t)
        ctx.want_alpha = want_alpha
        ctx.delta = float(delta)
        return rgb, alpha

    @staticmethod
    def backward(ctx, grad_rgb, grad_alpha):
        layers, mapping, occ, basis_t = ctx.saved_tensors
        f, nl, _, h, w = layers.shape
        k3 = mapping.shape[1]
        grad_rgb = _c(grad_rgb)
        if grad_alpha is not None:
            grad_alpha = _c(grad_alpha)
        gm, go = _zeros_like_each(mapping if ctx.needs_input_grad[1] else None, occ if ctx.needs_input_grad[2] else None)
        # 0: the shape is served by the generic kernel (or a test asked for it: WALDO_DEBUG_BWD_GENERIC)
        ws_bytes = _lib.load().waldo_warp_composite_bwd_workspace_bytes(f, nl, h, w, k3)
        ws = torch.empty(ws_bytes // 4, dtype=torch.int32, device=layers.device) if ws_bytes else None
        # with a workspace the two-kernel path writes every texel of grad_layers exactly once;
        # the generic kernel accumulates with atomics into a zero-filled buffer
        gl = torch.empty_like(layers) if ws_bytes else torch.zeros_like(layers)
        with _lib.on_device(layers.device):
            _lib.call("waldo_warp_composite_bwd", _lib.ptr(layers), _lib.ptr(basis_t),
                      _lib.ptr(mapping), _lib.ptr(occ), _lib.ptr(grad_rgb), _lib.ptr(grad_alpha),
                      _lib.ptr(gl), _lib.ptr(gm), _lib.ptr(go), _lib.ptr(ws), ws_bytes, f, nl, h,
                      w, k3, ctx.delta, _lib.current_stream(layers.device))
        return gl, gm, go, None, None, None


def warp_composite(layers, src_pts, occ, inverse_kernel, basis_t, return_alpha=False, delta=0.0):
    """Fused TPS grid -> bilinear warp of each 4-channel layer -> LVD.reduce_comp
    (models/modules/warp.py:49-55, F.grid_sample, models/nets/lvd.py:100-114).

    layers (F, L, 4, H, W) in [-1, 1]; src_pts (F*L, N, 2); occ (F, L, L);
    inverse_kernel (N+3, N+3); basis_t (N+3, H*W).  Returns rgb (F, 3, H, W) and, if asked,
    the composited alpha (F, L, H, W), both in [-1, 1].

    delta: the layers are sampled as ``F.grid_sample(x + delta, grid) - delta`` (lvd.py:548,559).
    0 (default) is the BASELINE pipeline of SURVEY 8d: taps outside a layer contribute 0 (alpha 0.5 /
    grey after ``reduce_comp``'s ``(x + 1) / 2``); 1 is ``Warper.layer_to_output``'s default:
    out-of-range taps read -1, i.e. alpha 0 / black.  Precision / NaN contract of the backward:
    include/waldo_hip.h."""
    f, nl = layers.shape[:2]
    needs_grad = torch.is_grad_enabled() and any(
        torch.is_tensor(t) and t.requires_grad for t in (layers, src_pts, occ, inverse_kernel, basis_t))
    # the one-launch forward pays for its launch saving with a mapping computation per (tile, frame):
    # worth it while the call is launch-bound (C2: 25 -> 19 us), 4 % slower at C4 size
    small = f * ((layers.shape[-2] + 15) // 16) * ((layers.shape[-1] + 15) // 16) <= FOLD_MAX_TILE_FRAMES
    if not needs_grad and small and layers.dim() == 5 and layers.shape[2] == 4 and src_pts.dim() == 3 and \
            _lib.load().waldo_warp_composite_pts_supported(nl, layers.shape[-2], layers.shape[-1],
                                                           src_pts.shape[1]):
        return _warp_composite_pts(layers, src_pts, occ, inverse_kernel, basis_t, bool(return_alpha),
                                   float(delta))
    mapping = tps_mapping(inverse_kernel, src_pts)
    chunk = _frames_per_call(f, nl, layers.shape[-2], layers.shape[-1], mapping.shape[1])
    if f <= chunk:
        rgb, alpha = _WarpComposite.apply(layers, mapping, occ, basis_t, bool(return_alpha), float(delta))
    else:  # frames are independent: long batches go in pieces (launch limits, bounded workspace)
        outs = [_WarpComposite.apply(layers[i:i + chunk], mapping[i * nl:(i + chunk) * nl], occ[i:i + chunk],
                                     basis_t, bool(return_alpha), float(delta)) for i in range(0, f, chunk)]
        rgb = torch.cat([o[0] for o in outs])
        alpha = torch.cat([o[1] for o in outs]) if return_alpha else None
    return (rgb, alpha) if return_alpha else rgb


def _warp_composite_pts(layers, src_pts, occ, inverse_kernel, basis_t, want_alpha, delta):
    """Forward without autograd, straight from the control points: ONE launch per call
    (waldo_warp_composite_pts_fwd; the TPS mapping is computed inside the kernel, same bits as
    tps_mapping + the two-step forward)."""
    _lib.check_cuda(layers, src_pts, occ, inverse_kernel, basis_t)
    layers, src_pts, occ = _c(layers.detach()), _c(src_pts.detach().float()), _c(occ.detach())
    inverse_kernel, basis_t = _c(inverse_kernel.detach()), _c(basis_t.detach())
    f, nl, _, h, w = layers.shape
    n = src_pts.shape[1]
    if src_pts.shape[0] != f * nl or tuple(occ.shape) != (f, nl, nl) or \
            tuple(basis_t.shape) != (n + 3, h * w) or tuple(inverse_kernel.shape) != (n + 3, n + 3):
        raise _lib.WaldoHipError(
            f"warp_composite: inconsistent shapes layers={tuple(layers.shape)} src_pts={tuple(src_pts.shape)} "
            f"occ={tuple(occ.shape)} basis_t={tuple(basis_t.shape)} inverse_kernel={tuple(inverse_kernel.shape)}")
    rgb = layers.new_empty(f, 3, h, w)
    alpha = layers.new_empty(f, nl, h, w) if want_alpha else None
    per = max(1, MAX_FL_PER_LAUNCH // nl)
    with _lib.on_device(layers.device):
        for i in range(0, f, per):
            j = min(f, i + per)
            _lib.call("waldo_warp_composite_pts_fwd", _lib.ptr(layers[i:j]), _lib.ptr(basis_t),
                      _lib.ptr(inverse_kernel), _lib.ptr(src_pts[i * nl:j * nl]), _lib.ptr(occ[i:j]),
                      _lib.ptr(rgb[i:j]), _lib.ptr(alpha[i:j]) if want_alpha else None, j - i, nl, h, w, n,
                      delta, _lib.current_stream(layers.device))
    return (rgb, alpha) if want_alpha else rgb


FOLD_MAX_TILE_FRAMES = 8192     # largest (frames x 16x16 tiles) the one-launch forward is used for
MAX_WORKSPACE_BYTES = 8 << 30   # backward workspace per call of the fused path
MAX_FL_PER_LAUNCH = 65535       # F * L limit of one launch (include/waldo_hip.h)


def _frames_per_call(f, nl, h, w, k3):
    """Largest number of frames one call of the fused path may take."""
    if f == 0:
        return 1
    per = max(1, MAX_FL_PER_LAUNCH // nl)
    query = _lib.load().waldo_warp_composite_bwd_workspace_bytes
    ws1, ws8 = query(1, nl, h, w, k3), query(8, nl, h, w, k3)
    per_frame = max((ws8 - ws1) / 7.0, 1.0) if ws8 > 0 else 0.0
    if per_frame:
        per = min(per, max(1, int(MAX_WORKSPACE_BYTES // per_frame)))
    return per
