"""Multi-GPU use of the path: frames are independent units, so ranks SHARD them.

The reference's only parallelism is data parallel over clips (DDP, tools/engine.py:46-49, batch
split by DistributedSampler tools/engine.py:63-64).  For the warp/composite path that means: every
output frame (b, t) depends only on its own L layers, control points and occ matrix (SURVEY.md
section 8e), so each rank composites a contiguous block of frames with no data-path collective;
only inference that needs all frames everywhere ends with ONE all-gather of the composited frames
(RCCL over xGMI through torch.distributed's "nccl" backend on ROCm; "gloo" on CPU for the tests).
One process per GPU, launched by torchrun; rendezvous on 127.0.0.1 unless told otherwise.
"""
import os

import torch
import torch.distributed as dist


def init_distributed(backend=None, single_rank_group=False):
    """env:// initialisation mirroring the reference's Engine (tools/engine.py:17-35) without its
    SLURM branch.  Returns (rank, local_rank, world_size).  A single process needs no group;
    ``single_rank_group`` creates one anyway (the reference's Engine does, tools/engine.py:35: the
    collectives of a world of one then still go through RCCL -- what the one-GPU test box can run)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or single_rank_group) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            kw["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend=backend, init_method="env://", **kw)
    return rank, local_rank, world


def shard_range(n_units, rank, world):
    """Contiguous block [start, stop) of `n_units` frames for `rank`: blocks of ceil(n/world),
    the last ranks may get fewer (or none)."""
    per = (n_units + world - 1) // world
    start = min(rank * per, n_units)
    return start, min(start + per, n_units)


def shard_frames(tensors, frames, rank, world, layers=None):
    """Slice per-frame inputs for this rank.  Each tensor's leading dim is `frames` (e.g. layers
    (F, L, 4, H, W), occ (F, L, L)) or `frames * layers` (control points (F*L, K, 2))."""
    s, e = shard_range(frames, rank, world)
    out = []
    for t in tensors:
        if t.shape[0] == frames:
            out.append(t[s:e])
        elif layers is not None and t.shape[0] == frames * layers:
            out.append(t[s * layers:e * layers])
        else:
            raise ValueError(f"leading dim {t.shape[0]} is neither F={frames} nor F*L")
    return out


def all_gather_frames(local, frames, group=None, collective_for_one=False):
    """Gather every rank's block of composited frames (n_local, C, H, W) into (frames, C, H, W) on
    every rank.  Blocks may be ragged (shard_range): the payload is padded to the common block size,
    gathered with one all_gather_into_tensor and trimmed.  A world of one returns its block as it is
    unless ``collective_for_one`` asks for the collective anyway (tests: the RCCL call on one GPU)."""
    if not dist.is_initialized() or (dist.get_world_size(group) == 1 and not collective_for_one):
        assert local.shape[0] == frames
        return local
    world = dist.get_world_size(group)
    per = (frames + world - 1) // world
    if local.shape[0] > per:
        raise ValueError("local block larger than ceil(frames / world)")
    if local.shape[0] < per:
        pad = local.new_zeros(per - local.shape[0], *local.shape[1:])
        local = torch.cat([local, pad], dim=0)
    if local.is_cuda and dist.get_backend(group) == "gloo":
        # gloo moves host memory (smoke tests of the multi-rank path on a box with fewer GPUs than
        # ranks); RCCL ("nccl") gathers device buffers directly over xGMI
        host = local.cpu().contiguous()
        out = host.new_empty(world * per, *host.shape[1:])
        dist.all_gather_into_tensor(out, host, group=group)
        return out[:frames].to(local.device)
    out = local.new_empty(world * per, *local.shape[1:])
    dist.all_gather_into_tensor(out, local.contiguous(), group=group)
    return out[:frames]


class PendingGather:
    """An all-gather of composited frames in flight (``all_gather_frames_async``).  ``wait()`` makes the current
    stream wait for it and returns the (frames, C, H, W) tensor.  Holds the buffers the collective reads and writes."""

    def __init__(self, work, out, local, frames, device=None):
        self.work, self.out, self.local, self.frames, self.device = work, out, local, frames, device

    def wait(self):
        if self.work is not None:
            self.work.wait()
            self.work = None
        res = self.out[:self.frames]
        return res.to(self.device) if self.device is not None else res


def all_gather_frames_async(local, frames, group=None):
    """``all_gather_frames`` without waiting: the collective runs on the communicator's own stream (RCCL) while the
    caller launches the next frames' kernels; ``.wait()`` on the returned ``PendingGather`` before the result is read.
    The frames of step i travel over xGMI while step i + 1 computes -- at the Cityscapes recipe 352 MB per rank
    against a 28 ms step."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        assert local.shape[0] == frames
        return PendingGather(None, local, local, frames)
    world = dist.get_world_size(group)
    per = (frames + world - 1) // world
    if local.shape[0] > per:
        raise ValueError("local block larger than ceil(frames / world)")
    if local.shape[0] < per:
        local = torch.cat([local, local.new_zeros(per - local.shape[0], *local.shape[1:])], dim=0)
    if local.is_cuda and dist.get_backend(group) == "gloo":  # (host staging: see all_gather_frames)
        host = local.cpu().contiguous()
        out = host.new_empty(world * per, *host.shape[1:])
        return PendingGather(dist.all_gather_into_tensor(out, host, group=group, async_op=True), out, host, frames,
                             device=local.device)
    local = local.contiguous()
    out = local.new_empty(world * per, *local.shape[1:])
    return PendingGather(dist.all_gather_into_tensor(out, local, group=group, async_op=True), out, local, frames)


def sharded_warp_composite(layers, src_pts, occ, inverse_kernel, basis_t, gather=True, delta=0.0):
    """Inference over all ranks: composite this rank's frames on its GPU (HIP kernels), then
    all-gather the RGB frames.  Inputs are the FULL (F, ...) tensors present on every rank (or
    already this rank's shard with gather=False)."""
    from . import functional as WF
    frames, nl = layers.shape[:2]
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    l, p, o = shard_frames([layers, src_pts, occ], frames, rank, world, layers=nl)
    rgb = WF.warp_composite(l, p, o, inverse_kernel, basis_t, delta=delta)
    return all_gather_frames(rgb, frames) if gather else rgb
