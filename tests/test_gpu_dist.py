"""GPU: the multi-rank inference path actually executed -- two fresh child processes (ranks) share
the one GPU of the test box, rendezvous over gloo on 127.0.0.1, each composites its contiguous block
of frames with the HIP kernels (waldo_amd.dist.sharded_warp_composite) and all-gathers the frames;
the result must equal the single-rank result bit for bit, ragged frame counts included.  (On the
8-GPU node the same code runs with backend "nccl" = RCCL, one rank per GPU: bench.py --mode infer.)"""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

from oracle import wif_oracle as O

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _rank(rank, world, port, frames, nl, h, w, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import torch.distributed as dist
    import waldo_amd
    from waldo_amd.dist import init_distributed, shard_range, sharded_warp_composite
    init_distributed(backend="gloo")
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    layers, pts, occ, _, _ = O.make_synthetic(frames, nl, h, w, seed=7)
    tps = waldo_amd.TPSWarp(h, w, O.get_grid(4, 4).view(-1, 2)).to(dev)
    with torch.no_grad():
        out = sharded_warp_composite(layers.to(dev), pts.to(dev), occ.to(dev), tps.inverse_kernel, tps.basis_t)
    torch.cuda.synchronize()
    maps = open("/proc/self/maps").read()
    q.put((rank, shard_range(frames, rank, world), out.cpu().numpy(), "libwaldo_hip.so" in maps))  # bytes, not shm handles
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("frames", [6, 5, 1])
def test_sharded_warp_composite_two_ranks_one_gpu(dev, frames):
    import waldo_amd
    from waldo_amd import functional as WF
    nl, h, w = 4, 32, 64
    layers, pts, occ, _, _ = O.make_synthetic(frames, nl, h, w, seed=7)
    tps = waldo_amd.TPSWarp(h, w, O.get_grid(4, 4).view(-1, 2)).to(dev)
    with torch.no_grad():
        single = WF.warp_composite(layers.to(dev), pts.to(dev), occ.to(dev), tps.inverse_kernel, tps.basis_t).cpu()
    ctx = mp.get_context("spawn")  # fresh interpreters: nothing of this process's GPU state is inherited
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank, args=(r, 2, port, frames, nl, h, w, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=300) for _ in procs), key=lambda x: x[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    blocks = [r[1] for r in res]
    assert [i for s, e in blocks for i in range(s, e)] == list(range(frames)), blocks
    for rank, _, out, native in res:
        assert native, "the rank did not load the HIP library"
        out = torch.from_numpy(out)
        assert out.shape == single.shape
        assert torch.equal(out, single), f"rank {rank}: gathered frames differ from the single-rank result"
