"""CPU, world_size 2 over gloo: the sharding / all-gather logic of the multi-GPU path
(waldo_amd/dist.py).  The per-frame compute is a stand-in here (no GPU in this container); the
GPU kernels themselves are covered by the -m gpu suites."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from waldo_amd.dist import all_gather_frames, init_distributed, shard_frames, shard_range


def test_shard_range_covers_everything():
    for n in (0, 1, 5, 8, 14, 112, 113):
        for world in (1, 2, 3, 4, 8):
            blocks = [shard_range(n, r, world) for r in range(world)]
            flat = [i for s, e in blocks for i in range(s, e)]
            assert flat == list(range(n)), (n, world, blocks)
            per = (n + world - 1) // world
            assert all(e - s <= per for s, e in blocks)


def test_shard_frames_slices_per_frame_and_per_layer_tensors():
    f, nl = 5, 3
    layers = torch.arange(f * nl * 2).view(f, nl, 2).float()
    pts = torch.arange(f * nl * 4).view(f * nl, 2, 2).float()
    occ = torch.arange(f * nl * nl).view(f, nl, nl).float()
    a, b, c = shard_frames([layers, pts, occ], f, 1, 2, layers=nl)
    assert torch.equal(a, layers[3:5]) and torch.equal(b, pts[9:15]) and torch.equal(c, occ[3:5])
    with pytest.raises(ValueError):
        shard_frames([torch.zeros(7, 1)], f, 0, 2, layers=nl)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, frames, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r, lr, w = init_distributed(backend="gloo")
    assert (r, w) == (rank, world)
    # stand-in for the composite: frame f -> constant image of value f (+ channel index)
    full = torch.arange(frames).float().view(frames, 1, 1, 1) + torch.arange(3).float().view(1, 3, 1, 1)
    full = full.expand(frames, 3, 4, 6).contiguous()
    (mine,) = shard_frames([full], frames, rank, world)
    out = all_gather_frames(mine.clone(), frames)
    ok = torch.equal(out, full)
    # the overlapped form bench.py's inference steps use: two gathers in flight one after the other, each waited
    # for a step late; same frames
    from waldo_amd.dist import all_gather_frames_async
    first = all_gather_frames_async(mine.clone(), frames)
    second = all_gather_frames_async((mine + 100.0).clone(), frames)
    ok = ok and torch.equal(first.wait(), full) and torch.equal(second.wait(), full + 100.0)
    ok = ok and torch.equal(first.wait(), full)  # waiting twice is harmless
    # the barrier + max-over-ranks timing reduction bench.py uses
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    q.put((rank, ok, mine.shape[0], t.item()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("frames", [4, 5, 1])
def test_all_gather_frames_world2_gloo(frames):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, frames, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _, _ in res), res
    assert sum(n for _, _, n, _ in res) == frames
    assert all(tmax == 2.0 for *_, tmax in res)


def test_bench_spawns_one_process_per_rank_without_a_launcher():
    """`python bench.py --gpus 2` with no launcher around it (how the driver calls `--gpus 1`): bench.py starts
    `python -m torch.distributed.run` as a child.  On this CPU-only box a rank then stops with "rank R of 2 needs a
    GPU" (the HIP path has no CPU fallback).  The launcher ends the other rank as soon as the first one has failed, so
    only ONE such line is certain -- but that line carries the world size the launcher set, which is what shows that
    two ranks were asked for, and the parent hands the launcher's failure on instead of exiting 2 on a world-size
    mismatch as it did before."""
    import subprocess
    import sys
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU box: tests/test_gpu_dist.py::test_bench_starts_its_own_ranks runs the real thing")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300, cwd=root)
    assert r.returncode not in (0, 2), (r.returncode, r.stderr[-500:])
    import re
    ranks = re.findall(r"rank (\d+) of 2 needs a GPU", r.stderr)
    assert len(ranks) >= 1 and set(ranks) <= {"0", "1"}, r.stderr[-1500:]


def test_cpu_budget_is_within_the_affinity_mask():
    import bench
    n = bench.cpu_budget()
    assert 1 <= n <= len(os.sched_getaffinity(0))


def test_unit_segments_cover_a_block_in_order():
    from waldo_amd.tools.demo import rec_unit_order, unit_segments
    for per_clip in (1, 5, 10, 14):
        for order in (None, rec_unit_order(per_clip, 4)):
            dealt = list(range(per_clip)) if order is None else order
            for u0 in range(0, 3 * per_clip + 1):
                for u1 in range(u0, 3 * per_clip + 1):
                    units = []
                    for b0, b1, frames in unit_segments(u0, u1, per_clip, order):
                        assert frames == sorted(frames) and len(set(frames)) == len(frames) > 0 and b0 < b1
                        assert set(frames) <= set(range(per_clip))
                        assert b1 - b0 == 1 or frames == list(range(per_clip))  # several clips only as whole clips
                        units += [b * per_clip + f for b in range(b0, b1) for f in frames]
                    want = [(u // per_clip) * per_clip + dealt[u % per_clip] for u in range(u0, u1)]
                    assert sorted(units) == sorted(want), (per_clip, u0, u1)
                    if order is None:
                        assert units == want
    assert unit_segments(7, 23, 10) == [(0, 1, [7, 8, 9]), (1, 2, list(range(10))), (2, 3, [0, 1, 2])]


def test_reconstruction_units_are_dealt_so_that_ranks_invert_the_same_number_of_new_grids():
    """dealt in frame order, a rank whose block of a clip held no context frame inverted 11 frames'
    grids, its neighbour 7.  `rec_unit_order` spreads the context frames: over the ranks that share a clip the number
    of frames beyond the context differs by at most one, and every rank's units still cover the job exactly once."""
    from waldo_amd.tools.demo import local_unit_ids, rec_unit_order
    assert rec_unit_order(14, 4) == [13, 12, 11, 0, 10, 9, 1, 8, 7, 6, 2, 5, 4, 3]
    assert sorted(rec_unit_order(9, 4)) == list(range(9)) and sorted(rec_unit_order(3, 4)) == [0, 1, 2]
    for b, t, ctx, world in ((4, 14, 4, 8), (8, 9, 4, 8), (4, 14, 4, 4), (1, 14, 4, 2), (2, 14, 4, 3), (1, 9, 4, 5)):
        ids = [local_unit_ids("rec", b, t, ctx, r, world) for r in range(world)]
        assert sorted(i for blk in ids for i in blk) == list(range(b * t))
        new = [len({(i // t, i % t) for i in blk if i % t >= ctx}) for blk in ids if blk]
        assert max(new) - min(new) <= (1 if (b * t) % world == 0 else t - ctx), (b, t, world, new)
        pred = [local_unit_ids("pred", b, t, ctx, r, world) for r in range(world)]
        assert [i for blk in pred for i in blk] == list(range(b * (t - ctx)))  # the prediction stays in frame order
    c5 = [local_unit_ids("rec", 4, 14, 4, r, 8) for r in range(8)]
    assert all(len(blk) == 7 and sum(1 for i in blk if i % 14 >= 4) == 5 for blk in c5)
    # ... and the rank that PREDICTS a clip's early frames reconstructs its late ones: the same distances from the context
    for r in range(8):
        far = sorted([i % 14 for i in c5[r] if i % 14 >= 4] + [4 + i % 10 for i in local_unit_ids("pred", 4, 14, 4, r, 8)])
        assert far == list(range(4, 14)), (r, far)


def _gather_predict_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist
    from waldo_amd.dist import init_distributed, shard_range
    from waldo_amd.tools.demo import gather_predict, local_unit_ids
    init_distributed(backend="gloo")
    b, t, ctx, hd, wd = 2, 7, 4, 3, 5
    tp = t - ctx
    real_vid = torch.arange(b * t * 3 * hd * wd, dtype=torch.float32).view(b, t, 3, hd, wd)
    # what predict() would return, and this rank's unit blocks of it as predict_sharded lays them out
    full = {"inp_pred_vid": torch.cat([real_vid[:, :ctx], 1000 + torch.rand(b, tp, 3, hd, wd, generator=torch.Generator().manual_seed(1))], 1),
            "pred_disocc": torch.rand(b, tp, 1, hd, wd, generator=torch.Generator().manual_seed(2)),
            "pred_flow": torch.rand(b, ctx, tp, 2, hd, wd, generator=torch.Generator().manual_seed(3)),
            "rec_vid": torch.rand(b, t, 3, hd, wd, generator=torch.Generator().manual_seed(4))}
    u0, u1 = shard_range(b * tp, rank, world)
    rec_ids = torch.tensor(local_unit_ids("rec", b, t, ctx, rank, world))  # (the reconstruction's dealing order)
    local = {"inp_pred_vid": full["inp_pred_vid"][:, ctx:].reshape(b * tp, 3, hd, wd)[u0:u1],
             "pred_disocc": full["pred_disocc"].reshape(b * tp, 1, hd, wd)[u0:u1],
             "pred_flow": full["pred_flow"].permute(0, 2, 1, 3, 4, 5).reshape(b * tp, ctx * 2, hd, wd)[u0:u1],
             "rec_vid": full["rec_vid"].reshape(b * t, 3, hd, wd)[rec_ids]}
    got = gather_predict(local, real_vid, ctx)
    q.put((rank, all(torch.equal(got[k], full[k]) for k in full), sorted(got)))
    dist.barrier()
    dist.destroy_process_group()


def test_gather_predict_world2_gloo():
    """The assembly of a split predict(): unit blocks (ragged: 6 predicted units over 2 ranks is even, 14 reconstructed
    ones are not a multiple of a clip) all-gathered and shaped as predict() returns them, pred_flow's (Tc, Tp) order
    included."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gather_predict_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res), res
