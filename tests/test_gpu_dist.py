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


def _rccl_rank(port, frames, q):
    os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    import waldo_amd
    from waldo_amd.dist import all_gather_frames, init_distributed
    init_distributed(backend="nccl", single_rank_group=True)  # "nccl" IS RCCL on ROCm
    dev = torch.device("cuda:0")
    nl, h, w = 4, 32, 64
    layers, pts, occ, _, _ = O.make_synthetic(frames, nl, h, w, seed=11)
    tps = waldo_amd.TPSWarp(h, w, O.get_grid(4, 4).view(-1, 2)).to(dev)
    from waldo_amd import functional as WF
    with torch.no_grad():
        rgb = WF.warp_composite(layers.to(dev), pts.to(dev), occ.to(dev), tps.inverse_kernel, tps.basis_t)
        out = all_gather_frames(rgb, frames, collective_for_one=True)  # one all_gather_into_tensor on device buffers
    torch.cuda.synchronize()
    maps = open("/proc/self/maps").read()
    q.put((dist.get_backend(), bool(out.is_cuda), out.data_ptr() != rgb.data_ptr(), torch.equal(out, rgb),
           "librccl" in maps, "libwaldo_hip.so" in maps))
    dist.barrier()
    dist.destroy_process_group()


def test_rccl_all_gather_runs_on_one_gpu(dev):
    """The first RCCL call of this code base must not happen on the driver's 8-GPU node: a freshly
    spawned child creates a world-size-1 "nccl" (= RCCL) group, composites frames with the HIP kernels
    and sends them through the same ``all_gather_into_tensor`` the N-rank inference path ends with
    (reference contract: tools/engine.py:35,86-92)."""
    ctx = mp.get_context("spawn")  # spawn, never re-exec or fork a process that touched the GPU
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_rank, args=(_free_port(), 5, q))
    p.start()
    backend, on_gpu, fresh_buffer, equal, rccl_mapped, hip_mapped = q.get(timeout=600)
    p.join(timeout=120)
    assert p.exitcode == 0
    assert backend == "nccl" and on_gpu and fresh_buffer, (backend, on_gpu, fresh_buffer)
    assert equal, "frames changed on their way through the collective"
    assert rccl_mapped, "librccl is not mapped: the collective did not go through RCCL"
    assert hip_mapped


@pytest.mark.parametrize("extra", [["--config", "C3", "--clips", "1"], ["--config", "C4", "--pipeline", "--clips", "1"]])
def test_bench_starts_its_own_ranks(extra):
    """`python bench.py --gpus 2` with NO launcher around it (as the driver calls `--gpus 1`): bench.py starts
    `python -m torch.distributed.run` itself as a child before anything touches the GPU, the two ranks share this
    box's one GPU (gloo rendezvous on 127.0.0.1, as the test above), and ONE JSON line with n_gpus = 2 comes back.
    Run from a fresh child process; this process's GPU state is not inherited."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--steps", "2",
           "--warmup", "1", "--no-cpu-baseline"] + extra
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["value"] > 0 and out["scaling"] == "weak"
    per_rank = out["config"]["frames_per_gpu"]
    assert abs(out["value"] - 2 * per_rank / (out["ms_per_step"] * 1e-3)) <= 1e-3 * out["value"]  # whole-job aggregate


def _predict_rank(rank, world, port, name, clips, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import torch.distributed as dist
    from waldo_amd.dist import init_distributed
    from waldo_amd.tools.pipeline import Pipeline
    init_distributed(backend="gloo")
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    pipe = Pipeline(name, clips, dev, seed=6, shard=(rank, world))
    local = pipe()
    got = pipe.gather(local, keys=("inp_pred_vid", "pred_flow", "pred_disocc"))
    torch.cuda.synchronize()
    q.put((rank, pipe.local_units("pred"), {k: v.cpu().numpy() for k, v in got.items()}))
    dist.barrier()
    dist.destroy_process_group()


def test_predict_sharded_two_ranks_one_gpu(dev):
    """ONE KITTI-recipe clip (256 x 832, 9 frames, 4 contexts: 5 predicted frames, a ragged 3 + 2 split) decoded by
    two ranks that share this box's GPU: each runs `predict_sharded` on its block of (b, t) units and the ranks
    all-gather (gloo here, RCCL on the node); `inp_pred_vid`, `pred_flow` and `pred_disocc` are bit-equal to the
    single-rank `predict` on every rank.  Reference: tools/engine.py:31-35,63-64 (process group, data-parallel split),
    models/synthesizer.py:434-472 (the calls being split)."""
    from waldo_amd.tools.pipeline import Pipeline
    name, clips = "C4", 1
    single = Pipeline(name, clips, dev, seed=6)()
    want = {k: single[k].cpu() for k in ("inp_pred_vid", "pred_flow", "pred_disocc")}
    del single
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_predict_rank, args=(r, 2, port, name, clips, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=600) for _ in procs), key=lambda x: x[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert [r[1] for r in res] == [(0, 3), (3, 5)]
    for rank, _, got in res:
        for k, v in want.items():
            g = torch.from_numpy(got[k])
            assert g.shape == v.shape, (rank, k, g.shape, v.shape)
            assert torch.equal(g, v), f"rank {rank}: {k} differs from the single-rank predict"


def test_bench_strong_scaling_line():
    """`bench.py --config C5 --pipeline --gpus 2 --scaling strong`: one job (here one Cityscapes clip) split over two
    ranks by output frames; ONE JSON line, "scaling": "strong", value = the job's frames over the slowest rank's time."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--steps", "2",
           "--warmup", "1", "--config", "C5", "--pipeline", "--scaling", "strong", "--clips", "1"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and out["value"] > 0
    assert abs(out["value"] - 14 / (out["ms_per_step"] * 1e-3)) <= 1e-3 * out["value"]  # the job's 14 frames, once


def test_default_multi_gpu_line_carries_the_north_star_block():
    """`bench.py --gpus 2` with NO other workload flag -- the one multi-GPU command the driver's scaling run issues (here
    with `--dist-backend gloo`: the two ranks share this box's GPU) -- prints ONE line: the C3 headline loop (no data-path
    collective) and, behind it, the `north_star` object: the all-gather of composited frames at C4 and C5 (bytes, the
    collective alone, exposed time with the overlap off and on), the C5 pipeline as one job split over the ranks
    (ms_per_step, per-rank compute, speed-up over rank 0 running the whole job) and the cross-rank bit-equality check.
    Reference: tools/engine.py:31-35,63-64,86-92; SURVEY.md section 8(e)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--steps", "3",
           "--warmup", "1"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1500, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["metric"] == "warped+composited frames/sec at 256x512, 8 layers; fwd+bwd" and out["n_gpus"] == 2
    assert out["steps"] == 3 and out["warmup"] == 1 and len(out["ms_per_step_blocks"]) == 5 and out["settle_ms"] >= 300
    ns = out["north_star"]
    assert ns["rccl_ranks"] == 2 and ns["backend"] == "gloo"
    for key in ("all_gather_C4", "all_gather_C5"):
        g = ns[key]
        assert isinstance(g, dict), g
        assert g["bytes_received_per_rank"] == g["bytes_sent_per_rank"] > 0
        assert g["all_gather_alone_ms"] > 0 and g["step_ms_overlap_off"] > 0 and g["step_ms_overlap_on"] > 0
    st = ns["strong_split"]
    assert isinstance(st, dict), st
    assert len(st["per_rank_compute_ms"]) == 2 and st["ms_per_step"] > 0 and st["whole_job_on_rank0_ms"] > 0
    assert st["speedup_over_one_gpu"] > 0
    assert ns["gather_bit_equal"] is True, ns["gather_bit_equal"]
    assert "headline before the north_star block" in r.stderr


def test_bench_wif_line():
    """`bench.py --config WIF`: the reference's WIF training step (Synthesizer.inpaint's call order at the
    train_wif.sh recipe) as one JSON line with the per-entry-point table -- the unrestricted flow passes, the frame
    warp and BOTH directions of the fusion kernel in it."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--config", "WIF", "--steps", "2", "--warmup", "1"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["grads_finite"] is True and out["value"] > 0 and out["config"]["layers"] == 17
    table = out["pipeline"]["entry_points"]
    for name in ("waldo_flow_ctx_alpha_fwd", "waldo_flow_ctx_warp_raw_fwd", "waldo_frame_warp_fuse_raw_fwd",
                 "waldo_wif_fuse_fwd", "waldo_wif_fuse_bwd", "waldo_inverse_warp_fwd"):
        assert name in table, sorted(table)
    assert table["waldo_wif_fuse_bwd"]["frac"] > 0 and out["roofline"]["frac"] > 0


def test_default_line_measures_its_hbm_traffic_in_the_run():
    """The default one-GPU command (what the driver runs) collects ``roofline.traffic`` itself: two rocprofv3 --pmc child
    runs of the same command (bench.py::live_traffic), and the bytes they report per backward launch lie between the
    algorithmic bytes and three times them (the K1 -> K2 records are real traffic of this design: 2.2 x)."""
    import json
    import shutil
    import subprocess
    import sys
    if shutil.which("rocprofv3") is None and not os.path.exists("/opt/rocm/bin/rocprofv3"):
        pytest.skip("no rocprofv3 on this box")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    roof = out["roofline"]
    if "under a profiler" in roof["traffic_source"]:
        pytest.skip("the test run is itself profiled: bench.py took the committed passes, as it says")
    if "live passes failed" in roof["traffic_source"]:  # (a box that does not grant the counters: the line stands on
        pytest.skip("no counters on this box: " + roof["traffic_source"][-200:])  # the committed passes and says so)
    assert roof["traffic_source"].startswith("rocprofv3 --kernel-trace --pmc"), roof["traffic_source"]
    assert roof["alg_bytes_per_launch"] < roof["traffic"] < 3 * roof["alg_bytes_per_launch"]
    fwd = roof["kernels"]["waldo_warp_composite_fwd"]
    assert fwd["alg_bytes"] < fwd["moved_bytes"] < 2 * fwd["alg_bytes"]
