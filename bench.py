#!/usr/bin/env python3
"""Headline benchmark: warped+composited frames/sec at 256x512, 8 layers, fwd+bwd.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[2], SURVEY.md 8(d) "C3"): per rank B clips x T=14 frames,
L=8 layers of 4x256x512 (RGB+alpha in [-1,1]), 16 TPS control points per layer,
occ = compute_occ(randn).  One step = one pass of the hot path over that batch:
  fwd  rgb = warp_composite(layers, pts, occ)      (TPS grid -> bilinear warp -> reduce_comp)
  bwd  rgb.square().mean().backward()              grads on layers and control points
Inputs are resident in HBM when the timed region starts.  Frames are independent, so ranks shard
them with no data-path collective ("scaling": "weak", per-GPU work fixed).

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline      dominant kernel, algorithmic bytes per launch / mean launch duration measured live
                with events on the launch stream, against the 8 TB/s HBM3E peak
  cpu_baseline  the oracle (a PyTorch-CPU restatement of the reference path, oracle/wif_oracle.py)
                timed on this host's cores on a bounded sample of the same workload
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def synth(frames, nl, h, w, device, seed):
    """Same recipe as oracle.wif_oracle.make_synthetic, generated on the device."""
    from waldo_amd.tools.utils import get_grid
    g = torch.Generator(device=device).manual_seed(seed)
    layers = torch.rand(frames, nl, 4, h, w, generator=g, device=device) * 2 - 1
    ctrl = get_grid(4, 4).view(1, 16, 2).to(device)
    sigma = float(os.environ.get("WALDO_BENCH_SIGMA", "0.05"))  # dev knob; the benchmark is 0.05
    pts = ctrl + sigma * torch.randn(frames * nl, 16, 2, generator=g, device=device)
    score = torch.randn(frames, nl - 1, generator=g, device=device)
    s = torch.exp(-score ** 2) + 1e-6
    occ = torch.zeros(frames, nl, nl, device=device)
    occ[:, 1:, 1:] = s[:, :, None] / (s[:, :, None] + s[:, None, :]) - 0.5 * torch.eye(nl - 1, device=device)
    occ[:, 1:, 0] = 1.0
    return layers, pts, occ


class _SquareMean(torch.autograd.Function):
    """The benchmark's loss, out.square().mean(), with its gradient 2 * out / N written as
    ONE elementwise kernel: autograd's own chain for it (expand, copy, two multiplies) is four passes
    over the 176 MB output, ~0.2 ms per step of pure loss scaffolding at the headline shape."""

    @staticmethod
    def forward(ctx, out):
        ctx.save_for_backward(out)
        return out.square().mean()

    @staticmethod
    def backward(ctx, g):
        (out,) = ctx.saved_tensors
        return out * (g * (2.0 / out.numel()))


def copy_bandwidth(device, nbytes=1 << 30, reps=10):
    """Measured device-to-device copy rate of this box (read + write bytes per second, GB/s): the
    second denominator SURVEY 8(d) asks for next to the 8 TB/s spec figure."""
    src = torch.empty(nbytes, dtype=torch.uint8, device=device)
    dst = torch.empty_like(src)
    dst.copy_(src)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        dst.copy_(src)
    e1.record()
    torch.cuda.synchronize()
    return 2.0 * nbytes * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9


def cpu_baseline(nl, h, w, frames, reps):
    """Oracle (kind "port") on the host cores: fwd+bwd frames/s on `frames` frames."""
    from oracle import wif_oracle as O
    ncpu = os.cpu_count() or 1
    layers, pts, occ, inv, rep = O.make_synthetic(frames, nl, h, w, seed=0)
    layers.requires_grad_()
    pts.requires_grad_()

    def once():
        layers.grad = pts.grad = None
        t0 = time.perf_counter()
        rgb, _ = O.warp_composite(layers, pts, occ, inv, rep)
        rgb.square().mean().backward()
        return time.perf_counter() - t0

    # PyTorch-CPU does not scale to every core of a big host on these shapes: probe a few
    # thread counts once and keep the fastest (this is the reference's best case)
    cands = sorted({c for c in (8, 16, 32, 64, ncpu) if c <= ncpu})
    probe = {}
    for c in cands:
        torch.set_num_threads(c)
        once()
        probe[c] = once()
        if probe[c] > 4 * min(probe.values()):
            break
    cores = min(probe, key=probe.get)
    torch.set_num_threads(cores)
    times = sorted(once() for _ in range(reps))
    med = times[len(times) // 2]
    return {"value": round(frames / med, 3), "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"{frames} frames of the same workload ({nl}x4x{h}x{w}, fwd+bwd), "
                      f"median of {reps} after warm-up, torch {torch.__version__} CPU, "
                      f"{cores} threads (fastest of {sorted(probe)} probed on {ncpu} logical CPUs)"}


def measured_traffic(entry_point, frames, nl, h, w):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/traffic.json, written
    by tools_dev/traffic.py from FETCH_SIZE / WRITE_SIZE with the gfx950 corrections of
    MI355X_MICROARCH.md), if one matches this workload; else null."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as fh:
            t = json.load(fh)
        if t.get("frames") == frames and t.get("layers") == nl and t.get("height") == h and t.get("width") == w:
            return t["bytes_per_launch"].get(entry_point)
    except (OSError, ValueError, KeyError):
        pass
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--clips", type=int, default=8, help="clips per GPU (T=14 frames each)")
    ap.add_argument("--height", type=int, default=256)
    ap.add_argument("--width", type=int, default=512)
    ap.add_argument("--layers", type=int, default=8)
    ap.add_argument("--frames-per-clip", type=int, default=14)
    ap.add_argument("--mode", choices=["train", "infer"], default="train",
                    help="train: fwd+bwd (the headline metric); infer: fwd only + RCCL all-gather "
                         "of the composited frames (not the headline; vs_baseline/roofline differ)")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl (= RCCL on ROCm) is the real thing; gloo only to smoke-test the "
                         "multi-rank code path on a box with fewer GPUs than ranks")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-frames", type=int, default=28)
    ap.add_argument("--cpu-reps", type=int, default=5)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torchrun",
                  file=sys.stderr)
        if world == 1 and args.gpus > 1:
            sys.exit(2)
    if not torch.cuda.is_available():
        print("bench.py: needs a GPU (the HIP path has no CPU fallback)", file=sys.stderr)
        sys.exit(1)
    if args.dist_backend == "gloo":
        local_rank = local_rank % torch.cuda.device_count()  # ranks may share a GPU in the smoke test
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        from waldo_amd.dist import init_distributed
        init_distributed(backend=args.dist_backend)

    import waldo_amd
    from waldo_amd import _lib, functional as WF
    from waldo_amd.tools.utils import get_grid

    nl, h, w = args.layers, args.height, args.width
    frames = args.clips * args.frames_per_clip
    tps = waldo_amd.TPSWarp(h, w, get_grid(4, 4).view(-1, 2)).to(device)
    layers, pts, occ = synth(frames, nl, h, w, device, seed=rank)
    layers.requires_grad_()
    pts.requires_grad_()

    from waldo_amd.dist import all_gather_frames

    def step():
        if args.mode == "infer":
            with torch.no_grad():
                rgb = WF.warp_composite(layers, pts, occ, tps.inverse_kernel, tps.basis_t)
                all_gather_frames(rgb, frames * world)
            return
        layers.grad = None
        pts.grad = None
        rgb = WF.warp_composite(layers, pts, occ, tps.inverse_kernel, tps.basis_t)
        _SquareMean.apply(rgb).backward()

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    with _lib.KernelTimer() as kt:
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        fence()
        elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], device=device if args.dist_backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        total_frames = frames * world
        value = total_frames / (elapsed / args.steps)
        ks = kt.summary()
        hw = h * w
        alg = {"waldo_warp_composite_fwd": (16 * nl + 12) * hw * frames,
               "waldo_warp_composite_bwd": (32 * nl + 12) * hw * frames}
        if args.mode == "infer":
            alg.pop("waldo_warp_composite_bwd")
        dom = max(alg, key=lambda k: ks[k][1])
        kern = {}
        for k in alg:
            gbs = alg[k] / (ks[k][1] * 1e-3) / 1e9
            kern[k] = {"launches": ks[k][0], "ms": round(ks[k][1], 4), "alg_bytes": alg[k],
                       "GBps": round(gbs, 1), "frac": round(gbs / HBM_PEAK_GBS, 4)}
        roof = {"bound": "hbm", "kernel": dom, "achieved": kern[dom]["GBps"], "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": kern[dom]["frac"], "traffic": measured_traffic(dom, frames, nl, h, w),
                "ms_per_launch": kern[dom]["ms"], "alg_bytes_per_launch": alg[dom],
                "kernels": kern}
        copy_gbs = copy_bandwidth(device)
        roof["copy_GBps"] = round(copy_gbs, 1)  # measured D2D copy rate of this box (read + write)
        roof["frac_of_copy"] = round(kern[dom]["GBps"] / copy_gbs, 4)
        out = {
            "metric": "warped+composited frames/sec at 256x512, 8 layers; fwd+bwd" if args.mode == "train"
            else "warped+composited frames/sec, fwd only + all-gather (not the headline metric)",
            "value": round(value, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"{'C3' if (nl, h, w, args.frames_per_clip) == (8, 256, 512, 14) else 'custom'}: "
                                   f"{args.clips} clips x {args.frames_per_clip} frames per GPU, "
                                   f"{nl} layers x 4x{h}x{w}, 16 TPS control points, " + ("fwd+bwd" if args.mode == "train" else "fwd"),
                       "frames_per_gpu": frames, "layers": nl, "height": h, "width": w,
                       "parallelism": f"frames sharded x{world}, no data-path collective"},
            "roofline": roof,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(nl, h, w, args.cpu_frames, args.cpu_reps)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
