"""BASELINE configs C4 / C5 as the reference runs them: the hot-path part of ``Synthesizer.predict``
(models/synthesizer.py:434-472) over whole clips at the sizes of the WIF recipes
(scripts/cityscapes/train_wif.sh: 128 x 256 layers, 512 x 1024 frames, 14-frame clips, 4 context frames;
scripts/kitti/train_wif.sh: 128 x 416 / 256 x 832, latent grid 8 x 26).

    producers (pose affine, decoder tail, compute_occ)  ->  Warper.forward (TPS grids + grid inversion)
    ->  decode_output (grid_to_flow_ctx + input_to_output)  ->  WIF fusion,   reconstruction AND prediction

The call order and every tensor between the steps are ``tools.demo.predict``'s (checked end to end
against the CPU restatement at the demo size, tests/test_demo.py); what is synthetic is what the
networks outside the path would hand over (``tools.demo.synthetic_network_outputs``) and the clip:
seeded frames and a piecewise-constant layout.  ``bench.py --config C5 --pipeline`` times it.
"""
import torch

from ..nets.lvd import Warper
from ..nets.wif import WIF
from . import demo

# name: (dim, aspect ratio, load_dim, latent grid, objects, frames per clip, context frames)
RECIPES = {
    "C4": (128, 3.25, 256, [8, 26], 7, 9, 4),    # KITTI 256 x 832, 8 layers, 9-frame clips
    "C5": (128, 2.0, 512, [8, 16], 11, 14, 4),   # Cityscapes 512 x 1024, 12 layers, 14-frame clips
}


def recipe_opt(name, **over):
    dim, ar, load_dim, latent, no, _, _ = RECIPES[name]
    return demo.demo_opt(dim=dim, aspect_ratio=ar, num_obj=no, num_lyt=20, load_dim=load_dim, latent_shape=latent,
                         obj_shape=[4, 4], patch_size=16, scale_factor=1, min_cls=0.1, use_lyt_opacity=True,
                         pad_obj_alpha=3, **over)


def synthetic_clip(opt, clips, frames, seed, device):
    """Frames in [-1, 1] and a layout of +-5 logits (data/base_dataset.py:173-183) that is constant on
    32 x 32-pixel blocks, at the full resolution."""
    g = torch.Generator(device=device).manual_seed(seed)
    hd = opt.load_dim if opt.load_dim > 0 else opt.dim
    wd = int(hd * opt.aspect_ratio)
    vid = torch.rand(clips, frames, 3, hd, wd, generator=g, device=device) * 2 - 1
    cls = torch.randint(0, opt.num_lyt, (clips, frames, (hd + 31) // 32, (wd + 31) // 32), generator=g, device=device)
    cls = cls.repeat_interleave(32, dim=2).repeat_interleave(32, dim=3)[:, :, :hd, :wd]
    lyt = torch.full((clips, frames, opt.num_lyt, hd, wd), -5.0, device=device)
    lyt.scatter_(2, cls.unsqueeze(2), 5.0)
    return vid, lyt


class Pipeline:
    """One rank's share of a C4 / C5 run: ``clips`` clips resident on ``device``; ``__call__`` runs
    ``predict`` on them and returns its dict (``inp_pred_vid`` (B, T, 3, Hd, Wd) is the product).

    ``shard=(rank, world)``: the ``clips`` clips are ONE job split over ``world`` ranks by (b, t) output units
    (``demo.predict_sharded``; every rank builds the same job from the same seed); ``__call__`` then returns this
    rank's unit blocks, ``gather`` the assembled dict on every rank."""

    def __init__(self, name, clips, device, seed=0, motion="calibrated", shard=None):
        self.name = name
        self.shard = shard
        self.motion = motion  # background motion of the stand-ins: demo.BG_MOTION
        self.opt = recipe_opt(name)
        self.frames, self.ctx_len = RECIPES[name][5], RECIPES[name][6]
        self.clips = clips
        self.warper = Warper(self.opt).to(device)
        self.wif = WIF(self.opt, unet=demo.UniformFusionUNet()).to(device)
        self.net = demo.synthetic_network_outputs(self.opt, clips, self.frames, self.ctx_len, seed=seed, device=device,
                                                  motion=motion)
        self.vid, self.lyt = synthetic_clip(self.opt, clips, self.frames, seed, device)

    def __call__(self, phases=("rec", "pred")):
        return self.run(self.vid, self.lyt, phases)

    def run(self, vid, lyt, phases=("rec", "pred")):
        """predict() (or this rank's share of it) on another clip of the same shape."""
        if self.shard is not None:
            return demo.predict_sharded(self.opt, self.warper, self.wif, vid, lyt, self.net, self.ctx_len,
                                        *self.shard, phases=phases)
        return demo.predict(self.opt, self.warper, self.wif, vid, lyt, self.net, self.ctx_len)

    def graphed(self, key="inp_pred_vid"):
        """The whole step as ONE HIP graph (waldo_amd.graphs.GraphedCall): a rank's share of a split job is a
        couple of hundred launches of a few microseconds each, queued slower than they run.  Returns a callable
        (vid, lyt) -> the step's ``key`` tensor, a static buffer the next replay overwrites."""
        from ..graphs import GraphedCall
        return GraphedCall(lambda vid, lyt: self.run(vid, lyt)[key], self.vid, self.lyt)

    def gather(self, local, keys=None, group=None):
        return demo.gather_predict(local, self.vid, self.ctx_len, keys=keys, group=group)

    def local_units(self, phase="pred"):
        """[start, stop) of this rank's (b, t) units of a phase ("rec": B * T, "pred": B * (T - Tc)) in the phase's
        dealing order (demo.phase_order: frame order for "pred")."""
        from ..dist import shard_range
        per_clip = self.frames if phase == "rec" else self.frames - self.ctx_len
        rank, world = self.shard if self.shard is not None else (0, 1)
        return shard_range(self.clips * per_clip, rank, world)

    def hd_algorithmic_bytes(self):
        """Bytes the full-resolution entry points have to move per ``predict`` (each input read once,
        each output written once; fp32), by C-ABI name -- the denominators of the bench line's table.  A context
        frame (its a01 planes for the flow pass, its C channels for the frame warp) is counted ONCE per (b, t),
        however many predicted frames gather from it.  Without autograd ``decode_output`` runs the ``_raw``
        entry points (the context alphas are composited straight into raw_output's slots): their counts have
        neither the L planes per (b, tc, tp) the frame warp would read nor the L it would copy, and one score
        plane written and read instead."""
        o = self.opt
        b, t, tc = self.clips, self.frames, self.ctx_len
        nl, ncls, c = o.num_obj + 1, o.num_lyt, 3 + o.num_lyt
        hw = o.dim * int(o.dim * o.aspect_ratio)
        hwd = o.load_dim * int(o.load_dim * o.aspect_ratio)
        ghost = 0 if o.allow_ghost else nl - 1
        out = {}
        merged = demo.MERGE_DECODES and not o.no_future and not getattr(o, "include_self", False)
        for i, tp in enumerate((t, t - tc)):  # reconstruction over all T frames, prediction over the T - Tc future ones
            m = b * tc * tp
            lr = m * (nl * 2 + ghost) * hw                      # per-layer flows and object masks, low resolution
            fcw_out = m * (2 + nl + 1 + 1) * hwd                # flow, alpha_ctx, disocc, layer maximum
            # (both decodes' units go through ONE launch of every full-resolution pass -- demo.decode_units -- so the
            # context's alphas and frames are read once per step, not once per decode)
            ctx_once = 0 if (merged and i == 1) else 1
            add = {
                # (a01 alone is written: without an inpainter -- opt.use_inpainter, off in the stand-in pipeline --
                # nothing reads `alpha` = 2 a01 - 1 and the pass is asked not to write it; the pass itself runs ONCE per
                # step, its result shared by the two decodes)
                "waldo_flow_ctx_alpha_fwd": (4 * (b * tc * nl * hw + b * tc * ncls * hwd
                                                  + (2 if getattr(o, "use_inpainter", False) else 1) * b * tc * nl * hwd)
                                             if i == 0 else 0),
                "waldo_flow_ctx_warp_fwd": 4 * (lr + ctx_once * b * tc * nl * hwd + fcw_out),
                "waldo_flow_ctx_warp_raw_fwd": 4 * (lr + ctx_once * b * tc * nl * hwd + fcw_out + m * hwd),  # + score
                "waldo_frame_warp_fuse_fwd": 4 * (ctx_once * b * tc * c * hwd + m * (2 + nl) * hwd + b * tp * (c + 1) * hwd
                                                  + m * (c + nl) * hwd),
                "waldo_frame_warp_fuse_raw_fwd": 4 * (ctx_once * b * tc * c * hwd + m * (2 + 1) * hwd + b * tp * (c + 1) * hwd
                                                      + m * c * hwd),
                # the fusion reads channels 0-2 and 4 of the raw frames and the UNet's four outputs (wif.py:49-54)
                "waldo_wif_fuse_fwd": 4 * (m * 4 * hwd + m * 4 * hwd + b * tp * 3 * hwd),
            }
            for k, v in add.items():
                out[k] = out.get(k, 0) + v
        return out
