"""The warp-path part of one WIF training step as the reference runs it -- BASELINE config 3's real call order
(``Synthesizer.inpaint``, models/synthesizer.py:517-576, 631-633, at scripts/cityscapes/train_wif.sh: 8 clips over 4
GPUs = 2 clips per GPU, ``vid_len 5`` with ``ctx_len 4`` -> one predicted frame, 16 objects + background, 20 layout
classes, 128 x 256 layers, 512 x 1024 frames):

    under torch.no_grad():   decoder tail -> pose heads' affine -> estimate_alpha_grid_occ (synthesizer.py:532)
                             real_input = cat([real_vid, real_lyt])                        (synthesizer.py:535)
                             decode_output with ctx_ts = 0 .. 3, pred_ts = [4]             (synthesizer.py:568-573)
                             -- train_wif.sh does NOT pass --s_restrict_to_ctx: the UNRESTRICTED Warper.grid_to_flow
                             (lvd.py:602-705: alphas composited on all T frames), then input_to_output
    with autograd:           inp_output = net_ii(raw_output)  = WIF.forward                (synthesizer.py:576)
                             loss (the `sharp_vid` L1 term, synthesizer.py:587-590), backward (synthesizer.py:631)

The networks outside the path are stand-ins: seeded decoder logits / pose head outputs / occlusion scores / class
distributions, and a 1 x 1 convolution (40 -> 5 channels, the UNet's widths at ``ii_score`` + ``ii_ab``, wif.py:19-23)
in the UNet's place so that ``waldo_wif_fuse_bwd`` receives and produces real gradients.  ``bench.py --config WIF``
times it.
"""
import torch
import torch.nn as nn

from .. import functional as WF
from ..nets import WIF, Warper, decode_output, estimate_alpha_grid_occ, flp
from ..nets.lvd import decoder_tail
from . import demo
from .pipeline import synthetic_clip


def wif_opt(**over):
    """The option fields the path reads, at the Cityscapes WIF recipe (train_wif.sh:12-16, 25-37)."""
    return demo.demo_opt(dim=128, aspect_ratio=2.0, num_obj=16, num_lyt=20, load_dim=512, latent_shape=[8, 16],
                         obj_shape=[4, 4], patch_size=16, scale_factor=1, min_cls=0.1, use_lyt_opacity=True,
                         pad_obj_alpha=3, **over)


class WifStep:
    """``clips`` clips of 5 frames resident on ``device``; ``__call__`` runs the no-grad decode, ``WIF.forward``, the
    loss and its backward once and returns the loss (the stand-in network's ``.grad`` hold the gradients)."""

    frames, ctx_len = 5, 4

    def __init__(self, clips, device, seed=0, motion="calibrated"):
        self.opt = o = wif_opt()
        self.clips, self.device = clips, device
        self.warper = Warper(o).to(device)
        torch.manual_seed(seed)
        self.unet = nn.Conv2d(3 + o.num_lyt + o.num_obj + 1, 5, 1).to(device)
        self.wif = WIF(o, unet=self.unet).to(device)
        self.net = demo.synthetic_network_outputs(o, clips, self.frames, self.ctx_len, seed=seed, device=device,
                                                  motion=motion)
        self.vid, self.lyt = synthetic_clip(o, clips, self.frames, seed, device)
        t, tc = self.frames, self.ctx_len
        # synthesizer.py:568-570: the context frames' indices, expanded over the predicted frames; the predicted frame
        self.ctx_ts = torch.arange(tc, device=device, dtype=torch.int64).view(1, -1, 1).expand(clips, -1, t - tc).contiguous()
        self.pred_ts = torch.arange(tc, t, device=device, dtype=torch.int64)
        self.raw_output = None

    def decode(self):
        """The no-grad half: what ``inpaint`` computes before ``net_ii`` (synthesizer.py:517-574)."""
        o, b, t = self.opt, self.clips, self.frames
        no = o.num_obj
        lo, lb = o.obj_shape[0] * o.obj_shape[1], o.latent_shape[0] * o.latent_shape[1]
        dev = self.device
        net = self.net
        with torch.no_grad():
            buf = demo.pose_buffers(o, dev)
            bg_alpha = demo._cached("bg_alpha", o, dev, lambda: torch.ones(1, 1, o.dim, int(o.dim * o.aspect_ratio), device=dev))
            obj_pose = flp.obj_pose_to_points(net["pred_obj_pose"], buf["tgt_pts_obj"], buf["mul_obj"], buf["bias_obj"])
            bg_pose = flp.bg_pose_to_points(net["pred_bg_pose"], buf["tgt_pts_bg"], buf["bias_bg"])
            obj_alpha = decoder_tail(net["raw"], init_bias=0.0, scale_factor=o.scale_factor)
            obj_alpha = obj_alpha.view(b, no, 1, *obj_alpha.shape[-2:])
            occ, obj_alpha, bga, grid = estimate_alpha_grid_occ(self.warper, obj_alpha, bg_alpha,
                                                                obj_pose.view(b, t, no, lo, 2), bg_pose.view(b, t, 1, lb, 2),
                                                                net["occ_score"], obj_alpha_mask=demo.obj_alpha_mask(o, dev))
            real_input = torch.cat([self.vid, self.lyt], dim=2)  # (all T frames: the unrestricted path composites on them)
            out = decode_output(self.warper, real_input, grid, occ, obj_alpha, bga, net["cls"], self.ctx_ts,
                                self.pred_ts, restrict_to_ctx=False)
        return out  # (output, flow, alpha_unflt, alpha, raw_alpha, raw_output, alpha_ctx)

    def __call__(self):
        for p in self.unet.parameters():
            p.grad = None
        rec_output, _, _, _, _, raw_output, _ = self.decode()
        self.raw_output = raw_output
        inp_output = self.wif(raw_output)                                   # synthesizer.py:576
        real = self.vid[:, self.ctx_len:]
        loss = (inp_output[:, :, :3] - real).abs().mean()                   # `sharp_vid` (synthesizer.py:587-590)
        loss.backward()                                                     # synthesizer.py:631
        return loss

    def grads_finite(self):
        return all(p.grad is not None and bool(torch.isfinite(p.grad).all()) for p in self.unet.parameters())

    def hd_algorithmic_bytes(self):
        """Bytes the full-resolution entry points have to move per step (each input read once, each output written
        once; fp32), by C-ABI name -- as ``Pipeline.hd_algorithmic_bytes``, for the UNRESTRICTED flow synthesis (the
        first pass composites on all T frames and writes ``alpha`` as well: ``inpaint`` keeps it in the tuple)."""
        o = self.opt
        b, t, tc = self.clips, self.frames, self.ctx_len
        tp = t - tc
        nl, ncls, c = o.num_obj + 1, o.num_lyt, 3 + o.num_lyt
        hw = o.dim * int(o.dim * o.aspect_ratio)
        hwd = o.load_dim * int(o.load_dim * o.aspect_ratio)
        m, u = b * tc * tp, b * tp
        lr = m * (nl * 2) * hw                                   # per-layer flows (no ghost mask on this path)
        return {
            "waldo_flow_ctx_alpha_fwd": 4 * (b * t * nl * hw + b * t * ncls * hwd + 2 * b * t * nl * hwd),
            "waldo_flow_ctx_warp_raw_fwd": 4 * (lr + b * tc * nl * hwd + m * (2 + nl + 1) * hwd + m * hwd),
            "waldo_frame_warp_fuse_raw_fwd": 4 * (b * tc * c * hwd + m * (2 + 1) * hwd + u * (c + 1) * hwd + m * c * hwd),
            "waldo_wif_fuse_fwd": 4 * (m * 4 * hwd + m * 4 * hwd + u * 3 * hwd),
            # reads vid channels 0-2 and 4, the network's 4 used outputs, out and grad_out; writes all 5 planes of
            # grad_net per (unit, context) (raw_output carries no gradient: grad_vid is not produced)
            "waldo_wif_fuse_bwd": 4 * (m * 4 * hwd + m * 4 * hwd + 2 * u * 3 * hwd + m * 5 * hwd),
        }
