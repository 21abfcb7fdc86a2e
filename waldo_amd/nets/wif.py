"""Drop-in for the hot-path part of the reference's models/nets/wif.py: ``WIF.forward``.

The UNet itself (models/modules/conv.py, MIOpen convolutions) is out of scope: the constructor
takes any ``nn.Module`` mapping (N, C_in, H, W) -> (N, 4|5, H, W) (or builds nothing when None is
given and ``forward`` is called with precomputed network outputs through ``fuse``).  The fusion
arithmetic around it runs in one hand-written gfx950 kernel (``waldo_wif_fuse_*``)."""
import torch.nn as nn

from .. import functional as WF


class WIF(nn.Module):
    """``forward(vid)`` with vid (B, Tc, T, C, H, W) -> (B, T, 3, H, W), reference wif.py:37-57.

    opt fields read: ``ii_score`` (must be true: the score-fusion variant every script uses),
    ``ii_ab``."""

    def __init__(self, opt, unet=None):
        super().__init__()
        self.score = opt.ii_score
        self.ab = opt.ii_ab
        self.opt = opt
        self.unet = unet

    def get_last_layer(self):
        return self.unet.from_emb.weight

    def fuse(self, vid_t, net_out):
        """vid_t (B, T, Tc, C, H, W) (already permuted), net_out (B, T, Tc, Co, H, W)."""
        return WF.wif_fuse(vid_t, net_out, ab=self.ab)

    def forward(self, vid):
        b, tc, t, c, h, w = vid.shape
        vid = vid.permute(0, 2, 1, 3, 4, 5).contiguous()
        if not self.score:
            out = self.unet(vid.reshape(b * t, tc * c, h, w))
            return out.reshape(b, t, -1, h, w)
        out = self.unet(vid.reshape(b * t * tc, c, h, w))
        return self.fuse(vid, out.reshape(b, t, tc, -1, h, w))
