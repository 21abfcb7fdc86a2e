"""dev: which host lines of the C5 / C4 pipeline make a COPY (contiguous() / reshape() of a non-contiguous tensor)?"""
import collections
import sys
import traceback

import torch

sys.path.insert(0, '.')
from waldo_amd.tools import pipeline  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "C5"
pipe = pipeline.Pipeline(name, 2, torch.device("cuda:0"))
pipe()
hits = collections.Counter()
orig_c, orig_r = torch.Tensor.contiguous, torch.Tensor.reshape


def where():
    for fr in reversed(traceback.extract_stack()[:-2]):
        if "waldo_amd" in fr.filename:
            return f"{fr.filename.split('waldo_amd/')[-1]}:{fr.lineno} {fr.line}"
    return "?"


def contiguous(self, *a, **k):
    if not self.is_contiguous():
        hits[("contiguous", where())] += self.numel() * 4
    return orig_c(self, *a, **k)


def reshape(self, *shape):
    out = orig_r(self, *shape)
    if out.data_ptr() != self.data_ptr() or (not self.is_contiguous() and out._base is None and out.numel() > 0 and out is not self):
        if not self.is_contiguous() and out._base is None:
            hits[("reshape", where())] += self.numel() * 4
    return out


torch.Tensor.contiguous, torch.Tensor.reshape = contiguous, reshape
with torch.no_grad():
    pipe()
for (kind, w), nbytes in hits.most_common(25):
    print(f"{nbytes / 1e6:10.1f} MB  {kind:10s} {w}")
