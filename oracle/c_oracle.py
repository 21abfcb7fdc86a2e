"""ctypes loader of the plain-C restatement (oracle/wif_oracle.c) -- TEST INFRASTRUCTURE ONLY.

``fused(...)`` runs TPS grid -> bilinear warp -> reduce_comp forward and backward in double precision
and returns numpy arrays.  Built by ``make -C oracle`` (``__graft_entry__.build()`` does that)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libwif_oracle.so")
_lib = None


def load():
    global _lib
    if _lib is None:
        src = os.path.join(_HERE, "wif_oracle.c")
        if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
            subprocess.run(["make", "-C", _HERE, "-s"], check=True)
        _lib = ctypes.CDLL(_SO)
        fp, dp = ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_double)
        _lib.waldo_oracle_fused.restype = ctypes.c_int
        _lib.waldo_oracle_fused.argtypes = [fp, fp, fp, fp] + [ctypes.c_int] * 5 + [fp, fp, ctypes.c_int, ctypes.c_double] + [dp] * 5
    return _lib


def _f32(a):
    return None if a is None else np.ascontiguousarray(np.asarray(a, dtype=np.float32))


def _ptr(a, t):
    return None if a is None else a.ctypes.data_as(ctypes.POINTER(t))


def fused(layers, pts, occ, ctrl, w_rgb=None, w_alpha=None, loss_sq=False, backward=True, delta=0.0):
    """layers (F,L,4,H,W), pts (F*L,N,2), occ (F,L,L), ctrl (N,2).  Loss: sum(rgb*w_rgb) +
    sum(alpha*w_alpha), or mean(rgb^2) with ``loss_sq``.  delta: grid_sample(x + delta) - delta
    (lvd.py:548,559).  Returns a dict of float64 arrays."""
    layers, pts, occ, ctrl, w_rgb, w_alpha = map(_f32, (layers, pts, occ, ctrl, w_rgb, w_alpha))
    f, nl, _, h, w = layers.shape
    n = ctrl.shape[0]
    out = {"rgb": np.empty((f, 3, h, w)), "alpha": np.empty((f, nl, h, w))}
    if backward:
        out.update(grad_layers=np.empty((f, nl, 4, h, w)), grad_pts=np.empty((f * nl, n, 2)),
                   grad_occ=np.empty((f, nl, nl)))
    d = ctypes.c_double
    rc = load().waldo_oracle_fused(
        _ptr(layers, ctypes.c_float), _ptr(pts, ctypes.c_float), _ptr(occ, ctypes.c_float),
        _ptr(ctrl, ctypes.c_float), f, nl, h, w, n, _ptr(w_rgb, ctypes.c_float), _ptr(w_alpha, ctypes.c_float),
        int(bool(loss_sq)), float(delta), _ptr(out["rgb"], d), _ptr(out["alpha"], d), _ptr(out.get("grad_layers"), d),
        _ptr(out.get("grad_pts"), d), _ptr(out.get("grad_occ"), d))
    if rc != 0:
        raise RuntimeError(f"waldo_oracle_fused failed with code {rc}")
    return out
