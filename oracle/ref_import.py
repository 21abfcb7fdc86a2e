"""Import harness for the upstream reference ops (TEST INFRASTRUCTURE ONLY).

Only usable in the build container, where the read-only reference tree lives
at /root/reference.  Nothing here travels to the GPU box in a useful form: the
functions raise ``ReferenceUnavailable`` when the tree is absent, and every
caller (``oracle/make_golden.py`` and the ``live_ref`` tests) skips then.

The reference package cannot be imported as a package under this image
(``models/modules/__init__.py:9`` pulls ``mat.py`` -> ``timm``; ``tools/utils.py:13``
imports ``torchvision``), so the hot-path files are loaded one by one under stub
parents, as SURVEY.md section 8(c) describes.  No reference source is copied: the
files are executed from where they lie.
"""
import importlib.util
import os
import sys
import types

REF_ROOT = os.environ.get("WALDO_REFERENCE_ROOT", "/root/reference")


class ReferenceUnavailable(RuntimeError):
    pass


_cache = {}


def available():
    return os.path.isfile(os.path.join(REF_ROOT, "models", "modules", "warp.py"))


def _stub(name, **attrs):
    mod = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(mod, k, v)
    sys.modules[name] = mod
    return mod


def _load(modname, relpath):
    spec = importlib.util.spec_from_file_location(modname, os.path.join(REF_ROOT, relpath))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[modname] = mod
    spec.loader.exec_module(mod)
    return mod


def load():
    """Returns a namespace with the reference's TPSWarp, InverseWarp, Warper, LVD
    (class only), PoseDecoder (class only), WIF, get_grid, get_gaussian_kernel, gather_time, scale."""
    if "ns" in _cache:
        return _cache["ns"]
    if not available():
        raise ReferenceUnavailable(f"reference tree not found at {REF_ROOT}")
    sys.dont_write_bytecode = True
    import torch

    saved = {k: sys.modules.get(k) for k in
             ("torchvision", "torchvision.transforms", "torchvision.utils", "torchvision.io",
              "torchvision.models", "lpips", "tools", "tools.utils", "models", "models.modules",
              "models.nets")}
    tv = _stub("torchvision")
    tv.transforms = _stub("torchvision.transforms", GaussianBlur=object)
    tv.utils = _stub("torchvision.utils")
    tv.io = _stub("torchvision.io")
    tv.models = _stub("torchvision.models")
    _stub("lpips")

    tools = _stub("tools")
    tools.__path__ = [os.path.join(REF_ROOT, "tools")]
    utils = _load("tools.utils", "tools/utils.py")
    tools.utils = utils

    models = _stub("models")
    models.__path__ = [os.path.join(REF_ROOT, "models")]
    modules = _stub("models.modules")
    modules.__path__ = [os.path.join(REF_ROOT, "models", "modules")]
    nets = _stub("models.nets")
    nets.__path__ = [os.path.join(REF_ROOT, "models", "nets")]
    models.modules, models.nets = modules, nets

    spectral = _load("models.modules.spectral", "models/modules/spectral.py")
    weight_init = _load("models.modules.weight_init", "models/modules/weight_init.py")
    transform = _load("models.modules.transform", "models/modules/transform.py")
    warp = _load("models.modules.warp", "models/modules/warp.py")
    conv = _load("models.modules.conv", "models/modules/conv.py")
    for name, src in (("TPSWarp", warp), ("InverseWarp", warp), ("CustomNorm", transform),
                      ("MultiBlocks", transform), ("Block", transform),
                      ("trunc_normal_", weight_init), ("init_weights", weight_init),
                      ("ConvPatchProj", conv), ("UNet", conv)):
        setattr(modules, name, getattr(src, name))
    lvd = _load("models.nets.lvd", "models/nets/lvd.py")

    # WIF.__init__ calls .cuda() on a buffer (models/nets/wif.py:31): a no-op while the file is executed, put back
    # afterwards (the callers build WIF with __new__ and never run that constructor; `cuda_is_noop` is there for one
    # that would)
    with cuda_is_noop():
        wif = _load("models.nets.wif", "models/nets/wif.py")
    flp = _load("models.nets.flp", "models/nets/flp.py")
    ns = types.SimpleNamespace(
        PoseDecoder=flp.PoseDecoder,
        TPSWarp=warp.TPSWarp, InverseWarp=warp.InverseWarp, kernel_distance=warp.kernel_distance,
        Warper=lvd.Warper, LVD=lvd.LVD, ImageDecoder=lvd.ImageDecoder, get_circle=lvd.get_circle,
        gather_time=lvd.gather_time, scale=lvd.scale,
        WIF=wif.WIF, UNet=conv.UNet, get_grid=utils.get_grid,
        get_gaussian_kernel=utils.get_gaussian_kernel, expand=utils.expand)
    _cache["ns"] = ns
    # keep the stubs out of the way of real imports done later by the test session
    for k in ("torchvision", "torchvision.transforms", "torchvision.utils", "torchvision.io",
              "torchvision.models", "lpips", "tools", "tools.utils", "models", "models.modules",
              "models.nets"):
        if saved[k] is None:
            sys.modules.pop(k, None)
        else:
            sys.modules[k] = saved[k]
    return ns


def warper_opt(**over):
    """SimpleNamespace with the 18 option fields Warper reads (models/nets/lvd.py:472-499)."""
    d = dict(latent_shape=[2, 4], obj_shape=[2, 2], time_dropout=False, num_obj=2, patch_size=4,
             scale_factor=1, dim=16, aspect_ratio=2, load_dim=32, num_perm_grid=1,
             normalize_alpha=False, use_lyt_filtering=False, use_lyt_opacity=False,
             weight_cls=False, min_cls=0.0, include_self=False, no_filter=False, allow_ghost=False)
    d.update(over)
    return types.SimpleNamespace(**d)


class cuda_is_noop:
    """Context manager: ``Tensor.cuda`` returns the tensor itself (this container has no GPU); the original method is
    put back on exit, whatever happened inside."""

    def __enter__(self):
        import torch
        self._orig = torch.Tensor.cuda
        torch.Tensor.cuda = lambda t, *a, **k: t
        return self

    def __exit__(self, *exc):
        import torch
        torch.Tensor.cuda = self._orig
        return False


class stable_sort:
    """Context manager: run the reference with ``Tensor.sort`` forced to ``stable=True``.

    InverseWarp's duplicate removal (models/modules/warp.py:113-117) keeps the first element of
    each run of equal keys after an UNSTABLE ``field.sort(dim=-1)``; which colliding source wins
    is therefore implementation-defined in the reference (it differs between torch builds and
    between CPU and GPU).  The build fixes the tie-break to "lowest source index wins", which
    is exactly the reference under a stable sort -- golden vectors for InverseWarp are
    generated inside this context, and say so."""

    def __enter__(self):
        import torch
        self._orig = torch.Tensor.sort
        orig = self._orig

        def _sort(t, *a, **k):
            k.setdefault("stable", True)
            return orig(t, *a, **k)
        torch.Tensor.sort = _sort
        return self

    def __exit__(self, *exc):
        import torch
        torch.Tensor.sort = self._orig
        return False
