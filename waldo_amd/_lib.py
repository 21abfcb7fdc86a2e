"""ctypes binding of the C-ABI HIP library (include/waldo_hip.h).

The product path has NO CPU fallback: if ``libwaldo_hip.so`` is missing or fails to load, every
op raises.  ``torch`` is imported before the library is opened so that the HIP runtime the
library binds to (SONAME libamdhip64.so.7) is the one PyTorch-ROCm already loaded -- device
pointers and streams are then shared between the two.
"""
import ctypes
import os
import threading

import torch  # noqa: F401  (must precede CDLL: see module docstring)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libwaldo_hip.so")
# ABI this binding was written against (include/waldo_hip.h: waldo_version() = major * 1000 + minor); a
# library of another version has other prototypes behind the same names and is refused by load()
ABI_VERSION = 1019

_c_f = ctypes.c_void_p  # device pointers travel as integers
_i64 = ctypes.c_int64
_int = ctypes.c_int
_flt = ctypes.c_float
_stream = ctypes.c_void_p

# name -> argtypes; mirrors include/waldo_hip.h one to one (tests check the header against this)
SIGNATURES = {
    "waldo_tps_mapping_fwd": [_c_f, _c_f, _c_f, _i64, _int, _stream],
    "waldo_tps_mapping_bwd": [_c_f, _c_f, _c_f, _i64, _int, _stream],
    "waldo_tps_grid_fwd": [_c_f, _c_f, _c_f, _i64, _i64, _int, _stream],
    "waldo_tps_grid_bwd": [_c_f, _c_f, _c_f, _i64, _i64, _int, _stream],
    "waldo_inverse_warp_fwd": [_c_f] * 14 + [_i64, _int, _int, _int, _int, _int, _int, _int, _stream],
    "waldo_inverse_warp_order_fwd": [_c_f] * 16 + [_i64, _int, _int, _int, _int, _int, _int, _int, _stream],
    "waldo_inverse_warp_bwd": [_c_f] * 9 + [_i64, _int, _int, _int, _int, _int, _int, _stream],
    "waldo_grid_sample2d_fwd": [_c_f, _c_f, _c_f, _i64, _int, _int, _int, _int, _int, _flt, _i64,
                                _i64, _i64, _i64, _stream],
    "waldo_grid_sample2d_ex_fwd": [_c_f, _c_f, _c_f, _c_f, _i64, _int, _int, _int, _int, _int, _flt, _i64,
                                   _i64, _i64, _i64, _i64, _i64, _i64, _flt, _flt, _stream],
    "waldo_grid_sample2d_bwd": [_c_f, _c_f, _c_f, _c_f, _c_f, _i64, _int, _int, _int, _int, _int,
                                _flt, _i64, _i64, _stream],
    "waldo_grid_sample2d_ex_bwd": [_c_f, _c_f, _c_f, _c_f, _c_f, _i64, _int, _int, _int, _int, _int,
                                   _flt, _i64, _i64, _i64, _i64, _i64, _flt, _flt, _stream],
    "waldo_occ_composite_fwd": [_c_f, _c_f, _c_f, _i64, _int, _i64, _i64, _stream],
    "waldo_occ_composite_bwd": [_c_f, _c_f, _c_f, _c_f, _c_f, _i64, _int, _i64, _i64, _stream],
    "waldo_compute_occ_fwd": [_c_f, _c_f, _i64, _int, _flt, _stream],
    "waldo_compute_occ_bwd": [_c_f, _c_f, _c_f, _i64, _int, _flt, _stream],
    "waldo_alpha_head_fwd": [_c_f] * 4 + [_i64, _int, _int, _int, _int, _flt, _int, _int, _stream],
    "waldo_alpha_head_bwd": [_c_f] * 5 + [_i64, _int, _int, _int, _int, _flt, _int, _int, _stream],
    "waldo_pose_affine_fwd": [_c_f] * 5 + [_i64, _int, _flt, _flt, _stream],
    "waldo_pose_affine_bwd": [_c_f] * 6 + [_i64, _int, _flt, _flt, _stream],
    "waldo_disocc_test_fwd": [_c_f, _c_f, _i64, _int, _int, _i64, _stream],
    "waldo_flow_ctx_alpha_fwd": [_c_f] * 7 + [_int] * 10 + [_stream],
    "waldo_flow_ctx_warp_fwd": [_c_f] * 12 + [_int] * 9 + [_stream],
    "waldo_frame_warp_fuse_fwd": [_c_f] * 7 + [_int] * 9 + [_flt, _stream],
    "waldo_flow_ctx_warp_raw_fwd": [_c_f] * 13 + [_int] * 11 + [_stream],
    "waldo_frame_warp_fuse_raw_fwd": [_c_f] * 7 + [_int] * 9 + [_flt, _stream],
    "waldo_flow_ctx_alpha_bwd": [_c_f] * 10 + [_int] * 10 + [_stream],
    "waldo_flow_ctx_warp_bwd": [_c_f] * 13 + [_int] * 9 + [_stream],
    "waldo_frame_warp_fuse_bwd": [_c_f] * 8 + [_int] * 9 + [_flt, _stream],
    "waldo_lyt_dist_fwd": [_c_f, _c_f, _i64, _i64, _c_f, _flt] + [_c_f] * 4 + [_i64] + [_int] * 7 + [_stream],
    "waldo_lyt_dist_bwd": [_c_f] * 3 + [_i64, _i64, _c_f, _flt] + [_c_f] * 6 + [_i64] + [_int] * 7 + [_stream],
    "waldo_wif_fuse_fwd": [_c_f, _c_f, _c_f, _i64, _int, _int, _int, _i64, _int, _stream],
    "waldo_wif_fuse_bwd": [_c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _i64, _int, _int, _int, _i64, _int,
                           _stream],
    "waldo_time_gather_fwd": [_c_f, _c_f, _c_f, _c_f, _c_f, _int, _int, _int, _int, _i64, _i64, _int, _stream],
    "waldo_time_gather_bwd": [_c_f, _c_f, _c_f, _c_f, _int, _int, _int, _int, _i64, _i64, _int, _stream],
    "waldo_downscale_frames_fwd": [_c_f, _c_f] + [_int] * 8 + [_stream],
    "waldo_mask_expand_fwd": [_c_f, _c_f, _c_f, _i64, _int, _int, _int, _int, _int, _flt, _stream],
    "waldo_points_in_polygon_fwd": [_c_f, _c_f, _int, _c_f, _i64, _stream],
    "waldo_inpaint_propagate_fwd": [_c_f] * 8 + [_int] + [_c_f] * 7 + [_i64, _int, _int, _int, _int, _stream],
    "waldo_inpaint_blend_fwd": [_c_f] * 4 + [_i64, _i64, _stream],
    "waldo_inpaint_holes_fwd": [_c_f, _i64, _i64, _i64, _i64, _c_f, _c_f, _i64, _int, _int, _int, _i64, _int, _flt, _stream],
    "waldo_warp_composite_fwd": [_c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _i64, _int, _int, _int, _int,
                                 _flt, _stream],
    "waldo_warp_composite_pts_fwd": [_c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _i64, _int, _int, _int, _int,
                                     _flt, _stream],
    "waldo_warp_composite_bwd": [_c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _i64,
                                 _i64, _int, _int, _int, _int, _flt, _stream],
}
PLAIN = {"waldo_version": (_int, []), "waldo_max_layers": (_int, []),
         "waldo_warp_composite_bwd_workspace_bytes": (_i64, [_i64, _int, _int, _int, _int]),
         "waldo_lyt_dist_workspace_bytes": (_i64, [_i64, _int, _int, _int, _int, _int]),
         "waldo_warp_composite_pts_supported": (_int, [_int, _int, _int, _int]),
         "waldo_last_error_string": (ctypes.c_char_p, []),
         "waldo_set_debug_option": (_int, [_int, _int]),
         "waldo_host_device_pointer": (_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p)])}

# include/waldo_hip.h: test-only switches between kernel variants (waldo_set_debug_option: the library's one piece of
# process-global state)
DEBUG_FWD_PLAIN = 0
DEBUG_IW_PASSES = 1
DEBUG_BWD_GENERIC = 2
INDEX_STATUS_WORDS = 4

_lock = threading.Lock()
_lib = None
_explicit = False  # the library was named by use_library(): the only way a timing-only ablation build is accepted


class WaldoHipError(RuntimeError):
    pass


def use_library(path):
    """Bind another build of the library (developer A/B runs: tools_dev/).  Only before the first call
    of any op -- the process keeps ONE library; nothing reads the environment."""
    global LIB_PATH, _explicit
    with _lock:
        if _lib is not None:
            raise WaldoHipError(f"use_library({path!r}): {LIB_PATH} is already loaded")
        LIB_PATH = os.path.abspath(path)
        _explicit = True


def load():
    """Open the library (once) and declare every prototype.  Raises if it is absent or of another ABI."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise WaldoHipError(
                f"{LIB_PATH} not found: build it with `python -m waldo_amd.build` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_LOCAL)
        lib.waldo_version.restype = _int
        lib.waldo_version.argtypes = []
        if lib.waldo_version() == 0 and _explicit:
            # a timing-only ablation build (csrc/waldo_common.hip.h: WALDO_ABL_*, may compute WRONG values): only ever
            # bound when a developer named it with use_library() / bench.py --lib, and never silently
            import sys
            print(f"waldo_amd: {LIB_PATH} is a TIMING-ONLY ablation build (waldo_version() == 0): its results are "
                  "not the product's", file=sys.stderr)
        elif lib.waldo_version() != ABI_VERSION:
            raise WaldoHipError(
                f"{LIB_PATH} has ABI version {lib.waldo_version()}, this binding needs {ABI_VERSION}: "
                "rebuild it with `python -m waldo_amd.build --force`")
        for name, argtypes in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.argtypes = argtypes
            fn.restype = _int
        for name, (res, argtypes) in PLAIN.items():
            fn = getattr(lib, name)
            fn.argtypes = argtypes
            fn.restype = res
        _lib = lib
    return _lib


class IndexStatus:
    """The caller's side of include/waldo_hip.h's "Frame-index status": four sticky int32 words in PINNED HOST memory
    that the kernels write, through the words' device address, when ``ctx_ts`` / ``pred_ts`` hold a frame index
    outside its time axis (they clamp it for memory safety and report; the reference's ``gather_time``,
    models/nets/lvd.py:462-467, fails with a device-side assert there).  Reading the words costs no device
    synchronisation, so an owner (``Warper``: one per module) checks them at every call for what EARLIER launches
    reported, and ``check(sync=True)`` gives the raise-at-once behaviour.  Replaces the host-side index bookkeeping
    of rounds 3-5 (one device -> host read per distinct index tensor and version)."""

    _live = None  # weak set of every IndexStatus of the process (check_all: after a HIP-graph replay)

    def __init__(self):
        import weakref
        self.words = torch.zeros(INDEX_STATUS_WORDS, dtype=torch.int32).pin_memory()
        dev = ctypes.c_void_p()
        if load().waldo_host_device_pointer(self.words.data_ptr(), ctypes.byref(dev)) != 0:
            msg = load().waldo_last_error_string()
            raise WaldoHipError(f"IndexStatus: {msg.decode() if msg else '?'}")
        self.ptr = dev.value
        self._np = self.words.numpy()
        if IndexStatus._live is None:
            IndexStatus._live = weakref.WeakSet()
        IndexStatus._live.add(self)

    def __deepcopy__(self, memo):  # (a copied module gets words of its own: the device address is not copyable)
        return IndexStatus()

    def __reduce__(self):
        return (IndexStatus, ())

    def check(self, sync=False):
        """Raise ``WaldoHipError`` if a kernel has reported an index outside its range since the last check (and clear
        the words).  ``sync``: wait for the current stream first, so that every launch queued so far has run.  A no-op
        while a HIP graph is being captured."""
        if torch.cuda.is_current_stream_capturing():
            return
        if sync:
            torch.cuda.current_stream().synchronize()
        if self._np[0] or self._np[2]:
            lim_c, bad_c, lim_p, bad_p = (int(v) for v in self._np)
            self._np[:] = 0
            what = [f"{name} holds the index {bad}, valid range is [0, {lim - 1}]"
                    for name, lim, bad in (("ctx_ts", lim_c, bad_c), ("pred_ts", lim_p, bad_p)) if lim]
            raise WaldoHipError("frame index outside its time axis (reported by a kernel; the frame was clamped): "
                                + "; ".join(what))

    @staticmethod
    def check_all(sync=False):
        for st in list(IndexStatus._live or ()):
            st.check(sync)
            sync = False


_default_status = None


def default_index_status():
    """The status words of callers that pass none: ``functional`` checks them with a synchronisation before it
    returns (the strict mode of a stand-alone call)."""
    global _default_status
    if _default_status is None:
        _default_status = IndexStatus()
    return _default_status


def ptr(t):
    """Device pointer of a tensor (None -> NULL), as the integer ctypes converts to ``void*`` itself."""
    if t is None:
        return None
    return t.data_ptr()


def check_cuda(*tensors):
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise WaldoHipError("waldo_amd ops need tensors on the GPU (cuda device); "
                                "there is no CPU fallback")
        if t.dtype != torch.float32:
            raise WaldoHipError(f"waldo_amd ops are fp32-only, got {t.dtype}")


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def current_stream(device):
    """The ``hipStream_t`` PyTorch launches on for ``device`` right now (inside ``torch.cuda.stream(...)``: that one).
    The raw query when the build has it: a library call is ~20 us of interpreter time around a ~5 us launch, and the
    LVD training step makes ~45 of them -- building a ``torch.cuda.Stream`` object per call was a quarter of that."""
    if _raw_stream is not None and device.index is not None:
        return _raw_stream(device.index)
    return torch.cuda.current_stream(device).cuda_stream


class on_device:
    """``with on_device(dev):`` = ``with torch.cuda.device(dev):`` (the kernels launch on the CURRENT device) that
    costs nothing when ``dev`` already is the current device -- the one-GPU-per-process case."""
    __slots__ = ("guard",)

    def __init__(self, device):
        self.guard = None if device.index is None or device.index == torch.cuda.current_device() else torch.cuda.device(device)

    def __enter__(self):
        if self.guard is not None:
            self.guard.__enter__()

    def __exit__(self, *exc):
        if self.guard is not None:
            return self.guard.__exit__(*exc)
        return False


class KernelTimer:
    """Optional per-entry-point device timing: records an event pair on the launch stream around
    every C-ABI call (bench.py uses it for the live roofline numbers).  Off by default."""

    def __init__(self):
        self.pairs = {}

    def __enter__(self):
        global _timer
        _timer = self
        return self

    def __exit__(self, *exc):
        global _timer
        _timer = None
        return False

    def summary(self):
        """name -> (launches, mean milliseconds); call after torch.cuda.synchronize()."""
        out = {}
        for name, evs in self.pairs.items():
            ms = [a.elapsed_time(b) for a, b in evs]
            out[name] = (len(ms), sum(ms) / max(len(ms), 1))
        return out


_timer = None


def call(name, *args):
    """Call an entry point; raise WaldoHipError with the library's message on failure."""
    lib = load()
    if _timer is not None:
        start = torch.cuda.Event(enable_timing=True)
        stop = torch.cuda.Event(enable_timing=True)
        start.record()
        rc = getattr(lib, name)(*args)
        stop.record()
        _timer.pairs.setdefault(name, []).append((start, stop))
    else:
        rc = getattr(lib, name)(*args)
    if rc != 0:
        msg = lib.waldo_last_error_string()
        raise WaldoHipError(f"{name} failed ({rc}): {msg.decode() if msg else '?'}")
