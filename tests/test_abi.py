"""CPU: the C-ABI library builds/loads and exports every symbol include/waldo_hip.h declares.
No compute call is made here (no GPU in the build container)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "waldo_hip.h")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(waldo_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def lib_path():
    from waldo_amd import build
    if not os.path.exists(build.LIB):
        build.build(verbose=False)
    return build.LIB


def test_header_declares_something():
    fns = declared_functions()
    assert "waldo_warp_composite_fwd" in fns and "waldo_warp_composite_bwd" in fns
    assert len(fns) >= 13


def test_library_exports_every_declared_symbol(lib_path):
    lib = ctypes.CDLL(lib_path)
    for name in declared_functions():
        assert hasattr(lib, name), f"{name} declared in waldo_hip.h but not exported"


def test_binding_matches_header(lib_path):
    from waldo_amd import _lib
    bound = set(_lib.SIGNATURES) | set(_lib.PLAIN)
    assert bound == set(declared_functions())
    # argument counts agree with the header prototypes
    src = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    for name, argtypes in _lib.SIGNATURES.items():
        m = re.search(name + r"\s*\((.*?)\)\s*;", src, flags=re.S)
        assert m, name
        nargs = len([a for a in m.group(1).split(",") if a.strip()])
        assert nargs == len(argtypes), (name, nargs, len(argtypes))


def test_plain_calls_without_gpu(lib_path):
    from waldo_amd import _lib
    lib = _lib.load()
    assert lib.waldo_version() >= 1000
    assert lib.waldo_max_layers() == 32


def test_argument_validation_needs_no_gpu(lib_path):
    """Bad shapes are rejected on the host before any launch, with a message."""
    from waldo_amd import _lib
    lib = _lib.load()
    rc = lib.waldo_warp_composite_fwd(None, None, None, None, None, None, 1, 99, 8, 8, 19, 0.0, None)
    assert rc == -1
    assert b"unsupported shape" in lib.waldo_last_error_string()
    rc = lib.waldo_grid_sample2d_fwd(None, None, None, 1, 0, 4, 4, 4, 4, 0.0, 1, 1, 1, 1, None)
    assert rc == -1
    # round 6's entry points (WIF.inpaint's dilation, polygon test and propagation; InverseWarp's kernel size)
    assert lib.waldo_mask_expand_fwd(None, None, None, 1, 8, 8, 3, 16, 0, 0.97, None) == -1       # steps: four bits
    assert b"bad arguments" in lib.waldo_last_error_string()
    assert lib.waldo_mask_expand_fwd(None, None, None, 1, 8, 8, 3, 15, 0, 0.97, None) == -1       # null planes
    assert lib.waldo_mask_expand_fwd(None, None, None, 0, 8, 8, 3, 15, 0, 0.97, None) == 0        # nothing to do
    assert lib.waldo_points_in_polygon_fwd(None, None, 17, None, 4, None) == -1                    # at most 16 corners
    assert lib.waldo_points_in_polygon_fwd(None, None, 4, None, 0, None) == 0
    assert lib.waldo_inpaint_propagate_fwd(*([None] * 8), 3, *([None] * 7), 1, 8, 8, 0, 0, None) == -1   # two objects at most
    assert b"entering" in lib.waldo_last_error_string()
    assert lib.waldo_inpaint_blend_fwd(None, None, None, None, 1, 64, None) == -1
    args = [None] * 14 + [1, 8, 8, 8, 8, 5, 1]
    assert lib.waldo_inverse_warp_fwd(*args, 4, None) == -1                                        # even window
    assert b"kernel size" in lib.waldo_last_error_string()


def test_product_has_no_cpu_fallback():
    import torch
    import waldo_amd
    from waldo_amd import functional as WF
    from waldo_amd._lib import WaldoHipError
    from waldo_amd.tools.utils import get_grid
    tps = waldo_amd.TPSWarp(8, 8, get_grid(4, 4).view(-1, 2))
    with pytest.raises(WaldoHipError):
        tps(get_grid(4, 4).view(1, 16, 2))
    with pytest.raises(WaldoHipError):
        WF.grid_sample(torch.zeros(1, 1, 4, 4), torch.zeros(1, 4, 4, 2))


def test_product_does_not_import_oracle():
    """The shipped package must never route through the oracle."""
    pkg = os.path.join(ROOT, "waldo_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in text.replace("no oracle", ""), os.path.join(dirpath, f)


def test_only_the_checker_legs_touch_the_oracle():
    """Outside ``tests/`` the oracle is imported in exactly two places -- ``__graft_entry__.smoke()`` (the checker of the
    smoke run) and ``bench.py``'s ``cpu_baseline`` child -- and by no development script (``tools_dev/``)."""
    import re
    pat = re.compile(r"^\s*(from\s+oracle\b|import\s+oracle\b)", re.M)
    for dirpath, _, files in os.walk(os.path.join(ROOT, "tools_dev")):
        for f in files:
            if f.endswith((".py", ".sh")):
                assert not pat.search(open(os.path.join(dirpath, f)).read()), os.path.join(dirpath, f)
    bench = open(os.path.join(ROOT, "bench.py")).read()
    hits = [m.start() for m in pat.finditer(bench)]
    assert len(hits) == 1
    inside = bench.rfind("\ndef ", 0, hits[0])
    assert bench[inside:].lstrip().startswith("def cpu_baseline_child"), bench[inside:inside + 60]
    entry = open(os.path.join(ROOT, "__graft_entry__.py")).read()
    hits = [m.start() for m in pat.finditer(entry)]
    assert hits and all(entry[entry.rfind("\ndef ", 0, h):].lstrip().startswith("def smoke") for h in hits)


def test_timing_only_ablations_cannot_reach_a_product_build(tmp_path):
    """A WALDO_ABL_* switch (timing-only, may compute wrong values) without -DWALDO_TIMING_ONLY_BUILD does not get
    past the preprocessor; with it, the library reports version 0."""
    import subprocess
    from waldo_amd import build
    src = os.path.join(build.CSRC, "runtime.hip")
    base = [build.HIPCC, "-E", "--offload-arch=gfx950", "--cuda-host-only", f"-I{build.INCLUDE}"]
    bad = subprocess.run(base + ["-DWALDO_ABL_REC_ALIAS=2", src], capture_output=True, text=True)
    assert bad.returncode != 0 and "WALDO_TIMING_ONLY_BUILD" in bad.stderr
    ok = subprocess.run(base + ["-DWALDO_ABL_REC_ALIAS=2", "-DWALDO_TIMING_ONLY_BUILD", src], capture_output=True,
                        text=True)
    assert ok.returncode == 0, ok.stderr[-500:]
    assert re.search(r"waldo_version\(void\)\s*\{\s*return 0;", ok.stdout)
    assert not re.search(r"waldo_version\(void\)\s*\{\s*return 1\d\d\d;", ok.stdout)


def test_binding_refuses_a_version_0_library_unless_named_explicitly(tmp_path):
    """What a timing-only build looks like to the binding: waldo_version() == 0.  The default path refuses it."""
    import subprocess
    import sys
    csrc = tmp_path / "v0.c"
    csrc.write_text("int waldo_version(void) { return 0; }\n")
    so = tmp_path / "libwaldo_hip.so"
    subprocess.run(["gcc", "-shared", "-fPIC", "-o", str(so), str(csrc)], check=True)
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from waldo_amd import _lib\n"
            "_lib.LIB_PATH = %r\n"
            "try:\n"
            "    _lib.load()\n"
            "except _lib.WaldoHipError as e:\n"
            "    assert 'ABI version 0' in str(e), e\n"
            "    print('refused')\n") % (ROOT, str(so))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "refused" in r.stdout, (r.stdout, r.stderr[-800:])


def test_entry_points_issue_no_memset_or_memcpy_nodes():
    """Fills and copies inside the library are kernels (waldo_common.hip.h: fill_words / copy_bytes): captured into a HIP
    graph a hipMemsetAsync becomes a memset NODE, and replaying a graph with the grid inversion's 107 MB one after an eager
    kernel faulted on ROCm 7.2 (round 5, tools_dev/repro_graph_parts.py)."""
    from waldo_amd import build
    for src in build.sources() + build._deps():
        text = re.sub(r"//[^\n]*", "", open(src).read())
        assert "hipMemsetAsync" not in text and "hipMemcpyAsync" not in text and "hipMemset(" not in text, src
