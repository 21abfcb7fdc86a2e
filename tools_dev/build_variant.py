"""dev: build a variant of the library with extra compiler flags for A/B timing.

    python tools_dev/build_variant.py NAME [--only unit,unit] -DWALDO_STAGE_AHEAD=3 ...   ->  tools_dev/_variants/NAME.so

Variants live under tools_dev/_variants/ (git-ignored), never next to the product library.  A -DWALDO_ABL_* flag
(a timing-only ablation, may compute wrong values: csrc/waldo_common.hip.h) adds -DWALDO_TIMING_ONLY_BUILD and
recompiles csrc/runtime.hip with it, so that the variant reports waldo_version() == 0 and only loads through
use_library() / bench.py --lib.

--only: recompile just those translation units (e.g. warp_composite_lp8,warp_composite_splat) with the
flags and link them with the PRODUCT objects of every other unit (build the product first).
Objects go to their own directory, so the product build is not disturbed."""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from waldo_amd import build as B  # noqa: E402

name, flags = sys.argv[1], sys.argv[2:]
only = None
if flags and flags[0] == "--only":
    only, flags = flags[1].split(","), flags[2:]
if any(f.startswith("-DWALDO_ABL_") for f in flags):
    flags = flags + ["-DWALDO_TIMING_ONLY_BUILD"]
    if only is not None and "runtime" not in only:
        only = only + ["runtime"]
prod_obj, prod_cflags = B.OBJ, list(B.CFLAGS)
# rejected kernel variants live in tools_dev/dropped/ and compile only into variant builds:
#   -DWALDO_VARIANT_FWD_PIPE   the staged forward with its frame loop software-pipelined (round 5; takes every launch it serves)
#   -DWALDO_VARIANT_FCB_ROWS   the lane-layer backward kernels of the full-resolution flow passes, 9 .. 17 layers (round 5)
B.CFLAGS = B.CFLAGS + flags + ["-I" + os.path.join(os.path.dirname(os.path.abspath(__file__)), "dropped")]
B.OBJ = os.path.join("/tmp", "waldo_variants", name)  # objects stay out of the tree (gpurun ships the tree)
B.LIB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_variants", name + ".so")
os.makedirs(B.OBJ, exist_ok=True)
os.makedirs(os.path.dirname(B.LIB), exist_ok=True)
if only is None:
    print(B.build(force=False, verbose=True))
else:
    objs = []
    for src in B.sources():
        base = os.path.splitext(os.path.basename(src))[0]
        if base in only:
            objs.append(B._compile(src, False)[0])
        else:
            obj = os.path.join(prod_obj, base + ".o")
            assert os.path.exists(obj), f"{obj}: build the product library first"
            objs.append(obj)
    r = subprocess.run([B.HIPCC, "-shared", "-fPIC", f"--offload-arch={B.ARCH}", "-o", B.LIB] + objs,
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    print(B.LIB)
