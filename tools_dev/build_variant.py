"""dev: build a variant of the library with extra compiler flags for A/B timing.

    python tools_dev/build_variant.py NAME -DWALDO_STAGE_AHEAD=3 ...   ->  waldo_amd/lib/abl/NAME.so

Objects go to their own directory, so the product build is not disturbed."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from waldo_amd import build as B  # noqa: E402

name, flags = sys.argv[1], sys.argv[2:]
B.CFLAGS = B.CFLAGS + flags
B.OBJ = os.path.join("/tmp", "waldo_variants", name)  # objects stay out of the tree (gpurun ships the tree)
B.LIB = os.path.join(B.LIBDIR, "abl", name + ".so")
os.makedirs(B.OBJ, exist_ok=True)
os.makedirs(os.path.dirname(B.LIB), exist_ok=True)
print(B.build(force=False, verbose=True))
