"""Build recipe for the C-ABI HIP library (gfx950 only).

    python -m waldo_amd.build            # incremental build of waldo_amd/lib/libwaldo_hip.so
    python -m waldo_amd.build --force

hipcc cross-compiles for gfx950 without a GPU.  Object files are cached under
waldo_amd/csrc/_obj (git-ignored); the .so is built in-tree so that it travels with the source
snapshot to the GPU box.
"""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libwaldo_hip.so")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"
CFLAGS = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
          f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function", f"-I{INCLUDE}"]


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _deps():
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs += [os.path.join(INCLUDE, f) for f in os.listdir(INCLUDE) if f.endswith(".h")]
    return sorted(hdrs)


def _stamp(src):
    h = hashlib.sha1()
    for p in [src] + _deps():
        with open(p, "rb") as fh:
            h.update(fh.read())
    h.update(" ".join(CFLAGS).encode())
    return h.hexdigest()


def _compile(src, force):
    os.makedirs(OBJ, exist_ok=True)
    base = os.path.splitext(os.path.basename(src))[0]
    obj = os.path.join(OBJ, base + ".o")
    stamp_file = obj + ".stamp"
    stamp = _stamp(src)
    if not force and os.path.exists(obj) and os.path.exists(stamp_file):
        with open(stamp_file) as fh:
            if fh.read() == stamp:
                return obj, False
    cmd = [HIPCC] + CFLAGS + ["-c", src, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    with open(stamp_file, "w") as fh:
        fh.write(stamp)
    return obj, True


def build(force=False, verbose=True, jobs=8):
    """Compile every .hip under csrc/ for gfx950 and link libwaldo_hip.so.  Returns its path."""
    if not os.path.exists(HIPCC):
        raise RuntimeError(f"hipcc not found at {HIPCC}; cannot build the HIP library")
    srcs = sources()
    with ThreadPoolExecutor(max_workers=jobs) as ex:
        results = list(ex.map(lambda s: _compile(s, force), srcs))
    objs = [o for o, _ in results]
    rebuilt = any(ch for _, ch in results)
    if rebuilt or force or not os.path.exists(LIB):
        os.makedirs(LIBDIR, exist_ok=True)
        cmd = [HIPCC, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
        if verbose:
            print(f"[waldo_amd.build] linked {LIB}")
    elif verbose:
        print(f"[waldo_amd.build] up to date: {LIB}")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
