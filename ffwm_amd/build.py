"""Build recipe for libffwm_hip.so (gfx950 only, in-tree).

``python -m ffwm_amd.build`` cross-compiles every ``csrc/*.hip`` with hipcc (no GPU needed) and
links ``ffwm_amd/lib/libffwm_hip.so``.  The .so is git-ignored but travels to the GPU box with
the repo snapshot.
"""
import concurrent.futures
import glob
import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(HERE, "build")
LIB = os.path.join(LIBDIR, "libffwm_hip.so")

# -ffp-contract=off : keep the reference's rounding sequence (no silent FMA fusion) so results
#                     are comparable bit-for-bit with the CPU oracle where the algorithm allows.
# -munsafe-fp-atomics: float/double atomicAdd -> native global_atomic_add_f32/f64 (no CAS loop).
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
               "-munsafe-fp-atomics", "-Wall", "-Wno-unused-function"]


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the HIP extension cannot be built")
    return exe


def _sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _digest():
    h = hashlib.sha256()
    for p in _sources() + sorted(glob.glob(os.path.join(CSRC, "*.hpp"))) + [
            os.path.join(HERE, "..", "include", "ffwm_hip.h")]:
        with open(p, "rb") as f:
            h.update(p.encode() + b"\0" + f.read())
    h.update(" ".join(HIPCC_FLAGS).encode())
    return h.hexdigest()


def _src_digest(src):
    h = hashlib.sha256()
    for p in [src] + sorted(glob.glob(os.path.join(CSRC, "*.hpp"))) + [os.path.join(HERE, "..", "include", "ffwm_hip.h")]:
        with open(p, "rb") as f:
            h.update(f.read())
    h.update(" ".join(HIPCC_FLAGS).encode())
    return h.hexdigest()


def _compile(src):
    obj = os.path.join(OBJDIR, os.path.basename(src)[:-4] + ".o")
    stamp, digest = obj + ".digest", _src_digest(src)
    if os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == digest:
        return obj                      # unchanged translation unit: keep the object
    cmd = [hipcc()] + HIPCC_FLAGS + ["-c", src, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    with open(stamp, "w") as f:
        f.write(digest)
    return obj


def build(force=False, verbose=False):
    os.makedirs(LIBDIR, exist_ok=True)
    os.makedirs(OBJDIR, exist_ok=True)
    stamp = os.path.join(LIBDIR, "libffwm_hip.digest")
    digest = _digest()
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read() == digest:
        return LIB
    srcs = _sources()
    if verbose:
        print("hipcc: compiling %d files for gfx950" % len(srcs))
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        objs = list(ex.map(_compile, srcs))
    cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
    with open(stamp, "w") as f:
        f.write(digest)
    return LIB


EXT_SRC = os.path.join(HERE, "csrc_ext", "ffwm_torch.cpp")
EXT_LIB = os.path.join(LIBDIR, "ffwm_torch_ext.so")


def ext_digest():
    """Build stamp of the torch extension: its source, the C ABI header and the PyTorch version it is compiled against."""
    import torch
    h = hashlib.sha256()
    for p in (EXT_SRC, os.path.join(HERE, "..", "include", "ffwm_hip.h")):
        with open(p, "rb") as f:
            h.update(f.read())
    h.update(torch.__version__.encode())
    return h.hexdigest()


def build_ext(force=False, verbose=False):
    """The C++ autograd bindings (csrc_ext/ffwm_torch.cpp, a pybind11 torch extension linked against libffwm_hip.so):
    g++ against the PyTorch-ROCm headers, in-tree, no GPU needed.  Returns the path of the module."""
    import sysconfig
    import torch
    build()
    stamp = EXT_LIB + ".digest"
    digest = ext_digest()
    if not force and os.path.exists(EXT_LIB) and os.path.exists(stamp) and open(stamp).read() == digest:
        return EXT_LIB
    tdir = os.path.dirname(torch.__file__)
    tinc, tlib = os.path.join(tdir, "include"), os.path.join(tdir, "lib")
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-w", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
           "-DTORCH_API_INCLUDE_EXTENSION_H", "-DTORCH_EXTENSION_NAME=ffwm_torch_ext",
           "-I" + tinc, "-I" + os.path.join(tinc, "torch/csrc/api/include"), "-I" + sysconfig.get_paths()["include"],
           "-I/opt/rocm/include", EXT_SRC, "-o", EXT_LIB, "-L" + tlib, "-L" + LIBDIR, "-Wl,-rpath,$ORIGIN", "-Wl,-rpath," + tlib,
           "-lffwm_hip", "-ltorch", "-ltorch_cpu", "-ltorch_hip", "-lc10", "-lc10_hip", "-ltorch_python"]
    if verbose:
        print("g++: compiling the torch extension (csrc_ext/ffwm_torch.cpp)")
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("building ffwm_torch_ext failed:\n%s\n%s" % (r.stdout[-3000:], r.stderr[-3000:]))
    with open(stamp, "w") as f:
        f.write(digest)
    return EXT_LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    print(build_ext(force="--force" in sys.argv, verbose=True))
