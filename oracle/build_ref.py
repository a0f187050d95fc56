"""TEST INFRASTRUCTURE ONLY -- builds the REFERENCE's three native extensions for gfx950 into oracle/_ref/.

    python oracle/build_ref.py            (needs /root/reference; a no-op with a message where it is absent)

What it does, and what it does not:

* The sources are compiled from where they lie under /root/reference/cuda/{resample2d_package,block_extractor,
  local_attn_reshape} (`*_kernel.cu` + `*_cuda.cc`, the same two files per extension the reference's own
  setup.py names).  Nothing is copied into this repository: intermediates live in a temporary directory that is
  removed afterwards; the only outputs are oracle/_ref/{resample2d,block_extractor,local_attn_reshape}_cuda.so
  (git-ignored, shipped to the GPU box like every other built .so).
* The `.cu` files include <ATen/cuda/CUDAContext.h>, which needs CUDA's cuda_runtime_api.h -- absent here.  No
  stand-in header is written.  Instead the files go through PyTorch-ROCm's OWN source translator
  (torch.utils.hipify, the step `torch.utils.cpp_extension.CUDAExtension` performs for every CUDA extension built
  against a ROCm wheel -- i.e. what `python setup.py install` of the reference does on this image): it rewrites
  the include to <ATen/hip/HIPContext.h>, `<<<...>>>` to hipLaunchKernelGGL and `at::cuda::getCurrentCUDAStream`
  to its HIP twin.  The `__global__` kernel bodies -- all of the arithmetic -- are untouched by it.
* ONE host-side token is patched in the translated intermediate: `AT_DISPATCH_FLOATING_TYPES(x.type(), ...)` ->
  `x.scalar_type()`.  The reference pins PyTorch 1.5 (README.md:14); the DeprecatedTypeProperties overload of the
  dispatch macro was removed from PyTorch 2.x.  This selects the same template instantiation and does not touch
  device code.

So oracle/_ref runs the reference's kernels themselves, compiled by hipcc for gfx950, on the MI355X.  It is used
by tests/golden/make_ref_ops_golden.py (golden vectors for the three ops, committed under tests/golden/) and by
the `-m gpu` parity tests as a live second checker; the product (ffwm_amd/) never loads it.
"""
import glob
import os
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_ref")
REF_CUDA = "/root/reference/cuda"
EXTS = (("resample2d_package", "resample2d"), ("block_extractor", "block_extractor"),
        ("local_attn_reshape", "local_attn_reshape"))


def so_path(name):
    return os.path.join(OUT, name + "_cuda.so")


def available():
    return all(os.path.exists(so_path(n)) for _, n in EXTS)


def build(force=False, verbose=False):
    """Returns True when oracle/_ref holds the three extensions afterwards."""
    if not os.path.isdir(REF_CUDA):
        if verbose:
            print("oracle/_ref: /root/reference absent -- using the prebuilt files" if available()
                  else "oracle/_ref: /root/reference absent and nothing prebuilt")
        return available()
    srcs = [p for d, _ in EXTS for p in glob.glob(os.path.join(REF_CUDA, d, "*.cu*")) + glob.glob(os.path.join(REF_CUDA, d, "*.cc"))]
    newest = max(os.path.getmtime(p) for p in srcs + [os.path.abspath(__file__)])
    if not force and available() and all(os.path.getmtime(so_path(n)) >= newest for _, n in EXTS):
        return True
    import torch
    from torch.utils.hipify import hipify_python
    tinc = os.path.join(os.path.dirname(torch.__file__), "include")
    tlib = os.path.join(os.path.dirname(torch.__file__), "lib")
    import sysconfig
    pyinc = sysconfig.get_paths()["include"]
    os.makedirs(OUT, exist_ok=True)
    tmp = tempfile.mkdtemp(prefix="ffwm_ref_build_")
    try:
        devnull = open(os.devnull, "w")
        stdout = sys.stdout
        try:
            sys.stdout = devnull if not verbose else stdout
            hipify_python.hipify(project_directory=REF_CUDA, output_directory=os.path.join(tmp, "hip"),
                                 includes=[REF_CUDA + "/*"], show_detailed=False, is_pytorch_extension=True)
        finally:
            sys.stdout = stdout
        common = ["-O2", "-std=c++17", "-fPIC", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1", "-DHIPBLAS_V2",
                  "-DTORCH_API_INCLUDE_EXTENSION_H", "-I" + tinc, "-I" + os.path.join(tinc, "torch/csrc/api/include"),
                  "-I" + pyinc, "-I/opt/rocm/include", "-w"]
        for d, name in EXTS:
            hip = os.path.join(tmp, "hip", d, name + "_kernel.hip")
            text = open(hip).read()
            assert ".type()" in text
            open(hip, "w").write(text.replace(".type()", ".scalar_type()"))   # the one host-side patch (see above)
            mod = "-DTORCH_EXTENSION_NAME=%s_cuda" % name
            ko, co = os.path.join(tmp, name + "_k.o"), os.path.join(tmp, name + "_c.o")
            subprocess.check_call(["hipcc", "--offload-arch=gfx950", mod] + common + ["-c", hip, "-o", ko])
            # the pybind wrapper is compiled from where it lies; it includes "<name>_kernel.cuh" from its own directory
            subprocess.check_call(["g++", mod] + common + ["-c", os.path.join(REF_CUDA, d, name + "_cuda.cc"), "-o", co])
            subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-o", so_path(name), ko, co, "-L" + tlib,
                                   "-Wl,-rpath," + tlib, "-ltorch", "-ltorch_cpu", "-ltorch_hip", "-lc10", "-lc10_hip",
                                   "-ltorch_python"])
            if verbose:
                print("built", so_path(name))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return available()


def load():
    """Import the three extension modules from oracle/_ref (torch must be imported first).  Returns a dict
    name -> module, or None when they are not built."""
    if not available():
        return None
    import importlib.util
    import torch  # noqa: F401
    mods = {}
    for _, name in EXTS:
        spec = importlib.util.spec_from_file_location(name + "_cuda", so_path(name))
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
        mods[name] = m
    return mods


if __name__ == "__main__":
    ok = build(force="--force" in sys.argv, verbose=True)
    print("oracle/_ref:", "ready" if ok else "not built")
