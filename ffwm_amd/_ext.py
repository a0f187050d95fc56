"""The C++ autograd bindings (ffwm_amd/csrc_ext/ffwm_torch.cpp -> ffwm_amd/lib/ffwm_torch_ext.so).

Same kernels, same C ABI as the ctypes path (`_lib`, `ops`, the Python autograd Functions); only the host-side cost per
call differs (a few microseconds instead of 14-35).  `get()` returns the module, or None when it was not built (the
callers then use the ctypes Functions); `FFWM_TORCH_EXT=0` switches it off for A/B measurements."""
import importlib.util
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
EXT_PATH = os.path.join(_HERE, "lib", "ffwm_torch_ext.so")
_mod = None
_tried = False


def get():
    global _mod, _tried
    if _tried:
        return _mod
    _tried = True
    if os.environ.get("FFWM_TORCH_EXT", "1") == "0" or not os.path.exists(EXT_PATH):
        return None
    import torch  # noqa: F401  (libtorch must be loaded first)
    # a module built from another version of the source / the C ABI header / PyTorch would be called with mismatched signatures:
    # it is only loaded when its build stamp matches what `build.build_ext` would produce now (else: the ctypes Functions)
    from . import build
    try:
        with open(EXT_PATH + ".digest") as f:
            stamp = f.read()
    except OSError:
        stamp = None
    if stamp != build.ext_digest():
        import warnings
        warnings.warn("ffwm_torch_ext.so is stale (source, header or PyTorch changed since it was built): using the ctypes bindings; "
                      "rebuild with `python -m ffwm_amd.build`")
        return None
    from . import _lib
    _lib.load()                               # libffwm_hip.so: fail loudly if the kernels themselves are missing
    spec = importlib.util.spec_from_file_location("ffwm_torch_ext", EXT_PATH)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    _mod = mod
    return _mod
