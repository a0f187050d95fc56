"""CPU-side checks of the drop-in boundary: the C-ABI library builds/loads and exports every symbol
include/ffwm_hip.h declares (no compute calls -- there is no GPU here), argument validation that
happens before any launch, and the isolation of the product path from the oracle."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def hiplib():
    from ffwm_amd import build, _lib
    build.build()
    return _lib.load()


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "ffwm_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ffwm_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported(hiplib):
    names = _declared_symbols()
    assert len(names) >= 15
    for n in names:
        assert hasattr(hiplib, n), "libffwm_hip.so does not export %s" % n


def test_binding_table_matches_header():
    from ffwm_amd import _lib
    assert _lib.EXPORTS == _declared_symbols()


def test_abi_version(hiplib):
    from ffwm_amd import _lib
    assert hiplib.ffwm_abi_version() == _lib.ABI_VERSION == 5          # round 6: ffwm_conv2d_forward takes a workspace (split reduction without atomics) and a second destination


def test_argument_errors_are_reported_before_launch(hiplib):
    # NULL pointers / bad dtype / bad sizes must fail cleanly without touching a device
    rc = hiplib.ffwm_block_extractor_forward(None, None, None, 1, 1, 4, 4, 4, 4, 3, 0, None)
    assert rc == -1 and b"NULL" in hiplib.ffwm_last_error()
    buf = (ctypes.c_float * 16)()
    p = ctypes.cast(buf, ctypes.c_void_p)
    rc = hiplib.ffwm_block_extractor_forward(p, p, p, 1, 1, 4, 4, 4, 4, 3, 7, None)
    assert rc == -2 and b"dtype" in hiplib.ffwm_last_error()
    rc = hiplib.ffwm_block_extractor_forward(p, p, p, 1, 1, 0, 4, 4, 4, 3, 0, None)
    assert rc == -1
    rc = hiplib.ffwm_resample2d_forward(p, p, p, 1, 1, 4, 4, 4, 4, 0, 1, 0, None)
    assert rc == -1 and b"kernel_size" in hiplib.ffwm_last_error()
    rc = hiplib.ffwm_warp_forward(p, p, p, 1, 1, 1 << 15, 1 << 15, 4, 4, 0, 0, None)
    assert rc == -3
    assert hiplib.ffwm_set_option(b"no_such_key", 1) == -1
    # the convolution entry points: argument checks come before any launch
    rc = hiplib.ffwm_conv3x3_winograd_forward(None, None, None, None, None, 1, 8, 4, 4, 8, 0, 0, 0.0, 0, None)
    assert rc == -1 and b"NULL" in hiplib.ffwm_last_error()
    rc = hiplib.ffwm_conv3x3_winograd_forward(p, p, None, p, p, 1, 8, 4, 4, 8, 5, 0, 0.0, 0, None)
    assert rc == -1 and b"data_gradient" in hiplib.ffwm_last_error()
    rc = hiplib.ffwm_conv3x3_winograd_forward(p, p, None, p, p, 1, 8, 4, 4, 8, 0, 0, 0.0, 1, None)
    assert rc == -2
    # transformed weights: 16 positions x 64-channel tiles x 8-channel chunks, fp32
    assert hiplib.ffwm_conv3x3_winograd_workspace_bytes(195, 195) == 4 * 25 * 16 * 2 * 64 * 4 * 4
    assert hiplib.ffwm_conv3x3_winograd_workspace_bytes(0, 5) == 0
    rc = hiplib.ffwm_conv2d_wgrad(None, None, None, 1, 8, 4, 4, 8, 4, 4, 3, 1, 1, 0, None)
    assert rc == -1
    rc = hiplib.ffwm_conv2d_wgrad_tiled(None, None, None, None, 1, 8, 4, 4, 8, 4, 4, 3, 1, 1, 0, 0, None)
    assert rc == -1 and b"NULL" in hiplib.ffwm_last_error()
    rc = hiplib.ffwm_conv2d_wgrad_tiled(p, p, p, None, 1, 8, 3, 3, 8, 3, 3, 3, 1, 1, 0, 0, None)        # 9 pixels per plane: not a multiple of 4
    assert rc == -1 and b"multiple of 4" in hiplib.ffwm_last_error()
    rc = hiplib.ffwm_l1_multi(None, 0, None, None, 1, 0, None)
    assert rc == -1


def test_cpu_tensors_are_refused_like_the_reference():
    # /root/reference/models/external_function.py:37-38,84-85 raise NotImplementedError on CPU
    from ffwm_amd import external_function as E
    with pytest.raises(NotImplementedError):
        E.BlockExtractor(3)(torch.rand(1, 2, 5, 5), torch.zeros(1, 2, 5, 5))
    with pytest.raises(NotImplementedError):
        E.LocalAttnReshape()(torch.rand(1, 9, 5, 5), 3)
    with pytest.raises(NotImplementedError):
        E.Resample2d(4, 1, sigma=2)(torch.rand(1, 2, 5, 5), torch.zeros(1, 2, 5, 5))
    with pytest.raises(NotImplementedError):
        E.WarpNet()(torch.rand(1, 2, 5, 5), torch.zeros(1, 2, 5, 5))
    with pytest.raises(AssertionError):
        E.LocalAttnReshapeFunction.apply(torch.rand(1, 8, 5, 5), 3)      # C != k*k
    with pytest.raises(AssertionError):
        E.BlockExtractorFunction.apply(torch.rand(1, 2, 5, 5), torch.zeros(1, 3, 5, 5), 3)   # df != 2


def test_compat_modules_expose_the_reference_pybind_names():
    import sys
    from ffwm_amd import compat
    saved = {k: sys.modules.pop(k) for k in list(sys.modules)
             if k in ("block_extractor_cuda", "local_attn_reshape_cuda", "resample2d_cuda")}
    try:
        compat.install()
        import block_extractor_cuda, local_attn_reshape_cuda, resample2d_cuda   # noqa: E401
        for m in (block_extractor_cuda, local_attn_reshape_cuda, resample2d_cuda):
            assert callable(m.forward) and callable(m.backward)
    finally:
        for k in ("block_extractor_cuda", "local_attn_reshape_cuda", "resample2d_cuda"):
            sys.modules.pop(k, None)
        sys.modules.update(saved)


def test_product_path_never_touches_the_oracle():
    """The oracle is test infrastructure: nothing under ffwm_amd/, and neither the timed legs of
    bench.py nor the kernels, may import or link it."""
    pkg = os.path.join(ROOT, "ffwm_amd")
    offenders = []
    for dirpath, _, files in os.walk(pkg):
        if os.path.basename(dirpath) in ("build", "lib", "__pycache__"):
            continue
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                if re.search(r"^\s*(import|from)\s+oracle\b", text, re.M) or "libffwm_oracle" in text \
                        or "ffwm_oracle" in text:
                    offenders.append(os.path.join(dirpath, f))
    assert not offenders, offenders
