"""ffwm_amd -- MI355X (gfx950) implementation of the flow-guided feature-warping hot path of
csyxwei/FFWM: hand-written HIP kernels behind the reference's own operator API.

    ffwm_amd.external_function   BlockExtractor / LocalAttnReshape / Resample2d (+Function.apply),
                                 WarpNet, WarpFlipCat
    ffwm_amd.compat              block_extractor_cuda / local_attn_reshape_cuda / resample2d_cuda shims
    ffwm_amd.ops                 tensor-level calls into the C ABI (include/ffwm_hip.h)
    ffwm_amd.build               hipcc build recipe for ffwm_amd/lib/libffwm_hip.so
"""
__version__ = "0.1.0"
