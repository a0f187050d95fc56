/*
 * ffwm_oracle.c -- TEST INFRASTRUCTURE ONLY (never shipped, never on the product path).
 *
 * CPU restatement of the reference CUDA kernels of csyxwei/FFWM's flow-warp hot path:
 *   cuda/block_extractor/block_extractor_kernel.cu:21-170
 *   cuda/local_attn_reshape/local_attn_reshape_kernel.cu:21-108
 *   cuda/resample2d_package/resample2d_kernel.cu:21-330
 *   models/base_networks.py:168-173 (WarpNet = F.grid_sample, PyTorch/ATen arithmetic)
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library; the HIP product path (ffwm_amd/) never does.
 *
 * Parity pin status (see DESIGN.md "Oracle"):
 *   block_extractor, local_attn_reshape, resample2d : PINNED TO THE REFERENCE'S OWN KERNELS.  oracle/build_ref.py
 *       compiles cuda/<op>/<op>_kernel.cu + <op>_cuda.cc from where they lie under /root/reference for gfx950
 *       (PyTorch-ROCm's hipify step + hipcc, no stand-in headers; see that file for the one host-side token it
 *       patches) into oracle/_ref/; tests/golden/make_ref_ops_golden.py ran them on an MI355X and
 *       tests/golden/reference_ops_gfx950.pt holds what they returned; tests/test_oracle_ref_golden.py checks every
 *       function here against those vectors (forward <= 1e-6 fp32 / 1e-13 fp64).  Before that round they were pinned
 *       by the reference's gradcheck recipes and range(9) known answer (cuda/[op]/test_[op].py) and by exact PyTorch
 *       identities (pixel_shuffle, shifted grid_sample(border, align_corners=True), unfold) -- those tests remain.
 *   warp (grid_sample)                   : pinned against torch's CPU grid_sample and the reference WarpNet fixture.
 *
 * Build: make -C oracle   ->  oracle/libffwm_oracle.so
 */
#include <math.h>
#include <stdint.h>

#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)

#define real float
#define FN(name) CAT(name, _f32)
#define FLOOR(v) floorf(v)
#define FMA(a, b, c) fmaf(a, b, c)
#include "ffwm_oracle_impl.inc"
#undef real
#undef FN
#undef FLOOR
#undef FMA

#define real double
#define FN(name) CAT(name, _f64)
#define FLOOR(v) floor(v)
#define FMA(a, b, c) fma(a, b, c)
#include "ffwm_oracle_impl.inc"
#undef real
#undef FN
#undef FLOOR
#undef FMA

int oracle_abi_version(void) { return 1; }
