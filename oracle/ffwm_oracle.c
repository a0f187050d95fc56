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
 *   block_extractor, local_attn_reshape : pinned by the reference's own gradcheck recipes and
 *       range(9) known answer (cuda/[op]/test_[op].py) + exact PyTorch identities
 *       (pixel_shuffle, shifted grid_sample(border, align_corners=True), unfold).
 *   warp (grid_sample)                   : pinned at run time against torch's CPU grid_sample.
 *   resample2d                           : PARITY UNPINNED by the reference (it ships no test and
 *       the call path is dead code); pinned only by autograd-of-forward
 *       self-consistency and hand-derived small cases.
 *
 * The reference CUDA sources cannot be compiled here (CUDA-only ATen headers, no nvcc):
 * there is no oracle/_ref build.
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
