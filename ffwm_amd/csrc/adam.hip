// adam.hip -- one Adam step over a FLAT parameter range.
//
// Reference: the three torch.optim.Adam optimizers of FFWMModel (models/ffwm_model.py:46-49,151-160) and
// FlowNetModel (models/flownet_model.py:33,74-78): Adam without weight decay / amsgrad, betas (0.5, 0.999).
// PyTorch's fused multi-tensor kernel walks ~600 separate parameter tensors per step (2.2 ms for the 108 M
// trainable parameters of the train step, 1.4 TB/s); here the parameters, their gradients (the data-parallel
// reducer's flat buckets, ffwm_amd/dp.py) and both moment buffers are four contiguous arrays, and a step is one
// perfectly coalesced streaming pass: 4 reads + 3 writes of 4 bytes per parameter, HBM-bound.
//
// Arithmetic = torch/aten/src/ATen/native/cuda/fused_adam_utils.cuh (fp32 opmath):
//     m = lerp(m, g, 1 - beta1);  v = beta2 v + (1 - beta2) g g
//     p -= (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps),   bc_i = 1 - beta_i^step
#include "common.hpp"

namespace ffwm {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, float beta1, float beta2, float eps,
                                         float step_size, float bc2_sqrt) {
    m = m + (g - m) * (1.f - beta1);
    v = beta2 * v + (1.f - beta2) * g * g;
    const float denom = sqrtf(v) / bc2_sqrt + eps;
    p = p - step_size * m / denom;
}

__global__ void __launch_bounds__(kBlock)
adam_flat_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, int64_t n,
                 float beta1, float beta2, float eps, float step_size, float bc2_sqrt) {
    const int64_t n4 = n >> 2;
    const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < n4; i += stride) {
        const f32x4 p4 = reinterpret_cast<f32x4*>(p)[i];
        const f32x4 g4 = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(g) + i);
        const f32x4 m4 = reinterpret_cast<f32x4*>(m)[i], v4 = reinterpret_cast<f32x4*>(v)[i];
        float pa[4] = {p4.x, p4.y, p4.z, p4.w}, ga[4] = {g4.x, g4.y, g4.z, g4.w};
        float ma[4] = {m4.x, m4.y, m4.z, m4.w}, va[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) adam_one(pa[q], ga[q], ma[q], va[q], beta1, beta2, eps, step_size, bc2_sqrt);
        const f32x4 pp = {pa[0], pa[1], pa[2], pa[3]}, mm = {ma[0], ma[1], ma[2], ma[3]}, vv = {va[0], va[1], va[2], va[3]};
        reinterpret_cast<f32x4*>(p)[i] = pp;
        reinterpret_cast<f32x4*>(m)[i] = mm;
        reinterpret_cast<f32x4*>(v)[i] = vv;
    }
    // tail (n not a multiple of 4)
    const int64_t t = (n4 << 2) + static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (t < n) adam_one(p[t], g[t], m[t], v[t], beta1, beta2, eps, step_size, bc2_sqrt);
}

// The step counter on the DEVICE (a step that sits inside a captured hipGraph: a replay runs no host code, so the bias corrections
// cannot come from a host integer).  state[0] = steps taken so far, state[1] = lr / bc1, state[2] = sqrt(bc2) of the current step,
// state[3] = learning-rate override (>= 0: replaces the kernel argument -- 0.0 freezes the weights; negative: none).
__global__ void adam_advance_kernel(double* __restrict__ state, double lr, double beta1, double beta2) {
    const double step = state[0] + 1.0;
    if (state[3] >= 0.0) lr = state[3];          // the host's current learning rate (a schedule under hipGraph replay)
    state[0] = step;
    state[1] = lr / (1.0 - pow(beta1, step));
    state[2] = sqrt(1.0 - pow(beta2, step));
}

__global__ void __launch_bounds__(kBlock)
adam_flat_dev_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, int64_t n,
                     float beta1, float beta2, float eps, const double* __restrict__ state) {
    const float step_size = static_cast<float>(state[1]), bc2_sqrt = static_cast<float>(state[2]);
    const int64_t n4 = n >> 2;
    const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < n4; i += stride) {
        const f32x4 p4 = reinterpret_cast<f32x4*>(p)[i];
        const f32x4 g4 = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(g) + i);
        const f32x4 m4 = reinterpret_cast<f32x4*>(m)[i], v4 = reinterpret_cast<f32x4*>(v)[i];
        float pa[4] = {p4.x, p4.y, p4.z, p4.w}, ga[4] = {g4.x, g4.y, g4.z, g4.w};
        float ma[4] = {m4.x, m4.y, m4.z, m4.w}, va[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) adam_one(pa[q], ga[q], ma[q], va[q], beta1, beta2, eps, step_size, bc2_sqrt);
        const f32x4 pp = {pa[0], pa[1], pa[2], pa[3]}, mm = {ma[0], ma[1], ma[2], ma[3]}, vv = {va[0], va[1], va[2], va[3]};
        reinterpret_cast<f32x4*>(p)[i] = pp;
        reinterpret_cast<f32x4*>(m)[i] = mm;
        reinterpret_cast<f32x4*>(v)[i] = vv;
    }
    const int64_t t = (n4 << 2) + static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (t < n) adam_one(p[t], g[t], m[t], v[t], beta1, beta2, eps, step_size, bc2_sqrt);
}

}  // namespace
}  // namespace ffwm

using namespace ffwm;

// The same step with the step counter in device memory: state = 4 doubles (steps taken so far -- start it at 0 --, two
// scratch values, the learning-rate override).  Two launches (a one-thread advance of the counter, the streaming pass); safe to capture in a hipGraph.
extern "C" int ffwm_adam_step_device(void* params, const void* grads, void* exp_avg, void* exp_avg_sq, int64_t n, double lr,
                                     double beta1, double beta2, double eps, void* state, int dtype, void* stream) {
    const char* fn = "ffwm_adam_step_device";
    FFWM_REQUIRE(dtype == FFWM_F32, FFWM_ERR_DTYPE, "%s: float32 only", fn);
    FFWM_REQUIRE(params && grads && exp_avg && exp_avg_sq && state, FFWM_ERR_ARG, "%s: NULL pointer", fn);
    FFWM_REQUIRE(n > 0, FFWM_ERR_ARG, "%s: need n > 0", fn);
    FFWM_REQUIRE((reinterpret_cast<uintptr_t>(params) | reinterpret_cast<uintptr_t>(grads) | reinterpret_cast<uintptr_t>(exp_avg) |
                  reinterpret_cast<uintptr_t>(exp_avg_sq)) % 16 == 0,
                 FFWM_ERR_ARG, "%s: the four arrays must be 16-byte aligned", fn);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int64_t n4 = (n + 3) / 4;
    int64_t blocks = (n4 + kBlock - 1) / kBlock;
    if (blocks > 256 * 16) blocks = 256 * 16;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(adam_advance_kernel, dim3(1), dim3(1), 0, st, static_cast<double*>(state), lr, beta1, beta2);
    LaunchScope ls("adam_flat", st, 28.0 * static_cast<double>(n));
    hipLaunchKernelGGL(adam_flat_dev_kernel, dim3(static_cast<unsigned>(blocks)), dim3(kBlock), 0, st, (float*)params, (const float*)grads,
                       (float*)exp_avg, (float*)exp_avg_sq, n, (float)beta1, (float)beta2, (float)eps, static_cast<const double*>(state));
    return check_launch(fn);
}

extern "C" int ffwm_adam_step(void* params, const void* grads, void* exp_avg, void* exp_avg_sq, int64_t n, double lr,
                              double beta1, double beta2, double eps, int64_t step, int dtype, void* stream) {
    const char* fn = "ffwm_adam_step";
    FFWM_REQUIRE(dtype == FFWM_F32, FFWM_ERR_DTYPE, "%s: float32 only", fn);
    FFWM_REQUIRE(params && grads && exp_avg && exp_avg_sq, FFWM_ERR_ARG, "%s: NULL pointer", fn);
    FFWM_REQUIRE(n > 0 && step >= 1, FFWM_ERR_ARG, "%s: need n > 0 and step >= 1 (n=%lld step=%lld)", fn, (long long)n, (long long)step);
    FFWM_REQUIRE((reinterpret_cast<uintptr_t>(params) | reinterpret_cast<uintptr_t>(grads) | reinterpret_cast<uintptr_t>(exp_avg) |
                  reinterpret_cast<uintptr_t>(exp_avg_sq)) % 16 == 0,
                 FFWM_ERR_ARG, "%s: the four arrays must be 16-byte aligned", fn);
    // bias corrections in double on the host, like torch's _fused_adam (step is a host integer here)
    const double bc1 = 1.0 - pow(beta1, static_cast<double>(step));
    const double bc2 = 1.0 - pow(beta2, static_cast<double>(step));
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int64_t n4 = (n + 3) / 4;
    int64_t blocks = (n4 + kBlock - 1) / kBlock;
    if (blocks > 256 * 16) blocks = 256 * 16;      // grid-stride: 16 workgroups per CU keep the loads in flight
    if (blocks < 1) blocks = 1;
    LaunchScope ls("adam_flat", st, 28.0 * static_cast<double>(n));
    hipLaunchKernelGGL(adam_flat_kernel, dim3(static_cast<unsigned>(blocks)), dim3(kBlock), 0, st, (float*)params, (const float*)grads,
                       (float*)exp_avg, (float*)exp_avg_sq, n, (float)beta1, (float)beta2, (float)eps, (float)(lr / bc1),
                       (float)sqrt(bc2));
    return check_launch(fn);
}
