// residual.hip -- the element-wise tails of netG's residual blocks and of its warp-attention gate as ONE pass each (gfx950).
//
// Reference: models/base_networks.py:207-233 (ResidualBlock.forward: activ(blocks(x) + input(x)), 13 blocks in netG: e0-e3, dres0-2
// x 2, att0-2) and :326-333 (FFWM.forward: skip = cat(w, flip(w)); att = att_i(skip); skip = skip * att -- att_i ends in a
// ResidualBlock with a sigmoid).  In PyTorch the tail of a block is an add and an activation (2 launches, 5 tensor passes
// forward; activation-backward 3 passes), and the gate adds a multiply (8 passes forward, 9 backward) over the largest
// activations of the generator (the 128 x 128 skip tensor is 67 MB at batch 8).
//
//   ffwm_add_act_forward / backward      y = act(a + b), act = LeakyReLU(slope) or sigmoid;  dz = g * act'(z) from y alone
//                                        (3 passes forward, 3 backward; d(a) = d(b) = dz, one tensor)
//   ffwm_sigmoid_gate_forward / backward att = sigmoid(a + b), y = x * att (5 passes);  dz = g x att (1 - att), dx = g att (5 passes)
//
// float4 streams, grid-stride, HBM-bound.  The arithmetic is ATen's: a + b rounded once, LeakyReLU z > 0 ? z : z * slope,
// sigmoid 1 / (1 + exp(-z)) in fp32.
#include "common.hpp"

namespace ffwm {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float sigmoid_f(float z) { return 1.f / (1.f + expf(-z)); }

template <int ACT>     // 1 = LeakyReLU, 3 = sigmoid
__device__ __forceinline__ float act_f(float z, float slope) {
    if constexpr (ACT == 1) return z > 0.f ? z : z * slope;
    else return sigmoid_f(z);
}
template <int ACT>     // g * act'(z) from y = act(z), in ATen's order of operations (LeakyReLU: slope > 0, so y and z have the same sign)
__device__ __forceinline__ float dact_f(float g, float y, float slope) {
    if constexpr (ACT == 1) return y > 0.f ? g : g * slope;
    else return (g * (1.f - y)) * y;
}

constexpr int kVecBlocks = 256 * 8;      // grid-stride: 8 workgroups per CU keep the loads in flight

template <int ACT>
__global__ void __launch_bounds__(kBlock)
add_act_fwd_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ y, int64_t n, float slope) {
    const int64_t n4 = n >> 2, stride = static_cast<int64_t>(gridDim.x) * kBlock;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < n4; i += stride) {
        const f32x4 av = reinterpret_cast<const f32x4*>(a)[i], bv = reinterpret_cast<const f32x4*>(b)[i];
        f32x4 r;
        r.x = act_f<ACT>(av.x + bv.x, slope); r.y = act_f<ACT>(av.y + bv.y, slope);
        r.z = act_f<ACT>(av.z + bv.z, slope); r.w = act_f<ACT>(av.w + bv.w, slope);
        reinterpret_cast<f32x4*>(y)[i] = r;
    }
    const int64_t t = (n4 << 2) + static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (t < n) y[t] = act_f<ACT>(a[t] + b[t], slope);
}

template <int ACT>
__global__ void __launch_bounds__(kBlock)
add_act_bwd_kernel(const float* __restrict__ y, const float* __restrict__ g, float* __restrict__ dz, int64_t n, float slope) {
    const int64_t n4 = n >> 2, stride = static_cast<int64_t>(gridDim.x) * kBlock;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < n4; i += stride) {
        const f32x4 yv = reinterpret_cast<const f32x4*>(y)[i], gv = reinterpret_cast<const f32x4*>(g)[i];
        f32x4 r;
        r.x = dact_f<ACT>(gv.x, yv.x, slope); r.y = dact_f<ACT>(gv.y, yv.y, slope);
        r.z = dact_f<ACT>(gv.z, yv.z, slope); r.w = dact_f<ACT>(gv.w, yv.w, slope);
        reinterpret_cast<f32x4*>(dz)[i] = r;
    }
    const int64_t t = (n4 << 2) + static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (t < n) dz[t] = dact_f<ACT>(g[t], y[t], slope);
}

__global__ void __launch_bounds__(kBlock)
gate_fwd_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ x, float* __restrict__ att,
                float* __restrict__ y, int64_t n) {
    const int64_t n4 = n >> 2, stride = static_cast<int64_t>(gridDim.x) * kBlock;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < n4; i += stride) {
        const f32x4 av = reinterpret_cast<const f32x4*>(a)[i], bv = reinterpret_cast<const f32x4*>(b)[i];
        const f32x4 xv = reinterpret_cast<const f32x4*>(x)[i];
        f32x4 s, r;
        s.x = sigmoid_f(av.x + bv.x); s.y = sigmoid_f(av.y + bv.y); s.z = sigmoid_f(av.z + bv.z); s.w = sigmoid_f(av.w + bv.w);
        r.x = xv.x * s.x; r.y = xv.y * s.y; r.z = xv.z * s.z; r.w = xv.w * s.w;
        reinterpret_cast<f32x4*>(att)[i] = s;
        reinterpret_cast<f32x4*>(y)[i] = r;
    }
    const int64_t t = (n4 << 2) + static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (t < n) {
        const float s = sigmoid_f(a[t] + b[t]);
        att[t] = s;
        y[t] = x[t] * s;
    }
}

__global__ void __launch_bounds__(kBlock)
gate_bwd_kernel(const float* __restrict__ x, const float* __restrict__ att, const float* __restrict__ g, float* __restrict__ dz,
                float* __restrict__ dx, int64_t n) {
    const int64_t n4 = n >> 2, stride = static_cast<int64_t>(gridDim.x) * kBlock;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < n4; i += stride) {
        const f32x4 xv = reinterpret_cast<const f32x4*>(x)[i], sv = reinterpret_cast<const f32x4*>(att)[i];
        const f32x4 gv = reinterpret_cast<const f32x4*>(g)[i];
        f32x4 z, d;
        // ATen's order: d(att) = g * x, then sigmoid_backward = (d(att) * (1 - att)) * att
        z.x = ((gv.x * xv.x) * (1.f - sv.x)) * sv.x; z.y = ((gv.y * xv.y) * (1.f - sv.y)) * sv.y;
        z.z = ((gv.z * xv.z) * (1.f - sv.z)) * sv.z; z.w = ((gv.w * xv.w) * (1.f - sv.w)) * sv.w;
        d.x = gv.x * sv.x; d.y = gv.y * sv.y; d.z = gv.z * sv.z; d.w = gv.w * sv.w;
        reinterpret_cast<f32x4*>(dz)[i] = z;
        reinterpret_cast<f32x4*>(dx)[i] = d;
    }
    const int64_t t = (n4 << 2) + static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (t < n) {
        dz[t] = ((g[t] * x[t]) * (1.f - att[t])) * att[t];
        dx[t] = g[t] * att[t];
    }
}

unsigned vec_grid(int64_t n) {
    int64_t blocks = ((n + 3) / 4 + kBlock - 1) / kBlock;
    if (blocks > kVecBlocks) blocks = kVecBlocks;
    if (blocks < 1) blocks = 1;
    return static_cast<unsigned>(blocks);
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace
}  // namespace ffwm

using namespace ffwm;

// act: 1 = LeakyReLU(negative_slope), 3 = sigmoid.  y may alias a or b.
extern "C" int ffwm_add_act_forward(const void* a, const void* b, void* y, int64_t n, int act, double negative_slope, int dtype,
                                    void* stream) {
    const char* fn = "ffwm_add_act_forward";
    FFWM_REQUIRE(dtype == FFWM_F32, FFWM_ERR_DTYPE, "%s: float32 only", fn);
    FFWM_REQUIRE(a && b && y && n > 0, FFWM_ERR_ARG, "%s: NULL pointer or empty tensor", fn);
    FFWM_REQUIRE(act == 1 || act == 3, FFWM_ERR_ARG, "%s: act must be 1 (LeakyReLU) or 3 (sigmoid), got %d", fn, act);
    FFWM_REQUIRE(act != 1 || negative_slope > 0, FFWM_ERR_ARG, "%s: the backward reads the sign of y: negative_slope must be > 0", fn);
    FFWM_REQUIRE(aligned16(a) && aligned16(b) && aligned16(y), FFWM_ERR_ARG, "%s: tensors must be 16-byte aligned", fn);
    hipStream_t st = static_cast<hipStream_t>(stream);
    LaunchScope ls("add_act_fwd", st, 12.0 * static_cast<double>(n));
    if (act == 1)
        hipLaunchKernelGGL((add_act_fwd_kernel<1>), dim3(vec_grid(n)), dim3(kBlock), 0, st, (const float*)a, (const float*)b, (float*)y, n,
                           (float)negative_slope);
    else
        hipLaunchKernelGGL((add_act_fwd_kernel<3>), dim3(vec_grid(n)), dim3(kBlock), 0, st, (const float*)a, (const float*)b, (float*)y, n, 0.f);
    return check_launch(fn);
}

// grad_z = grad_y * act'(z), from y = act(z) alone; grad_z may alias grad_y.
extern "C" int ffwm_add_act_backward(const void* y, const void* grad_y, void* grad_z, int64_t n, int act, double negative_slope,
                                     int dtype, void* stream) {
    const char* fn = "ffwm_add_act_backward";
    FFWM_REQUIRE(dtype == FFWM_F32, FFWM_ERR_DTYPE, "%s: float32 only", fn);
    FFWM_REQUIRE(y && grad_y && grad_z && n > 0, FFWM_ERR_ARG, "%s: NULL pointer or empty tensor", fn);
    FFWM_REQUIRE(act == 1 || act == 3, FFWM_ERR_ARG, "%s: act must be 1 (LeakyReLU) or 3 (sigmoid), got %d", fn, act);
    FFWM_REQUIRE(aligned16(y) && aligned16(grad_y) && aligned16(grad_z), FFWM_ERR_ARG, "%s: tensors must be 16-byte aligned", fn);
    hipStream_t st = static_cast<hipStream_t>(stream);
    LaunchScope ls("add_act_bwd", st, 12.0 * static_cast<double>(n));
    if (act == 1)
        hipLaunchKernelGGL((add_act_bwd_kernel<1>), dim3(vec_grid(n)), dim3(kBlock), 0, st, (const float*)y, (const float*)grad_y,
                           (float*)grad_z, n, (float)negative_slope);
    else
        hipLaunchKernelGGL((add_act_bwd_kernel<3>), dim3(vec_grid(n)), dim3(kBlock), 0, st, (const float*)y, (const float*)grad_y,
                           (float*)grad_z, n, 0.f);
    return check_launch(fn);
}

// att = sigmoid(a + b), y = x * att  (att is what the backward needs; att may alias a or b)
extern "C" int ffwm_sigmoid_gate_forward(const void* a, const void* b, const void* x, void* att, void* y, int64_t n, int dtype,
                                         void* stream) {
    const char* fn = "ffwm_sigmoid_gate_forward";
    FFWM_REQUIRE(dtype == FFWM_F32, FFWM_ERR_DTYPE, "%s: float32 only", fn);
    FFWM_REQUIRE(a && b && x && att && y && n > 0, FFWM_ERR_ARG, "%s: NULL pointer or empty tensor", fn);
    FFWM_REQUIRE(aligned16(a) && aligned16(b) && aligned16(x) && aligned16(att) && aligned16(y), FFWM_ERR_ARG,
                 "%s: tensors must be 16-byte aligned", fn);
    hipStream_t st = static_cast<hipStream_t>(stream);
    LaunchScope ls("sigmoid_gate_fwd", st, 20.0 * static_cast<double>(n));
    hipLaunchKernelGGL(gate_fwd_kernel, dim3(vec_grid(n)), dim3(kBlock), 0, st, (const float*)a, (const float*)b, (const float*)x,
                       (float*)att, (float*)y, n);
    return check_launch(fn);
}

// grad_z = grad_y * x * att * (1 - att) (the gradient of a and of b), grad_x = grad_y * att
extern "C" int ffwm_sigmoid_gate_backward(const void* x, const void* att, const void* grad_y, void* grad_z, void* grad_x, int64_t n,
                                          int dtype, void* stream) {
    const char* fn = "ffwm_sigmoid_gate_backward";
    FFWM_REQUIRE(dtype == FFWM_F32, FFWM_ERR_DTYPE, "%s: float32 only", fn);
    FFWM_REQUIRE(x && att && grad_y && grad_z && grad_x && n > 0, FFWM_ERR_ARG, "%s: NULL pointer or empty tensor", fn);
    FFWM_REQUIRE(aligned16(x) && aligned16(att) && aligned16(grad_y) && aligned16(grad_z) && aligned16(grad_x), FFWM_ERR_ARG,
                 "%s: tensors must be 16-byte aligned", fn);
    hipStream_t st = static_cast<hipStream_t>(stream);
    LaunchScope ls("sigmoid_gate_bwd", st, 20.0 * static_cast<double>(n));
    hipLaunchKernelGGL(gate_bwd_kernel, dim3(vec_grid(n)), dim3(kBlock), 0, st, (const float*)x, (const float*)att, (const float*)grad_y,
                       (float*)grad_z, (float*)grad_x, n);
    return check_launch(fn);
}
