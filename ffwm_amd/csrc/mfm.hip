// mfm.hip -- LightCNN's max-feature-map activation, forward and backward, one kernel each.
//
// Reference: lightcnn/light_cnn.py `mfm.forward`: out = torch.split(self.filter(x), out_channels, 1);
// return torch.max(out[0], out[1]) -- 29 such layers per LightCNN pass, two passes per train step (identity loss,
// models/ffwm_model.py:125-131).  ATen runs the maximum on two strided halves and its backward as five kernels
// (two comparisons, where, masked_fill, the concatenation of the split's backward); here
//     y[b,c,:]     = max(x[b,c,:], x[b,C+c,:])
//     dx[b,c,:]    = g * [a > b] + g/2 * [a == b],   dx[b,C+c,:] = g * [a < b] + g/2 * [a == b]
// (the tie rule of ATen's `maximum` derivative) are one coalesced float4 pass each.  An optional bias[2C] is the
// bias of the convolution / linear layer in front (the vendor library adds it with a separate read-modify-write
// pass over the activation): x + bias is formed on the fly, bit-identical to adding it first.
// The same file holds the fused bias + ReLU of the frozen VGG19 conv stack (models/losses.py:398-519).
#include "common.hpp"

namespace ffwm {
namespace {

// n4 = float4 per (b, c) row; rows = B * C; the partner row is C rows further inside the same sample
__global__ void __launch_bounds__(kBlock)
mfm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ bias, float* __restrict__ y, int64_t total, int C, int HW) {
    const int64_t half = static_cast<int64_t>(C) * HW;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < total; i += static_cast<int64_t>(gridDim.x) * kBlock) {
        const int64_t b = i / half, r = i - b * half;          // element r of the C*HW outputs of sample b
        const float* p = x + b * 2 * half + r;
        const int c = static_cast<int>(r / HW);
        const float ba = bias ? bias[c] : 0.f, bb = bias ? bias[C + c] : 0.f;
        y[i] = fmaxf(p[0] + ba, p[half] + bb);
    }
}

__global__ void __launch_bounds__(kBlock)
mfm_fwd4_kernel(const float4* __restrict__ x, const float* __restrict__ bias, float4* __restrict__ y, int64_t total4,
                int64_t half4, int C, int HW4) {     // HW4 = HW / 4: a float4 never straddles two channels
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < total4; i += static_cast<int64_t>(gridDim.x) * kBlock) {
        const int64_t b = i / half4, r = i - b * half4;
        const float4 a = x[b * 2 * half4 + r], c = x[b * 2 * half4 + half4 + r];
        const int ch = static_cast<int>(r / HW4);
        const float ba = bias ? bias[ch] : 0.f, bb = bias ? bias[C + ch] : 0.f;
        y[i] = float4{fmaxf(a.x + ba, c.x + bb), fmaxf(a.y + ba, c.y + bb), fmaxf(a.z + ba, c.z + bb), fmaxf(a.w + ba, c.w + bb)};
    }
}

__device__ __forceinline__ void mfm_grad(float a, float b, float g, float& da, float& db) {
    const float h = a == b ? 0.5f * g : g;          // maximum's derivative: ties share the gradient
    da = a < b ? 0.f : h;
    db = b < a ? 0.f : h;
}

__global__ void __launch_bounds__(kBlock)
mfm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ bias, const float* __restrict__ g, float* __restrict__ dx,
               int64_t total, int C, int HW) {
    const int64_t half = static_cast<int64_t>(C) * HW;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < total; i += static_cast<int64_t>(gridDim.x) * kBlock) {
        const int64_t b = i / half, r = i - b * half;
        const int64_t o = b * 2 * half + r;
        const int c = static_cast<int>(r / HW);
        const float ba = bias ? bias[c] : 0.f, bb = bias ? bias[C + c] : 0.f;
        float da, db;
        mfm_grad(x[o] + ba, x[o + half] + bb, g[i], da, db);
        dx[o] = da;
        dx[o + half] = db;
    }
}

__global__ void __launch_bounds__(kBlock)
mfm_bwd4_kernel(const float4* __restrict__ x, const float* __restrict__ bias, const float4* __restrict__ g, float4* __restrict__ dx,
                int64_t total4, int64_t half4, int C, int HW4) {
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < total4; i += static_cast<int64_t>(gridDim.x) * kBlock) {
        const int64_t b = i / half4, r = i - b * half4;
        const int64_t o = b * 2 * half4 + r;
        const float4 a = x[o], c = x[o + half4], gg = g[i];
        const int ch = static_cast<int>(r / HW4);
        const float ba = bias ? bias[ch] : 0.f, bb = bias ? bias[C + ch] : 0.f;
        float4 da, db;
        mfm_grad(a.x + ba, c.x + bb, gg.x, da.x, db.x);
        mfm_grad(a.y + ba, c.y + bb, gg.y, da.y, db.y);
        mfm_grad(a.z + ba, c.z + bb, gg.z, da.z, db.z);
        mfm_grad(a.w + ba, c.w + bb, gg.w, da.w, db.w);
        dx[o] = da;
        dx[o + half4] = db;
    }
}

// y = relu(h + bias[c]); in place (y == h) allowed
__global__ void __launch_bounds__(kBlock)
bias_relu_kernel(const float* __restrict__ h, const float* __restrict__ bias, float* __restrict__ y, int64_t total, int C, int HW) {
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < total; i += static_cast<int64_t>(gridDim.x) * kBlock) {
        const int c = static_cast<int>((i / HW) % C);
        y[i] = fmaxf(h[i] + bias[c], 0.f);
    }
}

__global__ void __launch_bounds__(kBlock)
bias_relu4_kernel(const float4* __restrict__ h, const float* __restrict__ bias, float4* __restrict__ y, int64_t total4, int C, int HW4) {
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < total4; i += static_cast<int64_t>(gridDim.x) * kBlock) {
        const float bc = bias[static_cast<int>((i / HW4) % C)];
        const float4 v = h[i];
        y[i] = float4{fmaxf(v.x + bc, 0.f), fmaxf(v.y + bc, 0.f), fmaxf(v.z + bc, 0.f), fmaxf(v.w + bc, 0.f)};
    }
}

int check_mfm(const char* fn, int64_t B, int64_t C, int64_t HW, int dtype) {
    FFWM_REQUIRE(dtype == FFWM_F32, FFWM_ERR_DTYPE, "%s: float32 only", fn);
    FFWM_REQUIRE(B > 0 && C > 0 && HW > 0, FFWM_ERR_ARG, "%s: sizes must be positive", fn);
    FFWM_REQUIRE(B * C * HW < (1LL << 40), FFWM_ERR_SIZE, "%s: tensor too large", fn);
    return FFWM_OK;
}

unsigned mfm_grid(int64_t n) {
    int64_t blocks = (n + kBlock - 1) / kBlock;
    return static_cast<unsigned>(blocks > 256 * 32 ? 256 * 32 : (blocks < 1 ? 1 : blocks));
}

}  // namespace
}  // namespace ffwm

using namespace ffwm;

extern "C" int ffwm_bias_relu_forward(const void* h, const void* bias, void* y, int64_t B, int64_t C, int64_t HW, int dtype,
                                     void* stream) {
    const char* fn = "ffwm_bias_relu_forward";
    if (int rc = check_mfm(fn, B, C, HW, dtype)) return rc;
    FFWM_REQUIRE(h && bias && y, FFWM_ERR_ARG, "%s: NULL tensor pointer", fn);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int64_t total = B * C * HW;
    LaunchScope ls("bias_relu_fwd", st, 4.0 * 2.0 * total);
    if (HW % 4 == 0 && (reinterpret_cast<uintptr_t>(h) | reinterpret_cast<uintptr_t>(y)) % 16 == 0)
        hipLaunchKernelGGL(bias_relu4_kernel, dim3(mfm_grid(total / 4)), dim3(kBlock), 0, st, (const float4*)h, (const float*)bias,
                           (float4*)y, total / 4, (int)C, (int)(HW / 4));
    else
        hipLaunchKernelGGL(bias_relu_kernel, dim3(mfm_grid(total)), dim3(kBlock), 0, st, (const float*)h, (const float*)bias, (float*)y,
                           total, (int)C, (int)HW);
    return check_launch(fn);
}

extern "C" int ffwm_mfm_forward(const void* x, const void* bias, void* y, int64_t B, int64_t C, int64_t HW, int dtype, void* stream) {
    const char* fn = "ffwm_mfm_forward";
    if (int rc = check_mfm(fn, B, C, HW, dtype)) return rc;
    FFWM_REQUIRE(x && y, FFWM_ERR_ARG, "%s: NULL tensor pointer", fn);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int64_t half = C * HW, total = B * half;
    LaunchScope ls("mfm_fwd", st, 4.0 * 3.0 * total);
    const bool vec = HW % 4 == 0 && (reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) % 16 == 0;
    if (vec)
        hipLaunchKernelGGL(mfm_fwd4_kernel, dim3(mfm_grid(total / 4)), dim3(kBlock), 0, st, (const float4*)x, (const float*)bias,
                           (float4*)y, total / 4, half / 4, (int)C, (int)(HW / 4));
    else
        hipLaunchKernelGGL(mfm_fwd_kernel, dim3(mfm_grid(total)), dim3(kBlock), 0, st, (const float*)x, (const float*)bias, (float*)y,
                           total, (int)C, (int)HW);
    return check_launch(fn);
}

extern "C" int ffwm_mfm_backward(const void* x, const void* bias, const void* grad_y, void* grad_x, int64_t B, int64_t C, int64_t HW,
                                 int dtype, void* stream) {
    const char* fn = "ffwm_mfm_backward";
    if (int rc = check_mfm(fn, B, C, HW, dtype)) return rc;
    FFWM_REQUIRE(x && grad_y && grad_x, FFWM_ERR_ARG, "%s: NULL tensor pointer", fn);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int64_t half = C * HW, total = B * half;
    LaunchScope ls("mfm_bwd", st, 4.0 * 5.0 * total);
    const bool vec = HW % 4 == 0 &&
                     (reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(grad_y) | reinterpret_cast<uintptr_t>(grad_x)) % 16 == 0;
    if (vec)
        hipLaunchKernelGGL(mfm_bwd4_kernel, dim3(mfm_grid(total / 4)), dim3(kBlock), 0, st, (const float4*)x, (const float*)bias,
                           (const float4*)grad_y, (float4*)grad_x, total / 4, half / 4, (int)C, (int)(HW / 4));
    else
        hipLaunchKernelGGL(mfm_bwd_kernel, dim3(mfm_grid(total)), dim3(kBlock), 0, st, (const float*)x, (const float*)bias,
                           (const float*)grad_y, (float*)grad_x, total, (int)C, (int)HW);
    return check_launch(fn);
}
