// local_attn_reshape.hip -- [B,k*k,H,W] <-> [B,1,k*H,k*W] window gather for gfx950.
//
// Replaces kernel_local_attn_reshape_update_output / _backward of
// /root/reference/cuda/local_attn_reshape/local_attn_reshape_kernel.cu:21-61,66-108 (one thread per
// output element: k consecutive threads read k different channel planes -> uncoalesced; backward
// uses atomicAdd although the map is a bijection).
//
// Here one thread owns one (b, ys, i, xs) ROW of a k x k window: it reads the k planes
// i*k .. i*k+k-1 at (ys, xs) (each read coalesced across the wave) and writes k CONTIGUOUS
// outputs.  Flattening (b, ys, i, xs) in that order makes the output offset exactly k * tid, so
// a wave writes one contiguous 64*k-element run.  Backward is the inverse permutation: plain
// stores (or read-modify-write in the reference's accumulate mode), no atomics.
#include "common.hpp"

namespace ffwm {
namespace {

template <typename T, int K>
struct __attribute__((packed, aligned(sizeof(T)))) PackedRow {
    T v[K];
};

template <typename T, int K>
__global__ void __launch_bounds__(kBlock)
lar_fwd_kernel(const T* __restrict__ in, T* __restrict__ out, int64_t nrows, int H, int W) {
    const int64_t plane = static_cast<int64_t>(H) * W;
    for (int64_t tid = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; tid < nrows;
         tid += static_cast<int64_t>(gridDim.x) * kBlock) {
        const int xs = static_cast<int>(tid % W);
        int64_t r = tid / W;
        const int i = static_cast<int>(r % K);
        r /= K;
        const int ys = static_cast<int>(r % H);
        const int64_t b = r / H;
        const T* p = in + (b * K * K + static_cast<int64_t>(i) * K) * plane + static_cast<int64_t>(ys) * W + xs;
        PackedRow<T, K> row;
#pragma unroll
        for (int j = 0; j < K; ++j) row.v[j] = p[j * plane];
        *reinterpret_cast<PackedRow<T, K>*>(out + tid * K) = row;
    }
}

template <typename T, int K, bool ACC>
__global__ void __launch_bounds__(kBlock)
lar_bwd_kernel(const T* __restrict__ gout, T* __restrict__ gin, int64_t nrows, int H, int W) {
    const int64_t plane = static_cast<int64_t>(H) * W;
    for (int64_t tid = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; tid < nrows;
         tid += static_cast<int64_t>(gridDim.x) * kBlock) {
        const int xs = static_cast<int>(tid % W);
        int64_t r = tid / W;
        const int i = static_cast<int>(r % K);
        r /= K;
        const int ys = static_cast<int>(r % H);
        const int64_t b = r / H;
        T* p = gin + (b * K * K + static_cast<int64_t>(i) * K) * plane + static_cast<int64_t>(ys) * W + xs;
        const PackedRow<T, K> row = *reinterpret_cast<const PackedRow<T, K>*>(gout + tid * K);
#pragma unroll
        for (int j = 0; j < K; ++j) {
            if (ACC) p[j * plane] += row.v[j];   // bijection: no other thread touches this address
            else p[j * plane] = row.v[j];
        }
    }
}

// Any kernel_size: the same row decomposition with a run-time k.
template <typename T>
__global__ void __launch_bounds__(kBlock)
lar_fwd_generic(const T* __restrict__ in, T* __restrict__ out, int64_t nrows, int H, int W, int k) {
    const int64_t plane = static_cast<int64_t>(H) * W;
    for (int64_t tid = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; tid < nrows;
         tid += static_cast<int64_t>(gridDim.x) * kBlock) {
        const int xs = static_cast<int>(tid % W);
        int64_t r = tid / W;
        const int i = static_cast<int>(r % k);
        r /= k;
        const int ys = static_cast<int>(r % H);
        const int64_t b = r / H;
        const T* p = in + (b * k * k + static_cast<int64_t>(i) * k) * plane + static_cast<int64_t>(ys) * W + xs;
        for (int j = 0; j < k; ++j) out[tid * k + j] = p[j * plane];
    }
}

template <typename T>
__global__ void __launch_bounds__(kBlock)
lar_bwd_generic(const T* __restrict__ gout, T* __restrict__ gin, int64_t nrows, int H, int W, int k,
                int acc, const GoStrides gs = GoStrides{0, 0, 0, 0}) {
    const int64_t plane = static_cast<int64_t>(H) * W;
    for (int64_t tid = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; tid < nrows;
         tid += static_cast<int64_t>(gridDim.x) * kBlock) {
        const int xs = static_cast<int>(tid % W);
        int64_t r = tid / W;
        const int i = static_cast<int>(r % k);
        r /= k;
        const int ys = static_cast<int>(r % H);
        const int64_t b = r / H;
        T* p = gin + (b * k * k + static_cast<int64_t>(i) * k) * plane + static_cast<int64_t>(ys) * W + xs;
        for (int j = 0; j < k; ++j) {
            // grad_output [B, 1, kH, kW]: row (b, ys k + i), column xs k + j -- through its strides when they were handed over
            const T g = gs.x == 0 ? gout[tid * k + j]
                                  : gout[b * gs.b + (static_cast<long long>(ys) * k + i) * gs.y + (static_cast<long long>(xs) * k + j) * gs.x];
            if (acc) p[j * plane] += g;
            else p[j * plane] = g;
        }
    }
}

unsigned grid_for(int64_t nrows) {
    const int64_t blocks = (nrows + kBlock - 1) / kBlock;
    return static_cast<unsigned>(blocks < 8192 ? blocks : 8192);   // grid-stride beyond 32 blocks/CU
}

template <typename T>
int launch_fwd(const T* in, T* out, int64_t B, int64_t H, int64_t W, int k, hipStream_t st) {
    const int64_t nrows = B * H * k * W;
    const double bytes = 2.0 * sizeof(T) * static_cast<double>(nrows) * k;
    const unsigned grid = grid_for(nrows);
    LaunchScope ls("local_attn_reshape_fwd", st, bytes);
#define FFWM_LAR_FWD(KK)                                                                          \
    case KK:                                                                                      \
        hipLaunchKernelGGL((lar_fwd_kernel<T, KK>), dim3(grid), dim3(kBlock), 0, st, in, out,      \
                           nrows, (int)H, (int)W);                                                \
        break;
    switch (k) {
        FFWM_LAR_FWD(1) FFWM_LAR_FWD(2) FFWM_LAR_FWD(3) FFWM_LAR_FWD(4) FFWM_LAR_FWD(5)
        FFWM_LAR_FWD(6) FFWM_LAR_FWD(7)
        default:
            hipLaunchKernelGGL((lar_fwd_generic<T>), dim3(grid), dim3(kBlock), 0, st, in, out, nrows,
                               (int)H, (int)W, k);
    }
#undef FFWM_LAR_FWD
    return check_launch("ffwm_local_attn_reshape_forward");
}

template <typename T>
int launch_bwd(const T* gout, T* gin, int64_t B, int64_t H, int64_t W, int k, int acc,
               hipStream_t st) {
    const int64_t nrows = B * H * k * W;
    const double bytes = 2.0 * sizeof(T) * static_cast<double>(nrows) * k;
    const unsigned grid = grid_for(nrows);
    LaunchScope ls("local_attn_reshape_bwd", st, bytes);
#define FFWM_LAR_BWD(KK)                                                                          \
    case KK:                                                                                      \
        if (acc)                                                                                  \
            hipLaunchKernelGGL((lar_bwd_kernel<T, KK, true>), dim3(grid), dim3(kBlock), 0, st,     \
                               gout, gin, nrows, (int)H, (int)W);                                 \
        else                                                                                      \
            hipLaunchKernelGGL((lar_bwd_kernel<T, KK, false>), dim3(grid), dim3(kBlock), 0, st,    \
                               gout, gin, nrows, (int)H, (int)W);                                 \
        break;
    switch (k) {
        FFWM_LAR_BWD(1) FFWM_LAR_BWD(2) FFWM_LAR_BWD(3) FFWM_LAR_BWD(4) FFWM_LAR_BWD(5)
        FFWM_LAR_BWD(6) FFWM_LAR_BWD(7)
        default:
            hipLaunchKernelGGL((lar_bwd_generic<T>), dim3(grid), dim3(kBlock), 0, st, gout, gin, nrows,
                               (int)H, (int)W, k, acc);
    }
#undef FFWM_LAR_BWD
    return check_launch("ffwm_local_attn_reshape_backward");
}

int check_dims(const char* fn, int64_t B, int64_t H, int64_t W, int k, int dtype) {
    FFWM_REQUIRE(dtype_ok(dtype), FFWM_ERR_DTYPE, "%s: dtype %d is not FFWM_F32/FFWM_F64", fn, dtype);
    FFWM_REQUIRE(B > 0 && H > 0 && W > 0 && k >= 1, FFWM_ERR_ARG,
                 "%s: sizes must be positive (B=%lld H=%lld W=%lld k=%d)", fn, (long long)B, (long long)H,
                 (long long)W, k);
    FFWM_REQUIRE(H < (1LL << 31) && W < (1LL << 31), FFWM_ERR_SIZE, "%s: H and W must fit 31 bits", fn);
    return FFWM_OK;
}

}  // namespace
}  // namespace ffwm

using namespace ffwm;

extern "C" int ffwm_local_attn_reshape_forward(const void* inputs, void* output, int64_t B, int64_t H,
                                               int64_t W, int kernel_size, int dtype, void* stream) {
    const char* fn = "ffwm_local_attn_reshape_forward";
    FFWM_REQUIRE(inputs && output, FFWM_ERR_ARG, "%s: NULL tensor pointer", fn);
    if (int rc = check_dims(fn, B, H, W, kernel_size, dtype)) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dtype == FFWM_F32)
        return launch_fwd<float>((const float*)inputs, (float*)output, B, H, W, kernel_size, st);
    return launch_fwd<double>((const double*)inputs, (double*)output, B, H, W, kernel_size, st);
}

extern "C" int ffwm_local_attn_reshape_backward(const void* grad_output, void* grad_inputs, int64_t B,
                                                int64_t H, int64_t W, int kernel_size, int accumulate,
                                                int dtype, void* stream) {
    const char* fn = "ffwm_local_attn_reshape_backward";
    FFWM_REQUIRE(grad_output && grad_inputs, FFWM_ERR_ARG, "%s: NULL tensor pointer", fn);
    if (int rc = check_dims(fn, B, H, W, kernel_size, dtype)) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dtype == FFWM_F32)
        return launch_bwd<float>((const float*)grad_output, (float*)grad_inputs, B, H, W, kernel_size,
                                 accumulate, st);
    return launch_bwd<double>((const double*)grad_output, (double*)grad_inputs, B, H, W, kernel_size,
                              accumulate, st);
}

// grad_output [B, 1, kH, kW] read through its element strides (NULL / contiguous: the entry point above); any other layout takes the
// run-time-k kernel.  Reference: local_attn_reshape_kernel.cu:66-108 reads gradOutput with DIM3_INDEX and the tensor's strides.
extern "C" int ffwm_local_attn_reshape_backward_strided(const void* grad_output, const int64_t* grad_output_strides, void* grad_inputs,
                                                        int64_t B, int64_t H, int64_t W, int kernel_size, int accumulate, int dtype,
                                                        void* stream) {
    const char* fn = "ffwm_local_attn_reshape_backward_strided";
    if (go_contiguous(grad_output_strides, 1, kernel_size * H, kernel_size * W))
        return ffwm_local_attn_reshape_backward(grad_output, grad_inputs, B, H, W, kernel_size, accumulate, dtype, stream);
    FFWM_REQUIRE(grad_output && grad_inputs, FFWM_ERR_ARG, "%s: NULL tensor pointer", fn);
    if (int rc = check_dims(fn, B, H, W, kernel_size, dtype)) return rc;
    for (int d = 0; d < 4; ++d)
        FFWM_REQUIRE(grad_output_strides[d] >= 0, FFWM_ERR_ARG, "%s: negative strides are not supported", fn);
    FFWM_REQUIRE(grad_output_strides[3] != 0, FFWM_ERR_ARG, "%s: a grad_output expanded along its last dimension (stride 0) is not supported", fn);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int64_t nrows = B * H * kernel_size * W;
    const GoStrides gs{grad_output_strides[0], grad_output_strides[1], grad_output_strides[2], grad_output_strides[3]};
    LaunchScope ls("local_attn_reshape_bwd_strided", st, 2.0 * (dtype == FFWM_F32 ? 4 : 8) * static_cast<double>(nrows) * kernel_size);
    if (dtype == FFWM_F32)
        hipLaunchKernelGGL((lar_bwd_generic<float>), dim3(grid_for(nrows)), dim3(kBlock), 0, st, (const float*)grad_output, (float*)grad_inputs,
                           nrows, (int)H, (int)W, kernel_size, accumulate, gs);
    else
        hipLaunchKernelGGL((lar_bwd_generic<double>), dim3(grid_for(nrows)), dim3(kBlock), 0, st, (const double*)grad_output, (double*)grad_inputs,
                           nrows, (int)H, (int)W, kernel_size, accumulate, gs);
    return check_launch(fn);
}
