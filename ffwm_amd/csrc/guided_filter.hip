// guided_filter.hip -- GuidedFilter(r, eps)(x, y) of FFWM's illumination-adaption path for gfx950.
//
// Reference: /root/reference/models/external_function.py:164-193 (diff_x / diff_y / BoxFilter: box sums
// as cumsum differences, rows first, then columns) and :239-277 (GuidedFilter.forward):
//     N = box(1); mean_x = box(x)/N; mean_y = box(y)/N; cov = box(x*y)/N - mean_x*mean_y;
//     var = box(x*x)/N - mean_x^2; A = cov/(var+eps); b = mean_y - A*mean_x;
//     out = (box(A)/N) * x + box(b)/N
// called by FFWMModel.forward on the generated 128 x 128 image every step and on the 64 / 32 px scales
// after 20000 iterations (models/ffwm_model.py:57-59,81,104-105).  In PyTorch that is ~100 launches
// forward and ~200 backward (cumsum, narrow, cat, sub, div ... per box filter).
//
// Here: one launch forward, one backward.  A 1024-thread block owns one (b, c) plane; the plane lives in
// LDS (two pitch-(W+1) buffers, bank-conflict-free along rows AND columns) and every box filter is
// done the reference's way -- inclusive prefix sums down the columns, window difference, prefix sums
// along the rows, window difference -- with wave-parallel scans (a wave per row / column, two elements
// per lane + 6 shuffle steps).  N is the closed form of box(1) (exact small integers, as in the
// reference).  Each thread owns up to 16 pixels (p = tid + 1024 q): global reads and writes are
// coalesced rows.  Planes up to 128 x 128 (2 x 66 KB of LDS).
#include "common.hpp"

namespace ffwm {
namespace {

constexpr int kGfThreads = 1024;
constexpr int kGfWaves = kGfThreads / kWave;
constexpr int kGfMaxPix = 16;         // pixels per thread: 128 * 128 / 1024
constexpr int kGfMaxDim = 128;

template <typename T>
struct GfCtx {
    T* buf0;
    T* buf1;
    int H, W, P, r, npix;
    int ij[kGfMaxPix];        // (row << 8 | column) of this thread's pixels, computed once (no per-pass integer division)
};

template <typename T>
__device__ __forceinline__ void gf_init(GfCtx<T>& c, unsigned char* smem, int H, int W, int r) {
    c.H = H; c.W = W; c.P = W + 1; c.r = r; c.npix = H * W;
    c.buf0 = reinterpret_cast<T*>(smem);
    c.buf1 = c.buf0 + static_cast<size_t>(H) * c.P;
#pragma unroll
    for (int k = 0; k < kGfMaxPix; ++k) {
        const int p = threadIdx.x + k * kGfThreads;
        const int i = p / W, j = p - i * W;
        c.ij[k] = (i << 8) | j;
    }
}

// inclusive prefix sums down every column (dim 2 of NCHW), in place: a wave per column, a lane owns
// 2 consecutive rows
template <typename T>
__device__ __forceinline__ void scan_cols(T* buf, const GfCtx<T>& c) {
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    for (int j = wave; j < c.W; j += kGfWaves) {
        const int i0 = 2 * lane;
        T a = i0 < c.H ? buf[i0 * c.P + j] : static_cast<T>(0);
        T b = i0 + 1 < c.H ? buf[(i0 + 1) * c.P + j] : static_cast<T>(0);
        b += a;
        T incl = b;
#pragma unroll
        for (int o = 1; o < kWave; o <<= 1) {
            const T t = __shfl_up(incl, o, kWave);
            if (lane >= o) incl += t;
        }
        const T excl = incl - b;
        if (i0 < c.H) buf[i0 * c.P + j] = a + excl;
        if (i0 + 1 < c.H) buf[(i0 + 1) * c.P + j] = b + excl;
    }
}

// inclusive prefix sums along every row (dim 3), in place
template <typename T>
__device__ __forceinline__ void scan_rows(T* buf, const GfCtx<T>& c) {
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    for (int i = wave; i < c.H; i += kGfWaves) {
        T* row = buf + i * c.P;
        const int j0 = 2 * lane;
        T a = j0 < c.W ? row[j0] : static_cast<T>(0);
        T b = j0 + 1 < c.W ? row[j0 + 1] : static_cast<T>(0);
        b += a;
        T incl = b;
#pragma unroll
        for (int o = 1; o < kWave; o <<= 1) {
            const T t = __shfl_up(incl, o, kWave);
            if (lane >= o) incl += t;
        }
        const T excl = incl - b;
        if (j0 < c.W) row[j0] = a + excl;
        if (j0 + 1 < c.W) row[j0 + 1] = b + excl;
    }
}

// box(q) for this thread's pixels; q[] in, box sums out (BoxFilter.forward, external_function.py:185-193)
template <typename T>
__device__ __forceinline__ void box_filter(T (&q)[kGfMaxPix], const GfCtx<T>& c) {
    const int r = c.r;
#pragma unroll
    for (int k = 0; k < kGfMaxPix; ++k) {
        const int p = threadIdx.x + k * kGfThreads;
        if (p < c.npix) {
            const int i = c.ij[k] >> 8, j = c.ij[k] & 255;
            c.buf0[i * c.P + j] = q[k];
        }
    }
    __syncthreads();
    scan_cols(c.buf0, c);                         // x.cumsum(dim=2)
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kGfMaxPix; ++k) {         // diff_x(., r)
        const int p = threadIdx.x + k * kGfThreads;
        if (p < c.npix) {
            const int i = c.ij[k] >> 8, j = c.ij[k] & 255;
            const int hi = i + r < c.H - 1 ? i + r : c.H - 1, lo = i - r - 1;
            T v = c.buf0[hi * c.P + j];
            if (lo >= 0) v -= c.buf0[lo * c.P + j];
            c.buf1[i * c.P + j] = v;
        }
    }
    __syncthreads();
    scan_rows(c.buf1, c);                         // .cumsum(dim=3)
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kGfMaxPix; ++k) {         // diff_y(., r)
        const int p = threadIdx.x + k * kGfThreads;
        if (p < c.npix) {
            const int i = c.ij[k] >> 8, j = c.ij[k] & 255;
            const int hi = j + r < c.W - 1 ? j + r : c.W - 1, lo = j - r - 1;
            T v = c.buf1[i * c.P + hi];
            if (lo >= 0) v -= c.buf1[i * c.P + lo];
            q[k] = v;
        }
    }
    // the next fill writes buf0 while slower waves may still read buf1 (different buffers), and the next
    // diff_x writes buf1 only after two more barriers: no barrier needed here
}

// N = box(1): the window clipped to the image
template <typename T>
__device__ __forceinline__ T box_count(int i, int j, const GfCtx<T>& c) {
    const int h = (i + c.r < c.H - 1 ? i + c.r : c.H - 1) - (i - c.r > 0 ? i - c.r : 0) + 1;
    const int w = (j + c.r < c.W - 1 ? j + c.r : c.W - 1) - (j - c.r > 0 ? j - c.r : 0) + 1;
    return static_cast<T>(h * w);
}

template <typename T>
__global__ void __launch_bounds__(kGfThreads)
gf_forward_kernel(const T* __restrict__ x, const T* __restrict__ y, T* __restrict__ out, T* __restrict__ saved,
                  int64_t planes, int H, int W, int r, T eps) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    GfCtx<T> c;
    gf_init(c, smem_raw, H, W, r);
    const size_t base = static_cast<size_t>(blockIdx.x) * c.npix;
    const size_t sstride = static_cast<size_t>(planes) * c.npix;      // saved[k] plane stride
    const T* xp = x + base;
    const T* yp = y + base;
    T* sv = saved + base;
    T q[kGfMaxPix], e1[kGfMaxPix];

#define FFWM_GF_EACH(...)                                                    \
    _Pragma("unroll") for (int k = 0; k < kGfMaxPix; ++k) {                  \
        const int p = threadIdx.x + k * kGfThreads;                          \
        if (p < c.npix) {                                                    \
            const int i = c.ij[k] >> 8, j = c.ij[k] & 255;                   \
            (void)i; (void)j;                                                \
            __VA_ARGS__                                                      \
        }                                                                    \
    }
    FFWM_GF_EACH(q[k] = xp[p];)
    box_filter(q, c);
    FFWM_GF_EACH(sv[p] = q[k] / box_count(i, j, c);)                                   // mean_x
    FFWM_GF_EACH(q[k] = yp[p];)
    box_filter(q, c);
    FFWM_GF_EACH(sv[sstride + p] = q[k] / box_count(i, j, c);)                         // mean_y
    FFWM_GF_EACH(q[k] = xp[p] * yp[p];)
    box_filter(q, c);
    FFWM_GF_EACH(e1[k] = q[k] / box_count(i, j, c) - sv[p] * sv[sstride + p];)         // cov_xy
    FFWM_GF_EACH(q[k] = xp[p] * xp[p];)
    box_filter(q, c);
    FFWM_GF_EACH(
        const T mx = sv[p];
        const T ve = (q[k] / box_count(i, j, c) - mx * mx) + eps;                      // var_x + eps
        const T A = e1[k] / ve;
        sv[2 * sstride + p] = A;
        sv[3 * sstride + p] = ve;
        e1[k] = sv[sstride + p] - A * mx;                                              // b
        q[k] = A;)
    box_filter(q, c);
    FFWM_GF_EACH(q[k] = q[k] / box_count(i, j, c); sv[4 * sstride + p] = q[k];)         // mean_A
    box_filter(e1, c);
    FFWM_GF_EACH(out[base + p] = q[k] * xp[p] + e1[k] / box_count(i, j, c);)            // mean_A * x + mean_b
}

// grad_x of out = GuidedFilter(x, y) (y is data: FFWM filters the generated image against the ground truth).
// box is self-adjoint (symmetric clipped windows), so the adjoint of z -> box(z)/N is g -> box(g/N).
template <typename T>
__global__ void __launch_bounds__(kGfThreads)
gf_backward_kernel(const T* __restrict__ x, const T* __restrict__ y, const T* __restrict__ saved,
                   const T* __restrict__ gout, T* __restrict__ gx, int64_t planes, int H, int W, int r) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    GfCtx<T> c;
    gf_init(c, smem_raw, H, W, r);
    const size_t base = static_cast<size_t>(blockIdx.x) * c.npix;
    const size_t sstride = static_cast<size_t>(planes) * c.npix;
    const T* xp = x + base;
    const T* yp = y + base;
    const T* gp = gout + base;
    const T* sv = saved + base;
    T q1[kGfMaxPix], q2[kGfMaxPix], q3[kGfMaxPix];

    FFWM_GF_EACH(q1[k] = gp[p] * xp[p] / box_count(i, j, c);)        // d mean_A -> d A
    box_filter(q1, c);
    FFWM_GF_EACH(q2[k] = gp[p] / box_count(i, j, c);)                // d mean_b -> d b
    box_filter(q2, c);
    FFWM_GF_EACH(
        const T mx = sv[p], my = sv[sstride + p], A = sv[2 * sstride + p], ve = sv[3 * sstride + p];
        const T n = box_count(i, j, c);
        const T gA = q1[k] - q2[k] * mx;             // b = mean_y - A mean_x
        T gmx = -q2[k] * A;
        const T gcov = gA / ve;                      // A = cov / (var + eps)
        const T gvar = -gA * A / ve;
        gmx += -gcov * my - 2 * gvar * mx;           // cov = E[xy] - mx my ; var = E[xx] - mx^2
        q1[k] = gmx / n;
        q2[k] = gcov / n;
        q3[k] = gvar / n;)
    box_filter(q1, c);                               // -> d x through mean_x
    FFWM_GF_EACH(q1[k] += gp[p] * sv[4 * sstride + p];)               // + g * mean_A
    box_filter(q2, c);                               // -> d (x y)
    FFWM_GF_EACH(q1[k] += yp[p] * q2[k];)
    box_filter(q3, c);                               // -> d (x x)
    FFWM_GF_EACH(gx[base + p] = q1[k] + 2 * xp[p] * q3[k];)
#undef FFWM_GF_EACH
}

int check_args(const char* fn, int64_t planes, int64_t H, int64_t W, int r, int dtype) {
    FFWM_REQUIRE(dtype_ok(dtype), FFWM_ERR_DTYPE, "%s: dtype %d is not FFWM_F32/FFWM_F64", fn, dtype);
    FFWM_REQUIRE(planes > 0 && H > 0 && W > 0 && r >= 0, FFWM_ERR_ARG, "%s: sizes must be positive", fn);
    FFWM_REQUIRE(H > 2 * r + 1 && W > 2 * r + 1, FFWM_ERR_ARG,
                 "%s: need H > 2r+1 and W > 2r+1 (H=%lld W=%lld r=%d), as the reference asserts", fn, (long long)H,
                 (long long)W, r);
    FFWM_REQUIRE(H <= kGfMaxDim && W <= kGfMaxDim, FFWM_ERR_SIZE,
                 "%s: planes up to %d x %d (LDS-resident), got %lld x %lld", fn, kGfMaxDim, kGfMaxDim, (long long)H, (long long)W);
    FFWM_REQUIRE(planes < (1LL << 31), FFWM_ERR_SIZE, "%s: too many planes", fn);
    return FFWM_OK;
}

}  // namespace
}  // namespace ffwm

using namespace ffwm;

extern "C" int ffwm_guided_filter_forward(const void* x, const void* y, void* output, void* saved, int64_t planes,
                                          int64_t H, int64_t W, int r, double eps, int dtype, void* stream) {
    const char* fn = "ffwm_guided_filter_forward";
    FFWM_REQUIRE(x && y && output && saved, FFWM_ERR_ARG, "%s: NULL tensor pointer", fn);
    if (int rc = check_args(fn, planes, H, W, r, dtype)) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const size_t esz = dtype == FFWM_F32 ? 4 : 8;
    const size_t lds = 2 * static_cast<size_t>(H) * (W + 1) * esz;
    FFWM_REQUIRE(lds <= static_cast<size_t>(kMaxLdsBytes), FFWM_ERR_SIZE, "%s: plane does not fit LDS", fn);
    LaunchScope ls("guided_filter_fwd", st, static_cast<double>(esz) * planes * H * W * 8.0);   // x, y in; out + 5 saved planes
    if (dtype == FFWM_F32) {
        allow_large_lds(reinterpret_cast<const void*>(gf_forward_kernel<float>));
        hipLaunchKernelGGL((gf_forward_kernel<float>), dim3(static_cast<unsigned>(planes)), dim3(kGfThreads), lds, st,
                           (const float*)x, (const float*)y, (float*)output, (float*)saved, planes, (int)H, (int)W, r,
                           static_cast<float>(eps));
    } else {
        allow_large_lds(reinterpret_cast<const void*>(gf_forward_kernel<double>));
        hipLaunchKernelGGL((gf_forward_kernel<double>), dim3(static_cast<unsigned>(planes)), dim3(kGfThreads), lds, st,
                           (const double*)x, (const double*)y, (double*)output, (double*)saved, planes, (int)H, (int)W, r, eps);
    }
    return check_launch(fn);
}

extern "C" int ffwm_guided_filter_backward(const void* x, const void* y, const void* saved, const void* grad_output,
                                           void* grad_x, int64_t planes, int64_t H, int64_t W, int r, int dtype,
                                           void* stream) {
    const char* fn = "ffwm_guided_filter_backward";
    FFWM_REQUIRE(x && y && saved && grad_output && grad_x, FFWM_ERR_ARG, "%s: NULL tensor pointer", fn);
    if (int rc = check_args(fn, planes, H, W, r, dtype)) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const size_t esz = dtype == FFWM_F32 ? 4 : 8;
    const size_t lds = 2 * static_cast<size_t>(H) * (W + 1) * esz;
    FFWM_REQUIRE(lds <= static_cast<size_t>(kMaxLdsBytes), FFWM_ERR_SIZE, "%s: plane does not fit LDS", fn);
    LaunchScope ls("guided_filter_bwd", st, static_cast<double>(esz) * planes * H * W * 9.0);   // x, y, g, 5 saved in; grad out
    if (dtype == FFWM_F32) {
        allow_large_lds(reinterpret_cast<const void*>(gf_backward_kernel<float>));
        hipLaunchKernelGGL((gf_backward_kernel<float>), dim3(static_cast<unsigned>(planes)), dim3(kGfThreads), lds, st,
                           (const float*)x, (const float*)y, (const float*)saved, (const float*)grad_output, (float*)grad_x,
                           planes, (int)H, (int)W, r);
    } else {
        allow_large_lds(reinterpret_cast<const void*>(gf_backward_kernel<double>));
        hipLaunchKernelGGL((gf_backward_kernel<double>), dim3(static_cast<unsigned>(planes)), dim3(kGfThreads), lds, st,
                           (const double*)x, (const double*)y, (const double*)saved, (const double*)grad_output,
                           (double*)grad_x, planes, (int)H, (int)W, r);
    }
    return check_launch(fn);
}
