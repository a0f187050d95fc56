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
// Here: four launches forward, four backward, every one over the WHOLE chip.  (Rounds 1-2 kept a plane in the LDS of one
// 1024-thread block: FFWM's call has 24 planes, i.e. 24 of 256 CUs busy for 110 us.)  A box filter is separable and the
// reference evaluates it that way -- cumsum down the columns, window difference, cumsum along the rows, window difference
// -- so the two directions are two kernels:
//   gf_cols_kernel: a block owns a strip of 32 columns of one plane for ONE quantity of the stage (x, y, xy, xx / A, b / ...;
//                   grid = planes x W/32 x quantities = 384 for the first stage of FFWM's call): the pointwise input is formed
//                   on the fly from coalesced 128-byte row segments, transposed through LDS, and a wave scans its eight columns
//                   side by side (two elements per lane, 6 shuffle steps, window difference by two more shuffles per element);
//   gf_rows_kernel: a WAVE owns one row (grid = planes x H / 4 = 768): coalesced row loads, the same register scan, and the
//                   stage's pointwise epilogue (means, cov, var, A, b ... / the quotient-rule terms of the backward) fused
//                   behind it.  No LDS, no barrier.
// The arithmetic (scan shape, order of the operations, N as the closed form of box(1)) is what the one-block kernel did.
// Intermediates live in the `saved` planes / the output (forward) and in grad_x + a two-plane workspace (backward).
// Planes up to 128 x 128 (a line is two elements per lane).
#include "common.hpp"

namespace ffwm {
namespace {

constexpr int kGfMaxDim = 128;
constexpr int kGfStrip = 32;          // columns per block of the column pass: a row segment is one 128-byte line
constexpr int kGfMaxQ = 4;            // quantities per stage

template <typename T>
struct GfArgs {
    const T* x;
    const T* y;
    const T* g;        // grad_output (backward)
    T* saved;          // [5, planes, H, W]  (read-only in the backward)
    T* out;            // output (forward) / grad_x (backward)
    T* ws;             // [2, planes, H, W] workspace (backward)
    int64_t planes;
    int H, W, r;
    T eps;
};

// N = box(1): the window clipped to the image
template <typename T>
__device__ __forceinline__ T box_count(int i, int j, int H, int W, int r) {
    const int h = (i + r < H - 1 ? i + r : H - 1) - (i - r > 0 ? i - r : 0) + 1;
    const int w = (j + r < W - 1 ? j + r : W - 1) - (j - r > 0 ? j - r : 0) + 1;
    return static_cast<T>(h * w);
}

// a, b = elements 2 lane, 2 lane + 1 of a line of L <= 128 values (zero beyond L) held by ONE wave; on return their box sums
// cumsum[min(k + r, L - 1)] - cumsum[k - r - 1]  (diff_x / diff_y of the reference on an inclusive cumsum).
template <typename T>
__device__ __forceinline__ void line_box(T& a, T& b, int lane, int L, int r) {
    b += a;
    T incl = b;
#pragma unroll
    for (int o = 1; o < kWave; o <<= 1) {
        const T t = __shfl_up(incl, o, kWave);
        if (lane >= o) incl += t;
    }
    const T excl = incl - b;
    const T ca = a + excl, cb = b + excl;         // inclusive cumsum at 2 lane, 2 lane + 1
    auto at = [&](int k) {                        // every lane takes part in the shuffles
        const T va = __shfl(ca, k >> 1, kWave), vb = __shfl(cb, k >> 1, kWave);
        return (k & 1) ? vb : va;
    };
    const int k0 = 2 * lane, k1 = k0 + 1;
    const int hi0 = k0 + r < L - 1 ? k0 + r : L - 1, hi1 = k1 + r < L - 1 ? k1 + r : L - 1;
    const int lo0 = k0 - r - 1, lo1 = k1 - r - 1;
    const T h0 = at(hi0), h1 = at(hi1);
    const T l0 = at(lo0 > 0 ? lo0 : 0), l1 = at(lo1 > 0 ? lo1 : 0);
    a = lo0 >= 0 ? h0 - l0 : h0;
    b = lo1 >= 0 ? h1 - l1 : h1;
}

// ---- column pass; a block = (plane, strip of 32 columns, ONE quantity q of the stage).
//   STAGE 0: (x, y, xy, xx) -> saved[0..3];  1: (A = saved[2], b = saved[4]) -> (out, saved[4]);
//         2: (g x / N, g / N) -> (grad_x, ws[0]);   3: (grad_x, ws[0], ws[1]) in place.
template <int STAGE>
struct GfStageQ { static constexpr int value = STAGE == 0 ? 4 : (STAGE == 3 ? 3 : 2); };

template <typename T, int STAGE>
__global__ void __launch_bounds__(kBlock)
gf_cols_kernel(const GfArgs<T> a, int strips) {
    constexpr int Q = GfStageQ<STAGE>::value;
    constexpr int NW = kBlock / kWave;
    constexpr int LPW = kGfStrip / NW;                          // lines per wave
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* tile = reinterpret_cast<T*>(smem_raw);                  // [kGfStrip][P], a column is a contiguous line
    const int H = a.H, W = a.W;
    const int P = (H | 1) + 1;                                 // 130 = 2 (mod 64) for H = 128: the transposing writes of a wave (32 columns x 2 rows) hit 64 banks
    unsigned bid = blockIdx.x;
    const int q = bid % Q;
    bid /= Q;
    const int strip = bid % strips;
    const int64_t plane = bid / strips;
    const size_t S = static_cast<size_t>(a.planes) * H * W;    // plane-set stride
    const size_t base = static_cast<size_t>(plane) * H * W;
    const int cc = threadIdx.x & (kGfStrip - 1), rr = threadIdx.x / kGfStrip;
    const int j = strip * kGfStrip + cc;
    const T* src = nullptr;
    T* dst = nullptr;
    if constexpr (STAGE == 0) dst = a.saved + q * S;
    if constexpr (STAGE == 1) { src = a.saved + (q == 0 ? 2 : 4) * S; dst = q == 0 ? a.out : a.saved + 4 * S; }
    if constexpr (STAGE == 2) dst = q == 0 ? a.out : a.ws;
    if constexpr (STAGE == 3) { dst = q == 0 ? a.out : a.ws + (q - 1) * S; src = dst; }

    // all of the thread's loads are issued before the first use (a rolled loop paid one memory round trip per row)
    constexpr int RPP = kBlock / kGfStrip;                      // rows per pass
    constexpr int NIT = kGfMaxDim / RPP;
    T v[NIT], v2[NIT];
    const bool inj = j < W;
    auto load_all = [&](const T* ptr, T (&dstv)[NIT]) {
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            const int i = rr + k * RPP;
            dstv[k] = (inj && i < H) ? ptr[base + static_cast<size_t>(i) * W + j] : static_cast<T>(0);
        }
    };
    if constexpr (STAGE == 0) {
        if (q == 1) {
            load_all(a.y, v);
        } else {
            load_all(a.x, v);
            if (q == 2) {
                load_all(a.y, v2);
#pragma unroll
                for (int k = 0; k < NIT; ++k) v[k] = v[k] * v2[k];
            } else if (q == 3) {
#pragma unroll
                for (int k = 0; k < NIT; ++k) v[k] = v[k] * v[k];
            }
        }
    } else if constexpr (STAGE == 2) {
        load_all(a.g, v);
        if (q == 0) load_all(a.x, v2);
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            const T n = box_count<T>(rr + k * RPP, inj ? j : 0, H, W, a.r);
            v[k] = q == 0 ? v[k] * v2[k] / n : v[k] / n;                             // d mean_A -> d A, d mean_b -> d b
        }
    } else {
        load_all(src, v);
    }
    if (inj) {
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            const int i = rr + k * RPP;
            if (i < H) tile[cc * P + i] = v[k];
        }
    }
    __syncthreads();
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    const int ncol = (W - strip * kGfStrip) < kGfStrip ? (W - strip * kGfStrip) : kGfStrip;
    const int i0 = 2 * lane;
    T va[LPW], vb[LPW];
#pragma unroll
    for (int k = 0; k < LPW; ++k) {                              // the wave's lines side by side: their shuffle chains interleave
        const T* ln = tile + (wave + k * NW) * P;
        const bool live = wave + k * NW < ncol;
        va[k] = live && i0 < H ? ln[i0] : static_cast<T>(0);
        vb[k] = live && i0 + 1 < H ? ln[i0 + 1] : static_cast<T>(0);
    }
#pragma unroll
    for (int k = 0; k < LPW; ++k) line_box(va[k], vb[k], lane, H, a.r);
#pragma unroll
    for (int k = 0; k < LPW; ++k) {
        T* ln = tile + (wave + k * NW) * P;
        if (wave + k * NW < ncol) {
            if (i0 < H) ln[i0] = va[k];
            if (i0 + 1 < H) ln[i0 + 1] = vb[k];
        }
    }
    __syncthreads();
    if (inj) {
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            const int i = rr + k * RPP;
            if (i < H) dst[base + static_cast<size_t>(i) * W + j] = tile[cc * P + i];
        }
    }
}

// ---- row pass + the stage's pointwise epilogue; a wave per row.
//   STAGE 0: saved[0..3] (column sums of x, y, xy, xx) -> saved[0..4] = mean_x, mean_y, A, var + eps, b
//         1: (out, saved[4]) (column sums of A, b)      -> saved[4] = mean_A, out = mean_A x + mean_b
//         2: (grad_x, ws[0]) (column sums of g x / N, g / N) -> (grad_x, ws[0], ws[1]) = (d mean_x, d E[xy], d E[xx]) / N
//         3: (grad_x, ws[0], ws[1]) -> grad_x
template <typename T, int STAGE>
__global__ void __launch_bounds__(kBlock)
gf_rows_kernel(const GfArgs<T> a) {
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    const int H = a.H, W = a.W;
    const int64_t row = static_cast<int64_t>(blockIdx.x) * (kBlock / kWave) + wave;
    if (row >= a.planes * H) return;
    const int i = static_cast<int>(row % H);
    const size_t S = static_cast<size_t>(a.planes) * H * W;
    const size_t base = static_cast<size_t>(row) * W;              // (plane * H + i) * W
    const int j0 = 2 * lane, j1 = j0 + 1;
    const bool in0 = j0 < W, in1 = j1 < W;
    const size_t p0 = base + (in0 ? j0 : 0), p1 = base + (in1 ? j1 : 0);
    auto box_of = [&](const T* src, T& s0, T& s1) {
        s0 = in0 ? src[p0] : static_cast<T>(0);
        s1 = in1 ? src[p1] : static_cast<T>(0);
        line_box(s0, s1, lane, W, a.r);
    };
    const T n0 = box_count<T>(i, in0 ? j0 : 0, H, W, a.r), n1 = box_count<T>(i, in1 ? j1 : 0, H, W, a.r);
    if constexpr (STAGE == 0) {
        T bx0, bx1, by0, by1, bxy0, bxy1, bxx0, bxx1;
        box_of(a.saved, bx0, bx1);
        box_of(a.saved + S, by0, by1);
        box_of(a.saved + 2 * S, bxy0, bxy1);
        box_of(a.saved + 3 * S, bxx0, bxx1);
        auto fin = [&](size_t p, T n, T bx, T by, T bxy, T bxx) {
            const T mx = bx / n, my = by / n;
            const T cov = bxy / n - mx * my;
            const T ve = (bxx / n - mx * mx) + a.eps;               // var_x + eps
            const T A = cov / ve;
            a.saved[p] = mx;
            a.saved[S + p] = my;
            a.saved[2 * S + p] = A;
            a.saved[3 * S + p] = ve;
            a.saved[4 * S + p] = my - A * mx;                       // b
        };
        if (in0) fin(p0, n0, bx0, by0, bxy0, bxx0);
        if (in1) fin(p1, n1, bx1, by1, bxy1, bxx1);
    } else if constexpr (STAGE == 1) {
        T bA0, bA1, bb0, bb1;
        box_of(a.out, bA0, bA1);
        box_of(a.saved + 4 * S, bb0, bb1);
        if (in0) { const T mA = bA0 / n0; a.saved[4 * S + p0] = mA; a.out[p0] = mA * a.x[p0] + bb0 / n0; }
        if (in1) { const T mA = bA1 / n1; a.saved[4 * S + p1] = mA; a.out[p1] = mA * a.x[p1] + bb1 / n1; }
    } else if constexpr (STAGE == 2) {
        T q10, q11, q20, q21;
        box_of(a.out, q10, q11);
        box_of(a.ws, q20, q21);
        auto fin = [&](size_t p, T n, T q1, T q2) {
            const T mx = a.saved[p], my = a.saved[S + p], A = a.saved[2 * S + p], ve = a.saved[3 * S + p];
            const T gA = q1 - q2 * mx;                              // b = mean_y - A mean_x
            T gmx = -q2 * A;
            const T gcov = gA / ve;                                 // A = cov / (var + eps)
            const T gvar = -gA * A / ve;
            gmx += -gcov * my - 2 * gvar * mx;                      // cov = E[xy] - mx my ; var = E[xx] - mx^2
            a.out[p] = gmx / n;
            a.ws[p] = gcov / n;
            a.ws[S + p] = gvar / n;
        };
        if (in0) fin(p0, n0, q10, q20);
        if (in1) fin(p1, n1, q11, q21);
    } else {
        T r10, r11, r20, r21, r30, r31;
        box_of(a.out, r10, r11);                                    // -> d x through mean_x
        box_of(a.ws, r20, r21);                                     // -> d (x y)
        box_of(a.ws + S, r30, r31);                                 // -> d (x x)
        auto fin = [&](size_t p, T r1, T r2, T r3) {
            T v = r1 + a.g[p] * a.saved[4 * S + p];                 // + g * mean_A
            v += a.y[p] * r2;
            a.out[p] = v + 2 * a.x[p] * r3;
        };
        if (in0) fin(p0, r10, r20, r30);
        if (in1) fin(p1, r11, r21, r31);
    }
}

template <typename T, int STAGE>
void gf_launch_cols(const GfArgs<T>& a, hipStream_t st) {
    constexpr int Q = GfStageQ<STAGE>::value;
    const int strips = (a.W + kGfStrip - 1) / kGfStrip;
    const int P = (a.H | 1) + 1;
    const size_t lds = static_cast<size_t>(kGfStrip) * P * sizeof(T);
    hipLaunchKernelGGL((gf_cols_kernel<T, STAGE>), dim3(static_cast<unsigned>(a.planes * strips * Q)), dim3(kBlock), lds, st, a, strips);
}
template <typename T, int STAGE>
void gf_launch_rows(const GfArgs<T>& a, hipStream_t st) {
    const int64_t rows = a.planes * a.H;
    const int per = kBlock / kWave;
    hipLaunchKernelGGL((gf_rows_kernel<T, STAGE>), dim3(static_cast<unsigned>((rows + per - 1) / per)), dim3(kBlock), 0, st, a);
}

template <typename T>
void gf_forward(const void* x, const void* y, void* out, void* saved, int64_t planes, int H, int W, int r, double eps, hipStream_t st) {
    GfArgs<T> a{static_cast<const T*>(x), static_cast<const T*>(y), nullptr, static_cast<T*>(saved), static_cast<T*>(out), nullptr,
                planes, H, W, r, static_cast<T>(eps)};
    gf_launch_cols<T, 0>(a, st);
    gf_launch_rows<T, 0>(a, st);
    gf_launch_cols<T, 1>(a, st);
    gf_launch_rows<T, 1>(a, st);
}
template <typename T>
void gf_backward(const void* x, const void* y, const void* saved, const void* g, void* gx, void* ws, int64_t planes, int H, int W, int r,
                 hipStream_t st) {
    GfArgs<T> a{static_cast<const T*>(x), static_cast<const T*>(y), static_cast<const T*>(g), const_cast<T*>(static_cast<const T*>(saved)),
                static_cast<T*>(gx), static_cast<T*>(ws), planes, H, W, r, static_cast<T>(0)};
    gf_launch_cols<T, 2>(a, st);
    gf_launch_rows<T, 2>(a, st);
    gf_launch_cols<T, 3>(a, st);
    gf_launch_rows<T, 3>(a, st);
}

int check_args(const char* fn, int64_t planes, int64_t H, int64_t W, int r, int dtype) {
    FFWM_REQUIRE(dtype_ok(dtype), FFWM_ERR_DTYPE, "%s: dtype %d is not FFWM_F32/FFWM_F64", fn, dtype);
    FFWM_REQUIRE(planes > 0 && H > 0 && W > 0 && r >= 0, FFWM_ERR_ARG, "%s: sizes must be positive", fn);
    FFWM_REQUIRE(H > 2 * r + 1 && W > 2 * r + 1, FFWM_ERR_ARG,
                 "%s: need H > 2r+1 and W > 2r+1 (H=%lld W=%lld r=%d), as the reference asserts", fn, (long long)H,
                 (long long)W, r);
    FFWM_REQUIRE(H <= kGfMaxDim && W <= kGfMaxDim, FFWM_ERR_SIZE,
                 "%s: planes up to %d x %d (a wave holds a line as two elements per lane), got %lld x %lld", fn, kGfMaxDim, kGfMaxDim, (long long)H, (long long)W);
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
    LaunchScope ls("guided_filter_fwd", st, static_cast<double>(esz) * planes * H * W * 8.0);   // x, y in; out + 5 saved planes
    if (dtype == FFWM_F32)
        gf_forward<float>(x, y, output, saved, planes, (int)H, (int)W, r, eps, st);
    else
        gf_forward<double>(x, y, output, saved, planes, (int)H, (int)W, r, eps, st);
    return check_launch(fn);
}

extern "C" int ffwm_guided_filter_backward(const void* x, const void* y, const void* saved, const void* grad_output,
                                           void* grad_x, void* workspace, int64_t planes, int64_t H, int64_t W, int r, int dtype,
                                           void* stream) {
    const char* fn = "ffwm_guided_filter_backward";
    FFWM_REQUIRE(x && y && saved && grad_output && grad_x && workspace, FFWM_ERR_ARG, "%s: NULL tensor pointer", fn);
    if (int rc = check_args(fn, planes, H, W, r, dtype)) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const size_t esz = dtype == FFWM_F32 ? 4 : 8;
    LaunchScope ls("guided_filter_bwd", st, static_cast<double>(esz) * planes * H * W * 9.0);   // x, y, g, 5 saved in; grad out
    if (dtype == FFWM_F32)
        gf_backward<float>(x, y, saved, grad_output, grad_x, workspace, planes, (int)H, (int)W, r, st);
    else
        gf_backward<double>(x, y, saved, grad_output, grad_x, workspace, planes, (int)H, (int)W, r, st);
    return check_launch(fn);
}
