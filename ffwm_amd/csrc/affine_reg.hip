// affine_reg.hip -- AffineRegularizationLoss of FlowNet pre-training as ONE kernel per flow scale.
//
// Reference: /root/reference/models/losses.py:181-223 (called per scale by MultiAffineRegularizationLoss,
// :163-179, from models/flownet_model.py:30-31,67-68).  For each of the two coordinate grids
// grid = (flow + 1) / 2 * 128 it runs
//     results     = conv2d(grid, K^T K as kz^2 filters of kz x kz)        -> [B, kz^2, h', w']
//     kernels_new = LocalAttnReshape(results, kz)                          -> [B, 1, kz h', kz w']
//     grid_H      = BlockExtractor(grid, constant flow kz//2)              -> [B, 1, kz h', kz w']
//     loss        = mean(avg_pool2d(grid_H * kernels_new, kz, kz)) * kz^2
// i.e. for every kz x kz window p of the grid the quadratic form q = p^T (K^T K) p, averaged over the
// windows: 6 launches forward and ~10 backward per grid, through kz^2-fold expanded intermediates
// (the only place the reference really runs block_extractor / local_attn_reshape -- those ops stay
// available on their own, ffwm_amd/losses.py composes them exactly like the reference).
//
// Here: one thread per window.  The grid tile of a 64 x 4 window block (+ kz-1 apron) and the matrix
// M = K^T K are staged in LDS; the thread forms r = M p row by row (M is read as wave-uniform LDS
// broadcasts), accumulates q = p . r, and -- because M is symmetric, dq/dp = 2 r -- scatters the
// gradient of the mean straight into grad_flow (one global atomic per window cell).  The loss is
// reduced per block and added to loss[0] with one atomic.  No intermediate tensor exists.
#include "common.hpp"

namespace ffwm {
namespace {

constexpr int kArTileX = 64, kArTileY = 4;

template <typename T, int KZ>
__global__ void __launch_bounds__(kBlock)
affine_reg_kernel(const T* __restrict__ flow, const T* __restrict__ M, T* __restrict__ loss, T* __restrict__ gflow,
                  int h, int w, int tiles_x, int tiles_y, T grad_scale) {
    constexpr int K2 = KZ * KZ;
    constexpr int SW = kArTileX + KZ - 1, SH = kArTileY + KZ - 1;
    __shared__ T Ms[K2 * K2];
    __shared__ T S[SH * SW];
    __shared__ T red[kBlock / kWave];
    unsigned t = blockIdx.x;
    const int tx = t % tiles_x;
    t /= tiles_x;
    const int ty = t % tiles_y;
    t /= tiles_y;
    const int ch = t & 1;            // 0 = x grid, 1 = y grid
    const int b = t >> 1;
    const int hw = h - KZ + 1, ww = w - KZ + 1;       // windows per column / row
    const T* fp = flow + (static_cast<size_t>(b) * 2 + ch) * h * w;
    const int x0 = tx * kArTileX, y0 = ty * kArTileY;
    for (int i = threadIdx.x; i < K2 * K2; i += kBlock) Ms[i] = M[i];
    for (int i = threadIdx.x; i < SH * SW; i += kBlock) {
        const int r = i / SW, c = i - r * SW;
        const int gy = y0 + r, gx = x0 + c;
        // flow2grid (losses.py:221-223): flow.add(1).div(2).mul(128)
        S[i] = (gy < h && gx < w) ? ((fp[static_cast<size_t>(gy) * w + gx] + 1) / 2) * 128 : static_cast<T>(0);
    }
    __syncthreads();
    const int lx = threadIdx.x & (kArTileX - 1), ly = threadIdx.x / kArTileX;
    const int wx = x0 + lx, wy = y0 + ly;
    const bool live = wx < ww && wy < hw;
    T q = 0;
    if (live) {
        T p[K2];
#pragma unroll
        for (int i = 0; i < KZ; ++i)
#pragma unroll
            for (int j = 0; j < KZ; ++j) p[i * KZ + j] = S[(ly + i) * SW + lx + j];
        T* gp = gflow ? gflow + (static_cast<size_t>(b) * 2 + ch) * h * w : nullptr;
#pragma unroll 1
        for (int a = 0; a < K2; ++a) {
            const T* mrow = Ms + a * K2;
            T r = 0;
#pragma unroll
            for (int c = 0; c < K2; ++c) r += mrow[c] * p[c];
            // p[a] with a dynamic (wave-uniform) index: re-read it from LDS instead of indexing registers
            const int i = a / KZ, j = a - i * KZ;
            const T pa = S[(ly + i) * SW + lx + j];
            q += pa * r;
            // d mean(q) / d grid = 2 r / (#windows); d grid / d flow = 64
            if (gp) atomic_add(gp + static_cast<size_t>(wy + i) * w + (wx + j), r * grad_scale);
        }
    }
    // block sum of q -> one atomic per block
    q = wave_sum(q);
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    if (lane == 0) red[wave] = q;
    __syncthreads();
    if (threadIdx.x == 0) {
        T s = 0;
#pragma unroll
        for (int k = 0; k < kBlock / kWave; ++k) s += red[k];
        atomic_add(loss, s);
    }
}

template <typename T>
int launch(const T* flow, const T* M, T* loss, T* gflow, int64_t B, int64_t h, int64_t w, int kz, double loss_scale,
           hipStream_t st) {
    const int hw = static_cast<int>(h) - kz + 1, ww = static_cast<int>(w) - kz + 1;
    const int tiles_x = (ww + kArTileX - 1) / kArTileX, tiles_y = (hw + kArTileY - 1) / kArTileY;
    const unsigned grid = static_cast<unsigned>(B * 2 * tiles_x * tiles_y);
    // the kernel accumulates sum_w q_w; the caller's loss is loss_scale * sum (loss_scale = 1 / #windows per
    // (b, grid) plane set = 1 / (B h' w')), and the gradient of that w.r.t. flow is 2 r * 64 * loss_scale
    const T gscale = static_cast<T>(2.0 * 64.0 * loss_scale);
    LaunchScope ls("affine_regularization", st, sizeof(T) * static_cast<double>(B) * 2 * h * w * (gflow ? 2.0 : 1.0));
#define FFWM_AR(KK)                                                                                          \
    case KK:                                                                                                 \
        hipLaunchKernelGGL((affine_reg_kernel<T, KK>), dim3(grid), dim3(kBlock), 0, st, flow, M, loss, gflow, \
                           (int)h, (int)w, tiles_x, tiles_y, gscale);                                        \
        break;
    switch (kz) {
        FFWM_AR(3) FFWM_AR(5) FFWM_AR(7)
        default:
            set_error("ffwm_affine_regularization: kernel size %d is not built (3, 5, 7: the sizes the reference uses)", kz);
            return FFWM_ERR_ARG;
    }
#undef FFWM_AR
    return check_launch("ffwm_affine_regularization");
}

}  // namespace
}  // namespace ffwm

using namespace ffwm;

extern "C" int ffwm_affine_regularization(const void* flow, const void* ktk, void* loss_sum, void* grad_flow,
                                          int64_t B, int64_t h, int64_t w, int kernel_size, double loss_scale,
                                          int dtype, void* stream) {
    const char* fn = "ffwm_affine_regularization";
    FFWM_REQUIRE(dtype_ok(dtype), FFWM_ERR_DTYPE, "%s: dtype %d is not FFWM_F32/FFWM_F64", fn, dtype);
    FFWM_REQUIRE(flow && ktk && loss_sum, FFWM_ERR_ARG, "%s: NULL tensor pointer", fn);
    FFWM_REQUIRE(B > 0 && h >= kernel_size && w >= kernel_size && kernel_size >= 1, FFWM_ERR_ARG,
                 "%s: need B > 0 and h, w >= kernel_size (B=%lld h=%lld w=%lld kz=%d)", fn, (long long)B, (long long)h,
                 (long long)w, kernel_size);
    FFWM_REQUIRE(h * w < (1LL << 30) && B * 2 * ((w + 63) / 64) * ((h + 3) / 4) < (1LL << 31), FFWM_ERR_SIZE,
                 "%s: tensor too large", fn);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dtype == FFWM_F32)
        return launch<float>((const float*)flow, (const float*)ktk, (float*)loss_sum, (float*)grad_flow, B, h, w, kernel_size,
                             loss_scale, st);
    return launch<double>((const double*)flow, (const double*)ktk, (double*)loss_sum, (double*)grad_flow, B, h, w,
                          kernel_size, loss_scale, st);
}
