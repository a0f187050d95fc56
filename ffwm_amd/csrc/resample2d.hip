// resample2d.hip -- Gaussian-weighted kernel_size x kernel_size flow resampler for gfx950.
//
// Replaces the three kernels of /root/reference/cuda/resample2d_package/resample2d_kernel.cu:
//   K1 kernel_resample2d_update_output   :21-95   (one thread per output element; dx/dy/sigma and
//                                                  4*(ks/2)^2 exp() recomputed for every channel)
//   K2 kernel_resample2d_backward_input1 :98-202  (same redundancy, 4*(ks/2)^2 atomics per element)
//   K3 kernel_resample2d_backward_input2 :204-330 (3 threads per pixel, each looping C x taps TWICE
//                                                  with 8 gathers per (channel, tap))
//
// Here every kernel is one thread per PIXEL looping over channels: the tap indices and the
// Gaussian weights (double-precision exp, as the reference's SAFE_DIV promotion implies) are
// computed once per pixel.  K3 produces all three gradients (dx, dy, sigma) from ONE pass over the
// channels: the four waves of a block split the channels, keep one accumulator per tap
// (sum_ch gO * in1[tap]) and combine them through LDS; the quotient-rule terms are applied once at
// the end.  ks in {2, 4, 6} (HALF = 1..3) is templated; any other even ks falls back to a literal
// per-element kernel.
#include <memory>

#include "common.hpp"
#include <type_traits>

namespace ffwm {
namespace {

// exp(SAFE_DIV(-v*v, 2*sigma*sigma)) -> T  (resample2d_kernel.cu:72-75); the macro's conditional
// has type double, so it is the double exp() that runs in both instantiations.
template <typename T>
__device__ __forceinline__ T gauss(T v, T sigma) {
    const T num = -v * v;
    const T den = 2 * sigma * sigma;
    return static_cast<T>(exp(safe_div<T>(num, den)));
}

template <typename T, int HALF>
struct RsTaps {
    static constexpr int N = 2 * HALF;   // entry 2f = "left/top" tap f, 2f+1 = "right/bottom" tap f
    unsigned col[N], row[N];             // byte offsets
    T wx[N], wy[N];                      // xL_P, xR_P / yT_P, yB_P
    T dx_[N], dy_[N];                    // xL_, xR_ / yT_, yB_ (distances)
    T sum;
};

// `trunc_alpha`: the reference's d_input1 kernel forms alpha/beta with C truncation, int(xf)
// (:137-138), everything else uses floor.
template <typename T, int HALF>
__device__ __forceinline__ void make_rs_taps(RsTaps<T, HALF>& t, T dx, T dy, T sigma, int x, int y,
                                             int Hi, int Wi, int dil, bool trunc_alpha) {
    const T xf = static_cast<T>(x) + dx;
    const T yf = static_cast<T>(y) + dy;
    const T flx = floor_t(xf), fly = floor_t(yf);
    T alpha, beta;
    if (trunc_alpha) {
        alpha = xf - static_cast<T>(clamp_index_wide(xf));
        beta = yf - static_cast<T>(clamp_index_wide(yf));
    } else {
        alpha = xf - flx;
        beta = yf - fly;
    }
    constexpr unsigned E = sizeof(T);
#pragma unroll
    for (int f = 0; f < HALF; ++f) {
        t.col[2 * f] = static_cast<unsigned>(clamp_index(flx - static_cast<T>(f * dil), Wi)) * E;
        t.col[2 * f + 1] = static_cast<unsigned>(clamp_index(flx + static_cast<T>((f + 1) * dil), Wi)) * E;
        t.row[2 * f] = static_cast<unsigned>(clamp_index(fly - static_cast<T>(f * dil), Hi)) * static_cast<unsigned>(Wi) * E;
        t.row[2 * f + 1] = static_cast<unsigned>(clamp_index(fly + static_cast<T>((f + 1) * dil), Hi)) * static_cast<unsigned>(Wi) * E;
        t.dx_[2 * f] = static_cast<T>(f * dil) + alpha;
        t.dx_[2 * f + 1] = static_cast<T>((1. + f) * dil) - alpha;
        t.dy_[2 * f] = static_cast<T>(f * dil) + beta;
        t.dy_[2 * f + 1] = static_cast<T>((1. + f) * dil) - beta;
        t.wx[2 * f] = gauss<T>(t.dx_[2 * f], sigma);
        t.wx[2 * f + 1] = gauss<T>(t.dx_[2 * f + 1], sigma);
        t.wy[2 * f] = gauss<T>(t.dy_[2 * f], sigma);
        t.wy[2 * f + 1] = gauss<T>(t.dy_[2 * f + 1], sigma);
    }
    T sum = 0;
#pragma unroll
    for (int fy = 0; fy < HALF; ++fy)
#pragma unroll
        for (int fx = 0; fx < HALF; ++fx) {
            const T yT = t.wy[2 * fy], yB = t.wy[2 * fy + 1], xL = t.wx[2 * fx], xR = t.wx[2 * fx + 1];
            sum += (yT * xL + yT * xR + yB * xL + yB * xR);   // :87
        }
    t.sum = sum;
}

// ------------------------------------------------------------------------------------ K1
template <typename T, int HALF>
__global__ void __launch_bounds__(kBlock)
rs_fwd_kernel(const T* __restrict__ in1, const T* __restrict__ in2, T* __restrict__ out, int C,
              int Hi, int Wi, int H, int W, int dil, int tiles_x, int tiles_y, int cslabs, int cs,
              int remap) {
    const TileCoord tc = decode_tile(tiles_x, tiles_y, cslabs, remap);
    const int x = tc.xf, y = tc.yf;
    if (x >= W || y >= H) return;
    const size_t plane = static_cast<size_t>(H) * W;
    const size_t poff = static_cast<size_t>(y) * W + x;
    const T* f = in2 + static_cast<size_t>(tc.b) * 3 * plane + poff;
    RsTaps<T, HALF> t;
    make_rs_taps<T, HALF>(t, f[0], f[plane], f[2 * plane], x, y, Hi, Wi, dil, false);

    const int c0 = tc.slab * cs;
    const int c1 = (c0 + cs < C) ? c0 + cs : C;
    const size_t iplane = static_cast<size_t>(Hi) * Wi;
    const unsigned ibytes = static_cast<unsigned>(iplane * sizeof(T));
    const T* ip = in1 + (static_cast<size_t>(tc.b) * C + c0) * iplane;
    T* op = out + (static_cast<size_t>(tc.b) * C + c0) * plane + poff;
    for (int c = c0; c < c1; ++c, ip += iplane, op += plane) {
        const rsrc_t r = make_rsrc(ip, ibytes);
        T val = 0;
#pragma unroll
        for (int fy = 0; fy < HALF; ++fy)
#pragma unroll
            for (int fx = 0; fx < HALF; ++fx) {
                const T yT = t.wy[2 * fy], yB = t.wy[2 * fy + 1], xL = t.wx[2 * fx], xR = t.wx[2 * fx + 1];
                const unsigned rT = t.row[2 * fy], rB = t.row[2 * fy + 1];
                const unsigned cL = t.col[2 * fx], cR = t.col[2 * fx + 1];
                val += yT * xL * buf_ld<T>(r, rT + cL);   // :82-85
                val += yT * xR * buf_ld<T>(r, rT + cR);
                val += yB * xL * buf_ld<T>(r, rB + cL);
                val += yB * xR * buf_ld<T>(r, rB + cR);
            }
        *op = static_cast<T>(safe_div<T>(val, t.sum));    // :93
    }
}

// ------------------------------------------------------------------------------------ K1, LDS-staged (fp32)
// The direct kernel above is bound by its per-lane gathers: 4*HALF^2 buffer loads per pixel and channel, every
// lane on its own cache line when the flow is not smooth (TA-bound: 0.63 TB/s at [8,64,512,512]).  Here
//   * a block owns a 64 x (4*RPT) pixel tile and ALL channels of its slab: the Gaussian weights (double exp, as the
//     reference's SAFE_DIV promotion implies) and tap positions are formed once per pixel and kept in registers;
//   * the block reduces the bounding box of its taps in UNCLAMPED source coordinates (wave shuffles + one LDS hop).
//     If every pixel is regular (finite, |coordinate| < 2^20) and the box fits BOXH x BOXW, the clamp-extended box
//     of FOUR channels at a time is staged in LDS with row-contiguous coalesced loads, channel-interleaved: a cell
//     is one float4 = the same source pixel of 4 channels.  A pixel's (2*HALF)^2 neighbourhood is then a dense
//     square at one LDS base + immediates, and ONE ds_read_b128 (256 B/clk/CU, the wide-read rate) feeds four
//     channel accumulators -- 4x fewer LDS instructions and half the LDS cycles per byte of per-channel ds_read_b32;
//   * the box of channel group g+1 is requested before group g is processed and committed after it (two buffers,
//     one barrier per group);
//   * otherwise the block falls back to direct gathers (block-uniform, data-dependent).
// Per tap the reference's arithmetic is kept: w = yP * xP (one rounding), val = fma(w, in, val) -- nvcc's default
// -fmad contraction of `val += yP * xP * in` (resample2d_kernel.cu:82-85) -- taps in the reference's order.
constexpr int kRsBoxW = 80;

template <int HALF, int RPT, bool DB>
__global__ void __launch_bounds__(kBlock)
rs_fwd_lds_kernel(const float* __restrict__ in1, const float* __restrict__ in2, float* __restrict__ out, int C,
                  int Hi, int Wi, int H, int W, int tiles_x, int tiles_y, int cslabs, int cs, int remap, int ablate) {
    constexpr int NW = kBlock / kWave;
    constexpr int NT = 2 * HALF;                 // taps per axis
    constexpr int TH = NW * RPT;                 // tile rows
    constexpr int BOXH = TH + 12;
    constexpr int NCELL = BOXH * kRsBoxW;
    constexpr int NI = (NCELL + kBlock - 1) / kBlock;
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    f32x4* tile = reinterpret_cast<f32x4*>(smem_raw);          // [DB ? 2 : 1][NCELL]
    __shared__ int red[4][NW];
    __shared__ int flag;

    unsigned tid = xcd_remap(blockIdx.x, gridDim.x, remap);
    const int tx = tid % tiles_x;
    tid /= tiles_x;
    const int ty = tid % tiles_y;
    tid /= tiles_y;
    const int slab = tid % cslabs;
    const int b = tid / cslabs;
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    const int x_raw = tx * kTileX + lane;
    const bool inx = x_raw < W;
    const int x = inx ? x_raw : W - 1;               // out-of-tile lanes shadow a valid pixel, never store
    if (threadIdx.x == 0) flag = 0;

    // ---- per-pixel taps and weights (resample2d_kernel.cu:47-80), once for all channels
    float w[RPT][NT * NT];                           // product weights, [row position][col position]
    float sum[RPT];
    int u0[RPT], v0[RPT], ys[RPT];
    bool iny[RPT];
    bool regular = true;
    const size_t plane = static_cast<size_t>(H) * W;
    const float* fb = in2 + static_cast<size_t>(b) * 3 * plane;
    int umin = 0x7fffffff, umax = -0x7fffffff, vmin = 0x7fffffff, vmax = -0x7fffffff;
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
        const int yraw = ty * TH + wave + r * NW;
        iny[r] = yraw < H;
        const int y = iny[r] ? yraw : H - 1;
        ys[r] = y;
        const size_t poff = static_cast<size_t>(y) * W + x;
        RsTaps<float, HALF> t;
        const float dx = fb[poff], dy = fb[plane + poff], sg = fb[2 * plane + poff];
        make_rs_taps<float, HALF>(t, dx, dy, sg, x, y, Hi, Wi, 1, false);
        sum[r] = t.sum;
        const float flx = floor_t(static_cast<float>(x) + dx), fly = floor_t(static_cast<float>(y) + dy);
        const float lim = static_cast<float>(1 << 20);
        const bool ok = (flx > -lim) && (flx < lim) && (fly > -lim) && (fly < lim);      // also rejects NaN
        regular = regular && ok;
        u0[r] = ok ? static_cast<int>(flx) - (HALF - 1) : 0;
        v0[r] = ok ? static_cast<int>(fly) - (HALF - 1) : 0;
        umin = min(umin, u0[r]); umax = max(umax, u0[r] + NT - 1);
        vmin = min(vmin, v0[r]); vmax = max(vmax, v0[r] + NT - 1);
        // position p <-> tap: p = HALF-1-f is the "left/top" tap f, p = HALF+f the "right/bottom" tap f
        float wxp[NT], wyp[NT];
#pragma unroll
        for (int f = 0; f < HALF; ++f) {
            wxp[HALF - 1 - f] = t.wx[2 * f]; wxp[HALF + f] = t.wx[2 * f + 1];
            wyp[HALF - 1 - f] = t.wy[2 * f]; wyp[HALF + f] = t.wy[2 * f + 1];
        }
#pragma unroll
        for (int pr = 0; pr < NT; ++pr)
#pragma unroll
            for (int pc = 0; pc < NT; ++pc) w[r][pr * NT + pc] = wyp[pr] * wxp[pc];
    }

    // ---- block-wide bounding box of the taps, unclamped coordinates
    umin = wave_min(umin); umax = wave_max(umax); vmin = wave_min(vmin); vmax = wave_max(vmax);
    if (lane == 0) { red[0][wave] = umin; red[1][wave] = umax; red[2][wave] = vmin; red[3][wave] = vmax; }
    __syncthreads();
    if (!regular) flag = 1;            // benign race: every writer stores 1
#pragma unroll
    for (int k = 0; k < NW; ++k) {
        umin = min(umin, red[0][k]); umax = max(umax, red[1][k]);
        vmin = min(vmin, red[2][k]); vmax = max(vmax, red[3][k]);
    }
    __syncthreads();
    const int bw = umax - umin + 1, bh = vmax - vmin + 1;
    const bool use_lds = (flag == 0) && bw <= kRsBoxW && bh <= BOXH;

    const int c0 = slab * cs;
    const int c1 = (c0 + cs < C) ? c0 + cs : C;
    const size_t iplane = static_cast<size_t>(Hi) * Wi;
    const unsigned ibytes = static_cast<unsigned>(iplane * sizeof(float));
    const unsigned obytes = static_cast<unsigned>(plane * sizeof(float));
    const float* ip = in1 + (static_cast<size_t>(b) * C + c0) * iplane;
    float* op = out + (static_cast<size_t>(b) * C + c0) * plane;
    unsigned obase[RPT];
#pragma unroll
    for (int r = 0; r < RPT; ++r)
        obase[r] = (inx && iny[r] && !(ablate & 2)) ? (static_cast<unsigned>(ys[r]) * W + static_cast<unsigned>(x)) * 4u : 0xFFFFFFF0u;  // OOB store is dropped

    // the reference's tap order: for fy, fx: TL, TR, BL, BR
    auto for_each_tap = [&](auto&& body) {
#pragma unroll
        for (int fy = 0; fy < HALF; ++fy)
#pragma unroll
            for (int fx = 0; fx < HALF; ++fx) {
                body(HALF - 1 - fy, HALF - 1 - fx);
                body(HALF - 1 - fy, HALF + fx);
                body(HALF + fy, HALF - 1 - fx);
                body(HALF + fy, HALF + fx);
            }
    };

    if (use_lds) {
        // staging map: thread t copies box cells t, t + 256, ...; cell i = (i / BOXW, i % BOXW) holds
        // in1[clamp(vmin + r)][clamp(umin + c)] of 4 channels.
        unsigned goff[NI];
#pragma unroll
        for (int k = 0; k < NI; ++k) {
            const int i = threadIdx.x + k * kBlock;
            const int r = i / kRsBoxW, cc = i - r * kRsBoxW;
            const int gy = min(max(vmin + r, 0), Hi - 1), gx = min(max(umin + cc, 0), Wi - 1);
            goff[k] = (r < bh && cc < bw && !(ablate & 1)) ? (static_cast<unsigned>(gy) * Wi + gx) * 4u : 0xFFFFFFF0u;   // OOB reads 0
        }
        f32x4 stage[NI];
        auto fetch = [&](int c) {                   // channels c .. c+3 (missing ones read 0)
            const float* p0 = ip + static_cast<size_t>(c - c0) * iplane;
            const rsrc_t r0 = make_rsrc(p0, ibytes);
            const rsrc_t r1 = make_rsrc(p0 + iplane, c + 1 < c1 ? ibytes : 0u);
            const rsrc_t r2 = make_rsrc(p0 + 2 * iplane, c + 2 < c1 ? ibytes : 0u);
            const rsrc_t r3 = make_rsrc(p0 + 3 * iplane, c + 3 < c1 ? ibytes : 0u);
#pragma unroll
            for (int k = 0; k < NI; ++k) {
                stage[k].x = buf_ld<float>(r0, goff[k]);
                stage[k].y = buf_ld<float>(r1, goff[k]);
                stage[k].z = buf_ld<float>(r2, goff[k]);
                stage[k].w = buf_ld<float>(r3, goff[k]);
            }
        };
        auto commit = [&](f32x4* buf) {
#pragma unroll
            for (int k = 0; k < NI; ++k) {
                const int i = threadIdx.x + k * kBlock;
                if (i < NCELL) buf[i] = stage[k];
            }
        };
        int lbase[RPT];
        float inv[RPT];
        bool tame = true;               // every sum of this wave's pixels is a normal, comfortably scaled float
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            lbase[r] = (v0[r] - vmin) * kRsBoxW + (u0[r] - umin);
            inv[r] = 1.0f / sum[r];
            tame = tame && (sum[r] >= 1e-30f) && (sum[r] <= 1e30f);
        }
        const bool fast_div = __builtin_amdgcn_ballot_w64(!tame) == 0;      // wave-uniform
        fetch(c0);
        commit(tile);
        __syncthreads();
        int p = 0;
        for (int c = c0; c < c1; c += 4) {
            const bool more = c + 4 < c1;
            if (more) fetch(c + 4);
            const f32x4* tb = tile + p * NCELL;
            const float* o0 = op + static_cast<size_t>(c - c0) * plane;
            const rsrc_t ro0 = make_rsrc(o0, obytes);
            const rsrc_t ro1 = make_rsrc(o0 + plane, c + 1 < c1 ? obytes : 0u);
            const rsrc_t ro2 = make_rsrc(o0 + 2 * plane, c + 2 < c1 ? obytes : 0u);
            const rsrc_t ro3 = make_rsrc(o0 + 3 * plane, c + 3 < c1 ? obytes : 0u);
#pragma unroll
            for (int r = 0; r < RPT; ++r) {
                const f32x4* nb = tb + lbase[r];
                f32x2 a01 = {0.f, 0.f}, a23 = {0.f, 0.f};       // v_pk_fma_f32: two channels per instruction
                for_each_tap([&](int pr, int pc) {
                    const f32x4 v = nb[pr * kRsBoxW + pc];
                    const float wt = w[r][pr * NT + pc];
                    const f32x2 ww = {wt, wt};
                    a01 = __builtin_elementwise_fma(ww, v.xy, a01);
                    a23 = __builtin_elementwise_fma(ww, v.zw, a23);
                });
                const float s = sum[r];
                float q0, q1, q2, q3;
                if (fast_div) {
                    // a / s with ONE IEEE reciprocal per pixel: q = RN(a * y), r = a - q * s (exact, fma), q' = RN(q + r * y)
                    // is the correctly rounded quotient when y = RN(1 / s) (Markstein); 3 instructions instead of the
                    // 12-instruction division sequence per output
                    const float y = inv[r];
                    q0 = a01.x * y; q1 = a01.y * y; q2 = a23.x * y; q3 = a23.y * y;
                    q0 = __builtin_fmaf(__builtin_fmaf(-q0, s, a01.x), y, q0);
                    q1 = __builtin_fmaf(__builtin_fmaf(-q1, s, a01.y), y, q1);
                    q2 = __builtin_fmaf(__builtin_fmaf(-q2, s, a23.x), y, q2);
                    q3 = __builtin_fmaf(__builtin_fmaf(-q3, s, a23.y), y, q3);
                } else {
                    q0 = static_cast<float>(safe_div<float>(a01.x, s)); q1 = static_cast<float>(safe_div<float>(a01.y, s));
                    q2 = static_cast<float>(safe_div<float>(a23.x, s)); q3 = static_cast<float>(safe_div<float>(a23.y, s));
                }
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, q0), ro0, obase[r], 0, 0);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, q1), ro1, obase[r], 0, 0);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, q2), ro2, obase[r], 0, 0);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, q3), ro3, obase[r], 0, 0);
            }
            if (more) {
                if constexpr (DB) {
                    commit(tile + (p ^ 1) * NCELL);
                    __syncthreads();
                    p ^= 1;
                } else {                 // one buffer: other blocks of the CU cover the two barriers
                    __syncthreads();
                    commit(tile);
                    __syncthreads();
                }
            }
        }
        return;
    }

    // ---- fallback: direct gathers (flow wider than the box / irregular pixel); the clamped tap offsets are
    // re-derived here instead of being carried through the LDS path's registers
    unsigned col[RPT][NT], row[RPT][NT];             // clamped byte offsets, position order
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
        const size_t poff = static_cast<size_t>(ys[r]) * W + x;
        const float fxv = static_cast<float>(x) + fb[poff], fyv = static_cast<float>(ys[r]) + fb[plane + poff];
        const float flx = floor_t(fxv), fly = floor_t(fyv);
#pragma unroll
        for (int f = 0; f < HALF; ++f) {
            col[r][HALF - 1 - f] = static_cast<unsigned>(clamp_index(flx - static_cast<float>(f), Wi)) * 4u;
            col[r][HALF + f] = static_cast<unsigned>(clamp_index(flx + static_cast<float>(f + 1), Wi)) * 4u;
            row[r][HALF - 1 - f] = static_cast<unsigned>(clamp_index(fly - static_cast<float>(f), Hi)) * static_cast<unsigned>(Wi) * 4u;
            row[r][HALF + f] = static_cast<unsigned>(clamp_index(fly + static_cast<float>(f + 1), Hi)) * static_cast<unsigned>(Wi) * 4u;
        }
    }
    for (int c = c0; c < c1; ++c, ip += iplane, op += plane) {
        const rsrc_t rs = make_rsrc(ip, ibytes);
        const rsrc_t ro = make_rsrc(op, obytes);
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            float val = 0;
            for_each_tap([&](int pr, int pc) {
                val = __builtin_fmaf(w[r][pr * NT + pc], buf_ld<float>(rs, row[r][pr] + col[r][pc]), val);
            });
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, static_cast<float>(safe_div<float>(val, sum[r]))), ro, obase[r], 0, 0);
        }
    }
}

// ------------------------------------------------------------------------------------ K2
template <typename T, int HALF>
__global__ void __launch_bounds__(kBlock)
rs_bwd1_kernel(const T* __restrict__ in2, const T* __restrict__ gout, T* __restrict__ gin1, int C,
               int Hi, int Wi, int H, int W, int dil, int quirk, int tiles_x, int tiles_y,
               int cslabs, int cs, int remap) {
    const TileCoord tc = decode_tile(tiles_x, tiles_y, cslabs, remap);
    const int x = tc.xf, y = tc.yf;
    if (x >= W || y >= H) return;
    const size_t plane = static_cast<size_t>(H) * W;
    const size_t poff = static_cast<size_t>(y) * W + x;
    const T* f = in2 + static_cast<size_t>(tc.b) * 3 * plane + poff;
    RsTaps<T, HALF> t;
    make_rs_taps<T, HALF>(t, f[0], f[plane], f[2 * plane], x, y, Hi, Wi, dil, quirk != 0);

    // normalised tap weights SAFE_DIV(w, sum) (:196-199), once per pixel
    T wn[HALF][HALF][4];
#pragma unroll
    for (int fy = 0; fy < HALF; ++fy)
#pragma unroll
        for (int fx = 0; fx < HALF; ++fx) {
            const T yT = t.wy[2 * fy], yB = t.wy[2 * fy + 1], xL = t.wx[2 * fx], xR = t.wx[2 * fx + 1];
            wn[fy][fx][0] = static_cast<T>(safe_div<T>(yT * xL, t.sum));
            wn[fy][fx][1] = static_cast<T>(safe_div<T>(yT * xR, t.sum));
            wn[fy][fx][2] = static_cast<T>(safe_div<T>(yB * xL, t.sum));
            wn[fy][fx][3] = static_cast<T>(safe_div<T>(yB * xR, t.sum));
        }

    const int c0 = tc.slab * cs;
    const int c1 = (c0 + cs < C) ? c0 + cs : C;
    const size_t iplane = static_cast<size_t>(Hi) * Wi;
    T* gp = gin1 + (static_cast<size_t>(tc.b) * C + c0) * iplane;
    const T* op = gout + (static_cast<size_t>(tc.b) * C + c0) * plane + poff;
    for (int c = c0; c < c1; ++c, gp += iplane, op += plane) {
        const T g = *op;
#pragma unroll
        for (int fy = 0; fy < HALF; ++fy)
#pragma unroll
            for (int fx = 0; fx < HALF; ++fx) {
                const unsigned rT = t.row[2 * fy], rB = t.row[2 * fy + 1];
                const unsigned cL = t.col[2 * fx], cR = t.col[2 * fx + 1];
                atomic_add_off(gp, rT + cL, wn[fy][fx][0] * g);
                atomic_add_off(gp, rT + cR, wn[fy][fx][1] * g);
                atomic_add_off(gp, rB + cL, wn[fy][fx][2] * g);
                atomic_add_off(gp, rB + cR, wn[fy][fx][3] * g);
            }
    }
}

// K2 without global atomics: a block owns `cg` whole (b, c) planes of grad_input1 in LDS, visits
// every pixel of image b once -- taps and normalised weights are formed once per pixel and reused for
// the cg channels -- accumulates with LDS atomics and adds the finished planes to grad_input1 with
// plain coalesced stores.  The LDS accumulator is DOUBLE whatever T is: on gfx950 ds_add_f64 retires a
// wave in ~9 clk where ds_add_f32 needs ~190 (tools/ubench/atomics.hip); rounded to T once, at the end.
constexpr int kPlaneThreads = 1024;

template <typename T, int HALF>
__global__ void __launch_bounds__(kPlaneThreads)
rs_bwd1_plane_kernel(const T* __restrict__ in2, const T* __restrict__ gout, T* __restrict__ gin1, int C,
                     int Hi, int Wi, int H, int W, int dil, int quirk, int cg, int groups, int nsplit) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    double* acc = reinterpret_cast<double*>(smem_raw);
    unsigned bid = blockIdx.x;
    const int split = bid % nsplit;          // with few planes the pixels are split over nsplit blocks per plane group
    bid /= nsplit;
    const int grp = bid % groups;
    const int b = bid / groups;
    const int c0 = grp * cg;
    const int nc = (c0 + cg <= C) ? cg : C - c0;
    const int ncell = Hi * Wi, npix = H * W;
    for (int i = threadIdx.x; i < nc * ncell; i += kPlaneThreads) acc[i] = 0;
    __syncthreads();
    const T* f = in2 + static_cast<size_t>(b) * 3 * npix;
    const T* g0 = gout + (static_cast<size_t>(b) * C + c0) * npix;
    for (int p = split * kPlaneThreads + threadIdx.x; p < npix; p += kPlaneThreads * nsplit) {
        const int y = p / W, x = p - y * W;
        RsTaps<T, HALF> t;
        make_rs_taps<T, HALF>(t, f[p], f[npix + p], f[2 * npix + p], x, y, Hi, Wi, dil, quirk != 0);
        constexpr unsigned E = sizeof(T);
#pragma unroll
        for (int fy = 0; fy < HALF; ++fy)
#pragma unroll
            for (int fx = 0; fx < HALF; ++fx) {
                const T yT = t.wy[2 * fy], yB = t.wy[2 * fy + 1], xL = t.wx[2 * fx], xR = t.wx[2 * fx + 1];
                const T w0 = static_cast<T>(safe_div<T>(yT * xL, t.sum)), w1 = static_cast<T>(safe_div<T>(yT * xR, t.sum));
                const T w2 = static_cast<T>(safe_div<T>(yB * xL, t.sum)), w3 = static_cast<T>(safe_div<T>(yB * xR, t.sum));
                const unsigned o0 = (t.row[2 * fy] + t.col[2 * fx]) / E, o1 = (t.row[2 * fy] + t.col[2 * fx + 1]) / E;
                const unsigned o2 = (t.row[2 * fy + 1] + t.col[2 * fx]) / E, o3 = (t.row[2 * fy + 1] + t.col[2 * fx + 1]) / E;
                for (int c = 0; c < nc; ++c) {
                    const T g = g0[static_cast<size_t>(c) * npix + p];
                    double* a = acc + c * ncell;
                    lds_add(a + o0, w0 * g);
                    lds_add(a + o1, w1 * g);
                    lds_add(a + o2, w2 * g);
                    lds_add(a + o3, w3 * g);
                }
            }
    }
    __syncthreads();
    T* dst = gin1 + (static_cast<size_t>(b) * C + c0) * ncell;
    if (nsplit == 1) {
        for (int i = threadIdx.x; i < nc * ncell; i += kPlaneThreads) dst[i] += static_cast<T>(acc[i]);
    } else {                                 // several blocks share the planes: one global atomic per non-zero cell
        for (int i = threadIdx.x; i < nc * ncell; i += kPlaneThreads) {
            const T v = static_cast<T>(acc[i]);
            if (v != 0) atomic_add(dst + i, v);
        }
    }
}

// ------------------------------------------------------------------------------------ K3
// Block = 64 consecutive x of one row (one lane per pixel) x 4 waves that split the channels.
template <typename T, int HALF>
__global__ void __launch_bounds__(kBlock)
rs_bwd2_kernel(const T* __restrict__ in1, const T* __restrict__ in2, const T* __restrict__ gout,
               T* __restrict__ gin2, int C, int Hi, int Wi, int H, int W, int dil, int tiles_x) {
    constexpr int NT = 4 * HALF * HALF;
    __shared__ T red[3][NT][kWave];
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x / kWave;
    unsigned t_id = blockIdx.x;
    const int tx = t_id % tiles_x;
    t_id /= tiles_x;
    const int y = t_id % H;
    const int b = t_id / H;
    const int x = tx * kWave + lane;
    const bool active = x < W;
    const int xc = active ? x : W - 1;   // inactive lanes shadow the last pixel, never store

    const size_t plane = static_cast<size_t>(H) * W;
    const size_t poff = static_cast<size_t>(y) * W + xc;
    const T* f = in2 + static_cast<size_t>(b) * 3 * plane + poff;
    const T sigma = f[2 * plane];
    RsTaps<T, HALF> t;
    make_rs_taps<T, HALF>(t, f[0], f[plane], sigma, xc, y, Hi, Wi, dil, false);

    T acc[NT];
#pragma unroll
    for (int k = 0; k < NT; ++k) acc[k] = 0;
    const size_t iplane = static_cast<size_t>(Hi) * Wi;
    const unsigned ibytes = static_cast<unsigned>(iplane * sizeof(T));
    const T* ip = in1 + (static_cast<size_t>(b) * C + wave) * iplane;
    const T* op = gout + (static_cast<size_t>(b) * C + wave) * plane + poff;
    for (int c = wave; c < C; c += kBlock / kWave, ip += (kBlock / kWave) * iplane, op += (kBlock / kWave) * plane) {
        const rsrc_t r = make_rsrc(ip, ibytes);
        const T g = *op;
#pragma unroll
        for (int fy = 0; fy < HALF; ++fy)
#pragma unroll
            for (int fx = 0; fx < HALF; ++fx) {
                const unsigned rT = t.row[2 * fy], rB = t.row[2 * fy + 1];
                const unsigned cL = t.col[2 * fx], cR = t.col[2 * fx + 1];
                T* a = acc + 4 * (fy * HALF + fx);
                a[0] += g * buf_ld<T>(r, rT + cL);
                a[1] += g * buf_ld<T>(r, rT + cR);
                a[2] += g * buf_ld<T>(r, rB + cL);
                a[3] += g * buf_ld<T>(r, rB + cR);
            }
    }
    // combine the 4 channel groups: waves 1..3 publish, wave 0 folds
    if (wave > 0) {
#pragma unroll
        for (int k = 0; k < NT; ++k) red[wave - 1][k][lane] = acc[k];
    }
    __syncthreads();
    if (wave != 0 || !active) return;
#pragma unroll
    for (int k = 0; k < NT; ++k) acc[k] += red[0][k][lane] + red[1][k][lane] + red[2][k][lane];

    // quotient rule, resample2d_kernel.cu:252-328, with sum_ch(gO * in1[tap]) factored out.
    const T ns2 = -sigma * sigma, s3 = sigma * sigma * sigma;
    T g1x = 0, g1y = 0, g1s = 0, sgx = 0, sgy = 0, sgs = 0, S = 0;
#pragma unroll
    for (int fy = 0; fy < HALF; ++fy)
#pragma unroll
        for (int fx = 0; fx < HALF; ++fx) {
            const T yT = t.wy[2 * fy], yB = t.wy[2 * fy + 1], xL = t.wx[2 * fx], xR = t.wx[2 * fx + 1];
            const T yT_ = t.dy_[2 * fy], yB_ = t.dy_[2 * fy + 1], xL_ = t.dx_[2 * fx], xR_ = t.dx_[2 * fx + 1];
            const T* a = acc + 4 * (fy * HALF + fx);   // TL, TR, BL, BR
            // d/d dx (:272-278)
            g1x += static_cast<T>(safe_div<T>(xL_ * yT * xL * a[0], ns2));
            g1x -= static_cast<T>(safe_div<T>(xR_ * yT * xR * a[1], ns2));
            g1x += static_cast<T>(safe_div<T>(xL_ * yB * xL * a[2], ns2));
            g1x -= static_cast<T>(safe_div<T>(xR_ * yB * xR * a[3], ns2));
            sgx += static_cast<T>(safe_div<T>(xL_ * yT * xL - xR_ * yT * xR + xL_ * yB * xL - xR_ * yB * xR, ns2));
            // d/d dy (:279-285)
            g1y += static_cast<T>(safe_div<T>(yT_ * yT * xL * a[0], ns2));
            g1y += static_cast<T>(safe_div<T>(yT_ * yT * xR * a[1], ns2));
            g1y -= static_cast<T>(safe_div<T>(yB_ * yB * xL * a[2], ns2));
            g1y -= static_cast<T>(safe_div<T>(yB_ * yB * xR * a[3], ns2));
            sgy += static_cast<T>(safe_div<T>(yT_ * yT * xL + yT_ * yT * xR - yB_ * yB * xL - yB_ * yB * xR, ns2));
            // d/d sigma (:286-293)
            const T dTL = yT_ * yT_ + xL_ * xL_, dTR = yT_ * yT_ + xR_ * xR_;
            const T dBL = yB_ * yB_ + xL_ * xL_, dBR = yB_ * yB_ + xR_ * xR_;
            g1s += static_cast<T>(safe_div<T>(dTL * yT * xL * a[0], s3));
            g1s += static_cast<T>(safe_div<T>(dTR * yT * xR * a[1], s3));
            g1s += static_cast<T>(safe_div<T>(dBL * yB * xL * a[2], s3));
            g1s += static_cast<T>(safe_div<T>(dBR * yB * xR * a[3], s3));
            sgs += static_cast<T>(safe_div<T>(dTL * yT * xL + dTR * yT * xR + dBL * yB * xL + dBR * yB * xR, s3));
            // grad2's common factor (:317-322)
            S += yT * xL * a[0];
            S += yT * xR * a[1];
            S += yB * xL * a[2];
            S += yB * xR * a[3];
        }
    const T sum = t.sum;
    T* gp = gin2 + static_cast<size_t>(b) * 3 * plane + poff;
    gp[0] = static_cast<T>(safe_div<T>(g1x, sum) - safe_div<T>(sgx * S, sum * sum));          // :328
    gp[plane] = static_cast<T>(safe_div<T>(g1y, sum) - safe_div<T>(sgy * S, sum * sum));
    gp[2 * plane] = static_cast<T>(safe_div<T>(g1s, sum) - safe_div<T>(sgs * S, sum * sum));
}

// ------------------------------------------------------------------------------------ K2 + K3, LDS tiles (fp32)
// The backward kernels on the forward's tile structure (rs_fwd_lds_kernel): a block owns a 64 x (4*RPT) pixel tile and a
// slab of channels, forms taps and Gaussian weights once per pixel, and works through the channels four at a time.
//
// rs_bwd2_lds_kernel (d_input2): the clamp-extended source box of 4 channels is staged in LDS exactly as in the forward;
// per tap ONE ds_read_b128 feeds acc[tap] += gO_c * in1_c[tap] for the 4 channels (the reference's K3 re-reads 8 gathers
// per (channel, tap) in two passes).  The quotient rule (:252-328) is linear in the 4*HALF^2 accumulators, so channel
// slabs (used when the tiles alone do not fill the chip) combine by atomically adding their partial results.
//
// rs_bwd1_tile_kernel (d_input1): the box is an ACCUMULATOR in LDS (double cells, one plane per channel of a group of 4: ds_add_f64
// retires a wave in ~9 clk, ds_add_f32 in ~190 -- tools/ubench/atomics.hip).  Every pixel adds its 4*HALF^2 normalised
// weights x 4 channel gradients into the box in UNCLAMPED coordinates; after the group's pixels the box is folded onto
// the clamped image and every non-zero cell goes to grad_input1 with ONE global atomic (~1.8 per pixel and channel
// instead of 4*HALF^2).  No plane-size limit (the plane kernel above needs a whole plane in LDS).
template <int HALF, int RPT>
__global__ void __launch_bounds__(kBlock)
rs_bwd2_lds_kernel(const float* __restrict__ in1, const float* __restrict__ in2, const float* __restrict__ gout,
                   float* __restrict__ gin2, int C, int Hi, int Wi, int H, int W, int tiles_x, int tiles_y, int cslabs, int cs,
                   int remap) {
    constexpr int NW = kBlock / kWave;
    constexpr int NT = 2 * HALF;
    constexpr int NA = NT * NT;
    constexpr int TH = NW * RPT;
    constexpr int BOXH = TH + 12;
    constexpr int NCELL = BOXH * kRsBoxW;
    constexpr int NI = (NCELL + kBlock - 1) / kBlock;
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    f32x4* tile = reinterpret_cast<f32x4*>(smem_raw);          // [2][NCELL]
    __shared__ int red[4][NW];
    __shared__ int flag;

    unsigned tid = xcd_remap(blockIdx.x, gridDim.x, remap);
    const int tx = tid % tiles_x;
    tid /= tiles_x;
    const int ty = tid % tiles_y;
    tid /= tiles_y;
    const int slab = tid % cslabs;
    const int b = tid / cslabs;
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    const int x_raw = tx * kTileX + lane;
    const bool inx = x_raw < W;
    const int x = inx ? x_raw : W - 1;
    if (threadIdx.x == 0) flag = 0;

    RsTaps<float, HALF> t[RPT];
    float sg[RPT];
    int u0[RPT], v0[RPT], ys[RPT];
    bool iny[RPT];
    bool regular = true;
    const size_t plane = static_cast<size_t>(H) * W;
    const float* fb = in2 + static_cast<size_t>(b) * 3 * plane;
    int umin = 0x7fffffff, umax = -0x7fffffff, vmin = 0x7fffffff, vmax = -0x7fffffff;
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
        const int yraw = ty * TH + wave + r * NW;
        iny[r] = yraw < H;
        const int y = iny[r] ? yraw : H - 1;
        ys[r] = y;
        const size_t poff = static_cast<size_t>(y) * W + x;
        const float dx = fb[poff], dy = fb[plane + poff];
        sg[r] = fb[2 * plane + poff];
        make_rs_taps<float, HALF>(t[r], dx, dy, sg[r], x, y, Hi, Wi, 1, false);
        const float flx = floor_t(static_cast<float>(x) + dx), fly = floor_t(static_cast<float>(y) + dy);
        const float lim = static_cast<float>(1 << 20);
        const bool ok = (flx > -lim) && (flx < lim) && (fly > -lim) && (fly < lim);
        regular = regular && ok;
        u0[r] = ok ? static_cast<int>(flx) - (HALF - 1) : 0;
        v0[r] = ok ? static_cast<int>(fly) - (HALF - 1) : 0;
        umin = min(umin, u0[r]); umax = max(umax, u0[r] + NT - 1);
        vmin = min(vmin, v0[r]); vmax = max(vmax, v0[r] + NT - 1);
    }
    umin = wave_min(umin); umax = wave_max(umax); vmin = wave_min(vmin); vmax = wave_max(vmax);
    if (lane == 0) { red[0][wave] = umin; red[1][wave] = umax; red[2][wave] = vmin; red[3][wave] = vmax; }
    __syncthreads();
    if (!regular) flag = 1;
#pragma unroll
    for (int k = 0; k < NW; ++k) {
        umin = min(umin, red[0][k]); umax = max(umax, red[1][k]);
        vmin = min(vmin, red[2][k]); vmax = max(vmax, red[3][k]);
    }
    __syncthreads();
    const int bw = umax - umin + 1, bh = vmax - vmin + 1;
    const bool use_lds = (flag == 0) && bw <= kRsBoxW && bh <= BOXH;

    const int c0 = slab * cs;
    const int c1 = (c0 + cs < C) ? c0 + cs : C;
    const size_t iplane = static_cast<size_t>(Hi) * Wi;
    const unsigned ibytes = static_cast<unsigned>(iplane * sizeof(float));
    const unsigned obytes = static_cast<unsigned>(plane * sizeof(float));
    const float* ip = in1 + (static_cast<size_t>(b) * C + c0) * iplane;
    const float* gp = gout + (static_cast<size_t>(b) * C + c0) * plane;
    unsigned poffb[RPT];
#pragma unroll
    for (int r = 0; r < RPT; ++r) poffb[r] = (static_cast<unsigned>(ys[r]) * W + static_cast<unsigned>(x)) * 4u;

    // sum over the channels of gO * in1[tap], per tap, in DOUBLE: the quotient rule below subtracts two nearly equal sums of
    // these (a narrow sigma makes the derivative weights large), so fp32 rounding of a running sum over the channels was
    // amplified to 1.5e-4 of the result against the reference's own kernels; a group of four channels is summed in fp32 from
    // zero (error relative to the GROUP's value) and added to the double accumulator
    double acc[RPT][NA];                     // reference order: k = 4 * (fy * HALF + fx) + {TL, TR, BL, BR}
#pragma unroll
    for (int r = 0; r < RPT; ++r)
#pragma unroll
        for (int k = 0; k < NA; ++k) acc[r][k] = 0.0;

    // visit the taps in the reference's order, handing the body (k, row position, column position)
    auto for_each_tap = [&](auto&& body) {
#pragma unroll
        for (int fy = 0; fy < HALF; ++fy)
#pragma unroll
            for (int fx = 0; fx < HALF; ++fx) {
                const int k = 4 * (fy * HALF + fx);
                body(k + 0, HALF - 1 - fy, HALF - 1 - fx);
                body(k + 1, HALF - 1 - fy, HALF + fx);
                body(k + 2, HALF + fy, HALF - 1 - fx);
                body(k + 3, HALF + fy, HALF + fx);
            }
    };

    if (use_lds) {
        unsigned goff[NI];
#pragma unroll
        for (int k = 0; k < NI; ++k) {
            const int i = threadIdx.x + k * kBlock;
            const int r = i / kRsBoxW, cc = i - r * kRsBoxW;
            const int gy = min(max(vmin + r, 0), Hi - 1), gx = min(max(umin + cc, 0), Wi - 1);
            goff[k] = (r < bh && cc < bw) ? (static_cast<unsigned>(gy) * Wi + gx) * 4u : 0xFFFFFFF0u;
        }
        f32x4 stage[NI];
        auto fetch = [&](int c) {
            const float* p0 = ip + static_cast<size_t>(c - c0) * iplane;
            const rsrc_t r0 = make_rsrc(p0, ibytes);
            const rsrc_t r1 = make_rsrc(p0 + iplane, c + 1 < c1 ? ibytes : 0u);
            const rsrc_t r2 = make_rsrc(p0 + 2 * iplane, c + 2 < c1 ? ibytes : 0u);
            const rsrc_t r3 = make_rsrc(p0 + 3 * iplane, c + 3 < c1 ? ibytes : 0u);
#pragma unroll
            for (int k = 0; k < NI; ++k) {
                stage[k].x = buf_ld<float>(r0, goff[k]);
                stage[k].y = buf_ld<float>(r1, goff[k]);
                stage[k].z = buf_ld<float>(r2, goff[k]);
                stage[k].w = buf_ld<float>(r3, goff[k]);
            }
        };
        auto commit = [&](f32x4* buf) {
#pragma unroll
            for (int k = 0; k < NI; ++k) {
                const int i = threadIdx.x + k * kBlock;
                if (i < NCELL) buf[i] = stage[k];
            }
        };
        int lbase[RPT];
#pragma unroll
        for (int r = 0; r < RPT; ++r) lbase[r] = (v0[r] - vmin) * kRsBoxW + (u0[r] - umin);
        fetch(c0);
        commit(tile);
        __syncthreads();
        int p = 0;
        for (int c = c0; c < c1; c += 4) {
            const bool more = c + 4 < c1;
            if (more) fetch(c + 4);
            const f32x4* tb = tile + p * NCELL;
            const float* g0 = gp + static_cast<size_t>(c - c0) * plane;
            const rsrc_t rg0 = make_rsrc(g0, obytes);
            const rsrc_t rg1 = make_rsrc(g0 + plane, c + 1 < c1 ? obytes : 0u);        // missing channels: g = 0
            const rsrc_t rg2 = make_rsrc(g0 + 2 * plane, c + 2 < c1 ? obytes : 0u);
            const rsrc_t rg3 = make_rsrc(g0 + 3 * plane, c + 3 < c1 ? obytes : 0u);
#pragma unroll
            for (int r = 0; r < RPT; ++r) {
                const float gx = buf_ld<float>(rg0, poffb[r]), gy = buf_ld<float>(rg1, poffb[r]);
                const float gz = buf_ld<float>(rg2, poffb[r]), gw = buf_ld<float>(rg3, poffb[r]);
                const f32x4* nb = tb + lbase[r];
                for_each_tap([&](int k, int pr, int pc) {
                    const f32x4 v = nb[pr * kRsBoxW + pc];
                    float a = gx * v.x;
                    a = __builtin_fmaf(gy, v.y, a);
                    a = __builtin_fmaf(gz, v.z, a);
                    a = __builtin_fmaf(gw, v.w, a);
                    acc[r][k] += static_cast<double>(a);
                });
            }
            if (more) {
                commit(tile + (p ^ 1) * NCELL);
                __syncthreads();
                p ^= 1;
            }
        }
    } else {
        // fallback: direct gathers through the clamped tap offsets
        for (int c = c0; c < c1; ++c, ip += iplane, gp += plane) {
            const rsrc_t rs = make_rsrc(ip, ibytes);
#pragma unroll
            for (int r = 0; r < RPT; ++r) {
                const float g = gp[poffb[r] / 4u];
                for_each_tap([&](int k, int pr, int pc) {
                    const int fyq = pr < HALF ? 2 * (HALF - 1 - pr) : 2 * (pr - HALF) + 1;      // position -> RsTaps entry
                    const int fxq = pc < HALF ? 2 * (HALF - 1 - pc) : 2 * (pc - HALF) + 1;
                    acc[r][k] += static_cast<double>(g) * static_cast<double>(buf_ld<float>(rs, t[r].row[fyq] + t[r].col[fxq]));
                });
            }
        }
    }

    // quotient rule, resample2d_kernel.cu:252-328, with sum_ch(gO * in1[tap]) factored out (as rs_bwd2_kernel), evaluated in
    // double from the fp32 weights (SAFE_DIV's zero tests on the values the reference's fp32 code would test)
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
        if (!(inx && iny[r])) continue;
        const float sigma = sg[r];
        const float ns2f = -sigma * sigma, s3f = sigma * sigma * sigma;
        const double ns2 = (ns2f == 0.f) ? 1e-8 : -static_cast<double>(sigma) * sigma;
        const double s3 = (s3f == 0.f) ? 1e-8 : static_cast<double>(sigma) * sigma * sigma;
        double g1x = 0, g1y = 0, g1s = 0, sgx = 0, sgy = 0, sgs = 0, S = 0;
#pragma unroll
        for (int fy = 0; fy < HALF; ++fy)
#pragma unroll
            for (int fx = 0; fx < HALF; ++fx) {
                const double yT = t[r].wy[2 * fy], yB = t[r].wy[2 * fy + 1], xL = t[r].wx[2 * fx], xR = t[r].wx[2 * fx + 1];
                const double yT_ = t[r].dy_[2 * fy], yB_ = t[r].dy_[2 * fy + 1], xL_ = t[r].dx_[2 * fx], xR_ = t[r].dx_[2 * fx + 1];
                const double* a = acc[r] + 4 * (fy * HALF + fx);
                const double wTL = yT * xL, wTR = yT * xR, wBL = yB * xL, wBR = yB * xR;
                g1x += (xL_ * wTL * a[0] - xR_ * wTR * a[1] + xL_ * wBL * a[2] - xR_ * wBR * a[3]) / ns2;
                sgx += (xL_ * wTL - xR_ * wTR + xL_ * wBL - xR_ * wBR) / ns2;
                g1y += (yT_ * wTL * a[0] + yT_ * wTR * a[1] - yB_ * wBL * a[2] - yB_ * wBR * a[3]) / ns2;
                sgy += (yT_ * wTL + yT_ * wTR - yB_ * wBL - yB_ * wBR) / ns2;
                const double dTL = yT_ * yT_ + xL_ * xL_, dTR = yT_ * yT_ + xR_ * xR_;
                const double dBL = yB_ * yB_ + xL_ * xL_, dBR = yB_ * yB_ + xR_ * xR_;
                g1s += (dTL * wTL * a[0] + dTR * wTR * a[1] + dBL * wBL * a[2] + dBR * wBR * a[3]) / s3;
                sgs += (dTL * wTL + dTR * wTR + dBL * wBL + dBR * wBR) / s3;
                S += wTL * a[0] + wTR * a[1] + wBL * a[2] + wBR * a[3];
            }
        const float sumf = t[r].sum;
        const double sum = (sumf == 0.f) ? 1e-8 : static_cast<double>(sumf);
        const double sum2 = (sumf * sumf == 0.f) ? 1e-8 : static_cast<double>(sumf) * sumf;
        float* op = gin2 + static_cast<size_t>(b) * 3 * plane + poffb[r] / 4u;
        const float rx = static_cast<float>(g1x / sum - sgx * S / sum2);
        const float ry = static_cast<float>(g1y / sum - sgy * S / sum2);
        const float rs_ = static_cast<float>(g1s / sum - sgs * S / sum2);
        if (cslabs == 1) {
            op[0] = rx; op[plane] = ry; op[2 * plane] = rs_;
        } else {                              // linear in the accumulators: the slabs' partial results add up (buffer zero-filled by the host)
            atomic_add(op, rx); atomic_add(op + plane, ry); atomic_add(op + 2 * plane, rs_);
        }
    }
}

// Which d_input1 kernel serves a call?  The tile kernel (a lane per pixel) is conflict-free when the 64 pixels of a wave's row
// share their integer displacement to within a cell -- any flow a network produces: 0.97 ms at [8,64,512,512] -- and pays 5 x for
// its LDS atomics under a random flow (1.73 ms); the tap-lane kernel below costs the same for every flow (1.53 ms).  This pre-pass
// counts the IRREGULAR 64-pixel row segments of the flow into a device counter; both kernels are launched and the one the
// count does not select returns at once (no host round trip).
__global__ void __launch_bounds__(kBlock)
rs_flow_irregular_kernel(const float* __restrict__ in2, int* __restrict__ count, int H, int W, int segs_x, int64_t nseg) {
    __shared__ int part[kBlock / kWave];
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    const size_t plane = static_cast<size_t>(H) * W;
    int mine = 0;                                            // grid-stride over the segments: ONE atomic per block at the end
    for (int64_t seg = static_cast<int64_t>(blockIdx.x) * (kBlock / kWave) + wave; seg < nseg;
         seg += static_cast<int64_t>(gridDim.x) * (kBlock / kWave)) {
        const int sx = static_cast<int>(seg % segs_x);
        const int64_t row = seg / segs_x;                      // b * H + y
        const int y = static_cast<int>(row % H);
        const int64_t b = row / H;
        const int xr = sx * kWave + lane;
        const int x = xr < W ? xr : W - 1;
        const float* f = in2 + static_cast<size_t>(b) * 3 * plane + static_cast<size_t>(y) * W + x;
        const float fx = floor_t(static_cast<float>(x) + f[0]) - static_cast<float>(x);
        const float fy = floor_t(static_cast<float>(y) + f[plane]) - static_cast<float>(y);
        const float lim = static_cast<float>(1 << 20);
        const bool ok = (fx > -lim) && (fx < lim) && (fy > -lim) && (fy < lim);
        const int du = ok ? static_cast<int>(fx) : (lane & 1 ? 4096 : -4096), dv = ok ? static_cast<int>(fy) : 0;
        if ((wave_max(du) - wave_min(du) > 1) || (wave_max(dv) - wave_min(dv) > 1)) ++mine;
    }
    if (lane == 0) part[wave] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
        int sum = 0;
#pragma unroll
        for (int k = 0; k < kBlock / kWave; ++k) sum += part[k];
        if (sum) atomicAdd(count, sum);
    }
}

// FIXED (round 5): the box holds 32-bit FIXED-POINT cells instead of doubles (ds_add_u32: half the LDS time of ds_add_f64 per
// conflict-free instruction, a wave's lanes spread over twice as many banks, half the LDS -- see be_bwd_tile2_kernel).  The scale is
// EXACT here: a block loads the grad_output values of its pixels for the four channels of a group before it adds anything, so the
// group's maximum |g| is known (one unsigned maximum over the magnitude bits, reduced over the block); every contribution is
// w / sum x g with w / sum <= 1, at most one per pixel and cell, so 64 TH contributions of < 2^(31 - log2(64 TH)) cannot overflow.
// One unit is <= max|g| / 2^21 (TH = 16).  A group with a NaN / Inf gradient takes the per-tap global-atomic path for its channels.
template <int HALF, int RPT, bool FIXED = false>
__global__ void __launch_bounds__(kBlock)
rs_bwd1_tile_kernel(const float* __restrict__ in2, const float* __restrict__ gout, float* __restrict__ gin1, int C, int Hi,
                    int Wi, int H, int W, int quirk, int tiles_x, int tiles_y, int cslabs, int cs, int remap, int ablate,
                    const int* __restrict__ sel = nullptr, int sel_limit = 0, int sel_want = 0) {
    if (sel && ((sel[0] < sel_limit) ? 1 : 0) != sel_want) return;      // the other kernel of the pair serves this call (rs_flow_irregular_kernel)
    constexpr int NW = kBlock / kWave;
    constexpr int NT = 2 * HALF;
    constexpr int TH = NW * RPT;
    constexpr int BOXH = TH + 12;
    constexpr int NCELL = BOXH * kRsBoxW;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    // [4][NCELL]: one PLANE per channel of the group.  (Round 2 interleaved the four channels of a cell: a wave's 64 lanes then sat
    // 32 bytes apart and used 16 of the 64 LDS banks -- an 8-way bank conflict on every ds_add_f64.  Planar, neighbouring lanes are
    // 8 bytes apart: the 64 lanes of a smooth flow cover all banks twice, the minimum for 512 bytes.)
    using AccT = typename std::conditional<FIXED, int, double>::type;
    AccT* box = reinterpret_cast<AccT*>(smem_raw);
    __shared__ int red[4][NW];
    __shared__ unsigned redm[NW];
    __shared__ int flag;

    unsigned tid = xcd_remap(blockIdx.x, gridDim.x, remap);
    const int tx = tid % tiles_x;
    tid /= tiles_x;
    const int ty = tid % tiles_y;
    tid /= tiles_y;
    const int slab = tid % cslabs;
    const int b = tid / cslabs;
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    const int x_raw = tx * kTileX + lane;
    const bool inx = x_raw < W;
    const int x = inx ? x_raw : W - 1;
    if (threadIdx.x == 0) flag = 0;

    float wn[RPT][NT * NT];                 // SAFE_DIV(w, sum) (:196-199), [row position][col position]
    int u0[RPT], v0[RPT], ys[RPT];
    bool live[RPT];
    bool regular = true;
    const size_t plane = static_cast<size_t>(H) * W;
    const float* fb = in2 + static_cast<size_t>(b) * 3 * plane;
    int umin = 0x7fffffff, umax = -0x7fffffff, vmin = 0x7fffffff, vmax = -0x7fffffff;
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
        const int yraw = ty * TH + wave + r * NW;
        live[r] = inx && yraw < H;
        const int y = yraw < H ? yraw : H - 1;
        ys[r] = y;
        const size_t poff = static_cast<size_t>(y) * W + x;
        const float dx = fb[poff], dy = fb[plane + poff], sgm = fb[2 * plane + poff];
        RsTaps<float, HALF> t;
        if (ablate & 8) {                  // bench-only: no Gaussian weights
#pragma unroll
            for (int f = 0; f < 2 * HALF; ++f) { t.wx[f] = dx; t.wy[f] = dy; }
            t.sum = sgm;
        } else {
            make_rs_taps<float, HALF>(t, dx, dy, sgm, x, y, Hi, Wi, 1, quirk != 0);
        }
        const float flx = floor_t(static_cast<float>(x) + dx), fly = floor_t(static_cast<float>(y) + dy);
        const float lim = static_cast<float>(1 << 20);
        const bool ok = (flx > -lim) && (flx < lim) && (fly > -lim) && (fly < lim);
        regular = regular && ok;
        u0[r] = ok ? static_cast<int>(flx) - (HALF - 1) : 0;
        v0[r] = ok ? static_cast<int>(fly) - (HALF - 1) : 0;
        umin = min(umin, u0[r]); umax = max(umax, u0[r] + NT - 1);
        vmin = min(vmin, v0[r]); vmax = max(vmax, v0[r] + NT - 1);
        float wxp[NT], wyp[NT];
#pragma unroll
        for (int f = 0; f < HALF; ++f) {
            wxp[HALF - 1 - f] = t.wx[2 * f]; wxp[HALF + f] = t.wx[2 * f + 1];
            wyp[HALF - 1 - f] = t.wy[2 * f]; wyp[HALF + f] = t.wy[2 * f + 1];
        }
#pragma unroll
        for (int pr = 0; pr < NT; ++pr)
#pragma unroll
            for (int pc = 0; pc < NT; ++pc)
                wn[r][pr * NT + pc] = static_cast<float>(safe_div<float>(wyp[pr] * wxp[pc], t.sum));
    }
    umin = wave_min(umin); umax = wave_max(umax); vmin = wave_min(vmin); vmax = wave_max(vmax);
    if (lane == 0) { red[0][wave] = umin; red[1][wave] = umax; red[2][wave] = vmin; red[3][wave] = vmax; }
    __syncthreads();
    if (!regular) flag = 1;
#pragma unroll
    for (int k = 0; k < NW; ++k) {
        umin = min(umin, red[0][k]); umax = max(umax, red[1][k]);
        vmin = min(vmin, red[2][k]); vmax = max(vmax, red[3][k]);
    }
    __syncthreads();
    const int bw = umax - umin + 1, bh = vmax - vmin + 1;
    const bool use_lds = (flag == 0) && bw <= kRsBoxW && bh <= BOXH;

    const int c0 = slab * cs;
    const int c1 = (c0 + cs < C) ? c0 + cs : C;
    const size_t iplane = static_cast<size_t>(Hi) * Wi;
    const unsigned obytes = static_cast<unsigned>(plane * sizeof(float));
    const float* gp = gout + (static_cast<size_t>(b) * C + c0) * plane;
    float* dp = gin1 + (static_cast<size_t>(b) * C + c0) * iplane;
    unsigned poffb[RPT];
#pragma unroll
    for (int r = 0; r < RPT; ++r)
        poffb[r] = live[r] ? (static_cast<unsigned>(ys[r]) * W + static_cast<unsigned>(x)) * 4u : 0xFFFFFFF0u;     // dead lanes read g = 0

    // per-tap global atomics through clamped offsets for the channels [cb, ce) of this block's pixels: the fallback of a block whose
    // taps do not fit the box, and (FIXED) of a channel group with a non-finite gradient
    auto global_taps = [&](int cb, int ce) {
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            if (!live[r]) continue;
            const float fxv = static_cast<float>(x) + fb[static_cast<size_t>(ys[r]) * W + x];
            const float fyv = static_cast<float>(ys[r]) + fb[plane + static_cast<size_t>(ys[r]) * W + x];
            const float flx = floor_t(fxv), fly = floor_t(fyv);
            unsigned col[NT], row[NT];
#pragma unroll
            for (int f = 0; f < HALF; ++f) {
                col[HALF - 1 - f] = static_cast<unsigned>(clamp_index(flx - static_cast<float>(f), Wi));
                col[HALF + f] = static_cast<unsigned>(clamp_index(flx + static_cast<float>(f + 1), Wi));
                row[HALF - 1 - f] = static_cast<unsigned>(clamp_index(fly - static_cast<float>(f), Hi)) * static_cast<unsigned>(Wi);
                row[HALF + f] = static_cast<unsigned>(clamp_index(fly + static_cast<float>(f + 1), Hi)) * static_cast<unsigned>(Wi);
            }
            for (int c = cb; c < ce; ++c) {
                const float g = gp[static_cast<size_t>(c - c0) * plane + poffb[r] / 4u];
                float* d = dp + static_cast<size_t>(c - c0) * iplane;
#pragma unroll
                for (int pr = 0; pr < NT; ++pr)
#pragma unroll
                    for (int pc = 0; pc < NT; ++pc) atomic_add(d + row[pr] + col[pc], wn[r][pr * NT + pc] * g);
            }
        }
    };
    int lbase[RPT];
#pragma unroll
    for (int r = 0; r < RPT; ++r) lbase[r] = (v0[r] - vmin) * kRsBoxW + (u0[r] - umin);
    // FIXED: 64 TH pixels, at most one contribution each per cell and channel, every one below 2^kShift after scaling
    // (one bit of headroom since round 6: 64 TH contributions of magnitude 2^kShift AFTER rounding would reach 2^31 exactly -- ADVICE r5.  The
    // group-shared scale costs small channels beside a large one relative precision: 2^-kShift of the GROUP's largest gradient per contribution;
    // the owned-tile kernel that serves the large calls since round 6 sizes its scale by the counted population and keeps 22 bits.)
    constexpr int kShift = 30 - 6 - (TH >= 16 ? 4 : (TH >= 8 ? 3 : 2));
    // channel groups in runs of 64: a group the box cannot take (FIXED: a non-finite gradient; all of them when the block's taps do not
    // fit the box) is noted in `exact` (block-uniform) and scattered per tap after the run -- ONE instance of that code path
    for (int cbase = c0; cbase < c1; cbase += 256) {
        const int cend = cbase + 256 < c1 ? cbase + 256 : c1;
        unsigned long long exact = use_lds ? 0ull : ~0ull;
        if (use_lds)
        for (int c = cbase; c < cend; c += 4) {
            if (!(ablate & 4)) {
                if constexpr (FIXED) {
                    for (int i = threadIdx.x; i < NCELL; i += kBlock) reinterpret_cast<int4*>(box)[i] = int4{0, 0, 0, 0};
                } else {
                    for (int i = threadIdx.x; i < NCELL * 2; i += kBlock) reinterpret_cast<double2*>(box)[i] = double2{0.0, 0.0};
                }
            }
            const float* g0 = gp + static_cast<size_t>(c - c0) * plane;
            const rsrc_t rg0 = make_rsrc(g0, obytes);
            const rsrc_t rg1 = make_rsrc(g0 + plane, c + 1 < c1 ? obytes : 0u);
            const rsrc_t rg2 = make_rsrc(g0 + 2 * plane, c + 2 < c1 ? obytes : 0u);
            const rsrc_t rg3 = make_rsrc(g0 + 3 * plane, c + 3 < c1 ? obytes : 0u);
            float g[RPT][4];
#pragma unroll
            for (int r = 0; r < RPT; ++r) {
                g[r][0] = buf_ld<float>(rg0, poffb[r]); g[r][1] = buf_ld<float>(rg1, poffb[r]);
                g[r][2] = buf_ld<float>(rg2, poffb[r]); g[r][3] = buf_ld<float>(rg3, poffb[r]);
            }
            float fx_inv = 1.f;
            bool exact_path = false;
            if constexpr (FIXED) {
                unsigned mb = 0;
#pragma unroll
                for (int r = 0; r < RPT; ++r)
#pragma unroll
                    for (int q = 0; q < 4; ++q) mb = max(mb, __float_as_uint(g[r][q]) & 0x7FFFFFFFu);
                mb = wave_max(mb);
                if (lane == 0) redm[wave] = mb;
                __syncthreads();                     // (also: the box is cleared)
#pragma unroll
                for (int k = 0; k < NW; ++k) mb = max(mb, redm[k]);
                exact_path = mb >= 0x7F800000u;      // a NaN / Inf gradient in the group: its channels scatter per tap, exactly
                int ex = 0;
                (void)frexpf(__uint_as_float(mb), &ex);          // max|g| < 2^ex
                const bool usable = mb != 0u && ex > -90 && !exact_path;
                const float sc = usable ? ldexpf(1.f, kShift - ex) : 0.f;
                fx_inv = usable ? ldexpf(1.f, ex - kShift) : 0.f;
#pragma unroll
                for (int r = 0; r < RPT; ++r)
#pragma unroll
                    for (int q = 0; q < 4; ++q) g[r][q] *= sc;          // (exact: a power of two; all zeros when there is nothing to add)
            } else {
                __syncthreads();
            }
            if (exact_path) {
                exact |= 1ull << ((c - cbase) >> 2);
                __syncthreads();                     // (redm is rewritten by the next group)
                continue;
            }
#pragma unroll
            for (int r = 0; r < RPT; ++r) {
                const float gx = g[r][0], gy = g[r][1], gz = g[r][2], gw = g[r][3];
                if (!live[r]) continue;
                if (ablate & 1) { if (gx + gy + gz + gw == 12345.f) box[0] = 1; continue; }
                AccT* nb = box + lbase[r];
#pragma unroll
                for (int pr = 0; pr < NT; ++pr)
#pragma unroll
                    for (int pc = 0; pc < NT; ++pc) {
                        const float wq = wn[r][pr * NT + pc];
                        AccT* cell = nb + (pr * kRsBoxW + pc);
                        if constexpr (FIXED) {
                            __hip_atomic_fetch_add(cell, __float2int_rn(wq * gx), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            __hip_atomic_fetch_add(cell + NCELL, __float2int_rn(wq * gy), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            __hip_atomic_fetch_add(cell + 2 * NCELL, __float2int_rn(wq * gz), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            __hip_atomic_fetch_add(cell + 3 * NCELL, __float2int_rn(wq * gw), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        } else {
                            lds_add(cell, wq * gx);
                            lds_add(cell + NCELL, wq * gy);
                            lds_add(cell + 2 * NCELL, wq * gz);
                            lds_add(cell + 3 * NCELL, wq * gw);
                        }
                    }
            }
            __syncthreads();
            // fold the box onto the clamped image: one global atomic per non-zero cell and channel
            const int nch = c1 - c < 4 ? c1 - c : 4;
            for (int i = threadIdx.x; i < bh * kRsBoxW; i += kBlock) {
                const int r = i / kRsBoxW, cc = i - r * kRsBoxW;
                if (cc >= bw) continue;
                const int gy = min(max(vmin + r, 0), Hi - 1), gx = min(max(umin + cc, 0), Wi - 1);
                float* dst = dp + static_cast<size_t>(c - c0) * iplane + static_cast<size_t>(gy) * Wi + gx;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float v = FIXED ? static_cast<float>(box[q * NCELL + i]) * fx_inv : static_cast<float>(box[q * NCELL + i]);
                    if (q < nch && v != 0.f && !(ablate & 2)) atomic_add(dst + static_cast<size_t>(q) * iplane, v);
                }
            }
            __syncthreads();
        }
        if (exact != 0ull)
            for (int c = cbase; c < cend; c += 4)
                if ((exact >> ((c - cbase) >> 2)) & 1ull) global_taps(c, c + 4 < cend ? c + 4 : cend);
    }
}

// ------------------------------------------------------------------------------------ K2, owned tiles (round 6)
// rs_bwd1_owned_kernel + rs_bwd1_far_kernel: d_input1 WITHOUT global atomics on the regular path.
//
// The tile kernel above lets the boxes of neighbouring blocks overlap and folds every non-zero box cell into grad_input1 with a
// global atomic: ~1.8 per pixel and channel, which issue at about one lane per clock and CU -- 445 us of the 1.23 ms at
// [8, 64, 512, 512] (profiles/r05_rs_bwd1_ablation.txt), an order of magnitude more per lane than an LDS atomic (4.3 clk per wave).
// Here every cell of grad_input1 has exactly ONE owner: a block of 8 waves owns an OW x OH tile of the input plane (54 x 38 at ks 4)
// and visits the 64 x 48 PIXELS of the tile grown by M = 3 + ks/2 -- every pixel whose floor offset is within +-3 of its own position
// and can therefore reach the tile.  A pixel near a tile edge is visited by up to four blocks (x 1.5 pixel visits, x 1.5 LDS
// atomics); each keeps the taps that land in ITS tile (after the reference's clamp to the image).  A tap ROW outside the tile is skipped
// under the exec mask, a tap COLUMN outside it goes to a per-lane dump cell behind the box (a shared ring cell was the first version:
// the 5 + 5 margin lanes of every wave row met on it and every LDS atomic paid a 5-way same-address conflict -- 2.0 ms instead of 0.9).  The box is flushed with plain coalesced
// stores: `=` when the caller says grad_input1 is uninitialised (reference_quirk bit 1: the zero-fill of the reference's wrapper,
// models/external_function.py:137, disappears as well), a read-modify-write otherwise.  No cell is touched by two blocks.
// rs_bwd1_far_kernel is the exact complement, at (pixel, tap) granularity: a pair whose pixel lies OUTSIDE the region of the tap's
// owner tile (flow wider than +-3, NaN, huge) goes out with a global atomic, after the tiles.  For a flow net's field it reads the flow
// and returns.
//
// Cells: 32-bit fixed point (see rs_bwd1_tile_kernel).  The scale of a 4-channel group is 2^(bits - e) with max|g| < 2^e taken over the
// block's pixels (all loaded before the first add) and bits = 31 - bitlength(pop): `pop` is the largest number of taps that meet on
// one own cell, COUNTED once per block in the box itself (the flow is the same for every channel) -- so pop 2^bits < 2^31 holds for
// any flow, contracting ones included, and a typical field gets 25-26 bits where the tile kernel's worst-case bound gives 21.
// A group with a NaN / Inf gradient scatters its own-tile taps with global atomics behind a flush of zeros.
// Weights: w_tap / sum = (wy / sum_y) (wx / sum_x) held as 4 + 4 factors per pixel (8 registers instead of 16): sum = sum_y sum_x up to
// rounding (resample2d_kernel.cu:87 adds the 16 products), SAFE_DIV's zero case kept per factor.
#ifndef FFWM_RS_OWN_THREADS
#define FFWM_RS_OWN_THREADS 512
#endif
#ifndef FFWM_RS_OWN_LDSW
#define FFWM_RS_OWN_LDSW 0          // 1: the per-pixel factors and origins in LDS instead of registers (one block per CU: use with 1024 threads)
#endif
template <int HALF>
struct RsOwn {
    static constexpr int NT = 2 * HALF;
    static constexpr int D = 3;                    // |floor offset| served on the fast path
    static constexpr int M = D + HALF;             // margin of the pixel region around the owned tile
#ifndef FFWM_RS_OWN_RH
#define FFWM_RS_OWN_RH 48          // measured: 48 rows (6 pixels per lane) 705 / 981 us smooth / random, 64 rows (8 per lane: 16 more registers of per-pixel state, scratch in the add loop) 786 / 947
#endif
    static constexpr int RW = 64, RH = FFWM_RS_OWN_RH;         // pixel region of a block
    static constexpr int OW = RW - 2 * M, OH = RH - 2 * M;
    static constexpr int NDUMP = 8;                // dump columns behind the tile's: where a tap COLUMN of another block's tile goes (never read)
    static constexpr int BP = 64;                  // box pitch: OW + NDUMP <= 64
    static constexpr int NCELL = BP * OH;
    static constexpr int THREADS = FFWM_RS_OWN_THREADS, NW = THREADS / 64, PPT = RH / NW;
    static_assert(OW + NDUMP <= BP, "box pitch");
};

// floor offset of a pixel's tap window: origin cell (u0, v0) of its NT x NT taps; ok = finite and small enough for int arithmetic.
// ONE definition for the tile and the far kernel: their predicates must agree bit for bit.
template <int HALF>
__device__ __forceinline__ bool rs_origin(float dx, float dy, int x, int y, int& u0, int& v0) {
    const float flx = floor_t(static_cast<float>(x) + dx), fly = floor_t(static_cast<float>(y) + dy);
    const float lim = static_cast<float>(1 << 20);
    const bool ok = (flx > -lim) && (flx < lim) && (fly > -lim) && (fly < lim);
    u0 = ok ? static_cast<int>(flx) - (HALF - 1) : 0;
    v0 = ok ? static_cast<int>(fly) - (HALF - 1) : 0;
    return ok;
}

// The normalised separable factors of one pixel (NT - 1 per axis, see rs_bwd1_owned_kernel) as a CALL, not inline: the eight
// double-precision exponentials of make_rs_taps want ~100 registers while they run, and inlined eight times into the kernel's prologue
// they pushed the persistent per-pixel state into scratch -- from where the add loop reloaded it, group after group.
template <int HALF>
struct RsFactors {
    float wy[2 * HALF - 1], wx[2 * HALF - 1];
    int degenerate;
};
template <int HALF>
__device__ __attribute__((noinline)) RsFactors<HALF> rs_pixel_factors(float dx, float dy, float sgm, int x, int y, int Hi, int Wi, int quirk, int ablate) {
    constexpr int NT = 2 * HALF;
    RsFactors<HALF> o;
    RsTaps<float, HALF> t;
    if (ablate & 8) {                      // bench-only: no Gaussian weights
#pragma unroll
        for (int f = 0; f < NT; ++f) { t.wx[f] = dx; t.wy[f] = dy; }
    } else {
        make_rs_taps<float, HALF>(t, dx, dy, sgm, x, y, Hi, Wi, 1, quirk != 0);
    }
    float wxp[NT], wyp[NT];
    float sx = 0.f, sy = 0.f;
#pragma unroll
    for (int f = 0; f < HALF; ++f) {
        wxp[HALF - 1 - f] = t.wx[2 * f]; wxp[HALF + f] = t.wx[2 * f + 1];
        wyp[HALF - 1 - f] = t.wy[2 * f]; wyp[HALF + f] = t.wy[2 * f + 1];
    }
#pragma unroll
    for (int f = 0; f < NT; ++f) { sx += wxp[f]; sy += wyp[f]; }
#pragma unroll
    for (int f = 0; f < NT - 1; ++f) {
        o.wx[f] = static_cast<float>(safe_div<float>(wxp[f], sx));
        o.wy[f] = static_cast<float>(safe_div<float>(wyp[f], sy));
    }
    // every weight of an axis underflowed (sigma -> 0 away from the taps): SAFE_DIV's zero arm makes all products 0 -- the pixel adds
    // nothing, and the "last = 1 - others" form must not invent a weight for it
    o.degenerate = (sx == 0.f || sy == 0.f) ? 1 : 0;
    return o;
}

typedef float rs_f2 __attribute__((ext_vector_type(2)));

// Cells, round 6 second version.  A contribution is formed by ONE fused multiply-add, fma(w, g 2^s, 1.5 2^23): the sum is rounded to
// the integer grid of the binade [2^23, 2^24) (round to nearest even, what v_cvt_i32_f32 did in a second instruction) and its BIT
// PATTERN is 0x4B400000 + k with k the signed integer contribution (|k| < 2^22).  ds_add_u32 of the patterns leaves
// n 0x4B400000 + sum k (mod 2^32) in a cell that took n contributions; n per cell is what the count pass leaves in a fifth plane (the
// flow, hence n, is the same for every channel), so the flush recovers sum k exactly.  Two channels share a v_pk_fma_f32.
template <int HALF>
__global__ void __launch_bounds__(RsOwn<HALF>::THREADS, 4)          // 16 waves per CU: 128 registers
rs_bwd1_owned_kernel(const float* __restrict__ in2, const float* __restrict__ gout, float* __restrict__ gin1, int C, int Hi, int Wi,
                     int H, int W, int quirk, int overwrite, int tiles_x, int tiles_y, int cslabs, int cs, int remap, int ablate) {
    using G = RsOwn<HALF>;
    constexpr int NT = G::NT, M = G::M, OW = G::OW, OH = G::OH, BP = G::BP, NCELL = G::NCELL, NW = G::NW, PPT = G::PPT;
    constexpr unsigned kMagicBits = 0x4B400000u;   // 1.5 * 2^23
#if FFWM_RS_OWN_LDSW
    // [4 channels + the tap count][OH][BP], then the per-pixel state: 2 (NT - 1) factor planes + the packed origins, [plane][pixel] with
    // pixel = r THREADS + thread (a wave reads 64 consecutive dwords: conflict-free).  One 16-wave block per CU, NO per-pixel registers.
    extern __shared__ __attribute__((aligned(16))) unsigned char rs_own_smem[];
    unsigned* const box = reinterpret_cast<unsigned*>(rs_own_smem);
    float* const fac = reinterpret_cast<float*>(rs_own_smem) + 5 * NCELL;
#define RS_WY(r, f) fac[((f) * PPT + (r)) * G::THREADS + threadIdx.x]
#define RS_WX(r, f) fac[((NT - 1 + (f)) * PPT + (r)) * G::THREADS + threadIdx.x]
#define RS_UV(r) reinterpret_cast<int*>(fac)[((2 * (NT - 1)) * PPT + (r)) * G::THREADS + threadIdx.x]
#else
    __shared__ unsigned box[5 * NCELL];            // [4 channels + the tap count][OH][BP]
#endif
    __shared__ unsigned redm[NW];
    __shared__ int redp[NW];
    unsigned* const cnt = box + 4 * NCELL;

    unsigned tid = xcd_remap(blockIdx.x, gridDim.x, remap);
    const int tx = tid % tiles_x;
    tid /= tiles_x;
    const int ty = tid % tiles_y;
    tid /= tiles_y;
    const int slab = tid % cslabs;
    const int b = tid / cslabs;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int X0 = tx * OW, Y0 = ty * OH;          // owned tile (cells of the input plane)
    const int x = X0 - M + lane;                   // this lane's pixel column
    const bool inx = x >= 0 && x < W;
    const int dumpc = OW + (lane & (G::NDUMP - 1));

    const size_t plane = static_cast<size_t>(H) * W;
    const float* fb = in2 + static_cast<size_t>(b) * 3 * plane;
    // the normalised factors of an axis sum to 1: NT - 1 of them are kept, the last one is 1 - the others (absolute error <= 2e-7,
    // the size of one fixed-point unit) -- 16 registers less per thread, which is what lets the add loop live in 128 without scratch
#if !FFWM_RS_OWN_LDSW
    float wyn[PPT][NT - 1], wxn[PPT][NT - 1];
    int uv[PPT];                                   // (v0 - Y0) << 16 | (u0 - X0) & 0xffff, clamped to +-2048 (enough: see colx)
#define RS_WY(r, f) wyn[r][f]
#define RS_WX(r, f) wxn[r][f]
#define RS_UV(r) uv[r]
#endif
    unsigned livemask = 0;
#pragma unroll
    for (int r = 0; r < PPT; ++r) {
        const int y = Y0 - M + wave + r * NW;
        const bool live_px = inx && y >= 0 && y < H;
        int u0 = 0, v0 = 0;
        bool ok = false;
#pragma unroll
        for (int f = 0; f < NT - 1; ++f) RS_WY(r, f) = RS_WX(r, f) = 0.f;
        bool degenerate = false;
        if (live_px) {
            const size_t poff = static_cast<size_t>(y) * W + x;
            const float dx = fb[poff], dy = fb[plane + poff], sgm = fb[2 * plane + poff];
            ok = rs_origin<HALF>(dx, dy, x, y, u0, v0);
            if (ok) {
                const RsFactors<HALF> fc = rs_pixel_factors<HALF>(dx, dy, sgm, x, y, Hi, Wi, quirk, ablate);
#pragma unroll
                for (int f = 0; f < NT - 1; ++f) { RS_WX(r, f) = fc.wx[f]; RS_WY(r, f) = fc.wy[f]; }
                degenerate = fc.degenerate != 0;
            }
        }
        const bool on = live_px && ok && !degenerate;
        if (on) livemask |= 1u << r;
        // (a pixel that is dead here -- outside the flow grid, or irregular: the far kernel's -- is skipped through `livemask`)
        const int ur = on ? min(max(u0 - X0, -2048), 2048) : 0;
        const int vr = on ? min(max(v0 - Y0, -2048), 2048) : 0;
        RS_UV(r) = (vr << 16) | (ur & 0xffff);
        __builtin_amdgcn_sched_barrier(0);
    }
    // tile-relative column / row of tap f of a pixel with packed origin `o`: the reference's clamp to the image, then the test against
    // the tile.  (The origin was clamped to +-2048 around the tile: a tap that far out lands on the image border or in another tile either
    // way, and the border cell is this tile's exactly when the unclamped coordinate would have put it there.)
    const int ixlo = -X0, ixhi = Wi - 1 - X0, iylo = -Y0, iyhi = Hi - 1 - Y0;
    auto colx = [&](int o, int f) { return min(max(static_cast<int>(static_cast<short>(o & 0xffff)) + f, ixlo), ixhi); };
    auto rowy = [&](int o, int f) { return min(max((o >> 16) + f, iylo), iyhi); };

    // ---- taps per own cell (the same for every channel): plane `cnt`
    for (int i = threadIdx.x; i < 5 * NCELL; i += G::THREADS) box[i] = 0;
    __syncthreads();
#pragma unroll
    for (int r = 0; r < PPT; ++r) {
        if (!((livemask >> r) & 1u)) continue;
        int cx[NT], ry[NT];
#pragma unroll
        for (int f = 0; f < NT; ++f) {
            cx[f] = colx(RS_UV(r), f);
            cx[f] = static_cast<unsigned>(cx[f]) < static_cast<unsigned>(OW) ? cx[f] : dumpc;
            ry[f] = rowy(RS_UV(r), f);
        }
#pragma unroll
        for (int pr = 0; pr < NT; ++pr) {
            if (static_cast<unsigned>(ry[pr]) >= static_cast<unsigned>(OH)) continue;
#pragma unroll
            for (int pc = 0; pc < NT; ++pc)
                __hip_atomic_fetch_add(cnt + ry[pr] * BP + cx[pc], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    __syncthreads();
    int pop = 0;
    for (int i = threadIdx.x; i < OW * OH; i += G::THREADS) {
        const int rr = i / OW, cc = i - rr * OW;
        pop = max(pop, static_cast<int>(cnt[rr * BP + cc]));
    }
    pop = wave_max(pop);
    if (lane == 0) redp[wave] = pop;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NW; ++k) pop = max(pop, redp[k]);
    // |k| < 2^bits per contribution with pop 2^bits < 2^31 and bits <= 22 (the magic-number binade)
    const int bits = __builtin_amdgcn_readfirstlane(min(22, 31 - (32 - __clz(pop))));

    const int c0 = slab * cs;
    const int c1 = (c0 + cs < C) ? c0 + cs : C;
    const size_t iplane = static_cast<size_t>(Hi) * Wi;
    const float* gp = gout + (static_cast<size_t>(b) * C + c0) * plane;
    float* dp = gin1 + (static_cast<size_t>(b) * C + c0) * iplane;
    const int ybase = Y0 - M + wave;               // this lane's pixel row r is ybase + 8 r
    const unsigned prow = static_cast<unsigned>(NW) * W;      // elements between two of its rows
    const unsigned p0 = static_cast<unsigned>(ybase * W + x); // row 0's element offset, modulo 2^32: p0 + r prow is exact for every pixel that exists (livemask)

    // max |g| (as magnitude bits) of this lane's pixels in the 4-channel group at channel c: the scale's first pass
    auto lane_max = [&](int c) {
        unsigned m = 0;
        const float* g0 = gp + static_cast<size_t>(c - c0) * plane;
        float v[PPT][4];
#pragma unroll
        for (int r = 0; r < PPT; ++r)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                v[r][q] = (((livemask >> r) & 1u) && c + q < c1) ? g0[static_cast<size_t>(q) * plane + static_cast<unsigned>(p0 + r * prow)] : 0.f;
#pragma unroll
        for (int r = 0; r < PPT; ++r)
#pragma unroll
            for (int q = 0; q < 4; ++q) m = max(m, __float_as_uint(v[r][q]) & 0x7FFFFFFFu);
        return m;
    };
    // ---- the scale of a group.  Exact would be max|g| over the block's pixels BEFORE the first add -- a pass over the group's gradients
    // whose HBM latency nothing covers (measured: 124 us of 810; as a burst behind the previous group's adds or a row per add step:
    // worse).  Only the block's FIRST group pays it.  Every later group starts OPTIMISTICALLY with the exponent of the previous group's
    // true maximum + 1, tracks its own true maximum while it adds (the values pass through the registers anyway), and checks afterwards:
    // a maximum above the assumed range (a contribution left the magic-number binade: the box holds garbage) or more than 3 bits below it
    // (precision) clears the box and repeats the group with the exact exponent.  Neighbouring channels of a gradient rarely differ by
    // 8 x; when they do the group costs twice, never correctness.
    unsigned mb = wave_max((ablate & 4) ? 0x3F800000u : lane_max(c0));
    if (lane == 0) redm[wave] = mb;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NW; ++k) mb = max(mb, redm[k]);
    int ex_assumed = 0;
    (void)frexpf(__uint_as_float(mb), &ex_assumed);     // max|g| of the first group < 2^ex_assumed (garbage for 0 / NaN / Inf: the check below repeats)
    __syncthreads();                                    // (redm is rewritten below)

    bool repeated = false;
    for (int c = c0; c < c1;) {
        const float* g0 = gp + static_cast<size_t>(c - c0) * plane;
        const int nch = c1 - c < 4 ? c1 - c : 4;
        bool exact_path = false, usable = false;
        float fx_inv = 0.f;
        // (buffer loads: a wave-uniform resource per channel + ONE 32-bit offset per pixel row.  Plain pointer loads let hipcc turn the 32
        // addresses of a group into 32 loop-carried 64-bit pointers -- 64 registers, and the per-pixel state went to scratch)
        const unsigned obytes = static_cast<unsigned>(plane * sizeof(float));
        const rsrc_t rg0 = make_rsrc(g0, obytes);
        const rsrc_t rg1 = make_rsrc(g0 + plane, nch > 1 ? obytes : 0u);
        const rsrc_t rg2 = make_rsrc(g0 + 2 * plane, nch > 2 ? obytes : 0u);
        const rsrc_t rg3 = make_rsrc(g0 + 3 * plane, nch > 3 ? obytes : 0u);
        auto load_row = [&](int r, float (&dst)[4]) {
            const unsigned po = ((livemask >> r) & 1u) ? static_cast<unsigned>(p0 + r * prow) * 4u : 0xFFFFFFF0u;      // dead pixels read 0
            dst[0] = buf_ld<float>(rg0, po); dst[1] = buf_ld<float>(rg1, po);
            dst[2] = buf_ld<float>(rg2, po); dst[3] = buf_ld<float>(rg3, po);
        };
        {
            const int ex = __builtin_amdgcn_readfirstlane(min(max(ex_assumed, -80), 120));
            const float sc = ldexpf(1.f, bits - ex);
            fx_inv = ldexpf(1.f, ex - bits);
            unsigned mt = 0;                            // this lane's true max|g| of the group
            float gn[4];
            load_row(0, gn);
#pragma unroll
            for (int r = 0; r < PPT; ++r) {
#pragma unroll
                for (int q = 0; q < 4; ++q) mt = max(mt, __float_as_uint(gn[q]) & 0x7FFFFFFFu);
                const rs_f2 ga = {gn[0] * sc, gn[1] * sc}, gb = {gn[2] * sc, gn[3] * sc};       // (exact: a power of two)
                if (r + 1 < PPT) load_row(r + 1, gn);
                __builtin_amdgcn_sched_barrier(0);
                if (!((livemask >> r) & 1u)) continue;
                // the pixel's origin and factors pass through an opaque register copy: everything derived from them (box offsets, 16
                // weight products) is channel-invariant, and hipcc would hoist all of it out of the channel loop -- 24 registers per pixel
                int o = RS_UV(r);
                asm volatile("" : "+v"(o));
                float wy4[NT], wx4[NT];
                float ry1 = 1.f, rx1 = 1.f;
#pragma unroll
                for (int f = 0; f < NT - 1; ++f) {
                    wy4[f] = RS_WY(r, f); wx4[f] = RS_WX(r, f);
                    asm volatile("" : "+v"(wy4[f]), "+v"(wx4[f]));
                    ry1 -= wy4[f]; rx1 -= wx4[f];
                }
                wy4[NT - 1] = ry1; wx4[NT - 1] = rx1;
                int cx[NT], ry[NT];
#pragma unroll
                for (int f = 0; f < NT; ++f) {
                    cx[f] = colx(o, f);
                    cx[f] = static_cast<unsigned>(cx[f]) < static_cast<unsigned>(OW) ? cx[f] : dumpc;          // another tile's column: a dump column
                    ry[f] = rowy(o, f);
                }
                const rs_f2 magic = {12582912.f, 12582912.f};
#pragma unroll
                for (int pr = 0; pr < NT; ++pr) {
                    if (static_cast<unsigned>(ry[pr]) >= static_cast<unsigned>(OH)) continue;              // a tap row of another tile
                    unsigned* rowp = box + ry[pr] * BP;
#pragma unroll
                    for (int pc = 0; pc < NT; ++pc) {
                        const float wq = wy4[pr] * wx4[pc];
                        const rs_f2 wq2 = {wq, wq};
                        const rs_f2 ka = __builtin_elementwise_fma(wq2, ga, magic), kb = __builtin_elementwise_fma(wq2, gb, magic);
                        unsigned* cell = rowp + cx[pc];
                        if (ablate & 1) { if (ka.x + ka.y + kb.x + kb.y == 12345.f) box[0] = 1; continue; }      // bench-only: no LDS atomics
                        __hip_atomic_fetch_add(cell, __float_as_uint(ka.x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        __hip_atomic_fetch_add(cell + NCELL, __float_as_uint(ka.y), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        __hip_atomic_fetch_add(cell + 2 * NCELL, __float_as_uint(kb.x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        __hip_atomic_fetch_add(cell + 3 * NCELL, __float_as_uint(kb.y), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);      // one pixel at a time: interleaved, the eight unrolled steps want 237 registers
            }
            mt = wave_max(mt);
            if (lane == 0) redm[wave] = mt;
            __syncthreads();                            // every contribution of the attempt is in the box; the waves' maxima are out
#pragma unroll
            for (int k = 0; k < NW; ++k) mt = max(mt, redm[k]);
            exact_path = mt >= 0x7F800000u;              // a NaN / Inf gradient among this group's pixels
            int ex_true = ex;
            if (mt != 0u && !exact_path) (void)frexpf(__uint_as_float(mt), &ex_true);       // max|g| < 2^ex_true
            ex_true = __builtin_amdgcn_readfirstlane(ex_true);
            const bool fits = ex_true <= ex && ex_true >= ex - 3 && ex_true > -80 && ex_true < 120;
            usable = !exact_path && (mt == 0u || fits);
            ex_assumed = mt != 0u && !exact_path ? ex_true + 1 : ex;       // the next group's assumption
            if (!usable && !exact_path && !(ablate & 4)) {
                if (repeated) {
                    exact_path = true;                  // (cannot happen: a repeat runs with the true exponent; defensive)
                } else {
                    // the SAME group again with the exact exponent: clear the four channel planes, do not advance c
                    repeated = true;
                    ex_assumed = ex_true;
                    for (int i = threadIdx.x; i < 4 * NCELL; i += G::THREADS) box[i] = 0;
                    __syncthreads();
                    continue;
                }
            }
        }
        repeated = false;
        if (!usable && !exact_path) usable = true;      // (ablate & 4 only)
        // flush the owned cells: plain coalesced stores, no other block touches them
        for (int i = threadIdx.x; i < OW * OH; i += G::THREADS) {
            const int rr = i / OW, cc = i - rr * OW;
            const int gy = Y0 + rr, gx = X0 + cc;
            const int bi = rr * BP + cc;
            const bool in_img = gy < Hi && gx < Wi;
            const unsigned nb = cnt[bi] * kMagicBits;
            float* dst = dp + static_cast<size_t>(c - c0) * iplane + static_cast<size_t>(gy) * Wi + gx;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float v = usable ? static_cast<float>(static_cast<int>(box[q * NCELL + bi] - nb)) * fx_inv : 0.f;
                box[q * NCELL + bi] = 0;
                if (q < nch && in_img && !(ablate & 2)) {
                    float* d = dst + static_cast<size_t>(q) * iplane;
                    if (overwrite) *d = v;
                    else if (v != 0.f) atomic_add(d, v);        // `+=` mode: a return-less atomic (one transaction), not a read-modify-write round trip (measured: +260 us)
                }
            }
        }
        if (exact_path) {
            // the group's own-tile taps by global atomics (this block is still the only writer of these cells: behind its own stores).
            // Everything is re-derived from the flow in a ROLLED loop: the rare path must not touch the register arrays of the hot one
            // (unrolled beside it, hipcc hoisted its products and offsets into 110 more registers and 900 bytes of scratch).
            __builtin_amdgcn_s_waitcnt(0);             // vmcnt(0) expcnt(0) lgkmcnt(0): the zeros are out
            __syncthreads();
#pragma unroll 1
            for (int r = 0; r < PPT; ++r) {
                const int y = ybase + r * NW;
                if (!(inx && y >= 0 && y < H)) continue;
                const size_t poff = static_cast<size_t>(y) * W + x;
                const float dx = fb[poff], dy = fb[plane + poff], sgm = fb[2 * plane + poff];
                int u0, v0;
                if (!rs_origin<HALF>(dx, dy, x, y, u0, v0)) continue;
                RsTaps<float, HALF> t;
                make_rs_taps<float, HALF>(t, dx, dy, sgm, x, y, Hi, Wi, 1, quirk != 0);
                float wxp[NT], wyp[NT];
                float sx = 0.f, sy = 0.f;
#pragma unroll
                for (int f = 0; f < HALF; ++f) {
                    wxp[HALF - 1 - f] = t.wx[2 * f]; wxp[HALF + f] = t.wx[2 * f + 1];
                    wyp[HALF - 1 - f] = t.wy[2 * f]; wyp[HALF + f] = t.wy[2 * f + 1];
                }
#pragma unroll
                for (int f = 0; f < NT; ++f) { sx += wxp[f]; sy += wyp[f]; }
#pragma unroll
                for (int f = 0; f < NT; ++f) {
                    wxp[f] = static_cast<float>(safe_div<float>(wxp[f], sx));
                    wyp[f] = static_cast<float>(safe_div<float>(wyp[f], sy));
                }
                float gq[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) gq[q] = q < nch ? gp[static_cast<size_t>(c - c0 + q) * plane + poff] : 0.f;
#pragma unroll
                for (int pr = 0; pr < NT; ++pr)
#pragma unroll
                    for (int pc = 0; pc < NT; ++pc) {
                        const int cc = min(max(u0 + pc, 0), Wi - 1) - X0, rr = min(max(v0 + pr, 0), Hi - 1) - Y0;
                        if (static_cast<unsigned>(cc) >= static_cast<unsigned>(OW) || static_cast<unsigned>(rr) >= static_cast<unsigned>(OH)) continue;
                        float* dst = dp + static_cast<size_t>(c - c0) * iplane + static_cast<size_t>(Y0 + rr) * Wi + (X0 + cc);
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            if (q < nch) atomic_add(dst + static_cast<size_t>(q) * iplane, (wyp[pr] * wxp[pc]) * gq[q]);
                    }
            }
        }
        __syncthreads();
        c += 4;
    }
}
#undef RS_WY
#undef RS_WX
#undef RS_UV

// The complement of rs_bwd1_owned_kernel: (pixel, tap) pairs whose pixel is not visited by the block that owns the tap's cell.
template <int HALF>
__global__ void __launch_bounds__(kBlock)
rs_bwd1_far_kernel(const float* __restrict__ in2, const float* __restrict__ gout, float* __restrict__ gin1, int C, int Hi, int Wi,
                   int H, int W, int quirk, int tiles_x, int tiles_y, int cslabs, int cs) {
    using G = RsOwn<HALF>;
    constexpr int NT = G::NT, M = G::M, OW = G::OW, OH = G::OH;
    const TileCoord tc = decode_tile(tiles_x, tiles_y, cslabs, 0);
    if (tc.xf >= W || tc.yf >= H) return;
    const int x = tc.xf, y = tc.yf;
    const size_t plane = static_cast<size_t>(H) * W;
    const float* fb = in2 + static_cast<size_t>(tc.b) * 3 * plane;
    const size_t poff = static_cast<size_t>(y) * W + x;
    const float dx = fb[poff], dy = fb[plane + poff];
    int u0, v0;
    const bool ok = rs_origin<HALF>(dx, dy, x, y, u0, v0);
    // which taps are far: the pixel is outside the region of the cell's owner tile (or irregular: every tap, through the reference's
    // saturating clamp of the float coordinate)
    unsigned farmask = 0;
    int colc[NT], rowc[NT];
    if (ok) {
        bool fx[NT], fy[NT];
#pragma unroll
        for (int f = 0; f < NT; ++f) {
            colc[f] = min(max(u0 + f, 0), Wi - 1);
            rowc[f] = min(max(v0 + f, 0), Hi - 1);
            const int ox = (colc[f] / OW) * OW, oy = (rowc[f] / OH) * OH;
            fx[f] = x < ox - M || x >= ox - M + G::RW;
            fy[f] = y < oy - M || y >= oy - M + G::RH;
        }
#pragma unroll
        for (int pr = 0; pr < NT; ++pr)
#pragma unroll
            for (int pc = 0; pc < NT; ++pc)
                if (fx[pc] || fy[pr]) farmask |= 1u << (pr * NT + pc);
    } else {
        farmask = NT * NT >= 32 ? 0xFFFFFFFFu : ((1u << (NT * NT)) - 1u);
        const float flx = floor_t(static_cast<float>(x) + dx), fly = floor_t(static_cast<float>(y) + dy);
#pragma unroll
        for (int f = 0; f < HALF; ++f) {
            colc[HALF - 1 - f] = clamp_index(flx - static_cast<float>(f), Wi);
            colc[HALF + f] = clamp_index(flx + static_cast<float>(f + 1), Wi);
            rowc[HALF - 1 - f] = clamp_index(fly - static_cast<float>(f), Hi);
            rowc[HALF + f] = clamp_index(fly + static_cast<float>(f + 1), Hi);
        }
    }
    if (farmask == 0) return;
    const float sgm = fb[2 * plane + poff];
    RsTaps<float, HALF> t;
    make_rs_taps<float, HALF>(t, dx, dy, sgm, x, y, Hi, Wi, 1, quirk != 0);
    float wxp[NT], wyp[NT];
#pragma unroll
    for (int f = 0; f < HALF; ++f) {
        wxp[HALF - 1 - f] = t.wx[2 * f]; wxp[HALF + f] = t.wx[2 * f + 1];
        wyp[HALF - 1 - f] = t.wy[2 * f]; wyp[HALF + f] = t.wy[2 * f + 1];
    }
    const size_t iplane = static_cast<size_t>(Hi) * Wi;
    const int c0 = tc.slab * cs;
    const int c1 = (c0 + cs < C) ? c0 + cs : C;
    const float* gp = gout + (static_cast<size_t>(tc.b) * C + c0) * plane + poff;
    float* dp = gin1 + (static_cast<size_t>(tc.b) * C + c0) * iplane;
    for (int c = c0; c < c1; ++c, gp += plane, dp += iplane) {
        const float g = *gp;
#pragma unroll
        for (int pr = 0; pr < NT; ++pr)
#pragma unroll
            for (int pc = 0; pc < NT; ++pc)
                if ((farmask >> (pr * NT + pc)) & 1u)
                    atomic_add(dp + static_cast<size_t>(rowc[pr]) * Wi + colc[pc], static_cast<float>(safe_div<float>(wyp[pr] * wxp[pc], t.sum)) * g);
    }
}

// rs_bwd1_taplane_kernel (d_input1, ks = 4): the same LDS box accumulator, other work assignment.  The tile kernel above gives a
// lane one PIXEL and adds one tap of 64 pixels per ds_add_f64: under a random flow the 64 target cells fall on random banks
// (measured at cfg-1: 18 of 34 us are the LDS atomics, 5x their conflict-free time).  Here a wave-instruction adds the 16 taps x 4
// channels of ONE pixel (lane = 16 ch + 4 pr + pc): 64 distinct cells, and with a box pitch of 88 (= 24 mod 32) and a channel
// plane stride = 4 (mod 32) the 64 doubles cover every one of the 32 eight-byte bank pairs exactly twice -- the minimum for
// 512 bytes -- for ANY flow.  The pixel's 16 normalised weights and 4 channel gradients are computed / loaded by its owner lane
// as before, handed over through a small wave-private staging area in LDS (conflict-free pitch 65), and the box origin of the
// pixel comes from the owner's register by v_readlane.
template <int RPT, int NW>
__global__ void __launch_bounds__(NW * kWave)
rs_bwd1_taplane_kernel(const float* __restrict__ in2, const float* __restrict__ gout, float* __restrict__ gin1, int C, int Hi,
                       int Wi, int H, int W, int quirk, int tiles_x, int tiles_y, int cslabs, int cs, int remap, int ablate,
                       const int* __restrict__ sel = nullptr, int sel_limit = 0, int sel_want = 0) {
    if (sel && ((sel[0] < sel_limit) ? 1 : 0) != sel_want) return;      // the other kernel of the pair serves this call
    constexpr int HALF = 2;
    constexpr int NTHR = NW * kWave;
    constexpr int NT = 2 * HALF;
    constexpr int NTAP = NT * NT;
    constexpr int TH = NW * RPT;
    constexpr int BOXH = TH + 12;
    constexpr int BP = 88;                                    // box pitch
    constexpr int NCELL = BOXH * BP;
    constexpr int PS = NCELL + ((4 - NCELL % 32) + 32) % 32;  // channel plane stride = 4 (mod 32)
    constexpr int SP = kWave + 1;                             // staging pitch
    static_assert(kWave == 4 * NTAP && BP % 32 == 24 && PS % 32 == 4, "bank mapping of the tap-lane scatter");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    double* box = reinterpret_cast<double*>(smem_raw);                           // [4][PS]
    float* wst = reinterpret_cast<float*>(box + 4 * PS);                         // [NW][NTAP][SP]
    float* gst = wst + NW * NTAP * SP;                                           // [NW][4][SP]
    __shared__ int red[4][NW];
    __shared__ int flag;

    unsigned tid = xcd_remap(blockIdx.x, gridDim.x, remap);
    const int tx = tid % tiles_x;
    tid /= tiles_x;
    const int ty = tid % tiles_y;
    tid /= tiles_y;
    const int slab = tid % cslabs;
    const int b = tid / cslabs;
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    const int x_raw = tx * kTileX + lane;
    const bool inx = x_raw < W;
    const int x = inx ? x_raw : W - 1;
    if (threadIdx.x == 0) flag = 0;

    float wn[RPT][NTAP];                    // SAFE_DIV(w, sum) (:196-199), [row position][col position]
    int u0[RPT], v0[RPT], ys[RPT];
    bool live[RPT];
    bool regular = true;
    const size_t plane = static_cast<size_t>(H) * W;
    const float* fb = in2 + static_cast<size_t>(b) * 3 * plane;
    int umin = 0x7fffffff, umax = -0x7fffffff, vmin = 0x7fffffff, vmax = -0x7fffffff;
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
        const int yraw = ty * TH + wave + r * NW;
        live[r] = inx && yraw < H;
        const int y = yraw < H ? yraw : H - 1;
        ys[r] = y;
        const size_t poff = static_cast<size_t>(y) * W + x;
        const float dx = fb[poff], dy = fb[plane + poff], sgm = fb[2 * plane + poff];
        RsTaps<float, HALF> t;
        make_rs_taps<float, HALF>(t, dx, dy, sgm, x, y, Hi, Wi, 1, quirk != 0);
        const float flx = floor_t(static_cast<float>(x) + dx), fly = floor_t(static_cast<float>(y) + dy);
        const float lim = static_cast<float>(1 << 20);
        const bool ok = (flx > -lim) && (flx < lim) && (fly > -lim) && (fly < lim);
        regular = regular && ok;
        u0[r] = ok ? static_cast<int>(flx) - (HALF - 1) : 0;
        v0[r] = ok ? static_cast<int>(fly) - (HALF - 1) : 0;
        umin = min(umin, u0[r]); umax = max(umax, u0[r] + NT - 1);
        vmin = min(vmin, v0[r]); vmax = max(vmax, v0[r] + NT - 1);
        float wxp[NT], wyp[NT];
#pragma unroll
        for (int f = 0; f < HALF; ++f) {
            wxp[HALF - 1 - f] = t.wx[2 * f]; wxp[HALF + f] = t.wx[2 * f + 1];
            wyp[HALF - 1 - f] = t.wy[2 * f]; wyp[HALF + f] = t.wy[2 * f + 1];
        }
#pragma unroll
        for (int pr = 0; pr < NT; ++pr)
#pragma unroll
            for (int pc = 0; pc < NT; ++pc)
                wn[r][pr * NT + pc] = static_cast<float>(safe_div<float>(wyp[pr] * wxp[pc], t.sum));
    }
    umin = wave_min(umin); umax = wave_max(umax); vmin = wave_min(vmin); vmax = wave_max(vmax);
    if (lane == 0) { red[0][wave] = umin; red[1][wave] = umax; red[2][wave] = vmin; red[3][wave] = vmax; }
    __syncthreads();
    if (!regular) flag = 1;
#pragma unroll
    for (int k = 0; k < NW; ++k) {
        umin = min(umin, red[0][k]); umax = max(umax, red[1][k]);
        vmin = min(vmin, red[2][k]); vmax = max(vmax, red[3][k]);
    }
    __syncthreads();
    const int bw = umax - umin + 1, bh = vmax - vmin + 1;
    const bool use_lds = (flag == 0) && bw <= BP && bh <= BOXH;

    const int c0 = slab * cs;
    const int c1 = (c0 + cs < C) ? c0 + cs : C;
    const size_t iplane = static_cast<size_t>(Hi) * Wi;
    const unsigned obytes = static_cast<unsigned>(plane * sizeof(float));
    const float* gp = gout + (static_cast<size_t>(b) * C + c0) * plane;
    float* dp = gin1 + (static_cast<size_t>(b) * C + c0) * iplane;
    unsigned poffb[RPT];
#pragma unroll
    for (int r = 0; r < RPT; ++r)
        poffb[r] = live[r] ? (static_cast<unsigned>(ys[r]) * W + static_cast<unsigned>(x)) * 4u : 0xFFFFFFF0u;     // dead lanes read g = 0

    if (use_lds) {
        int lbase[RPT];
#pragma unroll
        for (int r = 0; r < RPT; ++r) lbase[r] = (v0[r] - vmin) * BP + (u0[r] - umin);
        const int tap = lane & (NTAP - 1), chl = lane / NTAP;
        double* mycell = box + (tap / NT) * BP + (tap % NT) + chl * PS;
        float* wS = wst + wave * NTAP * SP;
        float* gS = gst + wave * 4 * SP;
        const float* wR = wS + tap * SP;
        const float* gR = gS + chl * SP;
        for (int c = c0; c < c1; c += 4) {
            for (int i = threadIdx.x; i < 2 * PS; i += NTHR) reinterpret_cast<double2*>(box)[i] = double2{0.0, 0.0};
            __syncthreads();
            const float* g0 = gp + static_cast<size_t>(c - c0) * plane;
            const rsrc_t rg0 = make_rsrc(g0, obytes);
            const rsrc_t rg1 = make_rsrc(g0 + plane, c + 1 < c1 ? obytes : 0u);
            const rsrc_t rg2 = make_rsrc(g0 + 2 * plane, c + 2 < c1 ? obytes : 0u);
            const rsrc_t rg3 = make_rsrc(g0 + 3 * plane, c + 3 < c1 ? obytes : 0u);
            float gv[RPT][4];
#pragma unroll
            for (int r = 0; r < RPT; ++r) {
                gv[r][0] = buf_ld<float>(rg0, poffb[r]); gv[r][1] = buf_ld<float>(rg1, poffb[r]);
                gv[r][2] = buf_ld<float>(rg2, poffb[r]); gv[r][3] = buf_ld<float>(rg3, poffb[r]);
            }
#pragma unroll
            for (int r = 0; r < RPT; ++r) {
                // the owner lanes hand their pixel's weights and gradients to the wave (a dead pixel hands g = 0)
#pragma unroll
                for (int k = 0; k < NTAP; ++k) wS[k * SP + lane] = wn[r][k];
#pragma unroll
                for (int q = 0; q < 4; ++q) gS[q * SP + lane] = (ablate & 16) ? 1.f : gv[r][q];
                __builtin_amdgcn_wave_barrier();
                const int lb = lbase[r];
                // batches of 8 pixels, the next batch's weights and gradients read before this batch's atomics are issued (the
                // staging area and the box are one LDS array to the compiler: it keeps program order, and a rolled loop waited
                // for a full LDS round trip per pixel)
                constexpr int NB = 8;
                float wa[NB], ga[NB], wb[NB], gb[NB];
                auto fetch = [&](int px0, float (&wv)[NB], float (&gq)[NB]) {
#pragma unroll
                    for (int k = 0; k < NB; ++k) { wv[k] = wR[px0 + k]; gq[k] = gR[px0 + k]; }
                };
                auto emit = [&](int px0, const float (&wv)[NB], const float (&gq)[NB]) {
#pragma unroll
                    for (int k = 0; k < NB; ++k) {
                        if (ablate & 1) { if (wv[k] * gq[k] == 12345.f) box[0] = 1; continue; }     // bench-only: no LDS atomics
                        lds_add(mycell + __builtin_amdgcn_readlane(lb, px0 + k), wv[k] * gq[k]);
                    }
                };
                fetch(0, wa, ga);
#pragma unroll
                for (int px0 = 0; px0 < kWave; px0 += 2 * NB) {
                    fetch(px0 + NB, wb, gb);
                    emit(px0, wa, ga);
                    if (px0 + 2 * NB < kWave) fetch(px0 + 2 * NB, wa, ga);
                    emit(px0 + NB, wb, gb);
                }
                __builtin_amdgcn_wave_barrier();
            }
            __syncthreads();
            // fold the box onto the clamped image: one global atomic per non-zero cell and channel
            const int nch = c1 - c < 4 ? c1 - c : 4;
            for (int i = threadIdx.x; i < bh * BP; i += NTHR) {
                const int r = i / BP, cc = i - r * BP;
                if (cc >= bw) continue;
                const int gy = min(max(vmin + r, 0), Hi - 1), gx = min(max(umin + cc, 0), Wi - 1);
                float* dst = dp + static_cast<size_t>(c - c0) * iplane + static_cast<size_t>(gy) * Wi + gx;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float v = static_cast<float>(box[q * PS + i]);
                    if (q < nch && v != 0.f && !(ablate & 2)) atomic_add(dst + static_cast<size_t>(q) * iplane, v);
                }
            }
            __syncthreads();
        }
        return;
    }
    // fallback: per-tap global atomics through clamped offsets
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
        if (!live[r]) continue;
        const float fxv = static_cast<float>(x) + fb[static_cast<size_t>(ys[r]) * W + x];
        const float fyv = static_cast<float>(ys[r]) + fb[plane + static_cast<size_t>(ys[r]) * W + x];
        const float flx = floor_t(fxv), fly = floor_t(fyv);
        unsigned col[NT], row[NT];
#pragma unroll
        for (int f = 0; f < HALF; ++f) {
            col[HALF - 1 - f] = static_cast<unsigned>(clamp_index(flx - static_cast<float>(f), Wi));
            col[HALF + f] = static_cast<unsigned>(clamp_index(flx + static_cast<float>(f + 1), Wi));
            row[HALF - 1 - f] = static_cast<unsigned>(clamp_index(fly - static_cast<float>(f), Hi)) * static_cast<unsigned>(Wi);
            row[HALF + f] = static_cast<unsigned>(clamp_index(fly + static_cast<float>(f + 1), Hi)) * static_cast<unsigned>(Wi);
        }
        for (int c = c0; c < c1; ++c) {
            const float g = gp[static_cast<size_t>(c - c0) * plane + poffb[r] / 4u];
            float* d = dp + static_cast<size_t>(c - c0) * iplane;
#pragma unroll
            for (int pr = 0; pr < NT; ++pr)
#pragma unroll
                for (int pc = 0; pc < NT; ++pc) atomic_add(d + row[pr] + col[pc], wn[r][pr * NT + pc] * g);
        }
    }
}

// ------------------------------------------------------------------- any even kernel_size
// Literal per-element kernels (the reference's decomposition, 64-bit safe indices).
template <typename T>
struct GenTap {
    int yT, yB, xL, xR;
    T xL_, xR_, yT_, yB_, xLP, xRP, yTP, yBP;
};
template <typename T>
__device__ __forceinline__ GenTap<T> gen_tap(T flx, T fly, T alpha, T beta, T sigma, int fx, int fy,
                                             int dil, int Hi, int Wi) {
    GenTap<T> g;
    g.yT = clamp_index(fly - static_cast<T>(fy * dil), Hi);
    g.yB = clamp_index(fly + static_cast<T>((fy + 1) * dil), Hi);
    g.xL = clamp_index(flx - static_cast<T>(fx * dil), Wi);
    g.xR = clamp_index(flx + static_cast<T>((fx + 1) * dil), Wi);
    g.xL_ = static_cast<T>(fx * dil) + alpha;
    g.xR_ = static_cast<T>((1. + fx) * dil) - alpha;
    g.yT_ = static_cast<T>(fy * dil) + beta;
    g.yB_ = static_cast<T>((1. + fy) * dil) - beta;
    g.xLP = gauss<T>(g.xL_, sigma);
    g.xRP = gauss<T>(g.xR_, sigma);
    g.yTP = gauss<T>(g.yT_, sigma);
    g.yBP = gauss<T>(g.yB_, sigma);
    return g;
}

template <typename T>
__global__ void __launch_bounds__(kBlock)
rs_fwd_generic(const T* __restrict__ in1, const T* __restrict__ in2, T* __restrict__ out, int64_t n,
               int C, int Hi, int Wi, int H, int W, int ks, int dil) {
    const int half = ks / 2;
    const size_t plane = static_cast<size_t>(H) * W, iplane = static_cast<size_t>(Hi) * Wi;
    for (int64_t index = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; index < n;
         index += static_cast<int64_t>(gridDim.x) * kBlock) {
        const int x = static_cast<int>(index % W);
        const int y = static_cast<int>((index / W) % H);
        const int64_t bc = index / plane;
        const int64_t b = bc / C;
        const T* f = in2 + b * 3 * plane + static_cast<size_t>(y) * W + x;
        const T sigma = f[2 * plane];
        const T xf = static_cast<T>(x) + f[0], yf = static_cast<T>(y) + f[plane];
        const T flx = floor_t(xf), fly = floor_t(yf);
        const T alpha = xf - flx, beta = yf - fly;
        const T* ip = in1 + bc * iplane;
        T val = 0, sum = 0;
        for (int fy = 0; fy < half; ++fy)
            for (int fx = 0; fx < half; ++fx) {
                const GenTap<T> g = gen_tap<T>(flx, fly, alpha, beta, sigma, fx, fy, dil, Hi, Wi);
                val += g.yTP * g.xLP * ip[static_cast<size_t>(g.yT) * Wi + g.xL];
                val += g.yTP * g.xRP * ip[static_cast<size_t>(g.yT) * Wi + g.xR];
                val += g.yBP * g.xLP * ip[static_cast<size_t>(g.yB) * Wi + g.xL];
                val += g.yBP * g.xRP * ip[static_cast<size_t>(g.yB) * Wi + g.xR];
                sum += (g.yTP * g.xLP + g.yTP * g.xRP + g.yBP * g.xLP + g.yBP * g.xRP);
            }
        out[index] = static_cast<T>(safe_div<T>(val, sum));
    }
}

template <typename T>
__global__ void __launch_bounds__(kBlock)
rs_bwd1_generic(const T* __restrict__ in2, const T* __restrict__ gout, T* __restrict__ gin1, int64_t n,
                int C, int Hi, int Wi, int H, int W, int ks, int dil, int quirk, const GoStrides gs = GoStrides{0, 0, 0, 0}) {
    const int half = ks / 2;
    const size_t plane = static_cast<size_t>(H) * W, iplane = static_cast<size_t>(Hi) * Wi;
    for (int64_t index = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; index < n;
         index += static_cast<int64_t>(gridDim.x) * kBlock) {
        const int x = static_cast<int>(index % W);
        const int y = static_cast<int>((index / W) % H);
        const int64_t bc = index / plane;
        const int64_t b = bc / C;
        const T* f = in2 + b * 3 * plane + static_cast<size_t>(y) * W + x;
        const T sigma = f[2 * plane];
        const T xf = static_cast<T>(x) + f[0], yf = static_cast<T>(y) + f[plane];
        const T flx = floor_t(xf), fly = floor_t(yf);
        const T alpha = quirk ? xf - static_cast<T>(clamp_index_wide(xf)) : xf - flx;
        const T beta = quirk ? yf - static_cast<T>(clamp_index_wide(yf)) : yf - fly;
        T sum = 0;
        for (int fy = 0; fy < half; ++fy)
            for (int fx = 0; fx < half; ++fx) {
                const GenTap<T> g = gen_tap<T>(flx, fly, alpha, beta, sigma, fx, fy, dil, Hi, Wi);
                sum += (g.yTP * g.xLP + g.yTP * g.xRP + g.yBP * g.xLP + g.yBP * g.xRP);
            }
        const T go = gs.x == 0 ? gout[index] : gout[b * gs.b + (bc - b * C) * gs.c + static_cast<long long>(y) * gs.y + static_cast<long long>(x) * gs.x];
        T* gp = gin1 + bc * iplane;
        for (int fy = 0; fy < half; ++fy)
            for (int fx = 0; fx < half; ++fx) {
                const GenTap<T> g = gen_tap<T>(flx, fly, alpha, beta, sigma, fx, fy, dil, Hi, Wi);
                atomic_add(gp + static_cast<size_t>(g.yT) * Wi + g.xL, static_cast<T>(safe_div<T>(g.yTP * g.xLP, sum) * go));
                atomic_add(gp + static_cast<size_t>(g.yT) * Wi + g.xR, static_cast<T>(safe_div<T>(g.yTP * g.xRP, sum) * go));
                atomic_add(gp + static_cast<size_t>(g.yB) * Wi + g.xL, static_cast<T>(safe_div<T>(g.yBP * g.xLP, sum) * go));
                atomic_add(gp + static_cast<size_t>(g.yB) * Wi + g.xR, static_cast<T>(safe_div<T>(g.yBP * g.xRP, sum) * go));
            }
    }
}

template <typename T>
__global__ void __launch_bounds__(kBlock)
rs_bwd2_generic(const T* __restrict__ in1, const T* __restrict__ in2, const T* __restrict__ gout,
                T* __restrict__ gin2, int64_t n, int C, int Hi, int Wi, int H, int W, int ks, int dil,
                const GoStrides gs = GoStrides{0, 0, 0, 0}) {
    const int half = ks / 2;
    const size_t plane = static_cast<size_t>(H) * W, iplane = static_cast<size_t>(Hi) * Wi;
    for (int64_t index = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; index < n;
         index += static_cast<int64_t>(gridDim.x) * kBlock) {
        const int x = static_cast<int>(index % W);
        const int y = static_cast<int>((index / W) % H);
        const int64_t bc = index / plane;
        const int c = static_cast<int>(bc % 3);
        const int64_t b = bc / 3;
        const size_t poff = static_cast<size_t>(y) * W + x;
        const T* f = in2 + b * 3 * plane + poff;
        const T sigma = f[2 * plane];
        const T xf = static_cast<T>(x) + f[0], yf = static_cast<T>(y) + f[plane];
        const T flx = floor_t(xf), fly = floor_t(yf);
        const T alpha = xf - flx, beta = yf - fly;
        const T ns2 = -sigma * sigma, s3 = sigma * sigma * sigma;
        const T* ip = in1 + b * C * iplane;
        // (grad_output through its strides when they were handed over: channel stride cs, base of pixel (b, y, x))
        const long long cs = gs.x == 0 ? static_cast<long long>(plane) : gs.c;
        const T* op = gs.x == 0 ? gout + b * C * plane + poff : gout + b * gs.b + static_cast<long long>(y) * gs.y + static_cast<long long>(x) * gs.x;
        T grad1 = 0, sumgrad = 0, sum = 0, S = 0;
        for (int fy = 0; fy < half; ++fy)
            for (int fx = 0; fx < half; ++fx) {
                const GenTap<T> g = gen_tap<T>(flx, fly, alpha, beta, sigma, fx, fy, dil, Hi, Wi);
                sum += (g.yTP * g.xLP + g.yTP * g.xRP + g.yBP * g.xLP + g.yBP * g.xRP);
                T a0 = 0, a1 = 0, a2 = 0, a3 = 0;
                for (int ch = 0; ch < C; ++ch) {
                    const T go = op[ch * cs];
                    const T* p = ip + ch * iplane;
                    a0 += go * p[static_cast<size_t>(g.yT) * Wi + g.xL];
                    a1 += go * p[static_cast<size_t>(g.yT) * Wi + g.xR];
                    a2 += go * p[static_cast<size_t>(g.yB) * Wi + g.xL];
                    a3 += go * p[static_cast<size_t>(g.yB) * Wi + g.xR];
                }
                if (c == 0) {
                    grad1 += static_cast<T>(safe_div<T>(g.xL_ * g.yTP * g.xLP * a0, ns2));
                    grad1 -= static_cast<T>(safe_div<T>(g.xR_ * g.yTP * g.xRP * a1, ns2));
                    grad1 += static_cast<T>(safe_div<T>(g.xL_ * g.yBP * g.xLP * a2, ns2));
                    grad1 -= static_cast<T>(safe_div<T>(g.xR_ * g.yBP * g.xRP * a3, ns2));
                    sumgrad += static_cast<T>(safe_div<T>(g.xL_ * g.yTP * g.xLP - g.xR_ * g.yTP * g.xRP + g.xL_ * g.yBP * g.xLP - g.xR_ * g.yBP * g.xRP, ns2));
                } else if (c == 1) {
                    grad1 += static_cast<T>(safe_div<T>(g.yT_ * g.yTP * g.xLP * a0, ns2));
                    grad1 += static_cast<T>(safe_div<T>(g.yT_ * g.yTP * g.xRP * a1, ns2));
                    grad1 -= static_cast<T>(safe_div<T>(g.yB_ * g.yBP * g.xLP * a2, ns2));
                    grad1 -= static_cast<T>(safe_div<T>(g.yB_ * g.yBP * g.xRP * a3, ns2));
                    sumgrad += static_cast<T>(safe_div<T>(g.yT_ * g.yTP * g.xLP + g.yT_ * g.yTP * g.xRP - g.yB_ * g.yBP * g.xLP - g.yB_ * g.yBP * g.xRP, ns2));
                } else {
                    const T dTL = g.yT_ * g.yT_ + g.xL_ * g.xL_, dTR = g.yT_ * g.yT_ + g.xR_ * g.xR_;
                    const T dBL = g.yB_ * g.yB_ + g.xL_ * g.xL_, dBR = g.yB_ * g.yB_ + g.xR_ * g.xR_;
                    grad1 += static_cast<T>(safe_div<T>(dTL * g.yTP * g.xLP * a0, s3));
                    grad1 += static_cast<T>(safe_div<T>(dTR * g.yTP * g.xRP * a1, s3));
                    grad1 += static_cast<T>(safe_div<T>(dBL * g.yBP * g.xLP * a2, s3));
                    grad1 += static_cast<T>(safe_div<T>(dBR * g.yBP * g.xRP * a3, s3));
                    sumgrad += static_cast<T>(safe_div<T>(dTL * g.yTP * g.xLP + dTR * g.yTP * g.xRP + dBL * g.yBP * g.xLP + dBR * g.yBP * g.xRP, s3));
                }
                S += g.yTP * g.xLP * a0;
                S += g.yTP * g.xRP * a1;
                S += g.yBP * g.xLP * a2;
                S += g.yBP * g.xRP * a3;
            }
        gin2[index] = static_cast<T>(safe_div<T>(grad1, sum) - safe_div<T>(sumgrad * S, sum * sum));
    }
}

// ------------------------------------------------------------------------------------ host
int check_dims(const char* fn, int64_t B, int64_t C, int64_t Hi, int64_t Wi, int64_t H, int64_t W,
               int ks, int dil, int dtype) {
    FFWM_REQUIRE(dtype_ok(dtype), FFWM_ERR_DTYPE, "%s: dtype %d is not FFWM_F32/FFWM_F64", fn, dtype);
    FFWM_REQUIRE(B > 0 && C > 0 && Hi > 0 && Wi > 0 && H > 0 && W > 0, FFWM_ERR_ARG,
                 "%s: sizes must be positive (B=%lld C=%lld Hi=%lld Wi=%lld H=%lld W=%lld)", fn, (long long)B,
                 (long long)C, (long long)Hi, (long long)Wi, (long long)H, (long long)W);
    FFWM_REQUIRE(ks >= 2 && dil >= 1, FFWM_ERR_ARG, "%s: need kernel_size >= 2 and dilation >= 1 (got %d, %d)",
                 fn, ks, dil);
    FFWM_REQUIRE(Hi * Wi < (1LL << 28) && H * W < (1LL << 28), FFWM_ERR_SIZE,
                 "%s: a single H*W plane must stay below 2^28 elements (32-bit byte offsets)", fn);
    const int64_t spatial = B * ((W + kTileX - 1) / kTileX) * ((H + kTileY - 1) / kTileY);
    FFWM_REQUIRE(spatial * C < (1LL << 31) && B * H * ((W + kWave - 1) / kWave) < (1LL << 31), FFWM_ERR_SIZE,
                 "%s: grid too large", fn);
    return FFWM_OK;
}

unsigned generic_grid(int64_t n) {
    const int64_t blocks = (n + kBlock - 1) / kBlock;
    return static_cast<unsigned>(blocks < 16384 ? blocks : 16384);
}

template <typename T>
int launch_fwd(const T* in1, const T* in2, T* out, int64_t B, int64_t C, int64_t Hi, int64_t Wi,
               int64_t H, int64_t W, int ks, int dil, hipStream_t st) {
    const double bytes = sizeof(T) * static_cast<double>(B) * H * W * (2.0 * C + 3.0);
    const int remap = options().xcd_remap;
    if constexpr (sizeof(T) == 4) {
        const int half = ks / 2;
        if (dil == 1 && half >= 1 && half <= 3 && options().rs_fwd_variant != 1) {
            // tile rows 4 * rpt; channels cut into slabs of a multiple of 4 until there are >= 2 blocks per CU
            const int tiles_x = static_cast<int>((W + kTileX - 1) / kTileX);
            const int v = options().rs_fwd_variant;
            int rpt = (v == 2 || v == 6) ? 4 : (v == 3 || v == 7) ? 1 : (v == 4 || v == 5) ? 2
                      : (B * tiles_x * ((H + 7) / 8) >= 2048 ? 2 : 1);      // measured: 64 x 8 tiles >= 64 x 16 > 64 x 4 at HBM-resident sizes
            const bool db = !(v >= 5 && v <= 7);
            const int tiles_y = static_cast<int>((H + 4 * rpt - 1) / (4 * rpt));
            const int64_t spatial = B * tiles_x * tiles_y;
            int cs = static_cast<int>((C + 3) / 4 * 4);
            while (cs > 4 && spatial * ((C + cs - 1) / cs) < 512) cs = (cs / 2 + 3) / 4 * 4;
            const int cslabs = static_cast<int>((C + cs - 1) / cs);
            const unsigned grid = static_cast<unsigned>(spatial * cslabs);
            const size_t lds = static_cast<size_t>(db ? 2 : 1) * (4 * rpt + 12) * kRsBoxW * 16;
            LaunchScope ls("resample2d_fwd_lds", st, bytes);
#define FFWM_RS_FWD_LDS(HH, RR, DD)                                                                        \
    do {                                                                                                   \
        allow_large_lds(reinterpret_cast<const void*>(rs_fwd_lds_kernel<HH, RR, DD>));                     \
        hipLaunchKernelGGL((rs_fwd_lds_kernel<HH, RR, DD>), dim3(grid), dim3(kBlock), lds, st, in1, in2, out, \
                           (int)C, (int)Hi, (int)Wi, (int)H, (int)W, tiles_x, tiles_y, cslabs, cs, remap, options().ablate); \
    } while (0)
#define FFWM_RS_FWD_LDS_H(RR, DD)                                                                          \
    do {                                                                                                   \
        if (half == 1) FFWM_RS_FWD_LDS(1, RR, DD); else if (half == 2) FFWM_RS_FWD_LDS(2, RR, DD); else FFWM_RS_FWD_LDS(3, RR, DD); \
    } while (0)
            if (rpt == 4) { if (db) FFWM_RS_FWD_LDS_H(4, true); else FFWM_RS_FWD_LDS_H(4, false); }
            else if (rpt == 2) { if (db) FFWM_RS_FWD_LDS_H(2, true); else FFWM_RS_FWD_LDS_H(2, false); }
            else { if (db) FFWM_RS_FWD_LDS_H(1, true); else FFWM_RS_FWD_LDS_H(1, false); }
#undef FFWM_RS_FWD_LDS_H
#undef FFWM_RS_FWD_LDS
            return check_launch("ffwm_resample2d_forward(lds)");
        }
    }
    const Geometry g = plan(B, C, H, W, 16);
    LaunchScope ls("resample2d_fwd", st, bytes);
#define FFWM_RS_FWD(HH)                                                                             \
    case 2 * HH:                                                                                    \
        hipLaunchKernelGGL((rs_fwd_kernel<T, HH>), dim3(g.grid), dim3(kBlock), 0, st, in1, in2, out, \
                           (int)C, (int)Hi, (int)Wi, (int)H, (int)W, dil, g.tiles_x, g.tiles_y,      \
                           g.cslabs, g.cs, remap);                                                  \
        break;
    switch (ks & ~1) {
        FFWM_RS_FWD(1) FFWM_RS_FWD(2) FFWM_RS_FWD(3)
        default: {
            const int64_t n = B * C * H * W;
            hipLaunchKernelGGL((rs_fwd_generic<T>), dim3(generic_grid(n)), dim3(kBlock), 0, st, in1, in2,
                               out, n, (int)C, (int)Hi, (int)Wi, (int)H, (int)W, ks, dil);
        }
    }
#undef FFWM_RS_FWD
    return check_launch("ffwm_resample2d_forward");
}

template <typename T>
int launch_bwd(const T* in1, const T* in2, const T* gout, T* gin1, T* gin2, int64_t B, int64_t C,
               int64_t Hi, int64_t Wi, int64_t H, int64_t W, int ks, int dil, int quirk_flags,
               hipStream_t st) {
    const int remap = options().xcd_remap;
    const size_t plane_lds = static_cast<size_t>(Hi) * Wi * sizeof(double);     // the LDS accumulator is double
    const int half = ks / 2;
    const int quirk = quirk_flags & 1;
    const bool overwrite = (quirk_flags & 2) != 0 && gin1 != nullptr;           // grad_input1 arrives uninitialised
    if constexpr (sizeof(T) == 4) {
        // Round 6: owned tiles (rs_bwd1_owned_kernel + rs_bwd1_far_kernel) for the calls the shared-cell tile kernel served: fp32, dilation 1,
        // kernel_size 2 / 4, >= 2^18 pixels, planes of >= 32 rows.  rs_bwd1_owned: 0 = on, 2 = off (rounds 3-5's kernels).
        if (gin1 && dil == 1 && (half == 1 || half == 2) && options().scatter_variant == 0 && options().rs_bwd1_variant == 0 &&
            options().rs_bwd1_owned != 2 && B * H * W >= (options().rs_bwd1_owned_min_pixels > 0 ? options().rs_bwd1_owned_min_pixels : (1 << 18)) && H >= 32 && Hi >= 32 &&
            static_cast<int64_t>(Hi) * Wi < (1LL << 29) && static_cast<int64_t>(H) * W < (1LL << 29)) {
            const double bytes1 = sizeof(T) * static_cast<double>(B) * (C * (static_cast<double>(H) * W + 2.0 * Hi * Wi) + 3.0 * H * W);
            const int ow = half == 1 ? RsOwn<1>::OW : RsOwn<2>::OW, oh = half == 1 ? RsOwn<1>::OH : RsOwn<2>::OH;
            const int tiles_x = static_cast<int>((Wi + ow - 1) / ow), tiles_y = static_cast<int>((Hi + oh - 1) / oh);
            const int64_t spatial = B * tiles_x * tiles_y;
            int cs = static_cast<int>((C + 3) / 4 * 4);
            // a block pays ~15 us for its pixels' weights (double-precision exponentials) and the population count before its first channel:
            // slabs as large as two rounds of the 512 resident blocks allow ([8,64,512,512], 800 tiles: 32 channels 722 us, 16: 815, 8: 907)
            const int min_blocks = options().rs_bwd1_owned_blocks > 0 ? options().rs_bwd1_owned_blocks : 1024;
            while (cs > 4 && spatial * ((C + cs - 1) / cs) < min_blocks) cs = (cs / 2 + 3) / 4 * 4;
            const int cslabs = static_cast<int>((C + cs - 1) / cs);
            LaunchScope ls("resample2d_bwd_input1_owned", st, bytes1);      // both launches: the tiles and their far complement
            {
                const unsigned grid = static_cast<unsigned>(spatial * cslabs);
#if FFWM_RS_OWN_LDSW
                const size_t lds1 = 4u * (5u * RsOwn<1>::NCELL + (2u * (RsOwn<1>::NT - 1) + 1u) * RsOwn<1>::PPT * RsOwn<1>::THREADS);
                const size_t lds2 = 4u * (5u * RsOwn<2>::NCELL + (2u * (RsOwn<2>::NT - 1) + 1u) * RsOwn<2>::PPT * RsOwn<2>::THREADS);
                allow_large_lds(reinterpret_cast<const void*>(rs_bwd1_owned_kernel<1>));
                allow_large_lds(reinterpret_cast<const void*>(rs_bwd1_owned_kernel<2>));
#else
                const size_t lds1 = 0, lds2 = 0;
#endif
                if (half == 1)
                    hipLaunchKernelGGL((rs_bwd1_owned_kernel<1>), dim3(grid), dim3(RsOwn<1>::THREADS), lds1, st, (const float*)in2, (const float*)gout,
                                       (float*)gin1, (int)C, (int)Hi, (int)Wi, (int)H, (int)W, quirk, overwrite ? 1 : 0, tiles_x, tiles_y, cslabs, cs, remap, options().ablate);
                else
                    hipLaunchKernelGGL((rs_bwd1_owned_kernel<2>), dim3(grid), dim3(RsOwn<2>::THREADS), lds2, st, (const float*)in2, (const float*)gout,
                                       (float*)gin1, (int)C, (int)Hi, (int)Wi, (int)H, (int)W, quirk, overwrite ? 1 : 0, tiles_x, tiles_y, cslabs, cs, remap, options().ablate);
            }
            if (int rc = check_launch("ffwm_resample2d_backward(input1, owned tiles)")) return rc;
            {
                // one thread per pixel with ALL channels: for a flow net's field the launch reads the flow and returns
                const Geometry gf = plan(B, C, H, W, static_cast<int>(C));
                if (half == 1)
                    hipLaunchKernelGGL((rs_bwd1_far_kernel<1>), dim3(gf.grid), dim3(kBlock), 0, st, (const float*)in2, (const float*)gout, (float*)gin1,
                                       (int)C, (int)Hi, (int)Wi, (int)H, (int)W, quirk, gf.tiles_x, gf.tiles_y, gf.cslabs, gf.cs);
                else
                    hipLaunchKernelGGL((rs_bwd1_far_kernel<2>), dim3(gf.grid), dim3(kBlock), 0, st, (const float*)in2, (const float*)gout, (float*)gin1,
                                       (int)C, (int)Hi, (int)Wi, (int)H, (int)W, quirk, gf.tiles_x, gf.tiles_y, gf.cslabs, gf.cs);
            }
            if (int rc = check_launch("ffwm_resample2d_backward(input1, far)")) return rc;
            if (!gin2) return FFWM_OK;
            gin1 = nullptr;
        }
    }
    // every other path ACCUMULATES into grad_input1: an uninitialised buffer is cleared here
    if (overwrite && gin1)
        if (zero_fill(gin1, sizeof(T) * static_cast<size_t>(B) * C * Hi * Wi, st)) return FFWM_ERR_LAUNCH;
    if constexpr (sizeof(T) == 4) {
        // fp32, dilation 1, kernel_size 2 / 4 / 6: the LDS-tile kernels (scatter_variant 1 = global atomics, 2 = plane kernel)
        if (dil == 1 && half >= 1 && half <= 3 && options().scatter_variant == 0) {
            const int tiles_x = static_cast<int>((W + kTileX - 1) / kTileX);
            auto slabs = [&](int64_t spatial, int& cs, int& cslabs) {
                cs = static_cast<int>((C + 3) / 4 * 4);
                while (cs > 4 && spatial * ((C + cs - 1) / cs) < 512) cs = (cs / 2 + 3) / 4 * 4;
                cslabs = static_cast<int>((C + cs - 1) / cs);
            };
            // d_input1: planes that fit LDS whole keep the plane kernel below (measured, cfg-1: 36 vs 47 us; [8,64,128,128]: 218 vs
            // 235 us); larger planes take the tile kernel ([8,64,512,512]: 3.1 ms vs 57 ms with per-tap global atomics)
            // d_input1, ks = 4.  Small calls: the tap-lane kernel (flow-independent).  Large calls (>= 2^18 pixels): the tile kernel when
            // the flow is smooth, the tap-lane kernel when it is not -- rs_flow_irregular_kernel counts, both are launched, one returns.
            const int variant = options().rs_bwd1_variant;
            int* sel = nullptr;
            int sel_limit = 0;
            // (round 5) variant 6: the tile kernel alone for every large call -- with fixed-point cells it no longer depends on the flow
            // (the default since the box holds fixed-point cells: [8,64,512,512] 1.23 ms random / 0.91 ms smooth against the adaptive
            // pair's 1.58 / 0.98 -- profiles/r05_rs_bwd1_fixed_point_ab.txt; rs_bwd1_fixed = 2 brings the pair of rounds 3-4 back)
            const bool tile_only = gin1 && (variant == 6 || (variant == 0 && options().rs_bwd1_fixed != 2)) && B * H * W >= (1 << 18) && H >= 32;
            bool adaptive = gin1 && half == 2 && variant == 0 && !tile_only && B * H * W >= (1 << 18) && H >= 32;
            if (adaptive) {
                sel = static_cast<int*>(stream_scratch(st));
                adaptive = sel != nullptr;
            }
            // one profiling scope over the whole sequence of an adaptive call (pre-pass + the kernel that works + the one that returns)
            std::unique_ptr<LaunchScope> auto_scope;
            const double bytes1 = sizeof(T) * static_cast<double>(B) * (C * (static_cast<double>(H) * W + 2.0 * Hi * Wi) + 3.0 * H * W);
            if (adaptive) {
                auto_scope.reset(new LaunchScope("resample2d_bwd_input1_auto", st, bytes1));
                const int segs_x = static_cast<int>((W + kWave - 1) / kWave);
                const int64_t nseg = B * H * segs_x;
                sel_limit = static_cast<int>(nseg / 4 > 0 ? nseg / 4 : 1);          // "smooth": fewer than a quarter of the row segments irregular
                if (zero_fill(sel, sizeof(int), st)) return FFWM_ERR_LAUNCH;
                const int per = kBlock / kWave;
                int64_t pre_blocks = (nseg + per - 1) / per;
                if (pre_blocks > 1024) pre_blocks = 1024;
                hipLaunchKernelGGL(rs_flow_irregular_kernel, dim3(static_cast<unsigned>(pre_blocks)), dim3(kBlock), 0, st, in2, sel,
                                   (int)H, (int)W, segs_x, nseg);
                if (int rc = check_launch("ffwm_resample2d_backward(flow regularity)")) return rc;
            }
            const bool run_taplane = gin1 && half == 2 && (variant == 0 || variant == 5) && !tile_only;
            const bool run_tile = gin1 && (adaptive || tile_only || (!run_taplane && (plane_lds > 131072 || variant == 2)));
            if (run_taplane) {
                // one pixel's 16 taps x 4 channels per LDS atomic instruction -- bank-conflict-free for any flow
                // default: 4 waves x 2 rows = 64 x 8 pixel tiles, two blocks per CU (56 KB box + 22 KB staging each); variant 5: 8 waves x 2 rows
                // (one block per CU, a quarter fewer fold atomics: measured 1.61 against 1.53 ms at [8,64,512,512])
                const int nw = variant == 5 ? 8 : 4, rpt = 2;
                const int th = nw * rpt;
                const int tiles_y = static_cast<int>((H + th - 1) / th);
                int cs, cslabs;
                slabs(B * tiles_x * tiles_y, cs, cslabs);
                const unsigned grid = static_cast<unsigned>(B * tiles_x * tiles_y * cslabs);
                const int ncell = (th + 12) * 88;
                const int ps = ncell + ((4 - ncell % 32) + 32) % 32;
                const size_t lds = static_cast<size_t>(4) * ps * sizeof(double) + static_cast<size_t>(nw) * (16 + 4) * 65 * sizeof(float);
                std::unique_ptr<LaunchScope> ls;
                if (!adaptive) ls.reset(new LaunchScope("resample2d_bwd_input1_taplane", st, bytes1));
                if (nw == 8) {
                    allow_large_lds(reinterpret_cast<const void*>(rs_bwd1_taplane_kernel<2, 8>));
                    hipLaunchKernelGGL((rs_bwd1_taplane_kernel<2, 8>), dim3(grid), dim3(8 * kWave), lds, st, in2, gout, gin1, (int)C, (int)Hi,
                                       (int)Wi, (int)H, (int)W, quirk, tiles_x, tiles_y, cslabs, cs, remap, options().ablate, sel, sel_limit, 0);
                } else {
                    allow_large_lds(reinterpret_cast<const void*>(rs_bwd1_taplane_kernel<2, 4>));
                    hipLaunchKernelGGL((rs_bwd1_taplane_kernel<2, 4>), dim3(grid), dim3(4 * kWave), lds, st, in2, gout, gin1, (int)C, (int)Hi,
                                       (int)Wi, (int)H, (int)W, quirk, tiles_x, tiles_y, cslabs, cs, remap, options().ablate, sel, sel_limit, 0);
                }
                if (int rc = check_launch("ffwm_resample2d_backward(input1, tap-lane)")) return rc;
            }
            if (run_tile) {
                const int rpt = H >= 32 ? (options().rs_bwd1_rpt == 2 ? 2 : 4) : 1;
                const int tiles_y = static_cast<int>((H + 4 * rpt - 1) / (4 * rpt));
                int cs, cslabs;
                slabs(B * tiles_x * tiles_y, cs, cslabs);
                const unsigned grid = static_cast<unsigned>(B * tiles_x * tiles_y * cslabs);
                const bool fixed_cells = options().rs_bwd1_fixed != 2;          // 32-bit fixed-point box cells (round 5); 2 = double cells
                const size_t lds = static_cast<size_t>(4 * rpt + 12) * kRsBoxW * 4 * (fixed_cells ? sizeof(int) : sizeof(double));
                std::unique_ptr<LaunchScope> ls;
                if (!adaptive) ls.reset(new LaunchScope("resample2d_bwd_input1_tile", st, bytes1));
#define FFWM_RS_B1T_(HH, RR, FX)                                                                              \
    do {                                                                                                      \
        allow_large_lds(reinterpret_cast<const void*>(rs_bwd1_tile_kernel<HH, RR, FX>));                      \
        hipLaunchKernelGGL((rs_bwd1_tile_kernel<HH, RR, FX>), dim3(grid), dim3(kBlock), lds, st, in2, gout, gin1, (int)C, \
                           (int)Hi, (int)Wi, (int)H, (int)W, quirk, tiles_x, tiles_y, cslabs, cs, remap, options().ablate, \
                           adaptive ? sel : nullptr, sel_limit, 1); \
    } while (0)
#define FFWM_RS_B1T(HH, RR) do { if (fixed_cells) FFWM_RS_B1T_(HH, RR, true); else FFWM_RS_B1T_(HH, RR, false); } while (0)
                if (rpt == 4) { if (half == 1) FFWM_RS_B1T(1, 4); else if (half == 2) FFWM_RS_B1T(2, 4); else FFWM_RS_B1T(3, 4); }
                else if (rpt == 2) { if (half == 1) FFWM_RS_B1T(1, 2); else if (half == 2) FFWM_RS_B1T(2, 2); else FFWM_RS_B1T(3, 2); }
                else { if (half == 1) FFWM_RS_B1T(1, 1); else if (half == 2) FFWM_RS_B1T(2, 1); else FFWM_RS_B1T(3, 1); }
#undef FFWM_RS_B1T
#undef FFWM_RS_B1T_
                if (int rc = check_launch("ffwm_resample2d_backward(input1, tile)")) return rc;
            }
            auto_scope.reset();
            if (run_taplane || run_tile) gin1 = nullptr;
            if (gin2) {
                const int tiles_y = static_cast<int>((H + 3) / 4);
                int cs, cslabs;
                slabs(B * tiles_x * tiles_y, cs, cslabs);
                const unsigned grid = static_cast<unsigned>(B * tiles_x * tiles_y * cslabs);
                const size_t lds = static_cast<size_t>(2) * 16 * kRsBoxW * 16;
                if (cslabs > 1)          // the slabs' partial results are added atomically
                    if (zero_fill(gin2, sizeof(T) * static_cast<size_t>(B) * 3 * H * W, st)) return FFWM_ERR_LAUNCH;
                const double bytes = sizeof(T) * static_cast<double>(B) * H * W * (2.0 * C + 6.0);
                LaunchScope ls("resample2d_bwd_input2_lds", st, bytes);
#define FFWM_RS_B2L(HH)                                                                                       \
    hipLaunchKernelGGL((rs_bwd2_lds_kernel<HH, 1>), dim3(grid), dim3(kBlock), lds, st, in1, in2, gout, gin2, (int)C, \
                       (int)Hi, (int)Wi, (int)H, (int)W, tiles_x, tiles_y, cslabs, cs, remap)
                if (half == 1) FFWM_RS_B2L(1); else if (half == 2) FFWM_RS_B2L(2); else FFWM_RS_B2L(3);
#undef FFWM_RS_B2L
                if (int rc = check_launch("ffwm_resample2d_backward(input2, lds)")) return rc;
                gin2 = nullptr;
            }
            if (!gin1 && !gin2) return FFWM_OK;
        }
    }
    if (gin1 && plane_lds <= 131072 && half >= 1 && half <= 3 && options().scatter_variant != 1) {      // fp64, dilation > 1, or scatter_variant 2
        const double bytes = sizeof(T) * static_cast<double>(B) * (C * (static_cast<double>(H) * W + 2.0 * Hi * Wi) + 3.0 * H * W);
        int cg = static_cast<int>(131072 / plane_lds);
        if (cg > C) cg = static_cast<int>(C);
        while (cg > 1 && B * ((C + cg - 1) / cg) < 512) cg = (cg + 1) / 2;   // keep >= 2 blocks per CU
        const int groups = static_cast<int>((C + cg - 1) / cg);
        int nsplit = 1;                      // the per-pixel Gaussian weights (double exp) dominate: fill the chip
        while (B * groups * nsplit < 256 && nsplit * 2 * kPlaneThreads <= H * W) nsplit *= 2;
        const unsigned grid = static_cast<unsigned>(B * groups * nsplit);
        const size_t lds = static_cast<size_t>(cg) * plane_lds;
        allow_large_lds(reinterpret_cast<const void*>(rs_bwd1_plane_kernel<T, 1>));
        allow_large_lds(reinterpret_cast<const void*>(rs_bwd1_plane_kernel<T, 2>));
        allow_large_lds(reinterpret_cast<const void*>(rs_bwd1_plane_kernel<T, 3>));
        {
            LaunchScope ls("resample2d_bwd_input1_plane", st, bytes);
            if (half == 1)
                hipLaunchKernelGGL((rs_bwd1_plane_kernel<T, 1>), dim3(grid), dim3(kPlaneThreads), lds, st, in2, gout, gin1,
                                   (int)C, (int)Hi, (int)Wi, (int)H, (int)W, dil, quirk, cg, groups, nsplit);
            else if (half == 2)
                hipLaunchKernelGGL((rs_bwd1_plane_kernel<T, 2>), dim3(grid), dim3(kPlaneThreads), lds, st, in2, gout, gin1,
                                   (int)C, (int)Hi, (int)Wi, (int)H, (int)W, dil, quirk, cg, groups, nsplit);
            else
                hipLaunchKernelGGL((rs_bwd1_plane_kernel<T, 3>), dim3(grid), dim3(kPlaneThreads), lds, st, in2, gout, gin1,
                                   (int)C, (int)Hi, (int)Wi, (int)H, (int)W, dil, quirk, cg, groups, nsplit);
        }
        if (int rc = check_launch("ffwm_resample2d_backward(input1, plane)")) return rc;
        gin1 = nullptr;
    }
    if (gin1) {
        const double bytes = sizeof(T) * static_cast<double>(B) * H * W * (2.0 * C + 3.0);
        const Geometry g = plan(B, C, H, W, 32);
        LaunchScope ls("resample2d_bwd_input1", st, bytes);
#define FFWM_RS_B1(HH)                                                                               \
    case 2 * HH:                                                                                     \
        hipLaunchKernelGGL((rs_bwd1_kernel<T, HH>), dim3(g.grid), dim3(kBlock), 0, st, in2, gout,     \
                           gin1, (int)C, (int)Hi, (int)Wi, (int)H, (int)W, dil, quirk, g.tiles_x,     \
                           g.tiles_y, g.cslabs, g.cs, remap);                                        \
        break;
        switch (ks & ~1) {
            FFWM_RS_B1(1) FFWM_RS_B1(2) FFWM_RS_B1(3)
            default: {
                const int64_t n = B * C * H * W;
                hipLaunchKernelGGL((rs_bwd1_generic<T>), dim3(generic_grid(n)), dim3(kBlock), 0, st, in2,
                                   gout, gin1, n, (int)C, (int)Hi, (int)Wi, (int)H, (int)W, ks, dil, quirk);
            }
        }
#undef FFWM_RS_B1
        if (int rc = check_launch("ffwm_resample2d_backward(input1)")) return rc;
    }
    if (gin2) {
        const double bytes = sizeof(T) * static_cast<double>(B) * H * W * (2.0 * C + 6.0);
        const int tiles_x = static_cast<int>((W + kWave - 1) / kWave);
        const unsigned grid = static_cast<unsigned>(B * H * tiles_x);
        LaunchScope ls("resample2d_bwd_input2", st, bytes);
#define FFWM_RS_B2(HH)                                                                               \
    case 2 * HH:                                                                                     \
        hipLaunchKernelGGL((rs_bwd2_kernel<T, HH>), dim3(grid), dim3(kBlock), 0, st, in1, in2, gout,  \
                           gin2, (int)C, (int)Hi, (int)Wi, (int)H, (int)W, dil, tiles_x);            \
        break;
        switch (ks & ~1) {
            FFWM_RS_B2(1) FFWM_RS_B2(2) FFWM_RS_B2(3)
            default: {
                const int64_t n = B * 3 * H * W;
                hipLaunchKernelGGL((rs_bwd2_generic<T>), dim3(generic_grid(n)), dim3(kBlock), 0, st, in1,
                                   in2, gout, gin2, n, (int)C, (int)Hi, (int)Wi, (int)H, (int)W, ks, dil);
            }
        }
#undef FFWM_RS_B2
        if (int rc = check_launch("ffwm_resample2d_backward(input2)")) return rc;
    }
    return FFWM_OK;
}

}  // namespace
}  // namespace ffwm

using namespace ffwm;

extern "C" int ffwm_resample2d_forward(const void* input1, const void* input2, void* output, int64_t B,
                                       int64_t C, int64_t Hi, int64_t Wi, int64_t H, int64_t W,
                                       int kernel_size, int dilation, int dtype, void* stream) {
    const char* fn = "ffwm_resample2d_forward";
    FFWM_REQUIRE(input1 && input2 && output, FFWM_ERR_ARG, "%s: NULL tensor pointer", fn);
    if (int rc = check_dims(fn, B, C, Hi, Wi, H, W, kernel_size, dilation, dtype)) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dtype == FFWM_F32)
        return launch_fwd<float>((const float*)input1, (const float*)input2, (float*)output, B, C, Hi, Wi,
                                 H, W, kernel_size, dilation, st);
    return launch_fwd<double>((const double*)input1, (const double*)input2, (double*)output, B, C, Hi, Wi,
                              H, W, kernel_size, dilation, st);
}

extern "C" int ffwm_resample2d_backward(const void* input1, const void* input2, const void* grad_output,
                                        void* grad_input1, void* grad_input2, int64_t B, int64_t C,
                                        int64_t Hi, int64_t Wi, int64_t H, int64_t W, int kernel_size,
                                        int dilation, int reference_quirk, int dtype, void* stream) {
    const char* fn = "ffwm_resample2d_backward";
    FFWM_REQUIRE(input1 && input2 && grad_output, FFWM_ERR_ARG, "%s: NULL tensor pointer", fn);
    if (int rc = check_dims(fn, B, C, Hi, Wi, H, W, kernel_size, dilation, dtype)) return rc;
    if (!grad_input1 && !grad_input2) return FFWM_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    FFWM_REQUIRE(reference_quirk >= 0 && reference_quirk <= 3, FFWM_ERR_ARG, "%s: reference_quirk is a 2-bit flag word", fn);
    if (dtype == FFWM_F32)
        return launch_bwd<float>((const float*)input1, (const float*)input2, (const float*)grad_output,
                                 (float*)grad_input1, (float*)grad_input2, B, C, Hi, Wi, H, W, kernel_size,
                                 dilation, reference_quirk, st);
    return launch_bwd<double>((const double*)input1, (const double*)input2, (const double*)grad_output,
                              (double*)grad_input1, (double*)grad_input2, B, C, Hi, Wi, H, W, kernel_size,
                              dilation, reference_quirk, st);
}

// grad_output read through its element strides (NULL / contiguous: the entry point above).  Any other layout takes the per-element
// kernels (resample2d_kernel.cu:98-330 read gradOutput with DIM3_INDEX and the tensor's strides): correct for any view, not tuned.
// grad_input1 += (reference_quirk bit 1: overwritten -- the library clears it first), grad_input2 is overwritten.
extern "C" int ffwm_resample2d_backward_strided(const void* input1, const void* input2, const void* grad_output,
                                                const int64_t* grad_output_strides, void* grad_input1, void* grad_input2, int64_t B,
                                                int64_t C, int64_t Hi, int64_t Wi, int64_t H, int64_t W, int kernel_size, int dilation,
                                                int reference_quirk, int dtype, void* stream) {
    const char* fn = "ffwm_resample2d_backward_strided";
    if (go_contiguous(grad_output_strides, C, H, W))
        return ffwm_resample2d_backward(input1, input2, grad_output, grad_input1, grad_input2, B, C, Hi, Wi, H, W, kernel_size, dilation,
                                        reference_quirk, dtype, stream);
    FFWM_REQUIRE(input1 && input2 && grad_output, FFWM_ERR_ARG, "%s: NULL tensor pointer", fn);
    if (int rc = check_dims(fn, B, C, Hi, Wi, H, W, kernel_size, dilation, dtype)) return rc;
    FFWM_REQUIRE(reference_quirk >= 0 && reference_quirk <= 3, FFWM_ERR_ARG, "%s: reference_quirk is a 2-bit flag word", fn);
    for (int d = 0; d < 4; ++d)
        FFWM_REQUIRE(grad_output_strides[d] >= 0, FFWM_ERR_ARG, "%s: negative strides are not supported", fn);
    FFWM_REQUIRE(grad_output_strides[3] != 0, FFWM_ERR_ARG, "%s: a grad_output expanded along its last dimension (stride 0) is not supported", fn);
    if (!grad_input1 && !grad_input2) return FFWM_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const GoStrides gs{grad_output_strides[0], grad_output_strides[1], grad_output_strides[2], grad_output_strides[3]};
    const size_t esz = dtype == FFWM_F32 ? 4 : 8;
    if (grad_input1 && (reference_quirk & 2))
        if (zero_fill(grad_input1, esz * static_cast<size_t>(B) * C * Hi * Wi, st)) return FFWM_ERR_LAUNCH;
    LaunchScope ls("resample2d_bwd_strided", st, static_cast<double>(esz) * B * (C * (static_cast<double>(H) * W + 2.0 * Hi * Wi) + 6.0 * H * W));
    const int ks = kernel_size & ~1, quirk = reference_quirk & 1;
    if (grad_input1) {
        const int64_t n = B * C * H * W;
        if (dtype == FFWM_F32)
            hipLaunchKernelGGL((rs_bwd1_generic<float>), dim3(generic_grid(n)), dim3(kBlock), 0, st, (const float*)input2, (const float*)grad_output,
                               (float*)grad_input1, n, (int)C, (int)Hi, (int)Wi, (int)H, (int)W, ks, dilation, quirk, gs);
        else
            hipLaunchKernelGGL((rs_bwd1_generic<double>), dim3(generic_grid(n)), dim3(kBlock), 0, st, (const double*)input2, (const double*)grad_output,
                               (double*)grad_input1, n, (int)C, (int)Hi, (int)Wi, (int)H, (int)W, ks, dilation, quirk, gs);
        if (int rc = check_launch(fn)) return rc;
    }
    if (grad_input2) {
        const int64_t n = B * 3 * H * W;
        if (dtype == FFWM_F32)
            hipLaunchKernelGGL((rs_bwd2_generic<float>), dim3(generic_grid(n)), dim3(kBlock), 0, st, (const float*)input1, (const float*)input2,
                               (const float*)grad_output, (float*)grad_input2, n, (int)C, (int)Hi, (int)Wi, (int)H, (int)W, ks, dilation, gs);
        else
            hipLaunchKernelGGL((rs_bwd2_generic<double>), dim3(generic_grid(n)), dim3(kBlock), 0, st, (const double*)input1, (const double*)input2,
                               (const double*)grad_output, (double*)grad_input2, n, (int)C, (int)Hi, (int)Wi, (int)H, (int)W, ks, dilation, gs);
        if (int rc = check_launch(fn)) return rc;
    }
    return FFWM_OK;
}
