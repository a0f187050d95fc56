// flownet_ops.hip -- the small layers of FlowNet's eval forward (BASELINE configs[1]) for gfx950.
//
// /root/reference/models/base_networks.py:59-165 at batch 6 is ~215 kernel launches per forward through the stock
// library path, 80 % of them shorter than 10 us: per conv block a convolution, a bias add, an eval BatchNorm and a
// LeakyReLU; per flow head an NHWC implicit-GEMM for TWO output channels wrapped in three layout transposes and a
// zero-fill, a bias add and a tanh; per flow upsampler a 2 -> 2 channel transposed convolution with the same wrapping;
// per decoder level a concatenation copy.  With BatchNorm folded into the conv weights (ffwm_amd/flownet_eval.py) what
// remains around the dense convolutions is done here, one launch each:
//   ffwm_bias_act_forward   y = act(h + bias[c]), act in {none, LeakyReLU, tanh}; written in place and / or into a channel
//                           slice of the next concatenation buffer (no torch.cat kernel)
//   ffwm_flow_head_forward  3x3 / pad 1 convolution C -> 2 + bias + tanh (predict_flow*, :45-49): HBM-bound on its input,
//                           channels split over the lanes for the tiny planes (1024 channels at 2 x 2)
//   ffwm_flow_up_forward    ConvTranspose2d(2, 2, 4, 2, 1) + bias (upsampled_flow*_to_*, :104-109) into the cat slice
#include "common.hpp"

namespace ffwm {
namespace {

template <int ACT>
__device__ __forceinline__ float act_f(float v, float slope) {
    if constexpr (ACT == 1) return v > 0.f ? v : v * slope;        // ATen leaky_relu: x > 0 ? x : x * slope
    if constexpr (ACT == 2) return tanhf(v);
    return v;
}

// one thread per 4 consecutive pixels of one (b, c) plane when HW % 4 == 0, else per pixel
template <int ACT, int VEC>
__global__ void __launch_bounds__(kBlock)
bias_act_kernel(const float* __restrict__ h, const float* __restrict__ bias, float* __restrict__ y, float* __restrict__ y2,
                int64_t total, int C, int HW, int64_t ybs, int64_t y2bs, float slope) {
    const int hwv = HW / VEC;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < total; i += static_cast<int64_t>(gridDim.x) * kBlock) {
        const int64_t plane = i / hwv;
        const int p = static_cast<int>(i - plane * hwv) * VEC;
        const int c = static_cast<int>(plane % C);
        const int64_t b = plane / C;
        const float bc = bias ? bias[c] : 0.f;
        const float* src = h + plane * HW + p;
        if constexpr (VEC == 4) {
            const float4 v = *reinterpret_cast<const float4*>(src);
            const float4 o = {act_f<ACT>(v.x + bc, slope), act_f<ACT>(v.y + bc, slope), act_f<ACT>(v.z + bc, slope), act_f<ACT>(v.w + bc, slope)};
            if (y) *reinterpret_cast<float4*>(y + b * ybs + static_cast<int64_t>(c) * HW + p) = o;
            if (y2) *reinterpret_cast<float4*>(y2 + b * y2bs + static_cast<int64_t>(c) * HW + p) = o;
        } else {
            const float o = act_f<ACT>(*src + bc, slope);
            if (y) y[b * ybs + static_cast<int64_t>(c) * HW + p] = o;
            if (y2) y2[b * y2bs + static_cast<int64_t>(c) * HW + p] = o;
        }
    }
}

// 3x3 / stride 1 / pad 1 convolution C -> 2 + bias + tanh.  A block = P pixels x S channel splits (P * S = 256):
// P = 64 for planes >= 64 pixels (a wave covers 64 consecutive pixels: coalesced rows), smaller powers of two for the
// 2 x 2 ... 4 x 4 planes so that the 1024 input channels are spread over all 256 threads.  Partial sums meet in LDS.
template <int P>
__global__ void __launch_bounds__(kBlock)
flow_head_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                 float* __restrict__ y, int C, int H, int W, int tiles) {
    constexpr int S = kBlock / P;
    __shared__ float red[2][S][P];
    const int px_l = threadIdx.x % P, split = threadIdx.x / P;
    const int tile = blockIdx.x % tiles, b = blockIdx.x / tiles;
    const int HW = H * W;
    const int p = tile * P + px_l;
    const bool live = p < HW;
    const int py = live ? p / W : 0, pxx = live ? p - py * W : 0;
    unsigned off[9];
    float msk[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const int yy = py + k / 3 - 1, xx = pxx + k % 3 - 1;
        const bool in = live && yy >= 0 && yy < H && xx >= 0 && xx < W;
        off[k] = in ? static_cast<unsigned>(yy * W + xx) : 0u;
        msk[k] = in ? 1.f : 0.f;
    }
    float a0 = 0.f, a1 = 0.f;
    const float* xb = x + static_cast<int64_t>(b) * C * HW;
#pragma unroll 4
    for (int c = split; c < C; c += S) {
        const float* xc = xb + static_cast<int64_t>(c) * HW;
        const float* w0 = w + static_cast<int64_t>(c) * 9;                       // w[0][c][:][:]
        const float* w1 = w + (static_cast<int64_t>(C) + c) * 9;                 // w[1][c][:][:]
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const float v = msk[k] != 0.f ? xc[off[k]] : 0.f;
            a0 = __builtin_fmaf(v, w0[k], a0);
            a1 = __builtin_fmaf(v, w1[k], a1);
        }
    }
    red[0][split][px_l] = a0;
    red[1][split][px_l] = a1;
    __syncthreads();
    if (split < 2 && live) {                      // thread (px, co = split) folds the S partial sums of its output
        float s = bias ? bias[split] : 0.f;
#pragma unroll
        for (int k = 0; k < S; ++k) s += red[split][k][px_l];
        y[(static_cast<int64_t>(b) * 2 + split) * HW + p] = tanhf(s);
    }
}

// ConvTranspose2d(2, 2, kernel 4, stride 2, pad 1) + bias: out[b, co, y, x] = bias[co] + sum_ci sum_ky,kx
// in[b, ci, (y + 1 - ky) / 2, (x + 1 - kx) / 2] * w[ci, co, ky, kx] over the taps whose division is exact and in range.
__global__ void __launch_bounds__(kBlock)
flow_up_kernel(const float* __restrict__ in, const float* __restrict__ w, const float* __restrict__ bias,
               float* __restrict__ out, int64_t total, int H, int W, int64_t obs) {
    const int Ho = 2 * H, Wo = 2 * W;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < total; i += static_cast<int64_t>(gridDim.x) * kBlock) {
        const int xo = static_cast<int>(i % Wo);
        const int yo = static_cast<int>((i / Wo) % Ho);
        const int64_t b = i / (static_cast<int64_t>(Wo) * Ho);
        float a0 = bias ? bias[0] : 0.f, a1 = bias ? bias[1] : 0.f;
        const float* ib = in + b * 2 * H * W;
#pragma unroll
        for (int ky = 0; ky < 4; ++ky) {
            const int ty = yo + 1 - ky;
            if (ty < 0 || (ty & 1) || (ty >> 1) >= H) continue;
#pragma unroll
            for (int kx = 0; kx < 4; ++kx) {
                const int tx = xo + 1 - kx;
                if (tx < 0 || (tx & 1) || (tx >> 1) >= W) continue;
                const int ip = (ty >> 1) * W + (tx >> 1);
                const float v0 = ib[ip], v1 = ib[H * W + ip];
                const int k = ky * 4 + kx;
                a0 = __builtin_fmaf(v0, w[k], a0);            // w[ci = 0][co = 0]
                a0 = __builtin_fmaf(v1, w[32 + k], a0);       // w[1][0]
                a1 = __builtin_fmaf(v0, w[16 + k], a1);       // w[0][1]
                a1 = __builtin_fmaf(v1, w[48 + k], a1);       // w[1][1]
            }
        }
        float* ob = out + b * obs + static_cast<int64_t>(yo) * Wo + xo;
        ob[0] = a0;
        ob[static_cast<int64_t>(Ho) * Wo] = a1;
    }
}

// ---- training: the backward of the two-channel layers (round 3).  Through the vendor library a flow head's data gradient is an
// NHWC implicit GEMM with two of its 64 output rows in use, wrapped in three layout transposes and a fill, and the 2 -> 2
// upsampler's is the same again: ~150 launches per train step of the two flow nets for a few MFLOP.
//
// flow head, y = tanh(conv3x3(x, w) + b):  gz = go * (1 - y^2)  [B, 2, H, W]  (written out: the weight gradient's row operand)
//                                          gx[b, c, p] = sum_{k, r, s} gz[b, k, p + (1 - r, 1 - s)] * w[k, c, r, s]
// P pixels x S = 256 / P channel splits per block: a thread forms the 18 gz values around its pixel once and walks its channels.
template <int P>
__global__ void __launch_bounds__(kBlock)
flow_head_bwd_kernel(const float* __restrict__ y, const float* __restrict__ go, const float* __restrict__ w, float* __restrict__ gz,
                     float* __restrict__ gx, int C, int H, int W, int tiles) {
    constexpr int S = kBlock / P;
    const int px_l = threadIdx.x % P, split = threadIdx.x / P;
    const int tile = blockIdx.x % tiles, b = blockIdx.x / tiles;
    const int HW = H * W;
    const int p = tile * P + px_l;
    if (p >= HW) return;
    const int py = p / W, pxx = p - py * W;
    const float* yb = y + static_cast<int64_t>(b) * 2 * HW;
    const float* gb = go + static_cast<int64_t>(b) * 2 * HW;
    float z0[9], z1[9];                 // gz at the pixel that tap (r, s) of the forward reached THIS pixel from: p + (1 - r, 1 - s)
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const int yy = py + 1 - k / 3, xx = pxx + 1 - k % 3;
        const bool in = yy >= 0 && yy < H && xx >= 0 && xx < W;
        const int o = in ? yy * W + xx : 0;
        const float y0 = yb[o], y1 = yb[HW + o];
        z0[k] = in ? gb[o] * (1.f - y0 * y0) : 0.f;
        z1[k] = in ? gb[HW + o] * (1.f - y1 * y1) : 0.f;
    }
    if (split == 0) {                   // k = 4 is the pixel itself
        gz[static_cast<int64_t>(b) * 2 * HW + p] = z0[4];
        gz[(static_cast<int64_t>(b) * 2 + 1) * HW + p] = z1[4];
    }
    float* gxb = gx + static_cast<int64_t>(b) * C * HW + p;
    for (int c = split; c < C; c += S) {
        const float* w0 = w + static_cast<int64_t>(c) * 9;
        const float* w1 = w + (static_cast<int64_t>(C) + c) * 9;
        float a = 0.f;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            a = __builtin_fmaf(z0[k], w0[k], a);
            a = __builtin_fmaf(z1[k], w1[k], a);
        }
        gxb[static_cast<int64_t>(c) * HW] = a;
    }
}

// 2 -> 2 upsampler, d(input): gx[b, ci, iy, ix] = sum_{co, ky, kx} go[b, co, 2 iy - 1 + ky, 2 ix - 1 + kx] * w[ci, co, ky, kx]
// (go may be a channel slice of the decoder's concatenation gradient: gobs = its batch stride)
__global__ void __launch_bounds__(kBlock)
flow_up_bwd_kernel(const float* __restrict__ go, const float* __restrict__ w, float* __restrict__ gx, int64_t total, int H, int W,
                   int64_t gobs) {
    const int Ho = 2 * H, Wo = 2 * W;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < total; i += static_cast<int64_t>(gridDim.x) * kBlock) {
        const int ix = static_cast<int>(i % W);
        const int iy = static_cast<int>((i / W) % H);
        const int64_t b = i / (static_cast<int64_t>(W) * H);
        const float* gb = go + b * gobs;
        float a0 = 0.f, a1 = 0.f;
#pragma unroll
        for (int ky = 0; ky < 4; ++ky) {
            const int oy = 2 * iy - 1 + ky;
            if (oy < 0 || oy >= Ho) continue;
#pragma unroll
            for (int kx = 0; kx < 4; ++kx) {
                const int ox = 2 * ix - 1 + kx;
                if (ox < 0 || ox >= Wo) continue;
                const float g0 = gb[oy * Wo + ox], g1 = gb[static_cast<int64_t>(Ho) * Wo + oy * Wo + ox];
                const int k = ky * 4 + kx;
                a0 = __builtin_fmaf(g0, w[k], a0);            // w[ci = 0][co = 0]
                a0 = __builtin_fmaf(g1, w[16 + k], a0);       // w[0][1]
                a1 = __builtin_fmaf(g0, w[32 + k], a1);       // w[1][0]
                a1 = __builtin_fmaf(g1, w[48 + k], a1);       // w[1][1]
            }
        }
        float* ob = gx + b * 2 * H * W + static_cast<int64_t>(iy) * W + ix;
        ob[0] = a0;
        ob[static_cast<int64_t>(H) * W] = a1;
    }
}

unsigned ew_grid(int64_t n) {
    int64_t blocks = (n + kBlock - 1) / kBlock;
    return static_cast<unsigned>(blocks > 256 * 32 ? 256 * 32 : (blocks < 1 ? 1 : blocks));
}


// ------------------------------------------------------------------------------ thin-channel layers, direct (round 6)
// FlowNet's full-resolution layers have 6 ... 34 input and 16 ... 64 output channels (conv0 6 -> 64, inter_conv1 34 -> 32, inter_conv0
// 18 -> 16 at 64^2 / 128^2): a 64 x 64 MFMA tile is mostly padding there (11-16 TFLOP/s on the Winograd kernel, 21-30 us per layer).  Here a lane owns ONE pixel and KT output channels in registers, walks the
// input channels with the 3 x 3 window in registers, and takes the weights -- pre-arranged [C][taps][K] by the host, wave-uniform -- as scalar
// operands of its FMAs.  A wave = 64 consecutive pixels of a row (coalesced loads and stores); no LDS, no barriers.
template <int KT, int G>
__global__ void __launch_bounds__(kBlock)
conv3x3_direct_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias, float* __restrict__ y,
                      int C, int H, int W, int K, int segs_x, int nseg, int act, float slope) {
    const int lane = threadIdx.x & (kWave - 1), wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    const int seg = blockIdx.x * (kBlock / kWave) + wave;
    if (seg >= nseg) return;
    const int kg = blockIdx.y * KT;
    const int xs = seg % segs_x, row = seg / segs_x;          // row = b * H + yy
    const int b = row / H, yy = row - b * H;
    const int xx = xs * kWave + lane;
    const bool live = xx < W;
    const int HW = H * W;
    const rsrc_t rx = make_rsrc(x + static_cast<size_t>(b) * C * HW, static_cast<unsigned>(static_cast<size_t>(C) * HW * 4));   // channels past C read 0
    unsigned off[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int y2 = yy + t / 3 - 1, x2 = xx + t % 3 - 1;
        off[t] = (live && y2 >= 0 && y2 < H && x2 >= 0 && x2 < W) ? static_cast<unsigned>(y2 * W + x2) * 4u : 0xFFFFFFF0u;
    }
    float acc[KT];
#pragma unroll
    for (int k = 0; k < KT; ++k) acc[k] = 0.f;
    const unsigned cstep = static_cast<unsigned>(HW) * 4u;
    // G channels' windows per step, the next step's requested before this one's FMAs (a wave has no neighbours to hide a load behind)
    float in[G][9], nx[G][9];
    auto request = [&](int c0, float (&d)[G][9]) {
#pragma unroll
        for (int gch = 0; gch < G; ++gch) {
            const unsigned co = static_cast<unsigned>(c0 + gch) * cstep;
#pragma unroll
            for (int t = 0; t < 9; ++t) d[gch][t] = buf_ld<float>(rx, (c0 + gch < C && off[t] != 0xFFFFFFF0u) ? off[t] + co : 0xFFFFFFF0u);
        }
    };
    request(0, in);
    for (int c = 0; c < C; c += G) {
        request(c + G, nx);
#pragma unroll
        for (int gch = 0; gch < G; ++gch) {
            if (c + gch >= C) break;
            const float* wp = w + (static_cast<size_t>(c + gch) * 9) * K + kg;          // wave-uniform: scalar loads
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int k = 0; k < KT; ++k) acc[k] = __builtin_fmaf(wp[t * K + k], in[gch][t], acc[k]);
        }
#pragma unroll
        for (int gch = 0; gch < G; ++gch)
#pragma unroll
            for (int t = 0; t < 9; ++t) in[gch][t] = nx[gch][t];
    }
    if (!live) return;
    float* yp = y + (static_cast<size_t>(b) * K + kg) * HW + static_cast<size_t>(yy) * W + xx;
#pragma unroll
    for (int k = 0; k < KT; ++k) {
        const float v = acc[k] + (bias ? bias[kg + k] : 0.f);
        yp[static_cast<size_t>(k) * HW] = act == 1 ? (v > 0.f ? v : v * slope) : v;
    }
}

}  // namespace
}  // namespace ffwm

using namespace ffwm;

extern "C" int ffwm_bias_act_forward(const void* h, const void* bias, void* y, void* y2, int64_t B, int64_t C, int64_t HW,
                                     int64_t y_batch_stride, int64_t y2_batch_stride, int act, double negative_slope,
                                     int dtype, void* stream) {
    const char* fn = "ffwm_bias_act_forward";
    FFWM_REQUIRE(dtype == FFWM_F32, FFWM_ERR_DTYPE, "%s: float32 only", fn);
    FFWM_REQUIRE(h && (y || y2), FFWM_ERR_ARG, "%s: NULL tensor pointer", fn);
    FFWM_REQUIRE(B > 0 && C > 0 && HW > 0 && HW < (1LL << 31) && act >= 0 && act <= 2, FFWM_ERR_ARG, "%s: bad sizes / activation", fn);
    FFWM_REQUIRE((!y || y_batch_stride >= C * HW) && (!y2 || y2_batch_stride >= C * HW), FFWM_ERR_ARG,
                 "%s: a destination batch stride is smaller than C * HW", fn);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const bool vec = HW % 4 == 0 && (reinterpret_cast<uintptr_t>(h) % 16 == 0) &&
                     (!y || (reinterpret_cast<uintptr_t>(y) % 16 == 0 && y_batch_stride % 4 == 0)) &&
                     (!y2 || (reinterpret_cast<uintptr_t>(y2) % 16 == 0 && y2_batch_stride % 4 == 0));
    const int64_t total = B * C * HW / (vec ? 4 : 1);
    LaunchScope ls("flownet_bias_act", st, 4.0 * B * C * HW * (1.0 + (y ? 1 : 0) + (y2 ? 1 : 0)));
#define FFWM_BA(ACT, VEC)                                                                                              \
    hipLaunchKernelGGL((bias_act_kernel<ACT, VEC>), dim3(ew_grid(total)), dim3(kBlock), 0, st, (const float*)h,        \
                       (const float*)bias, (float*)y, (float*)y2, total, (int)C, (int)HW, y_batch_stride, y2_batch_stride, \
                       (float)negative_slope)
    if (vec) {
        if (act == 0) FFWM_BA(0, 4); else if (act == 1) FFWM_BA(1, 4); else FFWM_BA(2, 4);
    } else {
        if (act == 0) FFWM_BA(0, 1); else if (act == 1) FFWM_BA(1, 1); else FFWM_BA(2, 1);
    }
#undef FFWM_BA
    return check_launch(fn);
}

extern "C" int ffwm_flow_head_forward(const void* x, const void* weight, const void* bias, void* y, int64_t B, int64_t C,
                                      int64_t H, int64_t W, int dtype, void* stream) {
    const char* fn = "ffwm_flow_head_forward";
    FFWM_REQUIRE(dtype == FFWM_F32, FFWM_ERR_DTYPE, "%s: float32 only", fn);
    FFWM_REQUIRE(x && weight && y, FFWM_ERR_ARG, "%s: NULL tensor pointer", fn);
    FFWM_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && H * W < (1LL << 28), FFWM_ERR_ARG, "%s: bad sizes", fn);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int64_t HW = H * W;
    LaunchScope ls("flownet_flow_head", st, 4.0 * (B * C * HW + 18.0 * C + 2.0 * B * HW));
#define FFWM_FH(P)                                                                                                     \
    do {                                                                                                               \
        const int tiles = static_cast<int>((HW + P - 1) / P);                                                          \
        hipLaunchKernelGGL((flow_head_kernel<P>), dim3(static_cast<unsigned>(B * tiles)), dim3(kBlock), 0, st,         \
                           (const float*)x, (const float*)weight, (const float*)bias, (float*)y, (int)C, (int)H, (int)W, tiles); \
    } while (0)
    // the widest pixel tile that still gives ~100 blocks: a head's C x 9 taps run serially in C / (256 / P) steps per thread, and the
    // 8 x 8 / 16 x 16 levels of FlowNet (256 / 128 channels, batch 6) had 6 / 24 blocks of 64 pixels walking 64 / 32 channels each
    auto blocks = [&](int64_t P) { return B * ((HW + P - 1) / P); };
    if (HW >= 64 && blocks(64) >= 96) FFWM_FH(64); else if (HW >= 16 && blocks(16) >= 96) FFWM_FH(16); else FFWM_FH(4);
#undef FFWM_FH
    return check_launch(fn);
}

extern "C" int ffwm_flow_up_forward(const void* flow, const void* weight, const void* bias, void* out, int64_t B, int64_t H,
                                    int64_t W, int64_t out_batch_stride, int dtype, void* stream) {
    const char* fn = "ffwm_flow_up_forward";
    FFWM_REQUIRE(dtype == FFWM_F32, FFWM_ERR_DTYPE, "%s: float32 only", fn);
    FFWM_REQUIRE(flow && weight && out, FFWM_ERR_ARG, "%s: NULL tensor pointer", fn);
    FFWM_REQUIRE(B > 0 && H > 0 && W > 0 && H * W < (1LL << 26) && out_batch_stride >= 8 * H * W, FFWM_ERR_ARG, "%s: bad sizes", fn);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int64_t total = B * 4 * H * W;
    LaunchScope ls("flownet_flow_up", st, 4.0 * (B * 2 * H * W + 2.0 * total));
    hipLaunchKernelGGL(flow_up_kernel, dim3(ew_grid(total)), dim3(kBlock), 0, st, (const float*)flow, (const float*)weight,
                       (const float*)bias, (float*)out, total, (int)H, (int)W, out_batch_stride);
    return check_launch(fn);
}

// Backward of ffwm_flow_head_forward: y = its output, grad_y [B, 2, H, W] contiguous.  Writes grad_z = grad_y * (1 - y^2) [B, 2, H, W]
// (the row operand of the weight gradient: ffwm_conv2d_wgrad_tiled(grad_z, x, ...); its sum over batch and pixels is the bias
// gradient) and grad_x [B, C, H, W].
extern "C" int ffwm_flow_head_backward(const void* y, const void* grad_y, const void* weight, void* grad_z, void* grad_x, int64_t B,
                                       int64_t C, int64_t H, int64_t W, int dtype, void* stream) {
    const char* fn = "ffwm_flow_head_backward";
    FFWM_REQUIRE(dtype == FFWM_F32, FFWM_ERR_DTYPE, "%s: float32 only", fn);
    FFWM_REQUIRE(y && grad_y && weight && grad_z && grad_x, FFWM_ERR_ARG, "%s: NULL tensor pointer", fn);
    FFWM_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && H * W < (1LL << 28), FFWM_ERR_ARG, "%s: bad sizes", fn);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int64_t HW = H * W;
    LaunchScope ls("flownet_flow_head_bwd", st, 4.0 * (B * C * HW + 18.0 * C + 6.0 * B * HW));
#define FFWM_FHB(P)                                                                                                    \
    do {                                                                                                               \
        const int tiles = static_cast<int>((HW + P - 1) / P);                                                          \
        hipLaunchKernelGGL((flow_head_bwd_kernel<P>), dim3(static_cast<unsigned>(B * tiles)), dim3(kBlock), 0, st,     \
                           (const float*)y, (const float*)grad_y, (const float*)weight, (float*)grad_z, (float*)grad_x, (int)C, (int)H, \
                           (int)W, tiles);                                                                             \
    } while (0)
    // (the forward's rule: the widest pixel tile that still gives ~100 blocks)
    auto blocks = [&](int64_t P) { return B * ((HW + P - 1) / P); };
    if (HW >= 64 && blocks(64) >= 96) FFWM_FHB(64); else if (HW >= 16 && blocks(16) >= 96) FFWM_FHB(16); else FFWM_FHB(4);
#undef FFWM_FHB
    return check_launch(fn);
}

// d(input) of ffwm_flow_up_forward: grad_out [B, 2, 2H, 2W] with batch stride grad_out_batch_stride (a channel slice of a
// concatenation's gradient is read in place), grad_x [B, 2, H, W] contiguous.
extern "C" int ffwm_flow_up_backward(const void* grad_out, const void* weight, void* grad_x, int64_t B, int64_t H, int64_t W,
                                     int64_t grad_out_batch_stride, int dtype, void* stream) {
    const char* fn = "ffwm_flow_up_backward";
    FFWM_REQUIRE(dtype == FFWM_F32, FFWM_ERR_DTYPE, "%s: float32 only", fn);
    FFWM_REQUIRE(grad_out && weight && grad_x, FFWM_ERR_ARG, "%s: NULL tensor pointer", fn);
    FFWM_REQUIRE(B > 0 && H > 0 && W > 0 && H * W < (1LL << 26) && grad_out_batch_stride >= 8 * H * W, FFWM_ERR_ARG, "%s: bad sizes", fn);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int64_t total = B * H * W;
    LaunchScope ls("flownet_flow_up_bwd", st, 4.0 * (B * 10.0 * H * W));
    hipLaunchKernelGGL(flow_up_bwd_kernel, dim3(ew_grid(total)), dim3(kBlock), 0, st, (const float*)grad_out, (const float*)weight,
                       (float*)grad_x, total, (int)H, (int)W, grad_out_batch_stride);
    return check_launch(fn);
}

// Thin-channel Conv2d(C, K, 3, 1, 1) by the direct kernel above.  weight_ctk: the layer's weights re-arranged by the host to [C][3][3][K] (from
// Conv2d's [K][C][3][3]), K a multiple of 8; act = 1: LeakyReLU(negative_slope) behind the bias; output [B, K, H, W] contiguous.
// (A direct kernel for the 4 x 4 / stride-2 transposed thin layers -- a lane per 2 x 2 output quad -- was built and measured 47-86 us against
// conv_fwd.hip's 27-34: not kept.  The 3 x 3 kernel wins where C <= 18: conv0 19 vs 24 us, inter_conv0 18 vs 30; at 34 -> 32 it loses, 26 vs 21.)
extern "C" int ffwm_conv_thin_forward(const void* x, const void* weight_ctk, const void* bias, void* y, int64_t B, int64_t C, int64_t H, int64_t W,
                                      int64_t K, int act, double negative_slope, int dtype, void* stream) {
    const char* fn = "ffwm_conv_thin_forward";
    FFWM_REQUIRE(dtype == FFWM_F32, FFWM_ERR_DTYPE, "%s: float32 only", fn);
    FFWM_REQUIRE(x && weight_ctk && y, FFWM_ERR_ARG, "%s: NULL tensor pointer", fn);
    FFWM_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && K > 0 && K % 8 == 0, FFWM_ERR_ARG, "%s: bad sizes (K must be a multiple of 8)", fn);
    FFWM_REQUIRE(C * H * W < (1LL << 29) && B * H * ((W + 63) / 64) < (1LL << 30), FFWM_ERR_SIZE, "%s: plane too large", fn);
    FFWM_REQUIRE(act == 0 || act == 1, FFWM_ERR_ARG, "%s: act must be 0 or 1", fn);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int segs_x = static_cast<int>((W + kWave - 1) / kWave);
    const int nseg = static_cast<int>(B * H * segs_x);
    const int per = kBlock / kWave;
    int kt = K % 16 == 0 ? 16 : 8;          // output channels per lane
    const int v = options().conv_thin_variant;           // bench: 1 = 8 channels per lane, 2 = 16, +4 = one channel per step instead of three
    if (v & 3) kt = (v & 3) == 1 ? 8 : (K % 16 == 0 ? 16 : 8);
    const dim3 grid(static_cast<unsigned>((nseg + per - 1) / per), static_cast<unsigned>(K / kt));
    LaunchScope ls("flownet_conv_thin", st, 4.0 * (B * C * H * W + static_cast<double>(B) * K * H * W), 2.0 * B * H * W * K * C * 9);
#define FFWM_CT2(KERNEL) hipLaunchKernelGGL(KERNEL, grid, dim3(kBlock), 0, st, (const float*)x, (const float*)weight_ctk, (const float*)bias, (float*)y, (int)C, \
                       (int)H, (int)W, (int)K, segs_x, nseg, act, (float)negative_slope)
    if (v & 4) { if (kt == 8) FFWM_CT2((conv3x3_direct_kernel<8, 1>)); else FFWM_CT2((conv3x3_direct_kernel<16, 1>)); }
    else { if (kt == 8) FFWM_CT2((conv3x3_direct_kernel<8, 3>)); else FFWM_CT2((conv3x3_direct_kernel<16, 3>)); }
#undef FFWM_CT2
    return check_launch(fn);
}
