// conv_fwd.hip -- fp32 MFMA implicit-GEMM convolution forward for the layers the vendor library serves badly (gfx950).
//
// FlowNet (/root/reference/models/base_networks.py:59-165) and netG's encoder (:274-347) are full of layers that MIOpen
// runs on NHWC implicit-GEMM kernels wrapped in NCHW<->NHWC layout transposes and zero-fills: the 3x3 / stride-2
// convolutions, the 4x4 / stride-2 / pad-1 transposed convolutions of the decoder, and the 2 x 2 ... 8 x 8 tail whose cost
// is streaming 5 - 38 MB of weights per layer for a few hundred output pixels.  In FlowNet's eval forward at batch 6
// those launches and their helpers are half of the time (profiles/r02_flownet_lean_kernel_stats.csv).
//
// One kernel, NCHW in and out, exact fp32 (v_mfma_f32_32x32x2_f32 = a k-ordered fp32 fma chain on the matrix cores):
//   out[b, k, p] = act(bias[k] + sum_{kd} Wm[k, kd] * Xg[kd, (b, p)])
//   * MODE 0  Conv2d(C, K, RxS, stride, pad): kd = (c, r, s), Xg = the zero-padded input window -- gathered on the fly, no
//             im2col buffer;
//   * MODE 1  ConvTranspose2d(C, K, 4, 2, 1): the four output-parity classes (oy & 1, ox & 1) are four independent 2x2
//             convolutions over the INPUT grid (kd = (c, a, b)): a parity class uses exactly the taps ky = 1 - py + 2a,
//             kx = 1 - px + 2b at input row y' + py - a, column x' + px - b -- no multiplication by structural zeros.
//   A workgroup (4 waves = 2 x 2 MFMA tiles) owns 64 output channels x 64 output pixels (pixels linearised over batch
//   and plane) and walks kd in chunks of 32: weights [64 x 32] and gathered activations [32 x 64] are staged in LDS
//   (global loads of chunk i+1 in flight under the 16 MFMAs per wave of chunk i), operands are ds_read_b32 at
//   conflict-free pitches.  Layers with few output pixels (the tail) are cut along kd (split-K over blockIdx.z) until the
//   chip is full: each slice adds its partial tile atomically into the zero-filled output and the bias / activation
//   epilogue runs as ffwm_bias_act_forward; otherwise the epilogue (bias + LeakyReLU / tanh, optional write into a channel
//   slice of a concatenation buffer) is fused here.
#include "common.hpp"

namespace ffwm {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct ConvGeo {
    int C, H, W;             // input planes
    int K, Ho, Wo;           // output channels; pixel grid of one launch class (MODE 1: the input grid)
    int stride, pad;
    int Kd;                  // reduction length: C * R * S (MODE 1: C * 4)
    int N;                   // B * Ho * Wo
    int n_tiles, k_tiles;
    int splitk, chunks;      // chunks (of CPC input channels) per split
    long long out_bs;        // output batch stride in elements
    int oH, oW;              // output plane (MODE 1: 2H x 2W)
    int act;
    float slope;
    unsigned x_bytes, w_bytes;   // sizes of the input / weight tensors (buffer resources; both < 2^31)
};

constexpr int kBP = 64;              // Bs pitch

// A chunk of the reduction = CPC whole input channels x all R*S taps, so that the (channel-in-chunk, r, s) of every staged
// element is FIXED for the life of the kernel: all address arithmetic happens once, before the loop, and a chunk costs
// one add per load.  (The first version decoded kd -> (c, r, s) per element and chunk on the scalar unit: 270 SALU
// instructions per chunk and wave, more issue time than the 16 MFMAs they fed -- SQ counters: SALU 7.6 M vs MFMA 0.44 M.)
template <int MODE, int R, int S>
__global__ void __launch_bounds__(kBlock)
conv_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias, float* __restrict__ out,
                const ConvGeo g) {
    constexpr int RS = MODE == 0 ? R * S : 4;
    constexpr int CPC = RS == 9 ? 4 : (RS == 16 ? 2 : 8);       // channels per chunk
    constexpr int KC = CPC * RS;                                  // 36 / 32 / 32 reduction steps per chunk (even: MFMA k = 2)
    constexpr int AP = KC + 1;                                    // As pitch (odd: conflict-free operand reads)
    constexpr int NE = KC * 64 / kBlock;                          // staged elements per thread and operand: 9 / 8 / 8
    __shared__ float As[64 * AP];
    __shared__ float Bs[KC * kBP];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int n0 = blockIdx.x * 64, k0 = blockIdx.y * 64;
    int zz = blockIdx.z;
    const int split = zz % g.splitk;
    zz /= g.splitk;
    const int py = MODE == 1 ? (zz >> 1) : 0, px = MODE == 1 ? (zz & 1) : 0;
    constexpr unsigned kOobOff = 0xFFFFFFF0u;

    // branch-free loads: buffer resources over the whole tensors + 32-bit byte offsets; an invalid element gets an offset
    // beyond the resource and the hardware returns 0
    const rsrc_t rx = make_rsrc(x, g.x_bytes);
    const rsrc_t rw = make_rsrc(w, g.w_bytes);

    // ---- B: the pixel this thread gathers for is fixed (nn = tid % 64), and so is each element's tap: kk = wave + 4 i
    const int plane_o = g.Ho * g.Wo;
    const int HW = g.H * g.W;
    const int gn = n0 + (threadIdx.x & 63);
    const bool gvalid = gn < g.N;
    const int gb = gvalid ? gn / plane_o : 0;
    const int gp = gvalid ? gn - gb * plane_o : 0;
    const int goy = gp / g.Wo, gox = gp - goy * g.Wo;
    const int iy0 = MODE == 0 ? goy * g.stride - g.pad : goy + py;     // MODE 1: iy = y' + py - a
    const int ix0 = MODE == 0 ? gox * g.stride - g.pad : gox + px;
    unsigned boff[NE];                   // byte offset of the element for channel chunk 0, or kOobOff
    int bcc[NE];                         // its channel within the chunk (wave-uniform)
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        const int kk = wave + 4 * i;
        const int cc = kk / RS, rs = kk - cc * RS;
        int iy, ix;
        if constexpr (MODE == 0) {
            iy = iy0 + rs / S; ix = ix0 + rs % S;
        } else {
            iy = iy0 - (rs >> 1); ix = ix0 - (rs & 1);
        }
        const bool ok = gvalid && iy >= 0 && iy < g.H && ix >= 0 && ix < g.W;
        bcc[i] = cc;
        boff[i] = ok ? ((static_cast<unsigned>(gb) * g.C + cc) * static_cast<unsigned>(HW) + static_cast<unsigned>(iy * g.W + ix)) * 4u : kOobOff;
    }
    // ---- A: element e = tid + 256 i of the [64 x KC] weight chunk: m = e / KC, kk = e % KC -- fixed as well
    unsigned aoff[NE];                   // byte offset for chunk 0, or kOobOff
    int acc_[NE], alds[NE];              // channel within the chunk; LDS slot
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        const int e = threadIdx.x + kBlock * i;
        const int m = e / KC, kk = e - m * KC;
        const int cc = kk / RS, rs = kk - cc * RS;
        acc_[i] = cc;
        alds[i] = m * AP + kk;
        if (k0 + m >= g.K) aoff[i] = kOobOff;
        else if (MODE == 0) aoff[i] = (static_cast<unsigned>(k0 + m) * static_cast<unsigned>(g.Kd) + kk) * 4u;
        else                       // ConvTranspose2d weight [C][K][4][4]: tap (a, b) = (rs >> 1, rs & 1) -> ky = 1 - py + 2a, kx = 1 - px + 2b
            aoff[i] = ((static_cast<unsigned>(cc) * g.K + (k0 + m)) * 16u + (1 - py + 2 * (rs >> 1)) * 4 + (1 - px + 2 * (rs & 1))) * 4u;
    }
    const unsigned a_step = (MODE == 0 ? static_cast<unsigned>(KC) : static_cast<unsigned>(CPC) * g.K * 16u) * 4u;   // bytes per chunk
    const unsigned b_step = static_cast<unsigned>(CPC) * static_cast<unsigned>(HW) * 4u;

    const int chunk_begin = split * g.chunks;
    const int chunk_end = min((g.C + CPC - 1) / CPC, chunk_begin + g.chunks);

    auto fetch = [&](int ch, float (&ra)[NE], float (&rb)[NE]) {
        const int c0 = ch * CPC;
        const unsigned ao = static_cast<unsigned>(ch) * a_step, bo = static_cast<unsigned>(ch) * b_step;
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            const bool cin = c0 + acc_[i] < g.C;                       // only the last chunk can run past C
            ra[i] = buf_ld<float>(rw, (cin && aoff[i] != kOobOff) ? aoff[i] + ao : kOobOff);
        }
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            const bool cin = c0 + bcc[i] < g.C;
            rb[i] = buf_ld<float>(rx, (cin && boff[i] != kOobOff) ? boff[i] + bo : kOobOff);
        }
    };
    auto commit = [&](const float (&ra)[NE], const float (&rb)[NE]) {
#pragma unroll
        for (int i = 0; i < NE; ++i) As[alds[i]] = ra[i];
#pragma unroll
        for (int i = 0; i < NE; ++i) Bs[(wave + 4 * i) * kBP + (threadIdx.x & 63)] = rb[i];
    };

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    // software pipeline, prefetch distance 2: the global loads of chunks i+1 and i+2 are in flight while the KC/2 MFMAs per
    // wave of chunk i run
    float ra0[NE], rb0[NE], ra1[NE], rb1[NE];
    auto compute = [&]() {
        const float* ap = As + (wm * 32 + l31) * AP + half;
        const float* bp = Bs + half * kBP + wn * 32 + l31;
#pragma unroll
        for (int q = 0; q < KC / 2; ++q)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[2 * q], bp[2 * q * kBP], acc, 0, 0, 0);
    };
    if (chunk_begin < chunk_end) fetch(chunk_begin, ra0, rb0);
    if (chunk_begin + 1 < chunk_end) fetch(chunk_begin + 1, ra1, rb1);
    for (int ch = chunk_begin; ch < chunk_end; ch += 2) {
        __syncthreads();                 // the previous chunk's operands are consumed
        commit(ra0, rb0);
        __syncthreads();
        if (ch + 2 < chunk_end) fetch(ch + 2, ra0, rb0);
        compute();
        if (ch + 1 >= chunk_end) break;
        __syncthreads();
        commit(ra1, rb1);
        __syncthreads();
        if (ch + 3 < chunk_end) fetch(ch + 3, ra1, rb1);
        compute();
    }

    // ---- epilogue.  C/D layout: col = lane & 31 (pixel), row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5) (channel)
    const int n = n0 + wn * 32 + l31;
    if (n >= g.N) return;
    const int b = n / plane_o;
    const int p = n - b * plane_o;
    size_t pix;
    if constexpr (MODE == 0) {
        pix = static_cast<size_t>(p);
    } else {
        const int oy = p / g.Wo, ox = p - oy * g.Wo;
        pix = static_cast<size_t>(2 * oy + py) * g.oW + (2 * ox + px);
    }
    float* ob = out + static_cast<size_t>(b) * g.out_bs + pix;
    const size_t oplane = static_cast<size_t>(g.oH) * g.oW;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int k = k0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (k >= g.K) continue;
        float v = acc[r];
        if (g.splitk > 1) {
            atomic_add(ob + static_cast<size_t>(k) * oplane, v);
        } else {
            if (bias) v += bias[k];
            if (g.act == 1) v = v > 0.f ? v : v * g.slope;
            else if (g.act == 2) v = tanhf(v);
            ob[static_cast<size_t>(k) * oplane] = v;
        }
    }
}

}  // namespace
}  // namespace ffwm

using namespace ffwm;

// Returns 1 through *needs_epilogue when the launch was cut along the reduction: the caller must have ZERO-FILLED the
// output and must apply bias / activation afterwards (ffwm_bias_act_forward).
extern "C" int ffwm_conv2d_forward(const void* input, const void* weight, const void* bias, void* output, int64_t B, int64_t C,
                                   int64_t H, int64_t W, int64_t K, int kernel, int stride, int pad, int transposed,
                                   int64_t out_batch_stride, int act, double negative_slope, int allow_split,
                                   int* needs_epilogue, int dtype, void* stream) {
    const char* fn = "ffwm_conv2d_forward";
    FFWM_REQUIRE(dtype == FFWM_F32, FFWM_ERR_DTYPE, "%s: float32 only", fn);
    FFWM_REQUIRE(input && weight && output, FFWM_ERR_ARG, "%s: NULL tensor pointer", fn);
    FFWM_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && K > 0 && act >= 0 && act <= 2, FFWM_ERR_ARG, "%s: bad sizes / activation", fn);
    ConvGeo g;
    g.C = static_cast<int>(C); g.H = static_cast<int>(H); g.W = static_cast<int>(W); g.K = static_cast<int>(K);
    g.stride = stride; g.pad = pad;
    int classes = 1;
    if (transposed) {
        FFWM_REQUIRE(kernel == 4 && stride == 2 && pad == 1, FFWM_ERR_ARG, "%s: transposed convolutions are 4x4 / stride 2 / pad 1 only", fn);
        g.Ho = g.H; g.Wo = g.W; g.oH = 2 * g.H; g.oW = 2 * g.W; g.Kd = g.C * 4;
        classes = 4;
    } else {
        FFWM_REQUIRE((kernel == 3 || kernel == 4) && (stride == 1 || stride == 2) && pad >= 0 && pad < kernel, FFWM_ERR_ARG,
                     "%s: 3x3 / 4x4 kernels with stride 1 / 2 only (got %d, %d, %d)", fn, kernel, stride, pad);
        g.Ho = (g.H + 2 * pad - kernel) / stride + 1;
        g.Wo = (g.W + 2 * pad - kernel) / stride + 1;
        FFWM_REQUIRE(g.Ho > 0 && g.Wo > 0, FFWM_ERR_ARG, "%s: empty output", fn);
        g.oH = g.Ho; g.oW = g.Wo; g.Kd = g.C * kernel * kernel;
    }
    FFWM_REQUIRE(B * g.Ho * g.Wo < (1LL << 31) && B * C * H * W < (1LL << 29) && K * static_cast<int64_t>(g.Kd) * (transposed ? 4 : 1) < (1LL << 29),
                 FFWM_ERR_SIZE, "%s: tensor too large (input and weight must stay below 2 GiB: 32-bit buffer offsets)", fn);
    g.x_bytes = static_cast<unsigned>(B * C * H * W * 4);
    g.w_bytes = static_cast<unsigned>(K * static_cast<int64_t>(g.Kd) * (transposed ? 4 : 1) * 4);
    FFWM_REQUIRE(out_batch_stride >= K * g.oH * g.oW, FFWM_ERR_ARG, "%s: output batch stride smaller than K * Ho * Wo", fn);
    g.N = static_cast<int>(B * g.Ho * g.Wo);
    g.n_tiles = (g.N + 63) / 64;
    g.k_tiles = (g.K + 63) / 64;
    const int cpc = transposed ? 8 : (kernel == 3 ? 4 : 2);          // channels per chunk (conv_fwd_kernel's CPC)
    const int chunks_total = (g.C + cpc - 1) / cpc;
    const int64_t tiles = static_cast<int64_t>(g.n_tiles) * g.k_tiles * classes;
    int splitk = 1;
    if (allow_split && tiles < 256) {
        splitk = static_cast<int>((768 + tiles - 1) / tiles);
        if (splitk > chunks_total) splitk = chunks_total;
        if (splitk < 1) splitk = 1;
    }
    g.chunks = (chunks_total + splitk - 1) / splitk;
    g.splitk = (chunks_total + g.chunks - 1) / g.chunks;
    g.out_bs = out_batch_stride;
    g.act = act; g.slope = static_cast<float>(negative_slope);
    if (needs_epilogue) *needs_epilogue = g.splitk > 1 ? 1 : 0;
    FFWM_REQUIRE(g.splitk == 1 || needs_epilogue, FFWM_ERR_ARG, "%s: a split launch needs the needs_epilogue out-parameter", fn);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const dim3 grid(static_cast<unsigned>(g.n_tiles), static_cast<unsigned>(g.k_tiles), static_cast<unsigned>(g.splitk * classes));
    const double flops = 2.0 * B * g.Ho * g.Wo * classes * static_cast<double>(g.K) * g.Kd;
    const double bytes = 4.0 * (static_cast<double>(B) * C * H * W + static_cast<double>(K) * g.Kd * classes + static_cast<double>(B) * K * g.oH * g.oW);
    LaunchScope ls(transposed ? "conv_fwd_mfma_transposed" : "conv_fwd_mfma", st, bytes, flops);
    const float* x = static_cast<const float*>(input);
    const float* wt = static_cast<const float*>(weight);
    const float* bs = static_cast<const float*>(bias);
    float* o = static_cast<float*>(output);
    if (transposed) hipLaunchKernelGGL((conv_fwd_kernel<1, 2, 2>), grid, dim3(kBlock), 0, st, x, wt, bs, o, g);
    else if (kernel == 3) hipLaunchKernelGGL((conv_fwd_kernel<0, 3, 3>), grid, dim3(kBlock), 0, st, x, wt, bs, o, g);
    else hipLaunchKernelGGL((conv_fwd_kernel<0, 4, 4>), grid, dim3(kBlock), 0, st, x, wt, bs, o, g);
    return check_launch(fn);
}
