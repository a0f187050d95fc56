// conv_bwd.hip -- fp32 MFMA weight gradient of the convolutions conv_fwd.hip serves (gfx950).
//
// The backward of FlowNet's / netG's stride-2, transposed and small-plane convolutions goes through the vendor's NHWC
// implicit-GEMM kernels, each wrapped in NCHW<->NHWC layout transposes and zero-fills: ~1000 launches per FFWM train
// step whose cost is mostly the dispatch gap behind each of them.  This kernel takes the weight
// gradient of those layers in ONE launch, NCHW in, weight layout out:
//
//   dW[k][(c, r, s)] = sum_{b, oy, ox} A[b, k, oy, ox] * X[b, c, oy * stride + r - pad, ox * stride + s - pad]
//
//   * Conv2d(C, K, RxS, stride, pad):      A = grad_output [B,K,Ho,Wo], X = input [B,C,H,W]        -> dW [K, C, R, S]
//   * ConvTranspose2d(Ci, Co, 4, 2, 1):    A = input [B,Ci,H,W],        X = grad_output [B,Co,2H,2W] -> dW [Ci, Co, 4, 4]
//     (the same sum with the roles of the two tensors swapped: out[2iy-1+ky] += x[iy] w[ky])
//
// GEMM view: M = K rows, N = C*R*S columns, reduction over the B*Ho*Wo pixels.  A workgroup (2 x 2 waves, one
// v_mfma_f32_32x32x2_f32 accumulator each) owns a 64 x 64 tile of dW and a slice of the pixels (blockIdx.z), walks its
// slice in chunks of 32 pixels -- A rows are contiguous along the pixels, the X window is gathered with the lanes along
// the pixels (coalesced) and written to LDS transposed at an odd pitch -- and adds its partial tile to the zero-filled
// dW with one atomic per element.  Exact fp32 products; the pixel slices meet in memory in arbitrary order.
#include "common.hpp"

namespace ffwm {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct WgGeo {
    int C, H, W;             // gathered tensor planes
    int K, Ho, Wo;           // row tensor: channels, plane
    int stride, pad;
    int N;                   // C * R * S
    int P;                   // Ho * Wo
    int chunks_per_img;      // ceil(P / 32)
    int chunks;              // chunks per z-slice
    int total_chunks;        // B * chunks_per_img
    unsigned a_bytes, x_bytes;
};

constexpr int kPC = 32;              // pixels per chunk
constexpr int kWAP = kPC + 1;        // As pitch
constexpr int kWBP = 65;             // Bs pitch (odd: the transposed store is conflict-free)

template <int R, int S>
__global__ void __launch_bounds__(kBlock)
conv_wgrad_generic_kernel(const float* __restrict__ a, const float* __restrict__ x, float* __restrict__ dw, const WgGeo g) {
    constexpr int RS = R * S;
    __shared__ float As[64 * kWAP];
    __shared__ float Bs[kPC * kWBP];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int n0 = blockIdx.x * 64, k0 = blockIdx.y * 64;
    constexpr unsigned kOobOff = 0xFFFFFFF0u;
    const rsrc_t ra_ = make_rsrc(a, g.a_bytes);
    const rsrc_t rx_ = make_rsrc(x, g.x_bytes);
    const int HW = g.H * g.W;

    // staged elements: e = tid + 256 i; kk = e % 32 (the pixel within the chunk: lanes run along the pixels), j = e / 32
    const int kk = threadIdx.x & 31, j0 = threadIdx.x >> 5;          // j = j0 + 8 i: the A row / the B column
    unsigned arow[8];                    // (k0 + j) * P in elements, or OOB
    int bc[8], br[8], bs[8];             // (c, r - pad, s - pad) of column n0 + j; c < 0: no such column
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int j = j0 + 8 * i;
        arow[i] = (k0 + j < g.K) ? static_cast<unsigned>(k0 + j) * static_cast<unsigned>(g.P) : kOobOff;
        const int n = n0 + j;
        if (n < g.N) {
            const int c = n / RS, rs = n - c * RS;
            bc[i] = c; br[i] = rs / S - g.pad; bs[i] = rs % S - g.pad;
        } else {
            bc[i] = -1; br[i] = 0; bs[i] = 0;
        }
    }

    const int ch_begin = blockIdx.z * g.chunks;
    const int ch_end = min(g.total_chunks, ch_begin + g.chunks);

    auto fetch = [&](int ch, float (&va)[8], float (&vb)[8]) {
        const int b = ch / g.chunks_per_img;
        const int p = (ch - b * g.chunks_per_img) * kPC + kk;            // this lane's pixel of image b
        const bool pin = p < g.P;
        const int oy = p / g.Wo, ox = p - oy * g.Wo;
        const int iy0 = oy * g.stride, ix0 = ox * g.stride;
        const unsigned abase = static_cast<unsigned>(b) * static_cast<unsigned>(g.K) * static_cast<unsigned>(g.P) + p;
        const unsigned xbase = static_cast<unsigned>(b) * static_cast<unsigned>(g.C) * static_cast<unsigned>(HW);
#pragma unroll
        for (int i = 0; i < 8; ++i) va[i] = buf_ld<float>(ra_, (pin && arow[i] != kOobOff) ? (abase + arow[i]) * 4u : kOobOff);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int iy = iy0 + br[i], ix = ix0 + bs[i];
            const bool ok = pin && bc[i] >= 0 && iy >= 0 && iy < g.H && ix >= 0 && ix < g.W;
            vb[i] = buf_ld<float>(rx_, ok ? (xbase + static_cast<unsigned>(bc[i]) * static_cast<unsigned>(HW) + static_cast<unsigned>(iy * g.W + ix)) * 4u
                                          : kOobOff);
        }
    };
    auto commit = [&](const float (&va)[8], const float (&vb)[8]) {
#pragma unroll
        for (int i = 0; i < 8; ++i) As[(j0 + 8 * i) * kWAP + kk] = va[i];
#pragma unroll
        for (int i = 0; i < 8; ++i) Bs[kk * kWBP + j0 + 8 * i] = vb[i];
    };

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    auto compute = [&]() {
        const float* ap = As + (wm * 32 + l31) * kWAP + half;
        const float* bp = Bs + half * kWBP + wn * 32 + l31;
#pragma unroll
        for (int q = 0; q < kPC / 2; ++q)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[2 * q], bp[2 * q * kWBP], acc, 0, 0, 0);
    };

    float a0[8], b0[8], a1[8], b1[8];
    if (ch_begin < ch_end) fetch(ch_begin, a0, b0);
    if (ch_begin + 1 < ch_end) fetch(ch_begin + 1, a1, b1);
    for (int ch = ch_begin; ch < ch_end; ch += 2) {
        __syncthreads();
        commit(a0, b0);
        __syncthreads();
        if (ch + 2 < ch_end) fetch(ch + 2, a0, b0);
        compute();
        if (ch + 1 >= ch_end) break;
        __syncthreads();
        commit(a1, b1);
        __syncthreads();
        if (ch + 3 < ch_end) fetch(ch + 3, a1, b1);
        compute();
    }

    // C/D layout: col = lane & 31 (n), row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5) (k)
    const int n = n0 + wn * 32 + l31;
    if (n >= g.N) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int k = k0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (k < g.K) atomic_add(dw + static_cast<size_t>(k) * g.N + n, acc[r]);
    }
}

}  // namespace
}  // namespace ffwm

using namespace ffwm;

extern "C" int ffwm_conv2d_wgrad(const void* rows, const void* gathered, void* grad_weight, int64_t B, int64_t K, int64_t Ho,
                                 int64_t Wo, int64_t C, int64_t H, int64_t W, int kernel, int stride, int pad, int dtype, void* stream) {
    const char* fn = "ffwm_conv2d_wgrad";
    FFWM_REQUIRE(dtype == FFWM_F32, FFWM_ERR_DTYPE, "%s: float32 only", fn);
    FFWM_REQUIRE(rows && gathered && grad_weight, FFWM_ERR_ARG, "%s: NULL tensor pointer", fn);
    FFWM_REQUIRE(B > 0 && K > 0 && C > 0 && Ho > 0 && Wo > 0 && H > 0 && W > 0, FFWM_ERR_ARG, "%s: sizes must be positive", fn);
    FFWM_REQUIRE((kernel == 3 || kernel == 4) && (stride == 1 || stride == 2) && pad >= 0 && pad < kernel, FFWM_ERR_ARG,
                 "%s: 3x3 / 4x4 kernels with stride 1 / 2 only", fn);
    FFWM_REQUIRE(Ho == (H + 2 * pad - kernel) / stride + 1 && Wo == (W + 2 * pad - kernel) / stride + 1, FFWM_ERR_ARG,
                 "%s: the row tensor's plane (%lld x %lld) is not the output plane of this convolution over %lld x %lld", fn,
                 (long long)Ho, (long long)Wo, (long long)H, (long long)W);
    FFWM_REQUIRE(B * K * Ho * Wo < (1LL << 29) && B * C * H * W < (1LL << 29), FFWM_ERR_SIZE, "%s: tensors must stay below 2 GiB", fn);
    WgGeo g;
    g.C = static_cast<int>(C); g.H = static_cast<int>(H); g.W = static_cast<int>(W);
    g.K = static_cast<int>(K); g.Ho = static_cast<int>(Ho); g.Wo = static_cast<int>(Wo);
    g.stride = stride; g.pad = pad;
    g.N = g.C * kernel * kernel;
    g.P = g.Ho * g.Wo;
    g.chunks_per_img = (g.P + kPC - 1) / kPC;
    g.total_chunks = static_cast<int>(B) * g.chunks_per_img;
    g.a_bytes = static_cast<unsigned>(B * K * Ho * Wo * 4);
    g.x_bytes = static_cast<unsigned>(B * C * H * W * 4);
    const int n_tiles = (g.N + 63) / 64, k_tiles = (g.K + 63) / 64;
    // pixel slices: enough workgroups to fill the chip, at least 4 chunks each
    int64_t slices = (1024 + static_cast<int64_t>(n_tiles) * k_tiles - 1) / (static_cast<int64_t>(n_tiles) * k_tiles);
    if (slices > g.total_chunks / 4) slices = g.total_chunks / 4;
    if (slices < 1) slices = 1;
    g.chunks = static_cast<int>((g.total_chunks + slices - 1) / slices);
    const unsigned nz = static_cast<unsigned>((g.total_chunks + g.chunks - 1) / g.chunks);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const double flops = 2.0 * B * g.P * static_cast<double>(g.K) * g.N;
    const double bytes = 4.0 * (static_cast<double>(B) * K * g.P + static_cast<double>(B) * C * H * W + static_cast<double>(K) * g.N);
    LaunchScope ls("conv_wgrad_mfma_generic", st, bytes, flops);
    const dim3 grid(static_cast<unsigned>(n_tiles), static_cast<unsigned>(k_tiles), nz);
    const float* a = static_cast<const float*>(rows);
    const float* x = static_cast<const float*>(gathered);
    float* dw = static_cast<float*>(grad_weight);
    if (kernel == 3) hipLaunchKernelGGL((conv_wgrad_generic_kernel<3, 3>), grid, dim3(kBlock), 0, st, a, x, dw, g);
    else hipLaunchKernelGGL((conv_wgrad_generic_kernel<4, 4>), grid, dim3(kBlock), 0, st, a, x, dw, g);
    return check_launch(fn);
}
