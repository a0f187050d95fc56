// conv_wgrad.hip -- weight gradient of a 3x3 / stride 1 / pad 1 convolution on the matrix cores
// (fp32-in / fp32-accumulate MFMA), NCHW fp32, no layout transposes, no im2col buffer.
//
// Reference: the conv stacks of FFWM.forward / FlowNet.forward (models/base_networks.py:59-165,274-347)
// are nn.Conv2d layers whose weight gradient ATen hands to the vendor library.  On gfx950 that library
// serves the reference's 195-channel layers (dres2: Conv2d(195, 195, 3, 1, 1) at 128^2 and 64^2, the
// largest single cost of the train step) with a 32x32-tile implicit GEMM behind two NCHW<->NHWC
// transposes: 1.7 ms per call, 52 TFLOP/s.  This kernel is the replacement for that one gradient:
//
//     dW[k][c][r][s] = sum_{b,y,x} gO[b][k][y][x] * X[b][c][y + r - 1][x + s - 1]          (zero padding)
//
// i.e. nine GEMMs  dW_rs = gO (K x P) . X_rs^T (P x C)  over the P = B H W pixels that share the A
// operand.  A workgroup of 2 x 2 waves owns a 64 (k) x 64 (c) tile of all nine taps and a slice of
// the pixels (one image, one 64-pixel column strip, a range of rows); a wave owns 32 x 32 x 9 = nine
// v_mfma_f32_32x32x2_f32 accumulators (144 registers).  Per output row the strip's gO row (64 k x 64 px)
// and ONE new X row (64 c x 66 px, rolling 4-slot window: rows y-1, y, y+1 live, y+2 landing) are staged
// global -> registers -> LDS while the previous row's 288 MFMAs per wave run, so the matrix pipe never
// waits for memory: 9 taps x 32 pixel pairs x 64 cycles = 18.4 k cycles of MFMA per 33 KB staged.
// Lanes 0-31 take pixel i and lanes 32-63 pixel 32 + i of the strip in MFMA step i (the sum over
// pixels does not care about the order), so a lane's A values of four consecutive steps are ONE
// ds_read_b128, and its B values for the three horizontal taps are a 6-float window of the X row.
// The pixel slices are combined with coalesced global atomics (dW is staged through LDS so that a
// wave adds 64 consecutive floats): the caller zero-fills dW, like every other gradient of this ABI.
#include <algorithm>

#include "common.hpp"

namespace ffwm {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kWgStrip = 64;              // pixels per strip row
constexpr int kWgPG = kWgStrip + 4;       // gO pitch in LDS (floats): 16 B aligned rows
constexpr int kWgPX = kWgStrip + 8;       // X pitch: index 3 = x0 - 1, 4..67 = x0 .. x0 + 63, 68 = x0 + 64
constexpr int kWgTile = 64;               // k and c extent of a workgroup tile (2 x 2 waves of 32 x 32)
constexpr int kWgGoFloats = kWgTile * kWgPG;         // one gO row buffer
constexpr int kWgXFloats = kWgTile * kWgPX;          // one X row slot
constexpr int kWgLdsFloats = 2 * kWgGoFloats + 4 * kWgXFloats;

struct WgradGeo {
    int K, C, H, W;
    int ktiles, ctiles, strips, chunks, rows_per_chunk;
};

__global__ void __launch_bounds__(kBlock, 1)
conv3x3_wgrad_kernel(const float* __restrict__ X, const float* __restrict__ gO, float* __restrict__ dW, WgradGeo g,
                     unsigned x_bytes, unsigned go_bytes) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* const go_lds = lds;                              // [2][64 k][kWgPG]
    float* const x_lds = lds + 2 * kWgGoFloats;             // [4][64 c][kWgPX]
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    const int half = lane >> 5, l31 = lane & 31;
    const int wk = wave >> 1, wc = wave & 1;                // the wave's 32 x 32 sub-tile

    // block -> (k tile, c tile, image, strip, row chunk); tiles vary fastest so that the blocks sharing
    // a pixel slice (same gO / X rows) run together and meet in L2
    unsigned t = blockIdx.x;
    const int kt = t % g.ktiles; t /= g.ktiles;
    const int ct = t % g.ctiles; t /= g.ctiles;
    const int chunk = t % g.chunks; t /= g.chunks;
    const int strip = t % g.strips;
    const int b = t / g.strips;
    const int k0 = kt * kWgTile, c0 = ct * kWgTile, x0 = strip * kWgStrip;
    const int ya = chunk * g.rows_per_chunk;
    const int yb = min(ya + g.rows_per_chunk, g.H);
    const size_t plane = static_cast<size_t>(g.H) * g.W;
    const rsrc_t rx = make_rsrc(X, x_bytes), rg = make_rsrc(gO, go_bytes);

    // ---- staging maps: thread -> 4 float4 of the gO row, 4 float4 of the X row, (threads < 128) one edge cell
    // float4 index f = threadIdx.x + q * 256: channel f / 16, pixels (f % 16) * 4 .. + 3
    unsigned go_off[4], x_off[4];          // byte offsets of (channel, x0 + 4 * (f % 16)) in row 0 of image b
    int go_dst[4], x_dst[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int f = threadIdx.x + q * kBlock;
        const int ch = f >> 4, x4 = (f & 15) * 4;
        const bool kin = k0 + ch < g.K, cin = c0 + ch < g.C;
        go_off[q] = kin ? static_cast<unsigned>(((static_cast<size_t>(b) * g.K + k0 + ch) * plane + x0 + x4) * 4) : 0xFFFFFFF0u;
        x_off[q] = cin ? static_cast<unsigned>(((static_cast<size_t>(b) * g.C + c0 + ch) * plane + x0 + x4) * 4) : 0xFFFFFFF0u;
        go_dst[q] = ch * kWgPG + x4;
        x_dst[q] = ch * kWgPX + 4 + x4;
    }
    // edge cells: thread e < 128 -> channel e / 2, side e & 1 (0: x0 - 1 at index 3, 1: x0 + 64 at index 68)
    const int ech = threadIdx.x >> 1, eside = threadIdx.x & 1;
    const int ex = eside ? x0 + kWgStrip : x0 - 1;
    const bool eok = threadIdx.x < 2 * kWgTile && c0 + ech < g.C && ex >= 0 && ex < g.W;
    const unsigned e_off = eok ? static_cast<unsigned>(((static_cast<size_t>(b) * g.C + c0 + ech) * plane + ex) * 4) : 0xFFFFFFF0u;
    const int e_dst = ech * kWgPX + (eside ? 4 + kWgStrip : 3);
    const unsigned row_bytes = static_cast<unsigned>(g.W) * 4;

    f32x4 sg[4], sx[4];
    float se = 0;
    auto fetch_x = [&](int yy) {          // row yy of X (zero outside the image)
        const bool in = yy >= 0 && yy < g.H;
        const unsigned ro = static_cast<unsigned>(in ? yy : 0) * row_bytes;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            unsigned w[4];
            buf_load_dwords<4>(rx, (in && x_off[q] != 0xFFFFFFF0u) ? x_off[q] + ro : 0xFFFFFFF0u, w);
            sx[q] = f32x4{__uint_as_float(w[0]), __uint_as_float(w[1]), __uint_as_float(w[2]), __uint_as_float(w[3])};
        }
        se = buf_ld<float>(rx, (in && eok) ? e_off + ro : 0xFFFFFFF0u);
    };
    auto commit_x = [&](int yy) {
        float* slot = x_lds + ((yy + 1) & 3) * kWgXFloats;
#pragma unroll
        for (int q = 0; q < 4; ++q) *reinterpret_cast<f32x4*>(slot + x_dst[q]) = sx[q];
        if (threadIdx.x < 2 * kWgTile) slot[e_dst] = se;
    };
    auto fetch_g = [&](int yy) {
        const unsigned ro = static_cast<unsigned>(yy) * row_bytes;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            unsigned w[4];
            buf_load_dwords<4>(rg, go_off[q] != 0xFFFFFFF0u ? go_off[q] + ro : 0xFFFFFFF0u, w);
            sg[q] = f32x4{__uint_as_float(w[0]), __uint_as_float(w[1]), __uint_as_float(w[2]), __uint_as_float(w[3])};
        }
    };
    auto commit_g = [&](int yy) {
        float* buf = go_lds + (yy & 1) * kWgGoFloats;
#pragma unroll
        for (int q = 0; q < 4; ++q) *reinterpret_cast<f32x4*>(buf + go_dst[q]) = sg[q];
    };

    f32x16 acc[9];
#pragma unroll
    for (int tp = 0; tp < 9; ++tp)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[tp][r] = 0.f;

    // ---- prologue: X rows ya - 1, ya, ya + 1 and gO row ya
    for (int yy = ya - 1; yy <= ya + 1; ++yy) {
        fetch_x(yy);
        commit_x(yy);
    }
    fetch_g(ya);
    commit_g(ya);
    __syncthreads();

    const int a_base = (wk * 32 + l31) * kWgPG + half * 32;
    const int b_base = (wc * 32 + l31) * kWgPX + half * 32 + 3;      // index of x - 1 for step 0
    for (int y = ya; y < yb; ++y) {
        const bool more = y + 1 < yb;
        if (more) {                        // lands during the MFMA loop
            fetch_x(y + 2);
            fetch_g(y + 1);
        }
        const float* ap = go_lds + (y & 1) * kWgGoFloats + a_base;
        const float* bp[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) bp[r] = x_lds + ((y + r) & 3) * kWgXFloats + b_base;     // slot of row y + r - 1
#pragma unroll 2
        for (int i = 0; i < 32; i += 4) {
            const f32x4 a4 = *reinterpret_cast<const f32x4*>(ap + i);
            float bw[3][6];
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                bw[r][0] = bp[r][i];
                const f32x4 m = *reinterpret_cast<const f32x4*>(bp[r] + i + 1);
                bw[r][1] = m.x; bw[r][2] = m.y; bw[r][3] = m.z; bw[r][4] = m.w;
                bw[r][5] = bp[r][i + 5];
            }
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                const float av = st == 0 ? a4.x : (st == 1 ? a4.y : (st == 2 ? a4.z : a4.w));
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int s = 0; s < 3; ++s)
                        acc[r * 3 + s] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bw[r][st + s], acc[r * 3 + s], 0, 0, 0);
            }
        }
        if (more) {
            commit_x(y + 2);
            commit_g(y + 1);
        }
        __syncthreads();
    }

    // ---- epilogue: the 64 x 64 x 9 tile goes through LDS so that a wave adds 64 consecutive floats of dW
    // (dW[k][c][r][s]: for one k the 64 c x 9 taps of this tile are 576 contiguous floats)
    // C/D layout: col (c) = lane & 31, row (k) = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
    float* out_lds = lds;                   // [64 k][64 c * 9] floats = 147 456 B: reuses the staging memory
#pragma unroll
    for (int tp = 0; tp < 9; ++tp)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int kk = wk * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            const int cc = wc * 32 + l31;
            out_lds[kk * (kWgTile * 9) + cc * 9 + tp] = acc[tp][r];
        }
    __syncthreads();
    const int cvalid = min(kWgTile, g.C - c0) * 9;      // floats of a k row that exist in dW
    for (int kk = wave; kk < kWgTile; kk += kBlock / kWave) {
        if (k0 + kk >= g.K) break;
        float* drow = dW + (static_cast<size_t>(k0 + kk) * g.C + c0) * 9;
        const float* srow = out_lds + kk * (kWgTile * 9);
        for (int e = lane; e < cvalid; e += kWave) atomic_add(drow + e, srow[e]);
    }
}

}  // namespace
}  // namespace ffwm

using namespace ffwm;

extern "C" int ffwm_conv3x3_wgrad(const void* input, const void* grad_output, void* grad_weight, int64_t B, int64_t C,
                                  int64_t K, int64_t H, int64_t W, int dtype, void* stream) {
    const char* fn = "ffwm_conv3x3_wgrad";
    FFWM_REQUIRE(dtype == FFWM_F32, FFWM_ERR_DTYPE, "%s: float32 only (fp32 MFMA)", fn);
    FFWM_REQUIRE(input && grad_output && grad_weight, FFWM_ERR_ARG, "%s: NULL tensor pointer", fn);
    FFWM_REQUIRE(B > 0 && C > 0 && K > 0 && H > 0 && W > 0 && W % kWgStrip == 0, FFWM_ERR_ARG,
                 "%s: need positive sizes and W a multiple of %d (B=%lld C=%lld K=%lld H=%lld W=%lld)", fn, kWgStrip,
                 (long long)B, (long long)C, (long long)K, (long long)H, (long long)W);
    FFWM_REQUIRE(B * C * H * W * 4 < (1LL << 32) - 64 && B * K * H * W * 4 < (1LL << 32) - 64, FFWM_ERR_SIZE,
                 "%s: tensors must stay below 4 GiB (32-bit buffer offsets)", fn);
    WgradGeo g;
    g.K = static_cast<int>(K); g.C = static_cast<int>(C); g.H = static_cast<int>(H); g.W = static_cast<int>(W);
    g.ktiles = static_cast<int>((K + kWgTile - 1) / kWgTile);
    g.ctiles = static_cast<int>((C + kWgTile - 1) / kWgTile);
    g.strips = static_cast<int>(W / kWgStrip);
    // pixel slices: enough workgroups for every CU (one workgroup per CU: 106 KB of LDS), rows split evenly
    const int64_t base = static_cast<int64_t>(g.ktiles) * g.ctiles * B * g.strips;
    int chunks = 1;
    while (base * chunks < 256 && chunks * 2 <= H && (H / (chunks * 2)) >= 4) chunks *= 2;
    g.chunks = chunks;
    g.rows_per_chunk = static_cast<int>((H + chunks - 1) / chunks);
    FFWM_REQUIRE(base * chunks < (1LL << 31), FFWM_ERR_SIZE, "%s: grid too large", fn);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const size_t lds = std::max(static_cast<size_t>(kWgLdsFloats), static_cast<size_t>(kWgTile) * kWgTile * 9) * sizeof(float);
    allow_large_lds(reinterpret_cast<const void*>(conv3x3_wgrad_kernel));
    // "bytes": operands once + result (the roofline that matters is MFMA: 2 * 9 * B H W C K flop)
    LaunchScope ls("conv3x3_wgrad", st, 4.0 * (static_cast<double>(B) * H * W * (C + K) + 9.0 * C * K));
    hipLaunchKernelGGL(conv3x3_wgrad_kernel, dim3(static_cast<unsigned>(base * chunks)), dim3(kBlock), lds, st,
                       (const float*)input, (const float*)grad_output, (float*)grad_weight, g,
                       static_cast<unsigned>(B * C * H * W * 4), static_cast<unsigned>(B * K * H * W * 4));
    return check_launch(fn);
}
