// conv_winograd.hip -- fp32 Winograd F(2x2, 3x3) convolution on the MFMA units (gfx950): forward and data gradient of
// the 3x3 / stride 1 / pad 1 layers of netG (/root/reference/models/base_networks.py:207-233, 293-298: dres0-2 at
// 195 channels x 128^2 and 64^2) and of FlowNet / LightCNN (:59-112).
//
// Those layers are ~40 % of the GPU time of an FFWM train step, spent in the vendor's VALU Winograd
// (miopenSp3AsmConv...f2x3: ~95 TFLOP/s direct-equivalent).  The same transform cuts the multiplications 2.25 x, and
// here the 16 per-position GEMMs run on v_mfma_f32_32x32x2_f32:
//
//   y[b, k, 2ty + i', 2tx + j'] = sum_ij At[i'][i] ( sum_c U[k, c, i, j] V[c, t, i, j] ) At[j'][j],  t = (b, ty, tx)
//   U[k, c] = G w[k, c] Gt   (4 x 4, prepared once per call by winograd_weights_kernel)
//   V[c, t] = Bt d[c, t] B   (4 x 4, d = the 4 x 4 input patch of tile t; formed in registers on the way to LDS)
//
// A workgroup (4 waves) owns 64 output channels x 64 tiles (= 256 output pixels per image channel) for ALL 16 positions:
// every wave holds 16 accumulators of 32 x 32 (256 AGPRs), so the output transform is lane-local and the result is
// stored once, NCHW, with bias + LeakyReLU in the same epilogue.  The reduction runs in chunks of 8 input channels:
//   * U is stored by the weight kernel exactly as LDS wants it ([k tile][chunk][position][channel half][64 k][4 c]), so
//     a chunk is one contiguous 32 KiB block: 8 dwordx4 loads + 8 ds_write_b128 per thread;
//   * each thread gathers the patch of one tile for two channels (32 dword loads through a buffer resource: the zero
//     padding and the channel tail are the hardware's out-of-range 0), transforms it (32 adds) and scatters the 16
//     positions to LDS -- lanes run along (4 channels, 16 tiles), consecutive dwords, no bank conflict;
//   * an MFMA takes A[k][c] and B[c][t] for TWO channels (k = 2): lanes 0-31 feed channel j, lanes 32-63 channel 4 + j of
//     the chunk, so ONE ds_read_b128 per operand feeds four MFMAs (0.5 LDS reads per MFMA; the direct kernel of
//     conv_fwd.hip needs 2).
// Loads of chunk i + 1 are in flight while the 64 MFMAs per wave of chunk i issue (4096 cycles between two barriers).
//
// Mode 1 is the data gradient of the same layer: dx = conv(dy, w') with w'[c][k][r][s] = w[k][c][2 - r][2 - s]; only the
// weight kernel differs.  fp32 throughout; the rounding differs from a direct sum in the last bits, like the vendor's
// Winograd that it replaces (tests: <= 2e-5 of the output scale against fp64).
#include "common.hpp"
#include <type_traits>

namespace ffwm {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kThinStride = 36;           // floats per input channel of the thin tail's weights: up to 4 output channels x 9 taps
constexpr int kWinoChunk = 16 * 2 * 64 * 4;      // floats of U (and of V) per chunk of 8 channels: 8192 = 32 KiB

struct WinoGeo {
    int C, H, W, K;          // K = the output channels THIS launch computes (a multiple of 64 when a thin tail follows)
    int Kout;                // channels of the output tensor
    int TH, TW, T;           // tiles per column / row / in total (B * TH * TW)
    int CH;                  // chunks of 8 input channels
    int KT, TT;              // tiles of 64 output channels / of 64 output tiles
    int act;
    float slope;
    unsigned x_bytes, u_bytes;
    int R, ROWS;             // raw-staged variant: tile rows per workgroup (64 / TW) and input rows it stages (2 R + 2)

};

// U[kt][ch][p][h][k64][c4] = (G w Gt)[p] of output channel kt * 64 + k64 and input channel ch * 8 + 4 h + c4; zero outside
// K x C.  mode 0: w is [K][C][3][3]; mode 1 (data gradient): w is the layer's own [C][K][3][3], used transposed and flipped.
__device__ __forceinline__ void winograd_weights_elem(const float* __restrict__ w, float* __restrict__ U, int K, int Kw, int C, int CH, int KT,
                                                      int mode, int tail, int e) {
    const int Cp = CH * 8;
    if (e >= KT * 64 * Cp) {
        // the thin tail's weights (output channels K .. K + tail): Wt[c][36] = (k, tap) -> w, rotated for the data gradient,
        // in the slot of the k tile the Winograd kernel no longer computes
        const int t = e - KT * 64 * Cp;
        if (t >= C * kThinStride) return;
        const int c = t / kThinStride, q = t - c * kThinStride, k = q / 9, tap = q - k * 9;
        float v = 0.f;
        if (k < tail) v = mode == 0 ? w[(static_cast<size_t>(K + k) * C + c) * 9 + tap] : w[(static_cast<size_t>(c) * Kw + K + k) * 9 + (8 - tap)];
        U[static_cast<size_t>(KT) * CH * kWinoChunk + t] = v;
        return;
    }
    const int k = e / Cp, c = e - k * Cp;
    float g[3][3];
    const bool in = k < K && c < C;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            float v = 0.f;
            if (in) v = mode == 0 ? w[(static_cast<size_t>(k) * C + c) * 9 + r * 3 + s]
                                  : w[(static_cast<size_t>(c) * Kw + k) * 9 + (2 - r) * 3 + (2 - s)];
            g[r][s] = v;
        }
    // rows: G g  (G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1])
    float t[4][3];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        t[0][s] = g[0][s];
        t[1][s] = 0.5f * (g[0][s] + g[1][s] + g[2][s]);
        t[2][s] = 0.5f * (g[0][s] - g[1][s] + g[2][s]);
        t[3][s] = g[2][s];
    }
    float* dst = U + (static_cast<size_t>(k >> 6) * CH + (c >> 3)) * kWinoChunk + (((c >> 2) & 1) * 64 + (k & 63)) * 4 + (c & 3);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float u0 = t[i][0];
        const float u1 = 0.5f * (t[i][0] + t[i][1] + t[i][2]);
        const float u2 = 0.5f * (t[i][0] - t[i][1] + t[i][2]);
        const float u3 = t[i][2];
        dst[(i * 4 + 0) * 512] = u0;
        dst[(i * 4 + 1) * 512] = u1;
        dst[(i * 4 + 2) * 512] = u2;
        dst[(i * 4 + 3) * 512] = u3;
    }
}

__global__ void __launch_bounds__(kBlock)
winograd_weights_kernel(const float* __restrict__ w, float* __restrict__ U, int K, int Kw, int C, int CH, int KT, int mode, int tail) {
    winograd_weights_elem(w, U, K, Kw, C, CH, KT, mode, tail, blockIdx.x * kBlock + threadIdx.x);
}

// The transforms of SEVERAL layers in one launch (ffwm_conv3x3_winograd_weights_multi): the weights of a spectrally normalised
// network are all known once its batched normalisation has run, so the ~45 per-call transform launches of netG's forward and data
// gradient (6 us each, every one in front of the convolution that needs it) become two or three launches at the start of the pass.
constexpr int kWinoMaxMulti = 24;
struct WinoWeightsItem {
    const float* w;
    float* U;
    int K, Kw, C, CH, KT, mode, tail;
    int begin;               // first workgroup of this item
};
struct WinoWeightsTable {
    int n;
    WinoWeightsItem it[kWinoMaxMulti];
};

__global__ void __launch_bounds__(kBlock)
winograd_weights_multi_kernel(const WinoWeightsTable tab) {
    int i = 0;
#pragma unroll
    for (int k = 1; k < kWinoMaxMulti; ++k)
        if (k < tab.n && static_cast<int>(blockIdx.x) >= tab.it[k].begin) i = k;
    const WinoWeightsItem& q = tab.it[i];
    winograd_weights_elem(q.w, q.U, q.K, q.Kw, q.C, q.CH, q.KT, q.mode, q.tail, (static_cast<int>(blockIdx.x) - q.begin) * kBlock + threadIdx.x);
}

constexpr int kWinoThreads = 512;        // 8 waves: two per SIMD, 128 accumulator + <= 128 other registers each
constexpr int kWinoTiles = 64;           // output tiles (2 x 2 pixels) per workgroup

// ---- epilogue: y = At m A per (k, tile).  C/D layout: col = lane & 31 (tile), row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5) (k).
// A wave holds rows i = 2 ph, 2 ph + 1 of m: it forms its share of y (2 x 2 values per register), the ph = 1 waves hand theirs
// to their ph = 0 partners through LDS (the staging buffers are free: the loop ended with a barrier).
__device__ __forceinline__ void winograd_epilogue(const f32x16 (&acc)[8], f32x4* smem, const float* __restrict__ bias, float* __restrict__ out,
                                                  const WinoGeo& g, int tt, int kt) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int ph = wave >> 2, wm = (wave >> 1) & 1, wn = wave & 1;
    const int HW = g.H * g.W;
    float part[16][4];               // [r][y00, y01, y10, y11]
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float za0 = acc[0][r] + acc[1][r] + acc[2][r], za1 = acc[1][r] - acc[2][r] - acc[3][r];      // row 2 ph
        const float zb0 = acc[4][r] + acc[5][r] + acc[6][r], zb1 = acc[5][r] - acc[6][r] - acc[7][r];      // row 2 ph + 1
        if (ph == 0) {               // y0 = z0 + z1 (+ z2), y1 = z1 (- z2 - z3)
            part[r][0] = za0 + zb0; part[r][1] = za1 + zb1; part[r][2] = zb0; part[r][3] = zb1;
        } else {                     // y0 = z2, y1 = -z2 - z3
            part[r][0] = za0; part[r][1] = za1; part[r][2] = -za0 - zb0; part[r][3] = -za1 - zb1;
        }
    }
    float* xch = reinterpret_cast<float*>(smem) + (wave & 3) * 4096 + lane;
    if (ph == 1) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int v = 0; v < 4; ++v) xch[(r * 4 + v) * 64] = part[r][v];
    }
    __syncthreads();
    if (ph == 1 || kt * 64 + wm * 32 >= g.K) return;
    const int tg = tt * kWinoTiles + wn * 32 + l31;
    if (tg >= g.T) return;
    const int b = tg / (g.TH * g.TW);
    const int rem = tg - b * (g.TH * g.TW);
    const int ty = rem / g.TW, tx = rem - ty * g.TW;
    const int oy = 2 * ty, ox = 2 * tx;
    const bool row1 = oy + 1 < g.H, col1 = ox + 1 < g.W;
    float* ob = out + (static_cast<size_t>(b) * g.Kout) * HW + static_cast<size_t>(oy) * g.W + ox;
    const bool vec = col1 && (g.W & 1) == 0;         // 8-byte aligned pair
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int k = kt * 64 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (k >= g.K) continue;
        const float bv = bias ? bias[k] : 0.f;
        float y[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            float t = part[r][v] + xch[(r * 4 + v) * 64] + bv;
            if (g.act == 1) t = t > 0.f ? t : t * g.slope;
            y[v] = t;
        }
        float* o = ob + static_cast<size_t>(k) * HW;
        if (vec) {
            *reinterpret_cast<float2*>(o) = make_float2(y[0], y[1]);
            if (row1) *reinterpret_cast<float2*>(o + g.W) = make_float2(y[2], y[3]);
        } else {
            o[0] = y[0];
            if (col1) o[1] = y[1];
            if (row1) {
                o[g.W] = y[2];
                if (col1) o[g.W + 1] = y[3];
            }
        }
    }
}

template <int ABL>          // ABL: timing experiments only (bit 0 no patch loads, 1 no U loads, 2 no LDS commits, 3 no operand reads; raw variant: 4 no output stores, 5 no epilogue)
__global__ void __launch_bounds__(kWinoThreads)
winograd_conv_kernel(const float* __restrict__ x, const float* __restrict__ U, const float* __restrict__ bias, float* __restrict__ out,
                     const WinoGeo g, int remap) {
    extern __shared__ f32x4 smem[];          // [2 buffers][U | V][16 positions][2 channel halves][64 k or tiles] x 4 channels: 128 KiB
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    // compute role: positions 8 ph .. 8 ph + 7 (rows 2 ph, 2 ph + 1 of the 4 x 4 transform domain) of the (wm, wn) quarter of the tile
    const int ph = wave >> 2, wm = (wave >> 1) & 1, wn = wave & 1;
    // consecutive logical ids = the k tiles of one strip of tiles, kept on one XCD (they gather the same input)
    const unsigned lid = xcd_remap(blockIdx.x, gridDim.x, remap);
    const int tt = static_cast<int>(lid) / g.KT, kt = static_cast<int>(lid) - tt * g.KT;
    constexpr unsigned kOobOff = 0xFFFFFFF0u;
    const rsrc_t rx = make_rsrc(x, g.x_bytes);
    const rsrc_t ru = make_rsrc(U, g.u_bytes);
    const int HW = g.H * g.W;

    // ---- staging role: lane = (c_lo, t_lo), wave = (channel half, t_hi): the patch of tile ts for channel 4 sh + c_lo of the chunk
    const int c_lo = lane & 3, ts = (wave & 3) * 16 + (lane >> 2), sh = wave >> 2;
    unsigned poff[16];               // byte offsets of the 4 x 4 patch in channel 0 of its image, or kOobOff
    {
        const int tg = tt * kWinoTiles + ts;
        const bool tv = tg < g.T;
        const int b = tv ? tg / (g.TH * g.TW) : 0;
        const int rem = tv ? tg - b * (g.TH * g.TW) : 0;
        const int ty = rem / g.TW, tx = rem - ty * g.TW;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int iy = 2 * ty - 1 + i, ix = 2 * tx - 1 + j;
                const bool ok = tv && iy >= 0 && iy < g.H && ix >= 0 && ix < g.W;
                poff[i * 4 + j] = ok ? (static_cast<unsigned>(b) * g.C * static_cast<unsigned>(HW) + static_cast<unsigned>(iy * g.W + ix)) * 4u : kOobOff;
            }
    }
    const unsigned u_base = (static_cast<unsigned>(kt) * g.CH) * (kWinoChunk * 4u) + threadIdx.x * 16u;

    // The side work of a chunk is cut into 2 fetch slices and 2 commit slices that ride between the MFMA groups
    float d[16] = {};
    u32x4 uw[4] = {};
    auto fetch_slice = [&](int ch, int s) {          // U rows 2s, 2s + 1; patch elements 8 s ..+8
        if (!(ABL & 2))
#pragma unroll
            for (int i = 2 * s; i < 2 * s + 2; ++i)
                uw[i] = __builtin_amdgcn_raw_buffer_load_b128(ru, u_base + static_cast<unsigned>(ch) * (kWinoChunk * 4u) + i * 8192u, 0, 0);
        const int c = ch * 8 + 4 * sh + c_lo;
        const unsigned co = static_cast<unsigned>(c) * static_cast<unsigned>(HW) * 4u;
        const bool cin = c < g.C;                    // the channel tail and every chunk past the last read 0
        if (!(ABL & 1))
#pragma unroll
            for (int q = 8 * s; q < 8 * s + 8; ++q) {
                const unsigned o = poff[q] + co;
                d[q] = buf_ld<float>(rx, (cin & (poff[q] != kOobOff)) ? o : kOobOff);
            }
    };
    auto commit_slice = [&](f32x4* Ub, f32x4* Vb, int s) {   // U rows 2s, 2s + 1; V rows 2s, 2s + 1
        if (ABL & 4) return;
#pragma unroll
        for (int i = 2 * s; i < 2 * s + 2; ++i) reinterpret_cast<u32x4*>(Ub)[threadIdx.x + kWinoThreads * i] = uw[i];
        float* dst = reinterpret_cast<float*>(Vb) + (sh * 64 + ts) * 4 + c_lo;
#pragma unroll
        for (int i = 2 * s; i < 2 * s + 2; ++i) {
            // row i of Bt d: (d0 - d2, d1 + d2, d2 - d1, d1 - d3), then the same along the columns
            float r[4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                r[j] = i == 0 ? d[0 + j] - d[8 + j] : i == 1 ? d[4 + j] + d[8 + j] : i == 2 ? d[8 + j] - d[4 + j] : d[4 + j] - d[12 + j];
            dst[(i * 4 + 0) * 512] = r[0] - r[2];
            dst[(i * 4 + 1) * 512] = r[1] + r[2];
            dst[(i * 4 + 2) * 512] = r[2] - r[1];
            dst[(i * 4 + 3) * 512] = r[1] - r[3];
        }
    };

    f32x16 acc[8];
#pragma unroll
    for (int p = 0; p < 8; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;

    // one chunk: 4 groups of 8 MFMAs (two positions); groups 0-1 carry the commit of chunk ch + 1 into the other buffer,
    // groups 2-3 the loads of chunk ch + 2; the operands of group i + 1 are read while group i issues
    auto step = [&](int ch, f32x4* Ub, f32x4* Vb, f32x4* Un, f32x4* Vn) {
        const f32x4* ap = Ub + (ph * 8) * 128 + half * 64 + wm * 32 + l31;
        const f32x4* bp = Vb + (ph * 8) * 128 + half * 64 + wn * 32 + l31;
        f32x4 oa[2][2], ob[2][2];
        oa[0][0] = ap[0]; ob[0][0] = bp[0]; oa[0][1] = ap[128]; ob[0][1] = bp[128];
#pragma unroll
        for (int grp = 0; grp < 4; ++grp) {
            const int cur = grp & 1, nxt = cur ^ 1;
            if (grp < 3 && !(ABL & 8)) {
                oa[nxt][0] = ap[(2 * grp + 2) * 128]; ob[nxt][0] = bp[(2 * grp + 2) * 128];
                oa[nxt][1] = ap[(2 * grp + 3) * 128]; ob[nxt][1] = bp[(2 * grp + 3) * 128];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[2 * grp] = __builtin_amdgcn_mfma_f32_32x32x2f32(oa[cur][0][j], ob[cur][0][j], acc[2 * grp], 0, 0, 0);
                acc[2 * grp + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(oa[cur][1][j], ob[cur][1][j], acc[2 * grp + 1], 0, 0, 0);
            }
            if (grp < 2) commit_slice(Un, Vn, grp);
            else fetch_slice(ch + 2, grp - 2);
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    };

    // buffer 0 = smem[0 .. 4096), buffer 1 = smem[4096 .. 8192) (f32x4 units); U first, V 2048 behind it
    fetch_slice(0, 0); fetch_slice(0, 1);
    commit_slice(smem, smem + 2048, 0); commit_slice(smem, smem + 2048, 1);
    fetch_slice(1, 0); fetch_slice(1, 1);
    __syncthreads();
    unsigned cur = 0;
    for (int ch = 0; ch < g.CH; ++ch) {
        f32x4* const ub = smem + cur;
        f32x4* const un = smem + (cur ^ 4096u);
        step(ch, ub, ub + 2048, un, un + 2048);
        cur ^= 4096u;
    }

    winograd_epilogue(acc, smem, bias, out, g, tt, kt);
}

// ---------------------------------------------------------------------------------------------------- raw-staged variant
// The gathers above cost the texture addresser 16 scattered dword loads per thread and chunk (TA busy 45 %, each input element
// fetched ~4 x through L1) next to the MFMAs' 60 %.  When the 64 tiles of a workgroup are whole tile rows of one image (W in
// {16, 32, 64, 128}: R = 64 / TW tile rows, 2 R + 2 input rows of W pixels, no horizontal halo -- it is all zero padding) the
// input window of a chunk is copied ONCE, row-contiguous dwordx4 -> LDS (2 loads per thread instead of 16), and the 4 x 4
// patches are read back from LDS (3 reads per row: the patch starts at an odd column; the border columns are masked).
//
// Pipeline of chunk n, one barrier per step: L(n) global -> registers in step n - 3, W(n) registers -> raw[n & 1] in step
// n - 2, T(n) raw -> Bt d B -> V[n & 1] and U(n) -> LDS in step n - 1, MFMAs in step n.  LDS: 128 KiB of operands + 2 x 16 KiB
// raw = the CU's 160 KiB exactly.
constexpr int kWinoRawFloats = 8 * 4 * 128;      // 8 channels x (2 R + 2) rows x W <= 4096 floats for every supported W

// The kernel is PERSISTENT: one workgroup per CU walks a contiguous range of (tile strip, k tile) pairs and the chunk
// pipeline runs straight through the boundaries -- the loads of the next pair's first chunks are in flight under the last
// MFMAs of this one.  Launched one workgroup per pair, every pair paid ~15 us of dead time (dispatch, two dependent load round
// trips before the first MFMA, the epilogue, uneven dynamic distribution) next to 2.3 us per chunk: 45 % of a 64-channel
// layer, 17 % of a 256-channel one.  Here the epilogue of a pair costs two barriers: the ph = 1 waves park their share of the
// output transform in the operand buffers the last step just finished with.
// SPLIT: the variant for calls with few pairs (WinoGeo::CS > 1).  A template parameter on purpose: the kernel lives on exactly 256
// registers, and the split's extra state (a chunk origin per load cursor) as run-time values pushed three more registers to
// scratch in the hot loop of EVERY call (measured: 146 -> 165 us per launch over the train step's layers).
template <int ABL, bool SPLIT = false>
__global__ void __launch_bounds__(kWinoThreads)
winograd_conv_raw_kernel(const float* __restrict__ x, const float* __restrict__ U, const float* __restrict__ bias, float* __restrict__ out,
                         const WinoGeo g, int remap, int split_n, int split_chunks) {
    // split_n / split_chunks (SPLIT only): the reduction cut into split_n pieces of split_chunks chunks each; a workgroup's unit is then
    // (strip, k tile, piece) and the pieces' partial outputs meet by atomics in the zero-filled output
    extern __shared__ f32x4 smem[];          // operands as above (8192 f32x4), then raw[2][kWinoRawFloats]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int ph = wave >> 2, wm = (wave >> 1) & 1, wn = wave & 1;
    constexpr unsigned kOobOff = 0xFFFFFFF0u;
    const rsrc_t rx = make_rsrc(x, g.x_bytes);
    const rsrc_t ru = make_rsrc(U, g.u_bytes);
    const int HW = g.H * g.W;
    float* const raw0 = reinterpret_cast<float*>(smem + 8192);
    float* const raw1 = raw0 + kWinoRawFloats;

    // this workgroup's pairs: [pair_begin, pair_begin + pair_cnt) of the tt-major list (the k tiles of a strip are consecutive:
    // the same CU re-reads the strip's input from L2); each XCD gets a contiguous range of the list
    const int CS = SPLIT ? split_n : 1;
    const int npairs = g.TT * g.KT * CS;
    const int L = static_cast<int>(xcd_remap(blockIdx.x, gridDim.x, remap));
    const int per = npairs / static_cast<int>(gridDim.x), extra = npairs - per * static_cast<int>(gridDim.x);
    const int pair_begin = L * per + min(L, extra), pair_cnt = per + (L < extra ? 1 : 0);
    const int pair_end = pair_begin + pair_cnt;
    if (pair_cnt == 0) return;
    const int CHn = SPLIT ? split_chunks : g.CH;                     // chunks per pair
    // ... rounded up to even: an odd count ends with a chunk of zeros -- input channels >= C read 0 through the range check, and the
    // U loads of that chunk are sent out of range too: they would fetch the NEXT k tile's first chunk or, behind the last k tile,
    // whatever lies in the workspace past the transformed weights -- and 0 x NaN from recycled memory is NaN (found in round 4 with a
    // NaN-poisoned allocator: every 195-channel layer, 25 chunks)
    const int CHp = (CHn + 1) & ~1;

    // ---- W stage role: quads q = tid, tid + 512 of the [8 channels][ROWS][W / 4] window
    const int W4 = g.W >> 2, nquads = 8 * g.ROWS * W4;
    const int strips_per_img = g.TH / g.R;
    int qc[2], qlds[2];      // channel within the chunk; float index in the raw buffer
    const int w4_shift = __builtin_ctz(static_cast<unsigned>(W4));
    const float quad_rcp = 1.f / static_cast<float>(g.ROWS * W4);      // quads per channel <= 128, quad index < 1024: (q + 0.5) * rcp truncates exactly
    auto quad_of = [&](int q, int& c, int& rr, int& col) {             // channel, window row, pixel column of quad q
        c = static_cast<int>((static_cast<float>(q) + 0.5f) * quad_rcp);
        const int rem = q - c * (g.ROWS * W4);
        rr = rem >> w4_shift;
        col = 4 * (rem & (W4 - 1));
    };
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int q = threadIdx.x + kWinoThreads * i;
        int c, rr, col;
        quad_of(q, c, rr, col);
        qc[i] = q < nquads ? c : 8;
        // the rows of channel c are stored rotated by 16 c pixels: the 4 channels a wave reads patches from sit 0 (mod 64) dwords
        // apart, unrotated they would share their banks (35 % of the LDS cycles were conflicts)
        qlds[i] = q < nquads ? (c * g.ROWS + rr) * g.W + ((col + 16 * c) & (g.W - 1)) : -1;
    }
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    // Per-lane values that are needed once per pair only (seat(), the output transform) are worked out THERE from the lane index, which
    // `lane_now` reads off the hardware in two instructions the optimiser cannot hoist: kept in registers across the chunk loop they
    // are spilled, and a scratch reload is a vector-memory load -- its wait also waits for every prefetched load in flight.
    auto lane_now = [&]() {
        int l;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
        return l;
    };
    // a load cursor: the (pair, chunk) a stream of loads has reached, and what depends on the pair
    struct Cursor {
        int pair, ch;
        unsigned qoff[2];            // byte offset of the quad in the first channel of the pair's first chunk, or kOobOff
        unsigned u_base;             // byte offset of this thread's first U quad of that chunk, or kOobOff past the last pair
    };
    const int cs_shift = CS == 4 ? 2 : (CS == 2 ? 1 : 0);          // CS is 1, 2 or 4 (winograd_splits)
    auto first_chunk = [&](int pair) { return SPLIT ? (pair & (CS - 1)) * split_chunks : 0; };     // of the pair's piece of the reduction
    auto seat = [&](Cursor& c) {
        if (c.pair >= pair_end) {
            c.qoff[0] = c.qoff[1] = kOobOff;
            c.u_base = kOobOff;
            return;
        }
        const int pk = SPLIT ? c.pair >> cs_shift : c.pair;
        const int ch0 = first_chunk(c.pair);
        const int tt = pk / g.KT, kt = pk - tt * g.KT;
        const int b = tt / strips_per_img, ty0 = (tt - b * strips_per_img) * g.R;
        const int tid = wave_u * 64 + lane_now();
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int q = tid + kWinoThreads * i;
            int qch, qrow, qcol;
            quad_of(q, qch, qrow, qcol);
            const int iy = 2 * ty0 - 1 + qrow;
            const bool ok = q < nquads && iy >= 0 && iy < g.H;
            c.qoff[i] = ok ? ((static_cast<unsigned>(b) * g.C + 8 * ch0 + qch) * static_cast<unsigned>(HW) + static_cast<unsigned>(iy * g.W + qcol)) * 4u : kOobOff;
        }
        c.u_base = (static_cast<unsigned>(kt) * g.CH + ch0) * (kWinoChunk * 4u) + threadIdx.x * 16u;
    };
    auto advance = [&](Cursor& c) {
        if (++c.ch == CHp) {
            c.ch = 0;
            ++c.pair;
            seat(c);
        }
    };
    Cursor cr, cu;
    cr.pair = cu.pair = pair_begin;
    cr.ch = cu.ch = 0;
    seat(cr);
    cu = cr;

    // ---- T stage role: lane = (c_lo, t_lo), wave = (channel half, t_hi): the patch of tile ts for channel 4 sh + c_lo of the chunk
    const int c_lo = lane & 3, ts = (wave & 3) * 16 + (lane >> 2), sh = wave >> 2;
    const int tr = ts / g.TW, tx = ts - tr * g.TW;
    const int prow = ((4 * sh + c_lo) * g.ROWS + 2 * tr) * g.W;                   // float index of the patch's first row
    const int prot = 16 * (4 * sh + c_lo);                                        // that channel's rotation
    const int pcm = (2 * tx - 1 + prot) & (g.W - 1), pc0 = (2 * tx + prot) & (g.W - 1), pcp = (2 * tx + 2 + prot) & (g.W - 1);
    const bool lcol = tx > 0, rcol = tx < g.TW - 1;

    // the raw window has two register sets, alternating with the chunk's parity: a load (HBM for a strip's first k tile) has ~1.75 steps
    // (9000+ cycles) to land.  The transformed weights come from L2 (every workgroup reads the same few k tiles): ONE set, each half
    // re-loaded one group after its commit, 0.75 steps ahead of its next use
    u32x4 rq[2][2] = {};
    u32x4 uw[4] = {};
    float d[16] = {};
    auto load_raw = [&](u32x4 (&q)[2]) {             // L: the cursor's chunk, then the cursor moves on
        if (!(ABL & 1)) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const bool cin = (first_chunk(cr.pair) + cr.ch) * 8 + qc[i] < g.C;
                q[i] = __builtin_amdgcn_raw_buffer_load_b128(rx, (cin & (cr.qoff[i] != kOobOff)) ? cr.qoff[i] + static_cast<unsigned>(cr.ch) * (32u * HW) : kOobOff, 0, 0);
            }
        }
        advance(cr);
    };
    auto write_raw = [&](float* raw, const u32x4 (&q)[2]) {      // W: registers -> raw window
        if (ABL & (4 | 256)) return;
#pragma unroll
        for (int i = 0; i < 2; ++i)
            if (qlds[i] >= 0) *reinterpret_cast<u32x4*>(raw + qlds[i]) = q[i];
    };
    auto load_u = [&](u32x4 (&u)[4], int s) {        // U rows 2s, 2s + 1 of the cursor's chunk; the cursor moves on after s = 1
        if (!(ABL & 2)) {
#pragma unroll
            for (int i = 2 * s; i < 2 * s + 2; ++i)
                u[i] = __builtin_amdgcn_raw_buffer_load_b128(ru, (cu.u_base != kOobOff && cu.ch < CHn) ? cu.u_base + static_cast<unsigned>(cu.ch) * (kWinoChunk * 4u) + i * 8192u : kOobOff, 0, 0);
        }
        if (s == 1) advance(cu);
    };
    auto read_patch = [&](const float* raw) {        // 3 reads per row: columns 2tx - 1 | 2tx, 2tx + 1 | 2tx + 2
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float* rp = raw + prow + i * g.W;
            const float a = rp[pcm], e = rp[pcp];
            const float2 m = *reinterpret_cast<const float2*>(rp + pc0);
            d[i * 4 + 0] = a;                 // the columns outside the image are masked where the transform starts (commit_slice 0):
            d[i * 4 + 1] = m.x;               // a select right behind its read would park the wave -- and its MFMAs -- on the LDS latency
            d[i * 4 + 2] = m.y;
            d[i * 4 + 3] = e;
        }
    };
    auto commit_slice = [&](f32x4* Ub, f32x4* Vb, const u32x4 (&u)[4], int s) {   // U rows 2s, 2s + 1; V rows 2s, 2s + 1
        if (ABL & 4) return;
        if (!(ABL & 64))
#pragma unroll
            for (int i = 2 * s; i < 2 * s + 2; ++i) reinterpret_cast<u32x4*>(Ub)[threadIdx.x + kWinoThreads * i] = u[i];
        if (ABL & 128) return;
        if (s == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                d[i * 4 + 0] = lcol ? d[i * 4 + 0] : 0.f;
                d[i * 4 + 3] = rcol ? d[i * 4 + 3] : 0.f;
            }
        }
        float* dst = reinterpret_cast<float*>(Vb) + (sh * 64 + ts) * 4 + c_lo;
#pragma unroll
        for (int i = 2 * s; i < 2 * s + 2; ++i) {
            float r[4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                r[j] = i == 0 ? d[0 + j] - d[8 + j] : i == 1 ? d[4 + j] + d[8 + j] : i == 2 ? d[8 + j] - d[4 + j] : d[4 + j] - d[12 + j];
            dst[(i * 4 + 0) * 512] = r[0] - r[2];
            dst[(i * 4 + 1) * 512] = r[1] + r[2];
            dst[(i * 4 + 2) * 512] = r[2] - r[1];
            dst[(i * 4 + 3) * 512] = r[1] - r[3];
        }
    };

    f32x16 acc[8];
#pragma unroll
    for (int p = 0; p < 8; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;

    // step of chunk n (parity P): MFMAs on buffers P; T(n + 1), U(n + 1) into buffers 1 - P; W(n + 2) into raw[P];
    // loads of U(n + 3) and raw chunk n + 4 into the register sets just emptied.
    // The step's ONE barrier stands between its third and fourth group of MFMAs, not at its end: by then every wave has issued its last
    // operand read of buffers P (the fourth group's operands are in registers) and its commits into buffers 1 - P, so the fourth group
    // runs with the operands of the NEXT step's first group being read under it -- the MFMA pipe does not wait for an LDS round trip
    // behind every barrier.  (Hazards: commits of chunk n + 1 [groups 0, 1 of step n] -> barrier n -> their first read [group 3 of
    // step n]; last read of buffers P [group 2 of step n] -> barrier n -> their next commits [groups 0, 1 of step n + 1]; write of
    // raw[P] [group 2] -> barrier n -> read_patch(raw[P]) [group 0 of step n + 1]; read_patch(raw[1 - P]) [group 0] -> barrier n ->
    // its next write [group 2 of step n + 1].)
    f32x4 oa[2][2] = {}, ob[2][2] = {};  // operands of the current / next group; [0] holds the first group's at a step's entry
    auto step = [&](auto parity) {
        constexpr int P = decltype(parity)::value, Q = 1 - P;
        f32x4* const Ub = smem + P * 4096;
        f32x4* const Vb = Ub + 2048;
        f32x4* const Un = smem + Q * 4096;
        f32x4* const Vn = Un + 2048;
        const int oidx = (ph * 8) * 128 + half * 64 + l31;
        const f32x4* ap = Ub + oidx + wm * 32;
        const f32x4* bp = Vb + oidx + wn * 32;
#pragma unroll
        for (int grp = 0; grp < 4; ++grp) {
            const int cur = grp & 1, nxt = cur ^ 1;
            if (!(ABL & 8)) {
                if (grp < 3) {
                    oa[nxt][0] = ap[(2 * grp + 2) * 128]; ob[nxt][0] = bp[(2 * grp + 2) * 128];
                    oa[nxt][1] = ap[(2 * grp + 3) * 128]; ob[nxt][1] = bp[(2 * grp + 3) * 128];
                } else {                 // behind the barrier: the next step's first operands, from the buffers committed in groups 0, 1
                    const f32x4* an = Un + oidx + wm * 32;
                    const f32x4* bn = Vn + oidx + wn * 32;
                    oa[nxt][0] = an[0]; ob[nxt][0] = bn[0]; oa[nxt][1] = an[128]; ob[nxt][1] = bn[128];
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[2 * grp] = __builtin_amdgcn_mfma_f32_32x32x2f32(oa[cur][0][j], ob[cur][0][j], acc[2 * grp], 0, 0, 0);
                acc[2 * grp + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(oa[cur][1][j], ob[cur][1][j], acc[2 * grp + 1], 0, 0, 0);
            }
            // the patches of the NEXT step's transform are read behind the barrier too (the window this step's group 2 wrote): a
            // group's side work then never waits for an LDS read it has just issued -- a wave that waits issues no MFMAs, and the two
            // waves of a SIMD run in step
            if (grp == 0) commit_slice(Un, Vn, uw, 0);
            else if (grp == 1) { commit_slice(Un, Vn, uw, 1); load_u(uw, 0); }
            else if (grp == 2) { write_raw(P ? raw1 : raw0, rq[P]); load_u(uw, 1); }
            else { if (!(ABL & (4 | 512))) read_patch(P ? raw1 : raw0); load_raw(rq[P]); }
            #pragma unroll
            for (int m = 0; m < 8; ++m) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // one MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);      // up to 6 VALU
                __builtin_amdgcn_sched_group_barrier(0x200, 2, 0);      // up to 2 LDS writes
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);      // up to 2 LDS reads
                __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);      // up to 2 VMEM reads
            }
            __builtin_amdgcn_sched_barrier(0);
            if (grp == 2) {
                __syncthreads();
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    auto first_operands = [&]() {        // of a parity-0 step, after a barrier behind the commits into buffers 0
        if (ABL & 8) return;
        const int oidx = (ph * 8) * 128 + half * 64 + l31;
        const f32x4* a0 = smem + oidx + wm * 32;
        const f32x4* b0 = smem + 2048 + oidx + wn * 32;
        oa[0][0] = a0[0]; ob[0][0] = b0[0]; oa[0][1] = a0[128]; ob[0][1] = b0[128];
    };

    // prologue of the stream: chunk 0 staged and transformed, chunk 1 in raw[1], U(1), U(2) and raw chunks 2, 3 in flight
    load_raw(rq[0]); load_u(uw, 0); load_u(uw, 1);
    load_raw(rq[1]);
    write_raw(raw0, rq[0]);
    __syncthreads();
    read_patch(raw0);
    commit_slice(smem, smem + 2048, uw, 0); commit_slice(smem, smem + 2048, uw, 1);
    write_raw(raw1, rq[1]);
    load_raw(rq[0]); load_raw(rq[1]);
    load_u(uw, 0); load_u(uw, 1);
    __syncthreads();
    first_operands();
    if (!(ABL & 4)) read_patch(raw1);

    // ---- output transform y = At (m A) of a pair.  C/D layout: col = lane & 31 (tile), row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5) (k).
    // A wave holds rows 2 ph, 2 ph + 1 of the 4 x 4 products m and forms z = m A of both (2 values per row).  y row 0 = z0 + z1 + z2 is
    // finished and stored by the ph = 0 wave, y row 1 = z1 - z2 - z3 by its ph = 1 partner on the same SIMD: each hands the other ONE z
    // row (32 floats per lane) through the parity-1 operand buffers (consumed by the step that just ended; next written by the commits of
    // the NEXT step, hence the second barrier).  Nothing here touches the vector memory counter: the bias comes by scalar loads (a k
    // index is wave-uniform up to the lane half) and the stores are buffer stores whose range check drops the channels past K -- a
    // `bias[k]` vector load per register made every one of the 16 store groups wait for the previous group's stores AND for the prefetched
    // loads of the next pair (vmcnt counts them all, in order): 5-7 us per pair, now ~2.
    float* const xch = reinterpret_cast<float*>(smem + 4096) + (wave & 3) * 4096 + lane;
    const int ph_u = wave_u >> 2, wm_u = (wave_u >> 1) & 1;
    const int tl = wn * 32 + l31;                    // tile within the strip: row tl / TW of its R tile rows
    const unsigned out_lane = (static_cast<unsigned>(wm * 32 + 4 * half) * static_cast<unsigned>(HW) +
                               static_cast<unsigned>((2 * (tl / g.TW) + ph) * g.W + 2 * (tl % g.TW))) * 4u;
    const float sgn = ph ? -1.f : 1.f;
    // the bias of a pair's output channels: lane j of every wave holds bias[k tile's first channel + 32 wm + (j & 31)], loaded a whole
    // pair ahead; the epilogue fetches the value of a register's channel from lane jr + 4 half by ds_bpermute, so it neither issues a
    // vector load nor waits on the vector memory counter
    auto load_bias = [&](int pair) {
        float v = 0.f;
        if (bias != nullptr && pair < pair_end && (!SPLIT || (pair & (CS - 1)) == 0)) {
            const int pk = SPLIT ? pair >> cs_shift : pair;
            const int k = (pk % g.KT) * 64 + wm * 32 + l31;
            if (k < g.K) v = bias[k];
        }
        return v;
    };
    float bvec = load_bias(pair_begin);
    const int bsel = 16 * half;
    for (int pair = pair_begin; pair < pair_end; ++pair) {
        for (int ch = 0; ch < CHp; ch += 2) {
            step(std::integral_constant<int, 0>());
            step(std::integral_constant<int, 1>());
        }
        const int pk = SPLIT ? pair >> cs_shift : pair;
        const int tt = pk / g.KT, kt = pk - tt * g.KT;
        if (ABL & 32) continue;          // timing experiment: no epilogue at all (the accumulators run on)
        if (ph_u) {                      // z row 2 for the partner's y row 0
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                xch[(r * 4 + 0) * 64] = acc[0][r] + acc[1][r] + acc[2][r];
                xch[(r * 4 + 1) * 64] = acc[1][r] - acc[2][r] - acc[3][r];
            }
        } else {                         // z row 1 for the partner's y row 1
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                xch[(r * 4 + 2) * 64] = acc[4][r] + acc[5][r] + acc[6][r];
                xch[(r * 4 + 3) * 64] = acc[5][r] - acc[6][r] - acc[7][r];
            }
        }
        __syncthreads();
        const int kq = kt * 64 + wm_u * 32;          // first output channel of this wave's quarter
        if (kq < g.K) {
            const int b_u = tt / strips_per_img;     // the 64 tiles of a strip are whole tile rows of ONE image
            const int ty0 = (tt - b_u * strips_per_img) * g.R;
            const int krem = g.K - kt * 64;
            float* const obase = out + (static_cast<size_t>(b_u) * g.Kout + static_cast<size_t>(kt) * 64) * HW;
            const rsrc_t ro = make_rsrc(obase, static_cast<unsigned>(krem < 64 ? krem : 64) * static_cast<unsigned>(HW) * 4u);
            unsigned voff = out_lane + static_cast<unsigned>(2 * ty0 * g.W) * 4u;
            asm volatile("" : "+v"(voff));           // (keeps the 16 channel offsets below as scalar addends: hoisted as 16 registers they spill)
            const float* const rcv = xch + (ph_u ? 2 * 64 : 0);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int jr = (r & 3) + 8 * (r >> 2);                   // k = kq + jr + 4 half
                const float za0 = acc[0][r] + acc[1][r] + acc[2][r], za1 = acc[1][r] - acc[2][r] - acc[3][r];      // z of row 2 ph
                const float zb0 = acc[4][r] + acc[5][r] + acc[6][r], zb1 = acc[5][r] - acc[6][r] - acc[7][r];      // z of row 2 ph + 1
                const float bv = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(bsel + 4 * jr, __builtin_bit_cast(int, bvec)));
                // ph 0: (z0 + z1) + z2;  ph 1: z1 - (z2 + z3)
                float y0 = __builtin_fmaf(sgn, za0 + zb0, rcv[(r * 4 + 0) * 64]) + bv;
                float y1 = __builtin_fmaf(sgn, za1 + zb1, rcv[(r * 4 + 1) * 64]) + bv;
                if (g.act == 1) {
                    y0 = y0 > 0.f ? y0 : y0 * g.slope;
                    y1 = y1 > 0.f ? y1 : y1 * g.slope;
                }
                const unsigned vo = voff + static_cast<unsigned>(jr) * static_cast<unsigned>(HW) * 4u;
                if (ABL & 16) {          // timing experiment: the epilogue without its global stores (one lane in a million keeps it alive)
                    if (y0 + y1 != 12345.678f) continue;
                }
                if (SPLIT) {
                    // a split of the reduction: the output transform is linear, the splits' shares meet in the zero-filled output
                    // (the host only splits a call without an activation; the bias rides with split 0)
                    if (kq + jr + 4 * half < g.K) {
                        float* o = obase + (vo >> 2);
                        atomic_add(o, y0); atomic_add(o + 1, y1);
                    }
                    continue;
                }
                const u32x2 yy = {__builtin_bit_cast(unsigned, y0), __builtin_bit_cast(unsigned, y1)};
                __builtin_amdgcn_raw_buffer_store_b64(yy, ro, vo, 0, 0);          // W is even: 8-byte aligned pairs
            }
        }
#pragma unroll
        for (int p = 0; p < 8; ++p)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;
        bvec = load_bias(pair + 1);
        __syncthreads();                 // the exchange is read before the next step's commits overwrite it
    }
    if (ABL & 32) {                      // keeps the accumulators of the epilogue-free timing variant alive
        float t = 0.f;
#pragma unroll
        for (int p = 0; p < 8; ++p)
#pragma unroll
            for (int r = 0; r < 16; ++r) t += acc[p][r];
        if (t == 12345.678f) out[threadIdx.x] = t;
    }
}

// ---------------------------------------------------------------------------------------------------- wave-specialised variant
// Round 5 (late).  The ablations of the kernel above (profiles/r05_winograd_pair_and_chunk_ablation.txt) say: the MFMAs with their operand
// reads and the barrier ALONE run at 98 % of the matrix pipe's bound; what costs 25-30 % is the side work -- window loads, transforms, LDS
// commits -- sitting in the SAME in-order instruction streams: a wave that waits for an LDS read or issues stores issues no MFMAs, and
// ~9.4 side instructions per 64-cycle MFMA slot and SIMD is the issue limit.  Here the roles are separate waves: 8 CONSUMER waves (the
// compute roles of the kernel above: operand reads + 32 MFMAs per chunk, nothing else) and 4 PRODUCER waves, one per SIMD, that do
// all the staging (each of their 256 threads carries what two threads carried above).  12 waves of 168 registers = 3 per SIMD.  A
// consumer's stream is the 98 % one; a producer has a whole chunk time for its ~200 instructions and may wait as it likes.  Same LDS
// plan (128 KiB operands + 2 x 16 KiB raw windows), same pipeline depths, one barrier per chunk, the same epilogue (the producers only
// pass its two barriers).  Calls with a split reduction keep the kernel above.
constexpr int kWsThreads = 768;
template <int ABL>
__global__ void __launch_bounds__(kWsThreads)
winograd_conv_ws_kernel(const float* __restrict__ x, const float* __restrict__ U, const float* __restrict__ bias, float* __restrict__ out,
                        const WinoGeo g, int remap) {
    extern __shared__ f32x4 smem[];          // operands (8192 f32x4), then raw[2][kWinoRawFloats]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    constexpr unsigned kOobOff = 0xFFFFFFF0u;
    const int HW = g.H * g.W;
    const int npairs = g.TT * g.KT;
    const int L = static_cast<int>(xcd_remap(blockIdx.x, gridDim.x, remap));
    const int per = npairs / static_cast<int>(gridDim.x), extra = npairs - per * static_cast<int>(gridDim.x);
    const int pair_begin = L * per + min(L, extra), pair_cnt = per + (L < extra ? 1 : 0);
    const int pair_end = pair_begin + pair_cnt;
    if (pair_cnt == 0) return;
    const int CHn = g.CH, CHp = (CHn + 1) & ~1;      // an odd count ends with a chunk of zeros (inputs AND weights out of range: see above)
    const int strips_per_img = g.TH / g.R;

    // The two roles are two separate loops (their registers must not be live together: 128 accumulators here, 64 staging registers
    // there); both execute the same sequence of barriers: 2 (prologue) + per pair CHp (steps) + 2 (epilogue).
    if (wave_u < 8) {
        // ============================================================ consumers: operand reads + MFMAs, the output transform
        const int l31 = lane & 31, half = lane >> 5;
        const int ph = wave >> 2, wm = (wave >> 1) & 1, wn = wave & 1;
        f32x16 acc[8];
#pragma unroll
        for (int p = 0; p < 8; ++p)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;
        auto step = [&](auto parity) {
            constexpr int P = decltype(parity)::value;
            if (ABL & 1) { __syncthreads(); return; }          // timing experiment: producers alone
            const f32x4* ap = smem + P * 4096 + (ph * 8) * 128 + half * 64 + wm * 32 + l31;
            const f32x4* bp = smem + P * 4096 + 2048 + (ph * 8) * 128 + half * 64 + wn * 32 + l31;
            f32x4 oa[2][2], ob[2][2];
            oa[0][0] = ap[0]; ob[0][0] = bp[0]; oa[0][1] = ap[128]; ob[0][1] = bp[128];
#pragma unroll
            for (int grp = 0; grp < 4; ++grp) {
                const int cur = grp & 1, nxt = cur ^ 1;
                if (grp < 3) {
                    oa[nxt][0] = ap[(2 * grp + 2) * 128]; ob[nxt][0] = bp[(2 * grp + 2) * 128];
                    oa[nxt][1] = ap[(2 * grp + 3) * 128]; ob[nxt][1] = bp[(2 * grp + 3) * 128];
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[2 * grp] = __builtin_amdgcn_mfma_f32_32x32x2f32(oa[cur][0][j], ob[cur][0][j], acc[2 * grp], 0, 0, 0);
                    acc[2 * grp + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(oa[cur][1][j], ob[cur][1][j], acc[2 * grp + 1], 0, 0, 0);
                }
            }
            __syncthreads();
        };
        __syncthreads();
        __syncthreads();
        float* const xch = reinterpret_cast<float*>(smem + 4096) + (wave & 3) * 4096 + lane;
        const int ph_u = wave_u >> 2, wm_u = (wave_u >> 1) & 1;
        const int tl = wn * 32 + l31;
        const unsigned out_lane = (static_cast<unsigned>(wm * 32 + 4 * half) * static_cast<unsigned>(HW) +
                                   static_cast<unsigned>((2 * (tl / g.TW) + ph) * g.W + 2 * (tl % g.TW))) * 4u;
        const float sgn = ph ? -1.f : 1.f;
        auto load_bias = [&](int pair) {
            float v = 0.f;
            if (bias != nullptr && pair < pair_end) {
                const int k = (pair % g.KT) * 64 + wm * 32 + l31;
                if (k < g.K) v = bias[k];
            }
            return v;
        };
        float bvec = load_bias(pair_begin);
        const int bsel = 16 * half;
        for (int pair = pair_begin; pair < pair_end; ++pair) {
            for (int ch = 0; ch < CHp; ch += 2) {
                step(std::integral_constant<int, 0>());
                step(std::integral_constant<int, 1>());
            }
            const int tt = pair / g.KT, kt = pair - tt * g.KT;
            if (ph_u) {                  // z row 2 for the partner's y row 0 (the exchange of the kernel above)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    xch[(r * 4 + 0) * 64] = acc[0][r] + acc[1][r] + acc[2][r];
                    xch[(r * 4 + 1) * 64] = acc[1][r] - acc[2][r] - acc[3][r];
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    xch[(r * 4 + 2) * 64] = acc[4][r] + acc[5][r] + acc[6][r];
                    xch[(r * 4 + 3) * 64] = acc[5][r] - acc[6][r] - acc[7][r];
                }
            }
            __syncthreads();
            const int kq = kt * 64 + wm_u * 32;
            if (kq < g.K) {
                const int b_u = tt / strips_per_img;
                const int ty0 = (tt - b_u * strips_per_img) * g.R;
                const int krem = g.K - kt * 64;
                float* const obase = out + (static_cast<size_t>(b_u) * g.Kout + static_cast<size_t>(kt) * 64) * HW;
                const rsrc_t ro = make_rsrc(obase, static_cast<unsigned>(krem < 64 ? krem : 64) * static_cast<unsigned>(HW) * 4u);
                unsigned voff = out_lane + static_cast<unsigned>(2 * ty0 * g.W) * 4u;
                asm volatile("" : "+v"(voff));
                const float* const rcv = xch + (ph_u ? 2 * 64 : 0);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int jr = (r & 3) + 8 * (r >> 2);
                    const float za0 = acc[0][r] + acc[1][r] + acc[2][r], za1 = acc[1][r] - acc[2][r] - acc[3][r];
                    const float zb0 = acc[4][r] + acc[5][r] + acc[6][r], zb1 = acc[5][r] - acc[6][r] - acc[7][r];
                    const float bv = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(bsel + 4 * jr, __builtin_bit_cast(int, bvec)));
                    float y0 = __builtin_fmaf(sgn, za0 + zb0, rcv[(r * 4 + 0) * 64]) + bv;
                    float y1 = __builtin_fmaf(sgn, za1 + zb1, rcv[(r * 4 + 1) * 64]) + bv;
                    if (g.act == 1) {
                        y0 = y0 > 0.f ? y0 : y0 * g.slope;
                        y1 = y1 > 0.f ? y1 : y1 * g.slope;
                    }
                    const unsigned vo = voff + static_cast<unsigned>(jr) * static_cast<unsigned>(HW) * 4u;
                    const u32x2 yy = {__builtin_bit_cast(unsigned, y0), __builtin_bit_cast(unsigned, y1)};
                    __builtin_amdgcn_raw_buffer_store_b64(yy, ro, vo, 0, 0);
                }
            }
#pragma unroll
            for (int p = 0; p < 8; ++p)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;
            bvec = load_bias(pair + 1);
            __syncthreads();
        }
        return;
    }

    // ================================================================ producers (threads 512 .. 767): every load, transform and LDS commit
    __builtin_amdgcn_s_setprio(3);           // the youngest waves of the workgroup lose every issue arbitration at equal priority -- and the barrier waits for them
    const rsrc_t rx = make_rsrc(x, g.x_bytes);
    const rsrc_t ru = make_rsrc(U, g.u_bytes);
    float* const raw0 = reinterpret_cast<float*>(smem + 8192);
    float* const raw1 = raw0 + kWinoRawFloats;
    const int pt = static_cast<int>(threadIdx.x) - 512;
    const int W4 = g.W >> 2, nquads = 8 * g.ROWS * W4;
    const int w4_shift = __builtin_ctz(static_cast<unsigned>(W4));
    const float quad_rcp = 1.f / static_cast<float>(g.ROWS * W4);
    auto quad_of = [&](int q, int& c, int& rr, int& col) {
        c = static_cast<int>((static_cast<float>(q) + 0.5f) * quad_rcp);
        const int rem = q - c * (g.ROWS * W4);
        rr = rem >> w4_shift;
        col = 4 * (rem & (W4 - 1));
    };
    int qc[4], qlds[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int q = pt + 256 * i;
        int c, rr, col;
        quad_of(q, c, rr, col);
        qc[i] = q < nquads ? c : 8;
        qlds[i] = q < nquads ? (c * g.ROWS + rr) * g.W + ((col + 16 * c) & (g.W - 1)) : -1;
    }
    struct Cursor {
        int pair, ch;
        unsigned qoff[4];
        unsigned u_base;
    };
    auto seat = [&](Cursor& c) {
        if (c.pair >= pair_end) {
#pragma unroll
            for (int i = 0; i < 4; ++i) c.qoff[i] = kOobOff;
            c.u_base = kOobOff;
            return;
        }
        const int tt = c.pair / g.KT, kt = c.pair - tt * g.KT;
        const int b = tt / strips_per_img, ty0 = (tt - b * strips_per_img) * g.R;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int q = pt + 256 * i;
            int qch, qrow, qcol;
            quad_of(q, qch, qrow, qcol);
            const int iy = 2 * ty0 - 1 + qrow;
            const bool ok = q < nquads && iy >= 0 && iy < g.H;
            c.qoff[i] = ok ? ((static_cast<unsigned>(b) * g.C + qch) * static_cast<unsigned>(HW) + static_cast<unsigned>(iy * g.W + qcol)) * 4u : kOobOff;
        }
        c.u_base = (static_cast<unsigned>(kt) * g.CH) * (kWinoChunk * 4u) + static_cast<unsigned>(pt) * 16u;
    };
    auto advance = [&](Cursor& c) {
        if (++c.ch == CHp) {
            c.ch = 0;
            ++c.pair;
            seat(c);
        }
    };
    Cursor cr, cu;
    cr.pair = cu.pair = pair_begin;
    cr.ch = cu.ch = 0;
    seat(cr);
    cu = cr;
    // patches: tile ts, channels c_lo and 4 + c_lo of the chunk
    const int c_lo = lane & 3, ts = (wave & 3) * 16 + (lane >> 2);
    const int tr = ts / g.TW, tx = ts - tr * g.TW;
    const bool lcol = tx > 0, rcol = tx < g.TW - 1;
    int prow[2], pcm[2], pc0[2], pcp[2];
#pragma unroll
    for (int sh = 0; sh < 2; ++sh) {
        prow[sh] = ((4 * sh + c_lo) * g.ROWS + 2 * tr) * g.W;
        const int prot = 16 * (4 * sh + c_lo);
        pcm[sh] = (2 * tx - 1 + prot) & (g.W - 1);
        pc0[sh] = (2 * tx + prot) & (g.W - 1);
        pcp[sh] = (2 * tx + 2 + prot) & (g.W - 1);
    }
    u32x4 rq[2][4] = {};
    u32x4 uw[8] = {};
    auto load_raw = [&](u32x4 (&q)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool cin = cr.ch * 8 + qc[i] < g.C;
            q[i] = __builtin_amdgcn_raw_buffer_load_b128(rx, (cin & (cr.qoff[i] != kOobOff)) ? cr.qoff[i] + static_cast<unsigned>(cr.ch) * (32u * HW) : kOobOff, 0, 0);
        }
        advance(cr);
    };
    auto write_raw = [&](float* raw, const u32x4 (&q)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (qlds[i] >= 0) *reinterpret_cast<u32x4*>(raw + qlds[i]) = q[i];
    };
    auto load_u = [&]() {
#pragma unroll
        for (int i = 0; i < 8; ++i)
            uw[i] = __builtin_amdgcn_raw_buffer_load_b128(ru, (cu.u_base != kOobOff && cu.ch < CHn) ? cu.u_base + static_cast<unsigned>(cu.ch) * (kWinoChunk * 4u) + i * 4096u : kOobOff, 0, 0);
        advance(cu);
    };
    auto commit_u = [&](f32x4* Ub) {
#pragma unroll
        for (int i = 0; i < 8; ++i) reinterpret_cast<u32x4*>(Ub)[pt + 256 * i] = uw[i];
    };
    auto transform = [&](const float* raw, f32x4* Vb) {      // both patches of this thread: raw window -> Bt d B -> V
#pragma unroll
        for (int sh = 0; sh < 2; ++sh) {
            float d[16];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float* rp = raw + prow[sh] + i * g.W;
                const float a = rp[pcm[sh]], e = rp[pcp[sh]];
                const float2 m = *reinterpret_cast<const float2*>(rp + pc0[sh]);
                d[i * 4 + 0] = lcol ? a : 0.f;
                d[i * 4 + 1] = m.x;
                d[i * 4 + 2] = m.y;
                d[i * 4 + 3] = rcol ? e : 0.f;
            }
            float* dst = reinterpret_cast<float*>(Vb) + (sh * 64 + ts) * 4 + c_lo;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float r[4];
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    r[j] = i == 0 ? d[0 + j] - d[8 + j] : i == 1 ? d[4 + j] + d[8 + j] : i == 2 ? d[8 + j] - d[4 + j] : d[4 + j] - d[12 + j];
                dst[(i * 4 + 0) * 512] = r[0] - r[2];
                dst[(i * 4 + 1) * 512] = r[1] + r[2];
                dst[(i * 4 + 2) * 512] = r[2] - r[1];
                dst[(i * 4 + 3) * 512] = r[1] - r[3];
            }
        }
    };
    // step of chunk n (parity P) while the consumers run their MFMAs on buffers P: U(n + 1) -> U buffer 1 - P and the load of U(n + 2);
    // the patches of chunk n + 1 from raw[1 - P] -> V buffer 1 - P; the window of chunk n + 2 (registers) -> raw[P] and the load of
    // chunk n + 4 into those registers
    auto pstep = [&](auto parity) {
        constexpr int P = decltype(parity)::value, Q = 1 - P;
        f32x4* const Un = smem + Q * 4096;
        if (ABL & 2) { __syncthreads(); return; }              // timing experiment: consumers alone
        if (!(ABL & 4)) commit_u(Un);
        if (!(ABL & 32)) load_u();
        if (!(ABL & 8)) transform(Q ? raw1 : raw0, Un + 2048);
        if (!(ABL & 16)) write_raw(P ? raw1 : raw0, rq[P]);
        if (!(ABL & 32)) load_raw(rq[P]);
        __syncthreads();
    };
    // prologue: chunk 0 staged and transformed into buffers 0, chunk 1's window in raw[1], U(1) and the windows of chunks 2, 3 in flight
    load_raw(rq[0]); load_u(); load_raw(rq[1]);
    write_raw(raw0, rq[0]);
    __syncthreads();
    transform(raw0, smem + 2048);
    commit_u(smem);
    write_raw(raw1, rq[1]);
    load_raw(rq[0]); load_raw(rq[1]);
    load_u();
    __syncthreads();
    for (int pair = pair_begin; pair < pair_end; ++pair) {
        for (int ch = 0; ch < CHp; ch += 2) {
            pstep(std::integral_constant<int, 0>());
            pstep(std::integral_constant<int, 1>());
        }
        __syncthreads();                 // the consumers' exchange of the output transform ...
        __syncthreads();                 // ... lives in the parity-1 buffers: no commit before they have read it
    }
}

// ---------------------------------------------------------------------------------------------------- thin tail
// The last 1-4 output channels of a layer whose channel count is just past a multiple of 64 (netG's residual blocks: 195 =
// 3 * 64 + 3) would cost a whole 64-channel tile of the Winograd kernel (a quarter of the launch).  They are a direct sum
// on the vector ALUs instead: a lane owns 4 consecutive pixels of a row and all KN channels, the 4 waves of a workgroup
// split the input channels (weights are wave-uniform: scalar loads feeding the FMAs) and meet in LDS.
struct ThinGeo {
    int C, H, W, Kout, k_off;
    int strips;              // B * H * W / 4
    int act;
    float slope;
    unsigned x_bytes;
};

constexpr int kThinWaves = 16;

template <int KN>
__global__ void __launch_bounds__(64 * kThinWaves)
conv3x3_thin_kernel(const float* __restrict__ x, const float* __restrict__ wt, const float* __restrict__ bias, float* __restrict__ out,
                    const ThinGeo g) {
    __shared__ float red[kThinWaves - 1][KN * 4][64];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int strip = blockIdx.x * 64 + lane;
    const bool sv = strip < g.strips;
    const int W4 = g.W >> 2, HW = g.H * g.W;
    const int row = sv ? strip / W4 : 0;                 // b * H + y
    const int x0 = sv ? (strip - row * W4) * 4 : 0;
    const int b = row / g.H, y = row - b * g.H;
    constexpr unsigned kOobOff = 0xFFFFFFF0u;
    const rsrc_t rx = make_rsrc(x, g.x_bytes);
    unsigned offm[3], off4[3], offp[3];                 // per input row: pixel x0 - 1, the aligned four, pixel x0 + 4 (channel 0)
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const int yy = y - 1 + r;
        const bool rv = sv && yy >= 0 && yy < g.H;
        const unsigned base = (static_cast<unsigned>(b) * g.C * static_cast<unsigned>(HW) + static_cast<unsigned>(yy * g.W + x0)) * 4u;
        off4[r] = rv ? base : kOobOff;
        offm[r] = (rv && x0 > 0) ? base - 4u : kOobOff;
        offp[r] = (rv && x0 + 4 < g.W) ? base + 16u : kOobOff;
    }
    float acc[KN][4];
#pragma unroll
    for (int k = 0; k < KN; ++k)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[k][j] = 0.f;
    auto load = [&](int c, float (&in)[3][6]) {          // a channel past the last reads 0
        const unsigned co = static_cast<unsigned>(c) * static_cast<unsigned>(HW) * 4u;
        const bool cv = c < g.C;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            // The pixels left and right of the aligned four are the neighbouring LANES' (strips are consecutive along a row): two lane
            // shuffles instead of two more gathers.  Round 5: the three loads per row cost the same in the texture-address unit (64
            // lanes 16 bytes apart = 16 cache lines each), and that unit bounded the kernel -- ~50 us whether the plane was 128 x 128
            // or 32 x 32 (tools/thin_tail_time.py).  Only a wave's first / last lane still loads its outer neighbour.
            unsigned q[4];           // (never __builtin_bit_cast a vector ELEMENT: clang reads the vector's first lane for each)
            buf_load_dwords<4>(rx, (cv & (off4[r] != kOobOff)) ? off4[r] + co : kOobOff, q);
            const float edge_l = buf_ld<float>(rx, (cv & (lane == 0) & (offm[r] != kOobOff)) ? offm[r] + co : kOobOff);
            const float edge_r = buf_ld<float>(rx, (cv & (lane == 63) & (offp[r] != kOobOff)) ? offp[r] + co : kOobOff);
            in[r][1] = __uint_as_float(q[0]); in[r][2] = __uint_as_float(q[1]);
            in[r][3] = __uint_as_float(q[2]); in[r][4] = __uint_as_float(q[3]);
            const float from_l = __shfl_up(in[r][4], 1, 64), from_r = __shfl_down(in[r][1], 1, 64);
            in[r][0] = offm[r] != kOobOff ? (lane == 0 ? edge_l : from_l) : 0.f;          // (offm / offp: the pixel exists in this row)
            in[r][5] = offp[r] != kOobOff ? (lane == 63 ? edge_r : from_r) : 0.f;
        }
    };
    auto fma_all = [&](int c, const float (&in)[3][6]) {
        const float* wp = wt + static_cast<size_t>(c) * kThinStride;          // wave-uniform: scalar loads
#pragma unroll
        for (int k = 0; k < KN; ++k)
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const float wv = wp[k * 9 + t];
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[k][j] = fmaf(wv, in[t / 3][j + t % 3], acc[k][j]);
            }
    };
    // channels wave, wave + 8, ...: the next channel's window is in flight while this one's 36 KN FMAs issue
    float in0[3][6], in1[3][6];
    load(wave, in0);
    for (int c = wave; c < g.C; c += 2 * kThinWaves) {
        load(c + kThinWaves, in1);
        fma_all(c, in0);
        if (c + kThinWaves >= g.C) break;
        load(c + 2 * kThinWaves, in0);
        fma_all(c + kThinWaves, in1);
    }
    if (wave > 0) {
#pragma unroll
        for (int k = 0; k < KN; ++k)
#pragma unroll
            for (int j = 0; j < 4; ++j) red[wave - 1][k * 4 + j][lane] = acc[k][j];
    }
    __syncthreads();
    if (wave > 0 || !sv) return;
#pragma unroll
    for (int k = 0; k < KN; ++k) {
        const float bv = bias ? bias[g.k_off + k] : 0.f;
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float t = acc[k][j];
#pragma unroll
            for (int q = 0; q < kThinWaves - 1; ++q) t += red[q][k * 4 + j][lane];
            t += bv;
            if (g.act == 1) t = t > 0.f ? t : t * g.slope;
            v[j] = t;
        }
        *reinterpret_cast<float4*>(out + (static_cast<size_t>(b) * g.Kout + g.k_off + k) * HW + static_cast<size_t>(y) * g.W + x0) =
            make_float4(v[0], v[1], v[2], v[3]);
    }
}

}  // namespace
}  // namespace ffwm

using namespace ffwm;

// Splits of the reduction for a raw-staged call that offers the persistent workgroups fewer (strip, k tile) pairs than half the
// CUs (256 -> 256 at 32 x 32, batch 8: 128 pairs; 512 -> 512 at 16 x 16: 64): the largest of 4 / 2 that keeps one round of
// workgroups, whole even chunk counts per split and >= 16 chunks (128 channels) each -- measured (tools/wino_split_check.py,
// profiles/r04_winograd_split.txt): with shorter splits the prologue / epilogue of a workgroup, the zero-fill and the atomics
// cost more than the idle CUs (64 -> 64 at 64 x 64: 35 -> 50 us), with 16 chunks 256 -> 256 at 32 x 32 goes 106 -> 92 us and
// 512 -> 512 at 16 x 16 199 -> 99 us.  Only without a fused activation (the epilogue of a split cannot apply one); 1 = no split.
static int winograd_splits(int pairs, int CH, int act, int cus) {
    if (act != 0 || !options().conv_wino_split || pairs <= 0) return 1;
    for (int cs = options().conv_wino_split == 2 ? 2 : 4; cs >= 2; cs -= 2) {          // (2: capped -- a + b is order-independent, a + b + c + d is not)
        if (pairs * cs > cus) continue;
        if (CH % cs != 0) continue;
        const int chs = CH / cs;
        if (chs < 16 || (chs & 1)) continue;
        return cs;
    }
    return 1;
}

extern "C" int ffwm_conv3x3_winograd_splits(int64_t B, int64_t C, int64_t H, int64_t W, int64_t K, int act) {
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || K <= 0) return 1;
    const bool rawshape = options().conv_wino_raw && (W == 16 || W == 32 || W == 64 || W == 128) && H % 2 == 0;
    if (!rawshape) return 1;
    const int TH = static_cast<int>((H + 1) / 2), TW = static_cast<int>((W + 1) / 2);
    if (!(TW <= 64 && 64 % TW == 0 && TH % (64 / TW) == 0)) return 1;
    const int tail = static_cast<int>(K % 64);
    const bool thin = (K > 64 || K <= 4) && tail >= 1 && tail <= 4 && W % 4 == 0 && options().conv_thin_tail;
    const int64_t Kw = thin ? K - tail : K;
    const int64_t pairs = ((B * TH * TW + kWinoTiles - 1) / kWinoTiles) * ((Kw + 63) / 64);
    if (pairs <= 0 || pairs > 4096) return 1;
    return winograd_splits(static_cast<int>(pairs), static_cast<int>((C + 7) / 8), act, device_cus());
}

extern "C" int64_t ffwm_conv3x3_winograd_workspace_bytes(int64_t K, int64_t C) {
    if (K <= 0 || C <= 0) return 0;
    return ((K + 63) / 64) * ((C + 7) / 8) * static_cast<int64_t>(kWinoChunk) * 4;
}

extern "C" int ffwm_conv3x3_winograd_weights_multi(const ffwm_wino_weights* items, int n, int dtype, void* stream) {
    const char* fn = "ffwm_conv3x3_winograd_weights_multi";
    FFWM_REQUIRE(dtype == FFWM_F32, FFWM_ERR_DTYPE, "%s: float32 only", fn);
    FFWM_REQUIRE(items || n == 0, FFWM_ERR_ARG, "%s: NULL item array", fn);
    FFWM_REQUIRE(n >= 0 && n <= 4096, FFWM_ERR_ARG, "%s: 0 .. 4096 items", fn);
    hipStream_t st = static_cast<hipStream_t>(stream);
    WinoWeightsTable tab;
    tab.n = 0;
    int blocks = 0;
    double bytes = 0;
    auto flush = [&]() -> int {
        if (tab.n == 0) return FFWM_OK;
        {
            LaunchScope ls("conv_winograd_weights_multi", st, bytes);
            hipLaunchKernelGGL(winograd_weights_multi_kernel, dim3(static_cast<unsigned>(blocks)), dim3(kBlock), 0, st, tab);
        }
        tab.n = 0; blocks = 0; bytes = 0;
        return check_launch(fn);
    };
    for (int i = 0; i < n; ++i) {
        const ffwm_wino_weights& w = items[i];
        FFWM_REQUIRE(w.weight && w.workspace && w.K > 0 && w.C > 0 && (w.data_gradient == 0 || w.data_gradient == 1), FFWM_ERR_ARG,
                     "%s: item %d: NULL pointer, non-positive size or data_gradient not 0 / 1", fn, i);
        FFWM_REQUIRE(ffwm_conv3x3_winograd_workspace_bytes(w.K, w.C) < (1LL << 31), FFWM_ERR_SIZE, "%s: item %d: workspace beyond 2 GiB", fn, i);
        // the same split of the output channels as ffwm_conv3x3_winograd_forward makes for this (K, W % 4)
        const int tail = static_cast<int>(w.K % 64);
        const bool thin = (w.K > 64 || w.K <= 4) && tail >= 1 && tail <= 4 && w.width_multiple_of_4 && options().conv_thin_tail;
        WinoWeightsItem& q = tab.it[tab.n];
        q.w = static_cast<const float*>(w.weight);
        q.U = static_cast<float*>(w.workspace);
        q.K = static_cast<int>(thin ? w.K - tail : w.K);
        q.Kw = static_cast<int>(w.K);
        q.C = static_cast<int>(w.C);
        q.CH = (q.C + 7) / 8;
        q.KT = (q.K + 63) / 64;
        q.mode = w.data_gradient;
        q.tail = thin ? tail : 0;
        q.begin = blocks;
        const int64_t elems = static_cast<int64_t>(q.KT) * 64 * q.CH * 8 + (thin ? w.C * kThinStride : 0);
        blocks += static_cast<int>((elems + kBlock - 1) / kBlock);
        bytes += 4.0 * (9.0 * w.K * w.C + 16.0 * elems);
        if (++tab.n == kWinoMaxMulti)
            if (int rc = flush()) return rc;
    }
    return flush();
}

extern "C" int ffwm_conv3x3_winograd_forward(const void* input, const void* weight, const void* bias, void* output, void* workspace,
                                             int64_t B, int64_t C, int64_t H, int64_t W, int64_t K, int data_gradient, int act,
                                             double slope, int dtype, void* stream) {
    const char* fn = "ffwm_conv3x3_winograd_forward";
    FFWM_REQUIRE(dtype == FFWM_F32, FFWM_ERR_DTYPE, "%s: float32 only", fn);
    FFWM_REQUIRE(input && weight && output && workspace, FFWM_ERR_ARG, "%s: NULL tensor pointer", fn);
    FFWM_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && K > 0, FFWM_ERR_ARG, "%s: sizes must be positive", fn);
    FFWM_REQUIRE(act == 0 || act == 1, FFWM_ERR_ARG, "%s: act must be 0 (none) or 1 (leaky relu)", fn);
    FFWM_REQUIRE(data_gradient >= 0 && data_gradient <= 3, FFWM_ERR_ARG, "%s: data_gradient must be 0 or 1, + 2 to reuse the workspace", fn);
    const bool reuse = (data_gradient & 2) != 0;         // the workspace holds the transformed weights of an earlier, identical call
    data_gradient &= 1;
    FFWM_REQUIRE(B * C * H * W < (1LL << 29) && B * K * H * W < (1LL << 40), FFWM_ERR_SIZE, "%s: the input must stay below 2 GiB", fn);
    // a tail of 1-4 channels past a multiple of 64 goes to the thin kernel instead of costing a 64-channel tile
    const int tail = static_cast<int>(K % 64);
    // (K <= 4 altogether -- netG's 195 -> 3 output layer, base_networks.py:312 -- is the thin kernel alone)
    const bool thin = (K > 64 || K <= 4) && tail >= 1 && tail <= 4 && W % 4 == 0 && options().conv_thin_tail;
    WinoGeo g;
    g.C = static_cast<int>(C); g.H = static_cast<int>(H); g.W = static_cast<int>(W);
    g.K = static_cast<int>(thin ? K - tail : K);
    g.Kout = static_cast<int>(K);
    g.TH = (g.H + 1) / 2; g.TW = (g.W + 1) / 2;
    const int64_t T = B * g.TH * g.TW;
    FFWM_REQUIRE(T < (1LL << 30), FFWM_ERR_SIZE, "%s: too many tiles", fn);
    g.T = static_cast<int>(T);
    g.CH = (g.C + 7) / 8;
    g.KT = (g.K + 63) / 64;
    g.TT = (g.T + kWinoTiles - 1) / kWinoTiles;
    g.act = act; g.slope = static_cast<float>(slope);
    g.x_bytes = static_cast<unsigned>(B * C * H * W * 4);
    const int64_t ub = ffwm_conv3x3_winograd_workspace_bytes(K, C);
    FFWM_REQUIRE(ub < (1LL << 31), FFWM_ERR_SIZE, "%s: weight workspace beyond 2 GiB", fn);
    g.u_bytes = static_cast<unsigned>(ub);
    hipStream_t st = static_cast<hipStream_t>(stream);
    float* U = static_cast<float*>(workspace);
    if (!reuse) {
        const int64_t n = static_cast<int64_t>(g.KT) * 64 * g.CH * 8 + (thin ? C * kThinStride : 0);
        LaunchScope ls("conv_winograd_weights", st, 4.0 * (9.0 * K * C + 16.0 * n));
        hipLaunchKernelGGL(winograd_weights_kernel, dim3(static_cast<unsigned>((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, st,
                           static_cast<const float*>(weight), U, g.K, static_cast<int>(K), g.C, g.CH, g.KT, data_gradient, thin ? tail : 0);
        const int rc = check_launch(fn);
        if (rc) return rc;
    }
    if (g.KT > 0) {
        // flops = the multiplications the MFMAs actually perform (16 per tile, channel pair), not the 36 of the direct sum
        const double flops = 2.0 * 16.0 * static_cast<double>(T) * g.K * C;
        const double bytes = 4.0 * (static_cast<double>(B) * C * H * W + static_cast<double>(B) * g.K * H * W) + static_cast<double>(ub);
        const unsigned nblk = static_cast<unsigned>(g.TT) * static_cast<unsigned>(g.KT);
        // whole tile rows of one image per workgroup -> the raw-staged variant
        const bool rawv = options().conv_wino_raw && (W == 16 || W == 32 || W == 64 || W == 128) && H % 2 == 0 && g.TW <= 64 && 64 % g.TW == 0 &&
                          g.TH % (64 / g.TW) == 0 && (2 * (64 / g.TW) + 2) * g.W * 8 <= kWinoRawFloats;
        g.R = rawv ? 64 / g.TW : 0;
        g.ROWS = 2 * g.R + 2;
        int split_n = 1;
        if (rawv) {
            split_n = winograd_splits(static_cast<int>(nblk), g.CH, act, device_cus());
        }
        // (a split call -- zero-fill + atomics, few pairs -- is a launch configuration of its own: its own profiling scope)
        LaunchScope ls(split_n > 1 ? (data_gradient ? "conv_winograd_dgrad_split" : "conv_winograd_fwd_split")
                                : (data_gradient ? "conv_winograd_dgrad" : "conv_winograd_fwd"), st, bytes, flops);
        if (split_n > 1 && zero_fill(output, static_cast<size_t>(B) * K * H * W * 4, st)) return FFWM_ERR_LAUNCH;
        if (rawv) {
            auto kern = winograd_conv_raw_kernel<0>;
            switch (options().ablate) {
                case 1: kern = winograd_conv_raw_kernel<1>; break;
                case 2: kern = winograd_conv_raw_kernel<2>; break;
                case 3: kern = winograd_conv_raw_kernel<3>; break;
                case 7: kern = winograd_conv_raw_kernel<7>; break;
                case 16: kern = winograd_conv_raw_kernel<16>; break;
                case 32: kern = winograd_conv_raw_kernel<32>; break;
                case 35: kern = winograd_conv_raw_kernel<35>; break;
                case 36: kern = winograd_conv_raw_kernel<36>; break;
                case 39: kern = winograd_conv_raw_kernel<39>; break;
                case 40: kern = winograd_conv_raw_kernel<40>; break;
                case 47: kern = winograd_conv_raw_kernel<47>; break;
                case 99: kern = winograd_conv_raw_kernel<35 + 64>; break;
                case 163: kern = winograd_conv_raw_kernel<35 + 128>; break;
                case 291: kern = winograd_conv_raw_kernel<35 + 256>; break;
                case 547: kern = winograd_conv_raw_kernel<35 + 512>; break;
                case 227: kern = winograd_conv_raw_kernel<35 + 64 + 128>; break;
                default: break;
            }
            if (split_n > 1) kern = winograd_conv_raw_kernel<0, true>;
            const int cus = device_cus();
            const unsigned units = nblk * static_cast<unsigned>(split_n);
            const unsigned pgrid = units < static_cast<unsigned>(cus) ? units : static_cast<unsigned>(cus);     // persistent: one workgroup per CU
            if (split_n == 1 && options().conv_wino_ws) {
                auto wsk = winograd_conv_ws_kernel<0>;
                if (options().ablate == 1) wsk = winograd_conv_ws_kernel<1>;
                if (options().ablate == 2) wsk = winograd_conv_ws_kernel<2>;
                if (options().ablate == 4) wsk = winograd_conv_ws_kernel<4>;
                if (options().ablate == 8) wsk = winograd_conv_ws_kernel<8>;
                if (options().ablate == 16) wsk = winograd_conv_ws_kernel<16>;
                if (options().ablate == 32) wsk = winograd_conv_ws_kernel<32>;
                if (options().ablate == 28) wsk = winograd_conv_ws_kernel<28>;
                allow_large_lds(reinterpret_cast<const void*>(wsk));
                hipLaunchKernelGGL(wsk, dim3(pgrid), dim3(kWsThreads), 4 * kWinoChunk * 4 + 2 * kWinoRawFloats * 4, st,
                                   static_cast<const float*>(input), U, static_cast<const float*>(bias), static_cast<float*>(output), g, options().xcd_remap);
                const int rc = check_launch(fn);
                if (rc || !thin) return rc;
            } else {
            allow_large_lds(reinterpret_cast<const void*>(kern));
            hipLaunchKernelGGL(kern, dim3(pgrid), dim3(kWinoThreads), 4 * kWinoChunk * 4 + 2 * kWinoRawFloats * 4, st, static_cast<const float*>(input), U,
                               static_cast<const float*>(bias), static_cast<float*>(output), g, options().xcd_remap, split_n, g.CH / split_n);
            const int rc = check_launch(fn);
            if (rc || !thin) return rc;
            }
        } else {
        auto kern = winograd_conv_kernel<0>;
        switch (options().ablate) {
            case 1: kern = winograd_conv_kernel<1>; break;
            case 2: kern = winograd_conv_kernel<2>; break;
            case 3: kern = winograd_conv_kernel<3>; break;
            case 4: kern = winograd_conv_kernel<4>; break;
            case 7: kern = winograd_conv_kernel<7>; break;
            case 15: kern = winograd_conv_kernel<15>; break;
            default: break;
        }
        allow_large_lds(reinterpret_cast<const void*>(kern));
        hipLaunchKernelGGL(kern, dim3(nblk), dim3(kWinoThreads), 4 * kWinoChunk * 4, st, static_cast<const float*>(input), U,
                           static_cast<const float*>(bias), static_cast<float*>(output), g, options().xcd_remap);
        const int rc = check_launch(fn);
        if (rc || !thin) return rc;
        }
    }
    ThinGeo t;
    t.C = g.C; t.H = g.H; t.W = g.W; t.Kout = g.Kout; t.k_off = g.K;
    t.strips = static_cast<int>(B * H * W / 4);
    t.act = act; t.slope = static_cast<float>(slope);
    t.x_bytes = g.x_bytes;
    LaunchScope ls("conv3x3_thin_tail", st, 4.0 * (static_cast<double>(B) * C * H * W + static_cast<double>(B) * tail * H * W));
    const dim3 grid(static_cast<unsigned>((t.strips + 63) / 64)), block(64 * kThinWaves);
    const float* xin = static_cast<const float*>(input);
    const float* wt = U + static_cast<size_t>(g.KT) * g.CH * kWinoChunk;
    const float* bs = static_cast<const float*>(bias);
    float* o = static_cast<float*>(output);
    switch (tail) {
        case 1: hipLaunchKernelGGL(conv3x3_thin_kernel<1>, grid, block, 0, st, xin, wt, bs, o, t); break;
        case 2: hipLaunchKernelGGL(conv3x3_thin_kernel<2>, grid, block, 0, st, xin, wt, bs, o, t); break;
        case 3: hipLaunchKernelGGL(conv3x3_thin_kernel<3>, grid, block, 0, st, xin, wt, bs, o, t); break;
        default: hipLaunchKernelGGL(conv3x3_thin_kernel<4>, grid, block, 0, st, xin, wt, bs, o, t); break;
    }
    return check_launch(fn);
}
