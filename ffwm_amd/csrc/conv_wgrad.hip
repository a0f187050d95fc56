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
// the pixels (a contiguous range of row steps of the linearised (image, 64-pixel column strip, row) space, so any
// number of slices balances: 9 tiles x 28 slices for the 195-channel layers); a wave owns 32 x 32 x 9 = nine
// v_mfma_f32_32x32x2_f32 accumulators (144 registers).  Per output row the strip's gO row (64 k x 64 px)
// and ONE new X row (64 c x 66 px, rolling 4-slot window: rows y-1, y, y+1 live, y+2 landing) are staged
// global -> registers -> LDS while the previous row's 288 MFMAs per wave run, so the matrix pipe never
// waits for memory: 9 taps x 32 pixel pairs x 64 cycles = 18.4 k cycles of MFMA per 33 KB staged.
// Lanes 0-31 take pixel i and lanes 32-63 pixel 32 + i of the strip in MFMA step i (the sum over
// pixels does not care about the order), so a lane's A values of four consecutive steps are ONE
// ds_read_b128, and its B values for the three horizontal taps are a 6-float window of the X row.
// The pixel slices are combined with coalesced global atomics (dW is staged through LDS so that a
// wave adds 64 consecutive floats): the caller zero-fills dW, like every other gradient of this ABI.
// The bias gradient (row sums of gO) is accumulated on the way by the waves that load each gO row anyway.
// Measured (MI355X, batch 8): 192 -> 192 @128^2 0.69 ms = 126 TFLOP/s (80 % of the 157.3 TFLOP/s fp32 MFMA peak),
// 195 -> 195 @128^2 0.82 ms (vendor library: 1.70 ms on its heuristic solver, 1.04 ms on its best one).
//
// Thin channel remainders.  195 = 128 + 64 + 3: a 64-wide tile for channels 192..194 would be 95 % padding
// (and 7 of the 16 workgroup tiles of a 195 x 195 weight would be ragged).  The full 64-tiles (192 x 192)
// run as above; the remainder runs the PACKED variant, where the N dimension of the MFMA holds
// (remainder channel, tap) pairs -- 3 channels x 9 taps = 27 of 32 columns, each lane of the B operand
// reading its own row slot and horizontal shift -- so the nine taps cost ONE accumulator and one MFMA per
// pixel pair instead of nine.  Remainder columns (input channels) use it directly; remainder rows (output
// channels) use it with the operands swapped (dW[k][c][r][s] = sum_p X[c][p] gO[k][p - (r-1, s-1)]:
// the same kernel with gO as the shifted operand and the taps flipped in the epilogue).
#include <algorithm>

#include "common.hpp"

namespace ffwm {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kWgStrip = 64;              // pixels per strip row
constexpr int kWgPG = kWgStrip + 4;       // pitch of the unshifted (A) operand rows in LDS (floats): 16 B aligned rows
constexpr int kWgPX = kWgStrip + 8;       // shifted (B) operand pitch: index 3 = x0 - 1, 4..67 = x0 .. x0 + 63, 68 = x0 + 64
constexpr int kWgTile = 64;               // channel extent of a workgroup tile on the B side (and on the A side: 2 x 2 waves)
constexpr int kWgXFloats = kWgTile * kWgPX;          // one B row slot

// A = the unshifted operand (M rows of the MFMA), S = the shifted, zero-padded operand (N columns).
// Normal orientation: A = grad_output (rows = output channels k), S = input (columns = input channels c).
struct WgradGeo {
    int At, St, H, W;                       // channel extents of the two tensors (strides)
    int a_begin, a_end, s_begin, s_end;     // the channel ranges this launch covers
    int atiles, stiles, strips, nsplit;
    int steps;                              // B * strips * H row steps, split evenly over nsplit workgroups per tile
    int out_stride_a, out_stride_s, flip;   // grad_weight offset = a * out_stride_a + s * out_stride_s + (flip ? 8 - tap : tap)
};

// PACKED = false: 2 x 2 waves, 64 A rows x 64 S channels x 9 taps (nine accumulators per wave).
// PACKED = true : 4 x 1 waves, 128 A rows x (<= 3 S channels x 9 taps packed into the 32 MFMA columns).
template <bool PACKED>
__global__ void __launch_bounds__(kBlock, 1)
conv3x3_wgrad_kernel(const float* __restrict__ Sp, const float* __restrict__ Ap, float* __restrict__ dW, WgradGeo g,
                     unsigned s_bytes, unsigned a_bytes, float* __restrict__ dbias) {
    constexpr int MT = PACKED ? 128 : 64;                   // A rows per workgroup
    constexpr int GQ = MT / 16;                             // float4 of the A row per thread
    constexpr int kAFloats = MT * kWgPG;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* const a_lds = lds;                               // [2][MT][kWgPG]
    float* const s_lds = lds + 2 * kAFloats;                // [4][64][kWgPX]
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    const int half = lane >> 5, l31 = lane & 31;
    const int wk = PACKED ? wave : wave >> 1, wc = PACKED ? 0 : wave & 1;     // the wave's 32-row / 32-column sub-tile

    // block -> (A tile, S tile, pixel slice); tiles vary fastest so that the blocks sharing a pixel slice
    // (same rows of both tensors) run together and meet in L2.  A slice is a contiguous range of ROW STEPS of
    // the linearised (image, strip, row) space, so any number of slices balances (9 tiles x 28 slices = 252
    // workgroups for the 195-channel layers); it is processed segment by segment (one image strip each).
    unsigned t = blockIdx.x;
    const int kt = t % g.atiles; t /= g.atiles;
    const int ct = t % g.stiles; t /= g.stiles;
    const int split = t;
    const int k0 = g.a_begin + kt * MT, c0 = g.s_begin + ct * kWgTile;
    const int st_begin = static_cast<int>(static_cast<int64_t>(g.steps) * split / g.nsplit);
    const int st_end = static_cast<int>(static_cast<int64_t>(g.steps) * (split + 1) / g.nsplit);
    const size_t plane = static_cast<size_t>(g.H) * g.W;
    const rsrc_t rx = make_rsrc(Sp, s_bytes), rg = make_rsrc(Ap, a_bytes);
    const unsigned row_bytes = static_cast<unsigned>(g.W) * 4;

    // ---- staging maps: thread -> GQ float4 of the A row, 4 float4 of the S row, (threads < 128) one edge cell
    // float4 index f = threadIdx.x + q * 256: channel f / 16, pixels (f % 16) * 4 .. + 3
    unsigned go_off[GQ], x_off[4];         // byte offsets of (channel, x0 + 4 * (f % 16)) in row 0 of the segment's image
    int go_dst[GQ], x_dst[4];
#pragma unroll
    for (int q = 0; q < GQ; ++q) {
        const int f = threadIdx.x + q * kBlock;
        go_dst[q] = (f >> 4) * kWgPG + (f & 15) * 4;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int f = threadIdx.x + q * kBlock;
        x_dst[q] = (f >> 4) * kWgPX + 4 + (f & 15) * 4;
    }
    // edge cells: thread e < 128 -> channel e / 2, side e & 1 (0: x0 - 1 at index 3, 1: x0 + 64 at index 68)
    const int ech = threadIdx.x >> 1, eside = threadIdx.x & 1;
    const int e_dst = ech * kWgPX + (eside ? 4 + kWgStrip : 3);
    bool eok = false;
    unsigned e_off = 0xFFFFFFF0u;
    auto set_segment = [&](int b, int x0) {
#pragma unroll
        for (int q = 0; q < GQ; ++q) {
            const int f = threadIdx.x + q * kBlock;
            const int ch = f >> 4, x4 = (f & 15) * 4;
            go_off[q] = k0 + ch < g.a_end ? static_cast<unsigned>(((static_cast<size_t>(b) * g.At + k0 + ch) * plane + x0 + x4) * 4)
                                          : 0xFFFFFFF0u;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int f = threadIdx.x + q * kBlock;
            const int ch = f >> 4, x4 = (f & 15) * 4;
            x_off[q] = c0 + ch < g.s_end ? static_cast<unsigned>(((static_cast<size_t>(b) * g.St + c0 + ch) * plane + x0 + x4) * 4)
                                         : 0xFFFFFFF0u;
        }
        const int ex = eside ? x0 + kWgStrip : x0 - 1;
        eok = threadIdx.x < 2 * kWgTile && c0 + ech < g.s_end && ex >= 0 && ex < g.W;
        e_off = eok ? static_cast<unsigned>(((static_cast<size_t>(b) * g.St + c0 + ech) * plane + ex) * 4) : 0xFFFFFFF0u;
    };

    f32x4 sg[GQ], sx[4];
    float se = 0;
    auto fetch_x = [&](int yy) {          // row yy of the shifted operand (zero outside the image)
        const bool in = yy >= 0 && yy < g.H;
        const unsigned ro = static_cast<unsigned>(in ? yy : 0) * row_bytes;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (PACKED && q > 0) {         // <= 3 channels: the first float4 round covers them
                sx[q] = f32x4{0.f, 0.f, 0.f, 0.f};
                continue;
            }
            unsigned w[4];
            buf_load_dwords<4>(rx, (in && x_off[q] != 0xFFFFFFF0u) ? x_off[q] + ro : 0xFFFFFFF0u, w);
            sx[q] = f32x4{__uint_as_float(w[0]), __uint_as_float(w[1]), __uint_as_float(w[2]), __uint_as_float(w[3])};
        }
        se = buf_ld<float>(rx, (in && eok) ? e_off + ro : 0xFFFFFFF0u);
    };
    auto commit_x = [&](int yy) {
        float* slot = s_lds + ((yy + 1) & 3) * kWgXFloats;
#pragma unroll
        for (int q = 0; q < (PACKED ? 1 : 4); ++q) *reinterpret_cast<f32x4*>(slot + x_dst[q]) = sx[q];
        if (threadIdx.x < 2 * kWgTile) slot[e_dst] = se;
    };
    auto fetch_g = [&](int yy) {
        const unsigned ro = static_cast<unsigned>(yy) * row_bytes;
#pragma unroll
        for (int q = 0; q < GQ; ++q) {
            unsigned w[4];
            buf_load_dwords<4>(rg, go_off[q] != 0xFFFFFFF0u ? go_off[q] + ro : 0xFFFFFFF0u, w);
            sg[q] = f32x4{__uint_as_float(w[0]), __uint_as_float(w[1]), __uint_as_float(w[2]), __uint_as_float(w[3])};
        }
    };
    auto commit_g = [&](int yy) {
        float* buf = a_lds + (yy & 1) * kAFloats;
#pragma unroll
        for (int q = 0; q < GQ; ++q) *reinterpret_cast<f32x4*>(buf + go_dst[q]) = sg[q];
    };

    constexpr int NACC = PACKED ? 1 : 9;
    f32x16 acc[NACC];
#pragma unroll
    for (int tp = 0; tp < NACC; ++tp)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[tp][r] = 0.f;

    const int a_base = (wk * 32 + l31) * kWgPG + half * 32;
    const int b_base = (wc * 32 + l31) * kWgPX + half * 32 + 3;      // index of x - 1 for step 0
    // bias gradient (sum of grad_output over the pixels): the waves of the first S tile that read a given A row
    // once (wc == 0) add up what they load for the MFMAs anyway
    const bool want_bias = !PACKED && dbias != nullptr && ct == 0 && wc == 0;
    float bias_sum = 0.f;
    // PACKED: column l31 = (channel l31 / 9, tap l31 % 9); columns 27..31 shadow column 0 and are never stored
    const int pcol = l31 < 27 ? l31 : 0;
    const int pch = pcol / 9, ptap = pcol - pch * 9, pr = ptap / 3, psx = ptap - pr * 3;
    const int p_base = pch * kWgPX + half * 32 + 3 + psx;
    for (int s0 = st_begin; s0 < st_end;) {
        // segment: rows [ya, yb) of one image strip
        const int img_strip = s0 / g.H;
        const int ya = s0 - img_strip * g.H;
        const int yb = min(g.H, ya + (st_end - s0));
        set_segment(img_strip / g.strips, (img_strip % g.strips) * kWgStrip);
        s0 += yb - ya;

        // ---- prologue: S rows ya - 1, ya, ya + 1 and A row ya
        for (int yy = ya - 1; yy <= ya + 1; ++yy) {
            fetch_x(yy);
            commit_x(yy);
        }
        fetch_g(ya);
        commit_g(ya);
        __syncthreads();

        for (int y = ya; y < yb; ++y) {
            const bool more = y + 1 < yb;
            if (more) {                        // lands during the MFMA loop
                fetch_x(y + 2);
                fetch_g(y + 1);
            }
            const float* ap = a_lds + (y & 1) * kAFloats + a_base;
            if constexpr (PACKED) {
                const float* bq = s_lds + ((y + pr) & 3) * kWgXFloats + p_base;       // this lane's row slot and shift
#pragma unroll 2
                for (int i = 0; i < 32; i += 4) {
                    const f32x4 a4 = *reinterpret_cast<const f32x4*>(ap + i);
                    const float b0 = bq[i], b1 = bq[i + 1], b2 = bq[i + 2], b3 = bq[i + 3];
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b0, acc[0], 0, 0, 0);
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b1, acc[0], 0, 0, 0);
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b2, acc[0], 0, 0, 0);
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b3, acc[0], 0, 0, 0);
                }
            } else {
                const float* bp[3];
#pragma unroll
                for (int r = 0; r < 3; ++r) bp[r] = s_lds + ((y + r) & 3) * kWgXFloats + b_base;     // slot of row y + r - 1
#pragma unroll 2
                for (int i = 0; i < 32; i += 4) {
                    const f32x4 a4 = *reinterpret_cast<const f32x4*>(ap + i);
                    if (want_bias) bias_sum += (a4.x + a4.y) + (a4.z + a4.w);      // the conv's bias gradient: row sums of A
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
                            for (int sx2 = 0; sx2 < 3; ++sx2)
                                acc[(r * 3 + sx2) % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bw[r][st + sx2], acc[(r * 3 + sx2) % NACC], 0, 0, 0);
                    }
                }
            }
            if (more) {
                commit_x(y + 2);
                commit_g(y + 1);
            }
            __syncthreads();
        }
    }

    if (want_bias) {
        bias_sum += __shfl_xor(bias_sum, 32, kWave);
        const int kk = k0 + wk * 32 + l31;
        if (half == 0 && kk < g.a_end) atomic_add(dbias + kk, bias_sum);
    }
    // C/D layout: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
    if constexpr (PACKED) {
        // 27 consecutive floats per (row, normal orientation); small either way: straight from the registers
        if (l31 < 27) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kk = k0 + wk * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (kk < g.a_end && g.s_begin + pch < g.s_end)
                    atomic_add(dW + static_cast<size_t>(kk) * g.out_stride_a + static_cast<size_t>(g.s_begin + pch) * g.out_stride_s +
                                   (g.flip ? 8 - ptap : ptap),
                               acc[0][r]);
            }
        }
    } else {
        // the 64 x 64 x 9 tile goes through LDS so that a wave adds 64 consecutive floats of dW
        // (dW[k][c][r][s]: for one k the 64 c x 9 taps of this tile are 576 contiguous floats)
        float* out_lds = lds;                   // [64 k][64 c * 9] floats = 147 456 B: reuses the staging memory
#pragma unroll
        for (int tp = 0; tp < NACC; ++tp)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kk = wk * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                const int cc = wc * 32 + l31;
                out_lds[kk * (kWgTile * 9) + cc * 9 + tp] = acc[tp][r];
            }
        __syncthreads();
        const int cvalid = min(kWgTile, g.s_end - c0) * 9;      // floats of a k row that this launch owns
        for (int kk = wave; kk < kWgTile; kk += kBlock / kWave) {
            if (k0 + kk >= g.a_end) break;
            float* drow = dW + static_cast<size_t>(k0 + kk) * g.out_stride_a + static_cast<size_t>(c0) * 9;
            const float* srow = out_lds + kk * (kWgTile * 9);
            for (int e = lane; e < cvalid; e += kWave) atomic_add(drow + e, srow[e]);
        }
    }
}

// Bias gradient of the few output channels the full tiles do not cover: one block per (channel, image) plane.
__global__ void __launch_bounds__(kBlock)
bias_rows_kernel(const float* __restrict__ gO, float* __restrict__ dbias, int K, int k_begin, int nk, int plane) {
    __shared__ float red[kBlock / kWave];
    const int k = k_begin + blockIdx.x % nk, b = blockIdx.x / nk;
    const float* p = gO + (static_cast<size_t>(b) * K + k) * plane;
    float s = 0.f;
    for (int i = threadIdx.x; i < plane; i += kBlock) s += p[i];
    s = wave_sum(s);
    if ((threadIdx.x & (kWave - 1)) == 0) red[threadIdx.x / kWave] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int w = 0; w < kBlock / kWave; ++w) t += red[w];
        atomic_add(dbias + k, t);
    }
}

// One launch: A rows [a_begin, a_end) x S channels [s_begin, s_end).
template <bool PACKED>
int launch_wgrad(const char* name, const float* S, const float* A, float* dW, int64_t B, int64_t St, int64_t At, int64_t H,
                 int64_t W, int64_t a_begin, int64_t a_end, int64_t s_begin, int64_t s_end, int64_t out_stride_a,
                 int64_t out_stride_s, int flip, hipStream_t st, float* dbias = nullptr) {
    if (a_begin >= a_end || s_begin >= s_end) return FFWM_OK;
    constexpr int MT = PACKED ? 128 : 64;
    WgradGeo g;
    g.At = static_cast<int>(At); g.St = static_cast<int>(St); g.H = static_cast<int>(H); g.W = static_cast<int>(W);
    g.a_begin = static_cast<int>(a_begin); g.a_end = static_cast<int>(a_end);
    g.s_begin = static_cast<int>(s_begin); g.s_end = static_cast<int>(s_end);
    g.atiles = static_cast<int>((a_end - a_begin + MT - 1) / MT);
    g.stiles = PACKED ? 1 : static_cast<int>((s_end - s_begin + kWgTile - 1) / kWgTile);
    g.strips = static_cast<int>(W / kWgStrip);
    g.steps = static_cast<int>(B * g.strips * H);
    g.out_stride_a = static_cast<int>(out_stride_a); g.out_stride_s = static_cast<int>(out_stride_s); g.flip = flip;
    // one workgroup per CU (>= 106 KB of LDS): as many pixel slices as that allows, at least 4 row steps each
    const int64_t tiles = static_cast<int64_t>(g.atiles) * g.stiles;
    int64_t nsplit = tiles >= 256 ? 1 : 256 / tiles;
    if (nsplit > g.steps / 4) nsplit = g.steps / 4 > 0 ? g.steps / 4 : 1;
    g.nsplit = static_cast<int>(nsplit);
    if (tiles * nsplit >= (1LL << 31)) {
        set_error("ffwm_conv3x3_wgrad: grid too large");
        return FFWM_ERR_SIZE;
    }
    const size_t staging = (2 * static_cast<size_t>(MT) * kWgPG + 4 * static_cast<size_t>(kWgXFloats)) * sizeof(float);
    const size_t lds = PACKED ? staging : std::max(staging, static_cast<size_t>(kWgTile) * kWgTile * 9 * sizeof(float));
    allow_large_lds(reinterpret_cast<const void*>(conv3x3_wgrad_kernel<PACKED>));
    // "bytes": operands once + result (the roofline that matters is MFMA: 2 * 9 * B H W C K flop)
    LaunchScope ls(name, st, 4.0 * (static_cast<double>(B) * H * W * ((s_end - s_begin) + (a_end - a_begin)) +
                                    9.0 * (s_end - s_begin) * (a_end - a_begin)),
                   2.0 * 9.0 * static_cast<double>(B) * H * W * (s_end - s_begin) * (a_end - a_begin));
    hipLaunchKernelGGL(conv3x3_wgrad_kernel<PACKED>, dim3(static_cast<unsigned>(tiles * nsplit)), dim3(kBlock), lds, st, S, A, dW,
                       g, static_cast<unsigned>(B * St * H * W * 4), static_cast<unsigned>(B * At * H * W * 4), dbias);
    return check_launch("ffwm_conv3x3_wgrad");
}

}  // namespace

// conv_wgrad_wino.hip
int launch_wgrad_wino(const float* X, const float* G, float* dW, float* dbias, int64_t B, int64_t C, int64_t K, int64_t H, int64_t W,
                      int64_t k_begin, int64_t k_end, int64_t c_begin, int64_t c_end, hipStream_t st);
}  // namespace ffwm

using namespace ffwm;

extern "C" int ffwm_conv3x3_wgrad_block(const void* input, const void* grad_output, void* grad_weight, void* grad_bias,
                                        int64_t B, int64_t C, int64_t K, int64_t H, int64_t W, int64_t k_begin, int64_t k_end,
                                        int64_t c_begin, int64_t c_end, int dtype, void* stream) {
    const char* fn = "ffwm_conv3x3_wgrad";
    FFWM_REQUIRE(dtype == FFWM_F32, FFWM_ERR_DTYPE, "%s: float32 only (fp32 MFMA)", fn);
    FFWM_REQUIRE(input && grad_output && grad_weight, FFWM_ERR_ARG, "%s: NULL tensor pointer", fn);
    FFWM_REQUIRE(B > 0 && C > 0 && K > 0 && H > 0 && W > 0 && W % kWgStrip == 0, FFWM_ERR_ARG,
                 "%s: need positive sizes and W a multiple of %d (B=%lld C=%lld K=%lld H=%lld W=%lld)", fn, kWgStrip,
                 (long long)B, (long long)C, (long long)K, (long long)H, (long long)W);
    FFWM_REQUIRE(0 <= k_begin && k_begin <= k_end && k_end <= K && 0 <= c_begin && c_begin <= c_end && c_end <= C, FFWM_ERR_ARG,
                 "%s: bad block [%lld, %lld) x [%lld, %lld) of a %lld x %lld weight", fn, (long long)k_begin, (long long)k_end,
                 (long long)c_begin, (long long)c_end, (long long)K, (long long)C);
    FFWM_REQUIRE(B * C * H * W * 4 < (1LL << 32) - 64 && B * K * H * W * 4 < (1LL << 32) - 64 && B * (W / kWgStrip) * H < (1LL << 31),
                 FFWM_ERR_SIZE, "%s: tensors must stay below 4 GiB (32-bit buffer offsets)", fn);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const float* X = (const float*)input;
    const float* G = (const float*)grad_output;
    float* dW = (float*)grad_weight;
    // thin remainders (<= 3 channels beyond the last full 64-tile) take the packed variant
    // (a range of at most 3 channels is all remainder: the first / last layers that read or write an RGB image)
    auto thin = [](int64_t lo, int64_t hi) {
        const int64_t n = hi - lo, r = n % kWgTile;
        return n <= 3 ? n : ((n > kWgTile && r > 0 && r <= 3) ? r : 0);
    };
    const int64_t km = k_end - thin(k_begin, k_end), cm = c_end - thin(c_begin, c_end);
    // full tiles: A = grad_output rows [k_begin, km), S = input channels [c_begin, cm)
    float* db = (float*)grad_bias;
    bool main_has_bias = db && km > k_begin && cm > c_begin;            // the full-tile launch sums its A rows on the way
    // the full tiles on the Winograd-domain kernel (conv_wgrad_wino.hip; it sums its dY rows on the way like the direct kernel) when it
    // serves the shape and -- conv_wgrad_wino 0 = auto -- every CU gets >= 16 chunks of it (1 = whenever served, 2 = never); a shape
    // it leaves comes back > 0 and takes the direct kernel
    int wino = 1;
    if (options().conv_wgrad_wino != 2 && km > k_begin && cm > c_begin) {
        wino = launch_wgrad_wino(X, G, dW, main_has_bias ? db : nullptr, B, C, K, H, W, k_begin, km, c_begin, cm, st);
        if (wino < 0) return wino;
    }
    if (wino != FFWM_OK)
        if (int rc = launch_wgrad<false>("conv3x3_wgrad", X, G, dW, B, C, K, H, W, k_begin, km, c_begin, cm, C * 9, 9, 0, st,
                                         main_has_bias ? db : nullptr))
            return rc;
    if (db) {
        const int64_t r0 = main_has_bias ? km : k_begin;                 // rows nobody summed yet
        if (r0 < k_end) {
            LaunchScope ls("conv_bias_rows", st, 4.0 * B * (k_end - r0) * H * W);
            hipLaunchKernelGGL(bias_rows_kernel, dim3(static_cast<unsigned>(B * (k_end - r0))), dim3(kBlock), 0, st, G, db, (int)K,
                               (int)r0, (int)(k_end - r0), (int)(H * W));
            if (int rc = check_launch(fn)) return rc;
        }
    }
    // remainder columns: every output channel x input channels [cm, c_end), taps packed into the MFMA columns
    if (int rc = launch_wgrad<true>("conv3x3_wgrad_packed", X, G, dW, B, C, K, H, W, k_begin, k_end, cm, c_end, C * 9, 9, 0, st)) return rc;
    // remainder rows: operands swapped (A = input channels [c_begin, cm), S = grad_output channels [km, k_end)), taps flipped
    return launch_wgrad<true>("conv3x3_wgrad_packed", G, X, dW, B, K, C, H, W, c_begin, cm, km, k_end, 9, C * 9, 1, st);
}

extern "C" int ffwm_conv3x3_wgrad(const void* input, const void* grad_output, void* grad_weight, void* grad_bias, int64_t B,
                                  int64_t C, int64_t K, int64_t H, int64_t W, int dtype, void* stream) {
    return ffwm_conv3x3_wgrad_block(input, grad_output, grad_weight, grad_bias, B, C, K, H, W, 0, K, 0, C, dtype, stream);
}
