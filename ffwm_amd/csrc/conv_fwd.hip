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
//   * MODE 2  d(input) of Conv2d(C, K, 3, 2, 1) (= ConvTranspose2d(K, C, 3, 2, 1, output_padding 1) with the same weight
//             tensor): parity classes again, with 1 or 2 taps per axis (an even output row meets r = 1 only, an odd one
//             r = 0 and r = 2); the missing taps of a class are zero weights.
//   * MODE 3  d(input) of Conv2d(C, K, 3, 1, 1): a 3x3 / stride-1 convolution of grad_output with the weight read
//             transposed and rotated in place, Wm[c, (k, r, s)] = W[k, c, 2 - r, 2 - s] -- no weight copy.
//   A workgroup (4 waves = 2 x 2 MFMA tiles) owns 64 output channels x 64 output pixels (pixels linearised over batch
//   and plane) and walks kd in chunks of 32: weights [64 x 32] and gathered activations [32 x 64] are staged in LDS
//   (global loads of chunk i+1 in flight under the 16 MFMAs per wave of chunk i), operands are ds_read_b32 at
//   conflict-free pitches.  Layers with few output pixels (the tail) are cut along kd (split-K over blockIdx.z) until the
//   chip is full.  Round 6: a slice STORES its partial tile into its own slot of a caller-provided workspace
//   ([splitk][B][K][oH * oW], nothing to zero-fill) and conv_split_reduce_kernel adds the slots in slice order, applies bias +
//   activation and writes the destination(s) -- no float atomics, so the result is bit-reproducible run to run (rounds 2-5 added
//   the slices atomically into a zero-filled output: 18 of 38 us of a tail layer were those atomics meeting on a few thousand
//   addresses).  Unsplit launches fuse the epilogue (bias + LeakyReLU / tanh, written to one or two destinations: channel
//   slices of concatenation buffers are valid) here.
#include "common.hpp"

namespace ffwm {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
#ifndef FFWM_CONV_CPC_MULT
#define FFWM_CONV_CPC_MULT 1
#endif
constexpr int kCpcMult = FFWM_CONV_CPC_MULT;      // chunk depth multiplier (experiment: 2 = 72 / 64 / 64 reduction steps per barrier pair)

struct ConvGeo {
    int C, H, W;             // input planes
    int K, Ho, Wo;           // output channels; pixel grid of one launch class (MODE 1: the input grid)
    int stride, pad;
    int Kd;                  // reduction length: C * R * S (MODE 1: C * 4)
    int N;                   // B * Ho * Wo
    int n_tiles, k_tiles;
    int kfast;               // linear workgroup index: channel tiles fastest (1) or pixel tiles fastest (0)
    int splitk, chunks;      // chunks (of CPC input channels) per split
    long long out_bs;        // output batch stride in elements (split launch: K * oH * oW, the workspace slot's)
    long long out2_bs;       // second destination's batch stride (unsplit launches with out2 != NULL)
    long long slot;          // split launch: elements per workspace slot = B * K * oH * oW
    int oH, oW;              // output plane (MODE 1: 2H x 2W)
    int act;
    float slope;
    unsigned x_bytes, w_bytes;   // sizes of the input / weight tensors (buffer resources; both < 2^31)
};

// A chunk of the reduction = CPC whole input channels x all R*S taps, so that the (channel-in-chunk, r, s) of every staged
// element is FIXED for the life of the kernel: all address arithmetic happens once, before the loop, and a chunk costs
// one add per load.  (The first version decoded kd -> (c, r, s) per element and chunk on the scalar unit: 270 SALU
// instructions per chunk and wave, more issue time than the 16 MFMAs they fed -- SQ counters: SALU 7.6 M vs MFMA 0.44 M.)
//
// TM x TN = the workgroup's tile of (output channels) x (output pixels), 64 or 128 each; the 2 x 2 waves own
// (TM/2) x (TN/2) sub-tiles = 1, 2 or 4 MFMA accumulators each.  A 128 x 128 tile issues 64 MFMAs per wave and chunk
// for the same two barriers, and one LDS operand read per MFMA instead of two.
template <int MODE, int R, int S, int TM, int TN>
__global__ void __launch_bounds__(kBlock)
conv_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias, float* __restrict__ out,
                float* __restrict__ out2, const ConvGeo g) {
    constexpr bool PARITY = MODE == 1 || MODE == 2;          // four output-parity classes over the input grid
    constexpr int RS = PARITY ? 4 : R * S;
    constexpr int CPC = (RS == 9 ? 4 : (RS == 16 ? 2 : 8)) * kCpcMult;       // channels per chunk
    constexpr int KC = CPC * RS;                                  // 36 / 32 / 32 reduction steps per chunk (even: MFMA k = 2)
    constexpr int AP = KC + 1;                                    // As pitch (odd: conflict-free operand reads)
    constexpr int NEA = TM * KC / kBlock, NEB = KC * TN / kBlock; // staged elements per thread
    constexpr int WMT = TM / 64, WNT = TN / 64;                   // 32 x 32 accumulators per wave: WMT x WNT
    constexpr int KSTEP = kBlock / TN;                            // B rows covered by one pass of the block: 4 or 2
    __shared__ float As[TM * AP];
    __shared__ float Bs[KC * TN];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    // Round 5: the (pixel tile, channel tile) pair comes from ONE linear index with the channel tiles fastest, XCD-remapped: the k_tiles
    // workgroups that gather the SAME activations are dispatched together on one XCD and meet in its L2.  (With blockIdx.x = pixel tile,
    // blockIdx.y = channel tile they ran a whole sweep of the pixels apart and every sweep re-fetched the input: PMC traffic 3.4-3.9 x
    // the algorithmic bytes)  g.kfast == 0: the old mapping.
    int n_tile, k_tile;
    if (g.kfast) {
        const unsigned t = xcd_remap(blockIdx.x, gridDim.x, 1);
        k_tile = static_cast<int>(t % static_cast<unsigned>(g.k_tiles));
        n_tile = static_cast<int>(t / static_cast<unsigned>(g.k_tiles));
    } else {
        n_tile = blockIdx.x % g.n_tiles;
        k_tile = blockIdx.x / g.n_tiles;
    }
    const int n0 = n_tile * TN, k0 = k_tile * TM;
    // The bias of this wave's TM / 2 output channels, one per lane, loaded HERE: the epilogue fetches a register's channel from lane
    // a * 32 + jr + 4 half by ds_bpermute.  (A `bias[k]` load per register in the epilogue made each of its 16-64 stores wait for that
    // load's round trip AND for the previous store's acknowledgement -- vmcnt counts both, in order.)
    float bvec = 0.f;
    if (bias != nullptr && g.splitk == 1) {
        const int kb_ = k0 + (static_cast<int>(threadIdx.x) >> 7) * (TM / 2) + (static_cast<int>(threadIdx.x) & 63);
        if ((static_cast<int>(threadIdx.x) & 63) < TM / 2 && kb_ < g.K) bvec = bias[kb_];
    }
    int zz = blockIdx.z;
    const int split = zz % g.splitk;
    zz /= g.splitk;
    const int py = PARITY ? (zz >> 1) : 0, px = PARITY ? (zz & 1) : 0;
    constexpr unsigned kOobOff = 0xFFFFFFF0u;

    // branch-free loads: buffer resources over the whole tensors + 32-bit byte offsets; an invalid element gets an offset
    // beyond the resource and the hardware returns 0
    const rsrc_t rx = make_rsrc(x, g.x_bytes);
    const rsrc_t rw = make_rsrc(w, g.w_bytes);

    // ---- B: the pixel this thread gathers for is fixed (nn = tid % TN), and so is each element's tap: kk = tid / TN + KSTEP i
    const int plane_o = g.Ho * g.Wo;
    const int HW = g.H * g.W;
    const int nn = threadIdx.x % TN, kb = threadIdx.x / TN;       // kb is wave-uniform
    const int gn = n0 + nn;
    const bool gvalid = gn < g.N;
    const int gb = gvalid ? gn / plane_o : 0;
    const int gp = gvalid ? gn - gb * plane_o : 0;
    const int goy = gp / g.Wo, gox = gp - goy * g.Wo;
    const int iy0 = PARITY ? goy + py : goy * g.stride - g.pad;     // MODE 1 / 2: iy = y' + py - a
    const int ix0 = PARITY ? gox + px : gox * g.stride - g.pad;
    unsigned boff[NEB];                  // byte offset of the element for channel chunk 0, or kOobOff
#pragma unroll
    for (int i = 0; i < NEB; ++i) {
        const int kk = kb + KSTEP * i;
        const int cc = kk / RS, rs = kk - cc * RS;
        int iy, ix;
        bool tap = true;
        if constexpr (!PARITY) {
            iy = iy0 + rs / S; ix = ix0 + rs % S;
        } else {
            iy = iy0 - (rs >> 1); ix = ix0 - (rs & 1);
            if constexpr (MODE == 2) tap = (py == 1 || (rs >> 1) == 0) && (px == 1 || (rs & 1) == 0);   // an even row / column has one tap
        }
        const bool ok = gvalid && tap && iy >= 0 && iy < g.H && ix >= 0 && ix < g.W;
        boff[i] = ok ? ((static_cast<unsigned>(gb) * g.C + cc) * static_cast<unsigned>(HW) + static_cast<unsigned>(iy * g.W + ix)) * 4u : kOobOff;
    }
    // ---- A: element e = tid + 256 i of the [TM x KC] weight chunk: m = e / KC, kk = e % KC -- fixed as well
    unsigned aoff[NEA];                  // byte offset for chunk 0, or kOobOff
#pragma unroll
    for (int i = 0; i < NEA; ++i) {
        const int e = threadIdx.x + kBlock * i;
        const int m = e / KC, kk = e - m * KC;
        const int cc = kk / RS, rs = kk - cc * RS;
        if (k0 + m >= g.K) aoff[i] = kOobOff;
        else if (MODE == 0) aoff[i] = (static_cast<unsigned>(k0 + m) * static_cast<unsigned>(g.Kd) + kk) * 4u;
        else if (MODE == 1)        // ConvTranspose2d weight [C][K][4][4]: tap (a, b) = (rs >> 1, rs & 1) -> ky = 1 - py + 2a, kx = 1 - px + 2b
            aoff[i] = ((static_cast<unsigned>(cc) * g.K + (k0 + m)) * 16u + (1 - py + 2 * (rs >> 1)) * 4 + (1 - px + 2 * (rs & 1))) * 4u;
        else if (MODE == 2) {      // weight [C_in][K_out][3][3] (the Conv2d's own [K][C][3][3]): even row: r = 1; odd row: a = 0 -> r = 0, a = 1 -> r = 2
            const int a = rs >> 1, bq = rs & 1;
            const bool tap = (py == 1 || a == 0) && (px == 1 || bq == 0);
            const int r = py == 0 ? 1 : 2 * a, sx = px == 0 ? 1 : 2 * bq;
            aoff[i] = tap ? ((static_cast<unsigned>(cc) * g.K + (k0 + m)) * 9u + r * 3 + sx) * 4u : kOobOff;
        } else                     // MODE 3: Wm[c_out = k0 + m][(k_in = cc, r, s)] = W[cc][k0 + m][2 - r][2 - s]
            aoff[i] = ((static_cast<unsigned>(cc) * g.K + (k0 + m)) * 9u + (8 - rs)) * 4u;
    }
    const unsigned a_step = (MODE == 0 ? static_cast<unsigned>(KC)
                                       : static_cast<unsigned>(CPC) * g.K * (MODE == 1 ? 16u : 9u)) * 4u;   // bytes per chunk
    const unsigned b_step = static_cast<unsigned>(CPC) * static_cast<unsigned>(HW) * 4u;

    const int chunk_begin = split * g.chunks;
    const int chunk_end = min((g.C + CPC - 1) / CPC, chunk_begin + g.chunks);

    auto fetch = [&](int ch, float (&ra)[NEA], float (&rb)[NEB]) {
        const int c0 = ch * CPC;
        const unsigned ao = static_cast<unsigned>(ch) * a_step, bo = static_cast<unsigned>(ch) * b_step;
        const bool tail = c0 + CPC > g.C;                                  // only the last chunk can run past C
#pragma unroll
        for (int i = 0; i < NEA; ++i) {
            const int kk = (threadIdx.x + kBlock * i) % KC;
            const bool cin = !tail || c0 + kk / RS < g.C;
            ra[i] = buf_ld<float>(rw, (cin && aoff[i] != kOobOff) ? aoff[i] + ao : kOobOff);
        }
#pragma unroll
        for (int i = 0; i < NEB; ++i) {
            const bool cin = !tail || c0 + (kb + KSTEP * i) / RS < g.C;
            rb[i] = buf_ld<float>(rx, (cin && boff[i] != kOobOff) ? boff[i] + bo : kOobOff);
        }
    };
    auto commit = [&](const float (&ra)[NEA], const float (&rb)[NEB]) {
#pragma unroll
        for (int i = 0; i < NEA; ++i) {
            const int e = threadIdx.x + kBlock * i;
            As[(e / KC) * AP + e % KC] = ra[i];
        }
#pragma unroll
        for (int i = 0; i < NEB; ++i) Bs[(kb + KSTEP * i) * TN + nn] = rb[i];
    };

    f32x16 acc[WMT][WNT];
#pragma unroll
    for (int a = 0; a < WMT; ++a)
#pragma unroll
        for (int b = 0; b < WNT; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // software pipeline, prefetch distance 2: the global loads of chunks i+1 and i+2 are in flight while the MFMAs of chunk i run
    float ra0[NEA], rb0[NEB], ra1[NEA], rb1[NEB];
    auto compute = [&]() {
        const float* ap = As + (wm * (TM / 2) + l31) * AP + half;
        const float* bp = Bs + half * TN + wn * (TN / 2) + l31;
#pragma unroll
        for (int q = 0; q < KC / 2; ++q) {
            float av[WMT], bv[WNT];
#pragma unroll
            for (int a = 0; a < WMT; ++a) av[a] = ap[a * 32 * AP + 2 * q];
#pragma unroll
            for (int b = 0; b < WNT; ++b) bv[b] = bp[2 * q * TN + b * 32];
#pragma unroll
            for (int a = 0; a < WMT; ++a)
#pragma unroll
                for (int b = 0; b < WNT; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[a], bv[b], acc[a][b], 0, 0, 0);
        }
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

    // ---- epilogue.  C/D layout: col = lane & 31 (pixel), row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5) (channel).
    // The bias fetch (ds_bpermute reads the SOURCE lane's register only while that lane is enabled) runs in uniform control flow, in
    // front of the per-lane range tests: round 5 had it behind `if (n >= N) continue; if (k >= K) continue;`, so a ragged last pixel
    // block (B * Ho * Wo % 32 != 0) or a ragged channel tile lost the bias of every channel whose source lane was masked off.
    const size_t oplane = static_cast<size_t>(g.oH) * g.oW;
    float* ob[WNT];
    float* ob2[WNT];
    bool nok[WNT];
#pragma unroll
    for (int bq = 0; bq < WNT; ++bq) {
        const int n = n0 + wn * (TN / 2) + bq * 32 + l31;
        nok[bq] = n < g.N;
        const int b = nok[bq] ? n / plane_o : 0;
        const int p = nok[bq] ? n - b * plane_o : 0;
        size_t pix;
        if constexpr (!PARITY) {
            pix = static_cast<size_t>(p);
        } else {
            const int oy = p / g.Wo, ox = p - oy * g.Wo;
            pix = static_cast<size_t>(2 * oy + py) * g.oW + (2 * ox + px);
        }
        ob[bq] = out + (g.splitk > 1 ? static_cast<size_t>(split) * g.slot : 0) + static_cast<size_t>(b) * g.out_bs + pix;
        ob2[bq] = out2 ? out2 + static_cast<size_t>(b) * g.out2_bs + pix : nullptr;
    }
    const bool fused = g.splitk == 1;
#pragma unroll
    for (int a = 0; a < WMT; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int kr = a * 32 + (r & 3) + 8 * (r >> 2);
            const int k = k0 + wm * (TM / 2) + kr + 4 * half;
            const float bk = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(16 * half + 4 * kr, __builtin_bit_cast(int, bvec)));   // all lanes on
#pragma unroll
            for (int bq = 0; bq < WNT; ++bq) {
                if (!nok[bq] || k >= g.K) continue;
                float v = acc[a][bq][r];
                if (fused) {
                    v += bk;
                    if (g.act == 1) v = v > 0.f ? v : v * g.slope;
                    else if (g.act == 2) v = tanhf(v);
                    if (ob2[bq]) ob2[bq][static_cast<size_t>(k) * oplane] = v;
                }
                ob[bq][static_cast<size_t>(k) * oplane] = v;        // split launch: the slice's own workspace slot, plain store
            }
        }
}

// The second half of a split launch: y[b, k, p] = act(bias[k] + slot_0 + slot_1 + ... ) in slice order (a fixed-order float sum:
// bit-reproducible), written to one or two destinations with their own batch strides.  One thread per 4 consecutive pixels when
// the plane allows, else per pixel.
template <int VEC>
__global__ void __launch_bounds__(kBlock)
conv_split_reduce_kernel(const float* __restrict__ ws, const float* __restrict__ bias, float* __restrict__ y, float* __restrict__ y2,
                         int64_t total, int splitk, int64_t slot, int K, int HW, int64_t ybs, int64_t y2bs, int act, float slope) {
    const int hwv = HW / VEC;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < total; i += static_cast<int64_t>(gridDim.x) * kBlock) {
        const int64_t plane = i / hwv;
        const int p = static_cast<int>(i - plane * hwv) * VEC;
        const int k = static_cast<int>(plane % K);
        const int64_t b = plane / K;
        const float* src = ws + plane * HW + p;
        float v[VEC];
        if constexpr (VEC == 4) {
            // eight slots requested before the first is added (the loads are independent, the sum keeps its slot order)
            float4 s4 = {0.f, 0.f, 0.f, 0.f};
            int s = 0;
            for (; s + 8 <= splitk; s += 8) {
                float4 t[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) t[j] = *reinterpret_cast<const float4*>(src + (s + j) * slot);
#pragma unroll
                for (int j = 0; j < 8; ++j) { s4.x += t[j].x; s4.y += t[j].y; s4.z += t[j].z; s4.w += t[j].w; }
            }
            for (; s < splitk; ++s) {
                const float4 t = *reinterpret_cast<const float4*>(src + s * slot);
                s4.x += t.x; s4.y += t.y; s4.z += t.z; s4.w += t.w;
            }
            v[0] = s4.x; v[1] = s4.y; v[2] = s4.z; v[3] = s4.w;
        } else {
            float a = src[0];
            for (int s = 1; s < splitk; ++s) a += src[s * slot];
            v[0] = a;
        }
        const float bk = bias ? bias[k] : 0.f;
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            float t = v[j] + bk;
            if (act == 1) t = t > 0.f ? t : t * slope;
            else if (act == 2) t = tanhf(t);
            v[j] = t;
        }
        const int64_t o = static_cast<int64_t>(k) * HW + p;
        if constexpr (VEC == 4) {
            const float4 o4 = {v[0], v[1], v[2], v[3]};
            *reinterpret_cast<float4*>(y + b * ybs + o) = o4;
            if (y2) *reinterpret_cast<float4*>(y2 + b * y2bs + o) = o4;
        } else {
            y[b * ybs + o] = v[0];
            if (y2) y2[b * y2bs + o] = v[0];
        }
    }
}

}  // namespace
}  // namespace ffwm

using namespace ffwm;

namespace {

unsigned ew_grid(int64_t n) {
    const int64_t blocks = (n + kBlock - 1) / kBlock;
    return static_cast<unsigned>(blocks > 256 * 32 ? 256 * 32 : (blocks < 1 ? 1 : blocks));
}

// Geometry, tile shape and reduction split of one call -- shared by the workspace query and the launch so that they cannot disagree.
struct ConvPlan {
    ConvGeo g;
    int tm, tn, classes, mode, kernel;
    int64_t out_elems;       // B * K * oH * oW
};

int conv_plan(const char* fn, int64_t B, int64_t C, int64_t H, int64_t W, int64_t K, int kernel, int stride, int pad, int mode,
              bool may_split, ConvPlan* pl) {
    FFWM_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && K > 0, FFWM_ERR_ARG, "%s: bad sizes", fn);
    ConvGeo& g = pl->g;
    g.C = static_cast<int>(C); g.H = static_cast<int>(H); g.W = static_cast<int>(W); g.K = static_cast<int>(K);
    g.stride = stride; g.pad = pad;
    int classes = 1;
    int64_t w_elems;
    // mode: 0 conv, 1 ConvTranspose2d(4, 2, 1), 2 d(input) of Conv2d(3, 2, 1), 3 d(input) of Conv2d(3, 1, 1)
    FFWM_REQUIRE(mode >= 0 && mode <= 3, FFWM_ERR_ARG, "%s: unknown mode %d", fn, mode);
    if (mode == 1 || mode == 2) {
        FFWM_REQUIRE((mode == 1 ? kernel == 4 : kernel == 3) && stride == 2 && pad == 1, FFWM_ERR_ARG,
                     "%s: mode %d is a %s / stride 2 / pad 1 operator", fn, mode, mode == 1 ? "4x4" : "3x3");
        g.Ho = g.H; g.Wo = g.W; g.oH = 2 * g.H; g.oW = 2 * g.W; g.Kd = g.C * 4;
        classes = 4;
        w_elems = C * K * kernel * kernel;
    } else {
        FFWM_REQUIRE((kernel == 3 || kernel == 4) && (stride == 1 || stride == 2) && pad >= 0 && pad < kernel, FFWM_ERR_ARG,
                     "%s: 3x3 / 4x4 kernels with stride 1 / 2 only (got %d, %d, %d)", fn, kernel, stride, pad);
        FFWM_REQUIRE(mode == 0 || (kernel == 3 && stride == 1 && pad == 1), FFWM_ERR_ARG, "%s: mode 3 is the data gradient of a 3x3 / stride 1 / pad 1 convolution", fn);
        g.Ho = (g.H + 2 * pad - kernel) / stride + 1;
        g.Wo = (g.W + 2 * pad - kernel) / stride + 1;
        FFWM_REQUIRE(g.Ho > 0 && g.Wo > 0, FFWM_ERR_ARG, "%s: empty output", fn);
        g.oH = g.Ho; g.oW = g.Wo; g.Kd = g.C * kernel * kernel;
        w_elems = C * K * kernel * kernel;
    }
    FFWM_REQUIRE(B * g.Ho * g.Wo < (1LL << 31) && B * C * H * W < (1LL << 29) && w_elems < (1LL << 29),
                 FFWM_ERR_SIZE, "%s: tensor too large (input and weight must stay below 2 GiB: 32-bit buffer offsets)", fn);
    g.x_bytes = static_cast<unsigned>(B * C * H * W * 4);
    g.w_bytes = static_cast<unsigned>(w_elems * 4);
    g.N = static_cast<int>(B * g.Ho * g.Wo);
    // tile (conv_tile_variant: 0 auto, 1 = 64 x 64, 2 = 128 x 64, 3 = 64 x 128, 4 = 128 x 128).  Measured per layer at batch 32
    // (tools/conv_layers.py): 64 x 128 pixels wins where >= 512 workgroups remain and on the transposed convolutions with >= 2048
    // pixels per parity class (the weight chunk is shared by twice the pixels); 128 output channels per workgroup never does
    // (128 x 128: 256 + 144 registers, one wave per SIMD)
    int tm = 64, tn = 64;
    {
        const int v = options().conv_tile_variant;
        auto wgs = [&](int a, int b) { return static_cast<int64_t>((g.K + a - 1) / a) * ((g.N + b - 1) / b) * classes; };
        if (v == 2 || v == 4) tm = 128;
        if (v == 3 || v == 4 || (v == 0 && (wgs(64, 128) >= 512 || (mode == 1 && g.N >= 2048)))) tn = 128;
    }
    g.n_tiles = (g.N + tn - 1) / tn;
    g.k_tiles = (g.K + tm - 1) / tm;
    const int cpc = ((mode == 1 || mode == 2) ? 8 : (kernel == 3 ? 4 : 2)) * kCpcMult;          // channels per chunk (conv_fwd_kernel's CPC)
    const int chunks_total = (g.C + cpc - 1) / cpc;
    const int64_t tiles = static_cast<int64_t>(g.n_tiles) * g.k_tiles * classes;
    int splitk = 1;
    if (may_split && tiles < 256) {
        // 768 workgroups; 384 for the transposed convolutions and for <= 32 output pixels (profiles/r05_conv_fwd_split_target.txt, measured
        // with the slices meeting by atomics; option conv_fwd_split_target overrides)
        // (round 6, slices meeting in workspace slots -- gpurun_out/conv_layers_split_r06.txt, per layer incl. the reduce pass: 256 for <= 96 output
        // pixels per class (conv5 / conv5_1 / conv6 / conv6_1 / deconv5 / deconv4: -1.5 ... -7 us each))
        const int by_shape = g.N <= 96 ? 256 : (mode == 1 ? 384 : 768);
        const int target = options().conv_fwd_split_target > 0 ? options().conv_fwd_split_target : by_shape;
        splitk = static_cast<int>((target + tiles - 1) / tiles);
        if (splitk > chunks_total) splitk = chunks_total;
        if (splitk < 1) splitk = 1;
    }
    g.chunks = (chunks_total + splitk - 1) / splitk;
    g.splitk = (chunks_total + g.chunks - 1) / g.chunks;
    pl->tm = tm; pl->tn = tn; pl->classes = classes; pl->mode = mode; pl->kernel = kernel;
    pl->out_elems = B * K * g.oH * g.oW;
    g.slot = pl->out_elems;
    return FFWM_OK;
}

}  // namespace

// Bytes of workspace with which ffwm_conv2d_forward cuts this layer along its reduction (0: the layer fills the chip unsplit).
extern "C" int64_t ffwm_conv2d_forward_workspace(int64_t B, int64_t C, int64_t H, int64_t W, int64_t K, int kernel, int stride, int pad,
                                                 int transposed) {
    ConvPlan pl;
    if (conv_plan("ffwm_conv2d_forward_workspace", B, C, H, W, K, kernel, stride, pad, transposed, true, &pl) != FFWM_OK) return -1;
    return pl.g.splitk > 1 ? static_cast<int64_t>(pl.g.splitk) * pl.out_elems * 4 : 0;
}

extern "C" int ffwm_conv2d_forward(const void* input, const void* weight, const void* bias, void* output, void* output2, int64_t B,
                                   int64_t C, int64_t H, int64_t W, int64_t K, int kernel, int stride, int pad, int transposed,
                                   int64_t out_batch_stride, int64_t out2_batch_stride, int act, double negative_slope,
                                   void* workspace, int64_t workspace_bytes, int dtype, void* stream) {
    const char* fn = "ffwm_conv2d_forward";
    FFWM_REQUIRE(dtype == FFWM_F32, FFWM_ERR_DTYPE, "%s: float32 only", fn);
    FFWM_REQUIRE(input && weight && output, FFWM_ERR_ARG, "%s: NULL tensor pointer", fn);
    FFWM_REQUIRE(act >= 0 && act <= 2, FFWM_ERR_ARG, "%s: bad activation", fn);
    ConvPlan pl;
    int rc = conv_plan(fn, B, C, H, W, K, kernel, stride, pad, transposed, workspace != nullptr, &pl);
    if (rc != FFWM_OK) return rc;
    if (pl.g.splitk > 1 && (workspace_bytes < static_cast<int64_t>(pl.g.splitk) * pl.out_elems * 4 ||
                            reinterpret_cast<uintptr_t>(workspace) % 16 != 0)) {
        rc = conv_plan(fn, B, C, H, W, K, kernel, stride, pad, transposed, false, &pl);        // too small a workspace: one slice
        if (rc != FFWM_OK) return rc;
    }
    ConvGeo& g = pl.g;
    const int mode = pl.mode, tm = pl.tm, tn = pl.tn, classes = pl.classes;
    const int64_t oplane = static_cast<int64_t>(g.oH) * g.oW;
    FFWM_REQUIRE(out_batch_stride >= K * oplane, FFWM_ERR_ARG, "%s: output batch stride smaller than K * Ho * Wo", fn);
    FFWM_REQUIRE(!output2 || out2_batch_stride >= K * oplane, FFWM_ERR_ARG, "%s: second output's batch stride smaller than K * Ho * Wo", fn);
    const bool split = g.splitk > 1;
    g.out_bs = split ? K * oplane : out_batch_stride;
    g.out2_bs = out2_batch_stride;
    g.act = act; g.slope = static_cast<float>(negative_slope);
    hipStream_t st = static_cast<hipStream_t>(stream);
    g.kfast = options().conv_fwd_kfast;
    const dim3 grid(static_cast<unsigned>(g.n_tiles) * static_cast<unsigned>(g.k_tiles), 1u, static_cast<unsigned>(g.splitk * classes));
    const double flops = 2.0 * B * g.Ho * g.Wo * classes * static_cast<double>(g.K) * g.Kd;
    const double bytes = 4.0 * (static_cast<double>(B) * C * H * W + static_cast<double>(K) * g.Kd * classes + static_cast<double>(B) * K * g.oH * g.oW);
    static const char* const kScope[4] = {"conv_fwd_mfma", "conv_fwd_mfma_transposed", "conv_dgrad_mfma_3x3s2", "conv_dgrad_mfma_3x3s1"};
    const float* x = static_cast<const float*>(input);
    const float* wt = static_cast<const float*>(weight);
    const float* bs = static_cast<const float*>(bias);
    float* o = split ? static_cast<float*>(workspace) : static_cast<float*>(output);
    float* o2 = split ? nullptr : static_cast<float*>(output2);
    {
        LaunchScope ls(kScope[mode], st, bytes, flops);
#define FFWM_CONV_LAUNCH(M, RR, SS)                                                                                          \
    do {                                                                                                                   \
        if (tm == 128 && tn == 128) hipLaunchKernelGGL((conv_fwd_kernel<M, RR, SS, 128, 128>), grid, dim3(kBlock), 0, st, x, wt, bs, o, o2, g); \
        else if (tm == 128) hipLaunchKernelGGL((conv_fwd_kernel<M, RR, SS, 128, 64>), grid, dim3(kBlock), 0, st, x, wt, bs, o, o2, g);          \
        else if (tn == 128) hipLaunchKernelGGL((conv_fwd_kernel<M, RR, SS, 64, 128>), grid, dim3(kBlock), 0, st, x, wt, bs, o, o2, g);          \
        else hipLaunchKernelGGL((conv_fwd_kernel<M, RR, SS, 64, 64>), grid, dim3(kBlock), 0, st, x, wt, bs, o, o2, g);                          \
    } while (0)
        if (mode == 1) FFWM_CONV_LAUNCH(1, 2, 2);
        else if (mode == 2) FFWM_CONV_LAUNCH(2, 2, 2);
        else if (mode == 3) FFWM_CONV_LAUNCH(3, 3, 3);
        else if (pl.kernel == 3) FFWM_CONV_LAUNCH(0, 3, 3);
        else FFWM_CONV_LAUNCH(0, 4, 4);
#undef FFWM_CONV_LAUNCH
        rc = check_launch(fn);
        if (rc != FFWM_OK) return rc;
    }
    if (!split) return FFWM_OK;
    // fixed-order reduction of the slices + bias + activation into the destination(s)
    float* y = static_cast<float*>(output);
    float* y2 = static_cast<float*>(output2);
    const bool vec = oplane % 4 == 0 && reinterpret_cast<uintptr_t>(y) % 16 == 0 && out_batch_stride % 4 == 0 &&
                     (!y2 || (reinterpret_cast<uintptr_t>(y2) % 16 == 0 && out2_batch_stride % 4 == 0));
    const int64_t total = pl.out_elems / (vec ? 4 : 1);
    LaunchScope ls("conv_fwd_split_reduce", st, 4.0 * pl.out_elems * (g.splitk + 1.0 + (y2 ? 1 : 0)));
    if (vec)
        hipLaunchKernelGGL((conv_split_reduce_kernel<4>), dim3(ew_grid(total)), dim3(kBlock), 0, st, static_cast<const float*>(workspace), bs, y, y2,
                           total, g.splitk, pl.out_elems, static_cast<int>(K), static_cast<int>(oplane), out_batch_stride, out2_batch_stride, act, g.slope);
    else
        hipLaunchKernelGGL((conv_split_reduce_kernel<1>), dim3(ew_grid(total)), dim3(kBlock), 0, st, static_cast<const float*>(workspace), bs, y, y2,
                           total, g.splitk, pl.out_elems, static_cast<int>(K), static_cast<int>(oplane), out_batch_stride, out2_batch_stride, act, g.slope);
    return check_launch(fn);
}
