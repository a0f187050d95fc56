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


// ------------------------------------------------------------------------------------------------ tiled variant (round 3)
// The same GEMM with the structure of the fast kernels of this library (conv_winograd.hip / conv_wgrad.hip):
//   * a workgroup (2 x 2 waves) owns a (64 WM) x (64 WN) tile of dW -- up to 128 x 128, four 32 x 32 accumulators per wave,
//     64 MFMAs per wave and chunk of 32 pixels behind ONE barrier (the 64 x 64 kernel above: 16 MFMAs behind two);
//   * pixels are linearised over (image, row, column): a chunk is always 32 real pixels, also for FlowNet's 2 x 2 ... 8 x 8
//     planes (the kernel above pads every image's plane to a multiple of 32);
//   * both operands sit in LDS as [row or column][pixel] at a pitch of 36 floats: the global loads land in their natural
//     order (A rows are contiguous along the pixels -> one dwordx4 + ds_write_b128 per 4 pixels; the gathered window has
//     the lanes along the pixels -> conflict-free ds_write_b32), and ONE ds_read_b128 per operand fragment feeds FOUR MFMA
//     steps: lanes 0-31 hold pixels 8g .. 8g+3, lanes 32-63 pixels 8g+4 .. 8g+7 of a row, MFMA step j multiplies pixel
//     pair (8g + j, 8g + 4 + j) -- any pairing is a valid order of the pixel sum as long as both operands use it.  At
//     pitch 36 the sixteen rows a ds_read_b128 group touches fall on sixteen distinct 4-bank groups: no conflicts;
//   * two LDS buffers: the chunk fetched one iteration ago is written to the other buffer before the MFMAs of the current
//     one start, its global loads had a whole chunk of MFMAs to land;
//   * the (c, r, s) of a thread's gathered columns is fixed for the life of the kernel, the border test of an element is
//     two bit tests on per-pixel row / column masks;
//   * bias gradient (grad_bias[k] = sum over the pixels of A[k]) from the A rows the n-tile-0 workgroups stage anyway;
//   * one z-slice: plain stores (no zero-fill needed); several: atomics into the buffer the host zero-filled.
struct Wg2Geo {
    int C, H, W, K, Ho, Wo, stride, pad;
    int N, P, HW;
    int total_px;            // B * P
    int chunks_total;        // ceil(total_px / 32)
    int chunks;              // chunks per z-slice
    int nz;
    unsigned a_bytes, x_bytes;
    FastDiv divP, divWo;     // by P (pixels of an output plane) and by Wo
};

constexpr int kW2PC = 32;        // pixels per chunk
constexpr int kW2P = 36;         // LDS pitch in floats (16-byte aligned rows, conflict-free b128 reads)

template <int R, int S, int WM, int WN>
__global__ void __launch_bounds__(kBlock, 2)
conv_wgrad_tile_kernel(const float* __restrict__ a, const float* __restrict__ x, float* __restrict__ dw, float* __restrict__ gbias,
                       const Wg2Geo g) {
    constexpr int RS = R * S;
    constexpr int TM = 64 * WM, TN = 64 * WN;
    constexpr int NQA = TM * 8 / kBlock;           // A quads (4 pixels) per thread: 2 or 4
    constexpr int NXB = TN / 8;                    // gathered elements per thread: 8 or 16
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float* lds = reinterpret_cast<float*>(smem_raw);
    // buffer p: As at lds + p * (TM + TN) * kW2P, Xs right behind it
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int n0 = blockIdx.x * TN, k0 = blockIdx.y * TM;
    constexpr unsigned kOobOff = 0xFFFFFFF0u;
    const rsrc_t ra_ = make_rsrc(a, g.a_bytes);
    const rsrc_t rx_ = make_rsrc(x, g.x_bytes);

    // ---- A: quad q8 of row (t >> 3) + 32 i
    const int q8 = threadIdx.x & 7, arow0 = threadIdx.x >> 3;
    unsigned arow_off[NQA];              // (k0 + row) * P elements, or OOB
#pragma unroll
    for (int i = 0; i < NQA; ++i) {
        const int k = k0 + arow0 + 32 * i;
        arow_off[i] = k < g.K ? static_cast<unsigned>(k) * static_cast<unsigned>(g.P) : kOobOff;
    }
    // ---- X: pixel kk of the chunk, columns (t >> 5) + 8 i
    const int kk = threadIdx.x & 31, xcol0 = threadIdx.x >> 5;
    int coloff[NXB];                     // c * HW + r * W + s of column n0 + j (elements)
    unsigned colsh[NXB];                 // 8 r + s: the column's bit in the per-pixel tap mask (31: no such column)
#pragma unroll
    for (int i = 0; i < NXB; ++i) {
        const int n = n0 + xcol0 + 8 * i;
        const bool cin = n < g.N;
        const int nc = cin ? n : 0;
        const int c = nc / RS, rs = nc - c * RS;
        const int r = rs / S, sx = rs - r * S;
        coloff[i] = c * g.HW + r * g.W + sx;
        colsh[i] = cin ? static_cast<unsigned>(8 * r + sx) : 31u;
    }

    const int ch_begin = blockIdx.z * g.chunks;
    const int ch_end = min(g.chunks_total, ch_begin + g.chunks);
    const bool want_bias = gbias != nullptr && blockIdx.x == 0;

    f32x4 va[NQA];
    float vx[NXB];
    float bsum[NQA];
#pragma unroll
    for (int i = 0; i < NQA; ++i) bsum[i] = 0.f;

    // The addresses of a chunk are formed in two steps so that the loads can be issued in slices between the MFMA groups of the
    // previous chunk: prep() decodes the chunk's pixels (two divisions per thread), issue_a / issue_x turn them into loads.
    // Everything is integer arithmetic on masks -- no branches (an `ok ? offset : OOB` per element compiled to an exec-mask
    // branch per load, which serialised the whole fetch in front of the MFMAs).  Round 5 (late): the step was bound by instruction
    // ISSUE (a 64 x 64 tile carried ~8 vector instructions of address arithmetic per MFMA and wave: two generic divisions per
    // chunk, two shifts + and + negate + and + or + multiply per gathered element): the divisions are multiply-high by host-made
    // constants, and a pixel's nine / sixteen taps share ONE inverted validity mask from which an element takes its bit
    // sign-extended (v_bfe_i32: 0 or all-ones) and ORs it onto its byte offset -- all-ones is out of range, the load returns 0
    // (tools/ubench/buf_soffset.hip, case G).  Three instructions per element.
    unsigned a_base, a_pin;              // A: element offset of the quad in row 0; all-ones when the quad exists
    int x_base;                          // X: element offset of tap (0, 0) of this lane's pixel
    int x_inv;                           // bit 8 r + s CLEAR: tap (r, s) of this lane's pixel lies inside the image (bit 31 always set)
    auto prep = [&](int ch) {
        {
            const unsigned gq = static_cast<unsigned>(ch) * kW2PC + q8 * 4;
            const unsigned b = fast_div(gq, g.divP);
            const unsigned p = gq - b * static_cast<unsigned>(g.P);
            a_pin = 0u - static_cast<unsigned>(gq < static_cast<unsigned>(g.total_px));
            a_base = b * static_cast<unsigned>(g.K) * static_cast<unsigned>(g.P) + p;
        }
        {
            const unsigned gq = static_cast<unsigned>(ch) * kW2PC + kk;
            const unsigned b = fast_div(gq, g.divP);
            const unsigned p = gq - b * static_cast<unsigned>(g.P);
            const unsigned pin = static_cast<unsigned>(gq < static_cast<unsigned>(g.total_px));
            const int oy = static_cast<int>(fast_div(p, g.divWo)), ox = static_cast<int>(p) - oy * g.Wo;
            const int iy0 = oy * g.stride - g.pad, ix0 = ox * g.stride - g.pad;
            unsigned rmask = 0, smask = 0;
#pragma unroll
            for (int r = 0; r < R; ++r) rmask |= static_cast<unsigned>(static_cast<unsigned>(iy0 + r) < static_cast<unsigned>(g.H)) << r;
#pragma unroll
            for (int q = 0; q < S; ++q) smask |= static_cast<unsigned>(static_cast<unsigned>(ix0 + q) < static_cast<unsigned>(g.W)) << q;
            unsigned taps = 0;
#pragma unroll
            for (int r = 0; r < R; ++r) taps |= (smask & (0u - ((rmask >> r) & 1u))) << (8 * r);
            x_inv = static_cast<int>(~(taps & (0u - pin)));
            x_base = static_cast<int>(b) * g.C * g.HW + iy0 * g.W + ix0;
        }
    };
    auto issue_a = [&](int i) {          // A: four consecutive pixels of one image (P % 4 == 0), 16-byte aligned
        const unsigned live = a_pin & (0u - static_cast<unsigned>(arow_off[i] != kOobOff));
        const unsigned off = (((a_base + arow_off[i]) * 4u) & live) | (kOobOff & ~live);
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(ra_, off, 0, 0);
        va[i][0] = __uint_as_float(v.x); va[i][1] = __uint_as_float(v.y); va[i][2] = __uint_as_float(v.z); va[i][3] = __uint_as_float(v.w);
    };
    auto issue_x = [&](int i) {          // X: this lane's pixel, column i of its NXB fixed (c, r, s) columns
        const unsigned dead = static_cast<unsigned>(__builtin_amdgcn_sbfe(x_inv, colsh[i], 1u));      // 0, or all-ones = out of range
        vx[i] = buf_ld<float>(rx_, (static_cast<unsigned>(x_base + coloff[i]) << 2) | dead);
    };
    auto fetch = [&](int ch) {
        prep(ch);
#pragma unroll
        for (int i = 0; i < NQA; ++i) issue_a(i);
#pragma unroll
        for (int i = 0; i < NXB; ++i) issue_x(i);
    };
    auto commit = [&](int buf) {
        float* As = lds + buf * (TM + TN) * kW2P;
        float* Xs = As + TM * kW2P;
#pragma unroll
        for (int i = 0; i < NQA; ++i) {
            *reinterpret_cast<f32x4*>(As + (arow0 + 32 * i) * kW2P + q8 * 4) = va[i];
            if (want_bias) bsum[i] += (va[i][0] + va[i][1]) + (va[i][2] + va[i][3]);
        }
#pragma unroll
        for (int i = 0; i < NXB; ++i) Xs[(xcol0 + 8 * i) * kW2P + kk] = vx[i];
    };

    f32x16 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // MFMAs of one chunk; `next` >= 0: the loads of chunk `next` are issued in four slices, one in front of each group of
    // 4 WM WN MFMAs (the staging registers were committed to LDS before this call)
    auto compute = [&](int buf, int next) {
        const float* As = lds + buf * (TM + TN) * kW2P;
        const float* Xs = As + TM * kW2P;
        const float* ap = As + (wm * 32 * WM + l31) * kW2P + 4 * half;
        const float* bp = Xs + (wn * 32 * WN + l31) * kW2P + 4 * half;
        if (next >= 0) prep(next);
#pragma unroll
        for (int gq = 0; gq < kW2PC / 8; ++gq) {
            f32x4 av[WM], bv[WN];
#pragma unroll
            for (int i = 0; i < WM; ++i) av[i] = *reinterpret_cast<const f32x4*>(ap + i * 32 * kW2P + 8 * gq);
#pragma unroll
            for (int j = 0; j < WN; ++j) bv[j] = *reinterpret_cast<const f32x4*>(bp + j * 32 * kW2P + 8 * gq);
            if (next >= 0) {
#pragma unroll
                for (int i = 0; i < NQA; ++i)
                    if (i % 4 == gq) issue_a(i);
#pragma unroll
                for (int i = 0; i < NXB; ++i)
                    if (i / (NXB / 4) == gq) issue_x(i);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int i = 0; i < WM; ++i)
#pragma unroll
                    for (int j = 0; j < WN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i][q], bv[j][q], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);          // keep the slices where they are
        }
    };

    if (ch_begin < ch_end) {
        fetch(ch_begin);
        commit(0);
        if (ch_begin + 1 < ch_end) fetch(ch_begin + 1);
        __syncthreads();
        for (int ch = ch_begin; ch < ch_end; ++ch) {
            const int cur = (ch - ch_begin) & 1;
            if (ch + 1 < ch_end) commit(cur ^ 1);          // fetched one iteration ago; the other buffer's readers passed the barrier
            compute(cur, ch + 2 < ch_end ? ch + 2 : -1);
            __syncthreads();
        }
    }

    // ---- epilogue.  C/D layout: col = lane & 31 (n), row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5) (k)
#pragma unroll
    for (int j = 0; j < WN; ++j) {
        const int n = n0 + wn * 32 * WN + j * 32 + l31;
        if (n >= g.N) continue;
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int k = k0 + wm * 32 * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (k >= g.K) continue;
                float* d = dw + static_cast<size_t>(k) * g.N + n;
                if (g.nz > 1) atomic_add(d, acc[i][j][r]);
                else *d = acc[i][j][r];
            }
    }
    if (want_bias) {
#pragma unroll
        for (int i = 0; i < NQA; ++i) {
            float v = bsum[i];
            v += __shfl_xor(v, 1, 64);
            v += __shfl_xor(v, 2, 64);
            v += __shfl_xor(v, 4, 64);
            const int k = k0 + arow0 + 32 * i;
            if (q8 == 0 && k < g.K) {
                if (g.nz > 1) atomic_add(gbias + k, v);
                else gbias[k] = v;
            }
        }
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
    FFWM_REQUIRE((kernel == 1 || kernel == 3 || kernel == 4) && (stride == 1 || stride == 2) && pad >= 0 && pad < kernel, FFWM_ERR_ARG,
                 "%s: 1x1 / 3x3 / 4x4 kernels with stride 1 / 2 only", fn);
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

// Tiled variant: grad_weight (and grad_bias, when given: Conv2d only -- the row sums of `rows`) are OVERWRITTEN; the
// library zero-fills them itself when the pixel range is cut into slices that meet by atomics.  Needs Ho * Wo % 4 == 0
// (16-byte row loads); returns FFWM_ERR_ARG otherwise (callers fall back to ffwm_conv2d_wgrad).
extern "C" int ffwm_conv2d_wgrad_tiled(const void* rows, const void* gathered, void* grad_weight, void* grad_bias, int64_t B, int64_t K,
                                       int64_t Ho, int64_t Wo, int64_t C, int64_t H, int64_t W, int kernel, int stride, int pad,
                                       int prezeroed, int dtype, void* stream) {
    const char* fn = "ffwm_conv2d_wgrad_tiled";
    FFWM_REQUIRE(dtype == FFWM_F32, FFWM_ERR_DTYPE, "%s: float32 only", fn);
    FFWM_REQUIRE(rows && gathered && grad_weight, FFWM_ERR_ARG, "%s: NULL tensor pointer", fn);
    FFWM_REQUIRE(B > 0 && K > 0 && C > 0 && Ho > 0 && Wo > 0 && H > 0 && W > 0, FFWM_ERR_ARG, "%s: sizes must be positive", fn);
    FFWM_REQUIRE((kernel == 1 || kernel == 3 || kernel == 4) && (stride == 1 || stride == 2) && pad >= 0 && pad < kernel, FFWM_ERR_ARG,
                 "%s: 1x1 / 3x3 / 4x4 kernels with stride 1 / 2 only", fn);
    FFWM_REQUIRE(Ho == (H + 2 * pad - kernel) / stride + 1 && Wo == (W + 2 * pad - kernel) / stride + 1, FFWM_ERR_ARG,
                 "%s: the row tensor's plane (%lld x %lld) is not the output plane of this convolution over %lld x %lld", fn,
                 (long long)Ho, (long long)Wo, (long long)H, (long long)W);
    FFWM_REQUIRE((Ho * Wo) % 4 == 0, FFWM_ERR_ARG, "%s: the row tensor's plane must hold a multiple of 4 pixels", fn);
    FFWM_REQUIRE((reinterpret_cast<uintptr_t>(rows) & 15) == 0, FFWM_ERR_ARG, "%s: the row tensor must be 16-byte aligned", fn);
    FFWM_REQUIRE(B * K * Ho * Wo < (1LL << 29) && B * C * H * W < (1LL << 29), FFWM_ERR_SIZE, "%s: tensors must stay below 2 GiB", fn);
    Wg2Geo g;
    g.C = static_cast<int>(C); g.H = static_cast<int>(H); g.W = static_cast<int>(W);
    g.K = static_cast<int>(K); g.Ho = static_cast<int>(Ho); g.Wo = static_cast<int>(Wo);
    g.stride = stride; g.pad = pad;
    g.N = g.C * kernel * kernel;
    g.P = g.Ho * g.Wo;
    g.HW = g.H * g.W;
    g.total_px = static_cast<int>(B) * g.P;
    g.chunks_total = (g.total_px + kW2PC - 1) / kW2PC;
    g.a_bytes = static_cast<unsigned>(B * K * Ho * Wo * 4);
    g.x_bytes = static_cast<unsigned>(B * C * H * W * 4);
    g.divP = make_fast_div(static_cast<unsigned>(g.P));
    g.divWo = make_fast_div(static_cast<unsigned>(g.Wo));
    // tile shape (64 or 128 rows x 64 or 128 columns of dW) and pixel slices: the candidate with the smallest estimated time, in
    // units of one 32 x 32 x 32-pixel MFMA block per wave (1024 cycles).  A workgroup pays ~6 units of prologue / epilogue, a
    // sliced launch pays the atomic flush of its tile, and two workgroups per CU are resident (512 per round): a launch of 567
    // workgroups runs as long as one of 1024, and FlowNet's 8 x 8 layers (16 chunks in all) are better off with 288 unsliced
    // 64 x 128 tiles than with 432 slices of three 128 x 128 chunks each.
    const int64_t wg_target = options().conv_wgrad_slice_target > 0 ? options().conv_wgrad_slice_target : 512;
    int wmt = 1, wnt = 1;
    double best = 1e30;
    for (int cm = 1; cm <= 2; ++cm)
        for (int cn = 1; cn <= 2; ++cn) {
            if ((cm == 2 && g.K <= 64) || (cn == 2 && g.N <= 64)) continue;
            const int64_t t = static_cast<int64_t>((g.K + 64 * cm - 1) / (64 * cm)) * ((g.N + 64 * cn - 1) / (64 * cn));
            int64_t sl = t >= wg_target ? 1 : wg_target / t;
            if (sl > g.chunks_total / 4) sl = g.chunks_total / 4;
            if (sl < 1) sl = 1;
            const int64_t per = (g.chunks_total + sl - 1) / sl;
            const int64_t nzc = (g.chunks_total + per - 1) / per;
            const int64_t rounds = (t * nzc + 511) / 512;
            const double est = rounds * (per * cm * cn * (cm * cn == 1 ? 1.6 : (cm * cn == 2 ? 1.25 : 1.0)) + 6.0 + (nzc > 1 ? 2.0 * cm * cn : 0.0));
            if (est < best) { best = est; wmt = cm; wnt = cn; }
        }
    const int k_tiles = (g.K + 64 * wmt - 1) / (64 * wmt), n_tiles = (g.N + 64 * wnt - 1) / (64 * wnt);
    const int64_t tiles = static_cast<int64_t>(k_tiles) * n_tiles;
    int64_t slices = tiles >= wg_target ? 1 : wg_target / tiles;
    if (slices > g.chunks_total / 4) slices = g.chunks_total / 4;
    if (slices < 1 || options().conv_wgrad_unsliced) slices = 1;
    g.chunks = static_cast<int>((g.chunks_total + slices - 1) / slices);
    g.nz = (g.chunks_total + g.chunks - 1) / g.chunks;
    hipStream_t st = static_cast<hipStream_t>(stream);
    float* dw = static_cast<float*>(grad_weight);
    float* gb = static_cast<float*>(grad_bias);
    if (g.nz > 1 && !prezeroed && !options().conv_wgrad_prezeroed) {
        const size_t nw = static_cast<size_t>(g.K) * g.N;
        const bool joined = gb == dw + nw;          // one buffer (ops.conv2d_wgrad_tiled allocates them together): one memset
        if (zero_fill(dw, sizeof(float) * (nw + (joined ? static_cast<size_t>(g.K) : 0)), st)) return FFWM_ERR_LAUNCH;
        if (gb && !joined && zero_fill(gb, sizeof(float) * static_cast<size_t>(g.K), st)) return FFWM_ERR_LAUNCH;
    }
    const double flops = 2.0 * B * g.P * static_cast<double>(g.K) * g.N;
    const double bytes = 4.0 * (static_cast<double>(B) * K * g.P + static_cast<double>(B) * C * H * W + static_cast<double>(K) * g.N);
    LaunchScope ls("conv_wgrad_mfma_tiled", st, bytes, flops);
    const dim3 grid(static_cast<unsigned>(n_tiles), static_cast<unsigned>(k_tiles), static_cast<unsigned>(g.nz));
    const float* a = static_cast<const float*>(rows);
    const float* x = static_cast<const float*>(gathered);
#define FFWM_WG2(RR, MM, NN)                                                                                                  \
    do {                                                                                                                      \
        auto kfn = conv_wgrad_tile_kernel<RR, RR, MM, NN>;                                                                    \
        const size_t lds = 2u * (64 * MM + 64 * NN) * kW2P * sizeof(float);                                                   \
        allow_large_lds(reinterpret_cast<const void*>(kfn));                                                                  \
        hipLaunchKernelGGL(kfn, grid, dim3(kBlock), lds, st, a, x, dw, gb, g);                                                \
    } while (0)
#define FFWM_WG2_K(RR)                                                                                                        \
    do {                                                                                                                      \
        if (wmt == 2 && wnt == 2) FFWM_WG2(RR, 2, 2);                                                                         \
        else if (wmt == 2) FFWM_WG2(RR, 2, 1);                                                                                \
        else if (wnt == 2) FFWM_WG2(RR, 1, 2);                                                                                \
        else FFWM_WG2(RR, 1, 1);                                                                                              \
    } while (0)
    if (kernel == 3) FFWM_WG2_K(3);
    else if (kernel == 4) FFWM_WG2_K(4);
    else FFWM_WG2_K(1);          // 1x1 (the shortcut convolution of netG's residual blocks, base_networks.py:213): a plain [K x P] . [P x C] product
#undef FFWM_WG2_K
#undef FFWM_WG2
    return check_launch(fn);
}
