// conv_wgrad_wino.hip -- weight gradient of a 3x3 / stride 1 / pad 1 convolution in the Winograd F(2x2, 3x3) domain on the
// fp32 MFMA units (gfx950).  The direct kernel of conv_wgrad.hip runs netG's large layers (models/base_networks.py:207-233,
// 293-298: dres2 / att2 at 128 x 128, dres1 / att1 at 64 x 64) at 0.72-0.80 of the fp32 MFMA peak and is the second largest cost
// of the captured train step (5.9 ms); the fp32 MFMA peak equals the vector peak, so the only way past it is fewer
// multiplications.  With Y_t = At [ (G w Gt) (.) (Bt d_t B) ] A per 2 x 2 output tile t,
//
//     dW = Gt [ sum_t (A dY_t At) (.) (Bt d_t B) ] G          (A: 4 x 2, B: 4 x 4, G: 4 x 3)
//
// i.e. 16 position-wise GEMMs  dU_p[k][c] = sum_t dM_p[k][t] V_p[c][t]  with the reduction over the tiles: 16 instead of 36
// multiplications per (k, c, tile).  Both operands are transformed on the way to LDS (the forward kernel of conv_winograd.hip
// transforms one and reads the other prepared).
//
//   * A workgroup of 8 waves owns 64 output channels x 64 input channels for ALL 16 positions (a wave: 8 positions x 32 x 32 =
//     8 accumulators of v_mfma_f32_32x32x2_f32, 128 registers) and a contiguous range of CHUNKS of 8 tiles (a strip of 8 tiles
//     along x: 16 x 2 pixels of dY, 18 x 4 of the input).
//   * Per chunk, thread (tile t8 = tid & 7, channel ch = tid >> 3) loads the 2 x 2 dY tile of output channel k0 + ch (two 8-byte
//     loads) and the 4 x 4 input patch of input channel c0 + ch (a dword, an 8-byte pair and a dword per row; the rows outside the
//     image are the buffer range check's zeros, the two columns outside are masked), forms A dY At (12 adds) and Bt d B (32
//     adds) and writes the 2 x 16 values to LDS as [position][tile half][64 channels][4 tiles]: an MFMA operand for two tiles is
//     then one lane's float of a ds_read_b128, exactly the forward kernel's layout with tiles in the place of input channels.
//   * Step n: 32 MFMAs per wave on LDS buffer n & 1; in their shadow chunk n + 1 (requested two steps ago) is transformed into the
//     other buffer and chunk n + 3 is requested into the registers just freed.  The step is branch-free (chunks past the slice
//     load and commit zeros), so the scheduler can interleave it: one barrier per chunk.
//   * Epilogue: Gt dU G per (k, c) -- a wave holds two of the four rows i of dU, so it forms its share of the nine taps, the two
//     shares meet in LDS (64 x 64 x 9 floats) and the block adds its tile to grad_weight with coalesced global atomics (rows of
//     576 consecutive floats): the pixel slices of a (k, c) tile meet in memory; the caller zero-fills, as for the direct kernel.
// fp32 throughout; sums over up to 32768 tiles per element: <= 2e-5 of the result's scale against fp64 (tests).
#include "common.hpp"
#include <type_traits>

namespace ffwm {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kWwThreads = 512;
// f32x4 per operand buffer: [16 positions][2 tile halves][64 channels (+ 8 of padding)] x 4 tiles.  The padding shifts the second tile
// half by 32 banks: a wave's ds_write_b32 covers (8 channels x 4 tiles) of BOTH halves, which sat on the same 32 banks at a half
// stride of 256 floats -- a 2-way conflict on every one of the 32 writes per thread and chunk (SQ_LDS_BANK_CONFLICT = 33 % of the
// LDS cycles, profiles/r04_wgrad_wino_counters.txt); 4 x 36 KiB = 144 KiB, what the epilogue's staging needs anyway.
constexpr int kWwHalf = 72;                         // f32x4 per tile half of a position
constexpr int kWwPos = 2 * kWwHalf;                 // f32x4 per position
constexpr int kWwOperand = 16 * kWwPos;
constexpr unsigned kWwOob = 0xFFFFFFF0u;

struct WwGeo {
    int C, K, H, W;                  // tensor extents
    int k_begin, k_end, c_begin, c_end;
    int KT, CT;                      // 64-channel tiles of the two ranges
    int TH, TWC;                     // tile rows, chunks (8 tiles) per tile row
    int chunks;                      // B * TH * TWC
    int nsplit, remap;
    unsigned x_bytes, g_bytes;
};

__global__ void __launch_bounds__(kWwThreads)
conv3x3_wgrad_wino_kernel(const float* __restrict__ X, const float* __restrict__ G, float* __restrict__ dW, float* __restrict__ dbias,
                          const WwGeo g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    f32x4* const smem = reinterpret_cast<f32x4*>(smem_raw);      // [2 buffers][A: kWwOperand | B: kWwOperand]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int ph = wave >> 2, wm = (wave >> 1) & 1, wn = wave & 1;

    // the (k, c) tiles of one pixel slice are consecutive logical ids: with xcd_remap they run on ONE XCD and share its L2
    // (workgroup b runs on XCD b % 8: unremapped, the nine tiles of a slice of a 192 x 192 weight fetch it through eight L2s)
    unsigned t = xcd_remap(blockIdx.x, gridDim.x, g.remap);
    const int kt = t % g.KT; t /= g.KT;
    const int ct = t % g.CT; t /= g.CT;
    const int split = static_cast<int>(t);
    const int k0 = g.k_begin + kt * 64, c0 = g.c_begin + ct * 64;
    const int ch_begin = static_cast<int>(static_cast<int64_t>(g.chunks) * split / g.nsplit);
    const int ch_end = static_cast<int>(static_cast<int64_t>(g.chunks) * (split + 1) / g.nsplit);
    const rsrc_t rx = make_rsrc(X, g.x_bytes), rg = make_rsrc(G, g.g_bytes);
    const int HW = g.H * g.W;

    // ---- producer role: tile t8 of the chunk, channel chn of both 64-channel ranges
    const int t8 = threadIdx.x & 7, chn = threadIdx.x >> 3;
    const bool k_ok = k0 + chn < g.k_end, c_ok = c0 + chn < g.c_end;
    const unsigned gk_base = static_cast<unsigned>(k0 + chn) * static_cast<unsigned>(HW);       // floats; + b * K * HW
    const unsigned xc_base = static_cast<unsigned>(c0 + chn) * static_cast<unsigned>(HW);
    const int wr_off = (t8 >> 2) * (4 * kWwHalf) + chn * 4 + (t8 & 3);       // float index inside a position block of 4 * kWwPos floats

    // the load cursor: chunk index and its (image, tile row, chunk of the row), advanced without divisions
    int cur = ch_begin;
    int cb = cur / (g.TH * g.TWC), cty, ctxc;
    {
        const int rem = cur - cb * (g.TH * g.TWC);
        cty = rem / g.TWC;
        ctxc = rem - cty * g.TWC;
    }
    struct Regs {
        float gy[4];            // dY tile [row][col]
        float xp[16];           // input patch rows 2ty - 1 .. 2ty + 2: [4 i + 1], [4 i + 2] = cols 2tx, 2tx + 1; [4 i] = the gathered outer pixel
    };
    float bsum = 0.f;           // bias gradient: this thread's share of sum(dY[k0 + chn]) (the ct == 0 workgroups keep it)
    // Round 5 (late): the addresses of a chunk's loads are a PER-THREAD constant (channel, tile t8 of the chunk) + a WAVE-UNIFORM offset
    // (image, tile row, chunk of the row) that rides in the loads' scalar offset; what is left per chunk on the vector ALU is one select
    // per load (a row outside the image / a chunk past the slice -> the range check's zeros).  Before, every thread worked out every
    // address of every chunk (~60 VALU + ~80 SALU instructions per wave and chunk next to 32 MFMAs and the transforms): the step was
    // bound by instruction issue, not by the matrix pipe.  The two outer-column gathers of a row are ONE load (tile 0 of the chunk takes
    // the pixel on its left, tile 7 the one on its right, the others nothing): 10 loads per thread and chunk instead of 14.
    const unsigned gv = k_ok ? (gk_base + static_cast<unsigned>(2 * t8)) * 4u : kWwOob;
    const unsigned xv = c_ok ? (xc_base + static_cast<unsigned>(2 * t8)) * 4u : kWwOob;
    // (through a resource that starts one float BEFORE the tensor: the range check adds the vector and the scalar offset in more than 32
    // bits, so "- 4" as a wrapped vector offset is out of range for channel 0 -- measured, tools/ubench/buf_soffset.hip)
    const rsrc_t rxe = make_rsrc(reinterpret_cast<const char*>(X) - 4, g.x_bytes + 4u);
    const unsigned xe = (c_ok && (t8 == 0 || t8 == 7)) ? (xc_base + static_cast<unsigned>(2 * t8)) * 4u + (t8 == 0 ? 0u : 12u) : kWwOob;
    // the scalar offsets run along with the cursor (bytes, modulo 2^32): dY row 2 ty and input row 2 ty - 1 of the chunk's first pixel
    // column.  A chunk to the right is + 64 bytes; a tile row down + 4 W more (2 W pixels on, W back); the next image (K - 1) / (C - 1)
    // planes more -- three scalar adds per chunk instead of the products of (image, tile row, chunk)
    unsigned sg = (static_cast<unsigned>(cb) * g.K * HW + static_cast<unsigned>(2 * cty * g.W) + static_cast<unsigned>(16 * ctxc)) * 4u;
    unsigned sxr = (static_cast<unsigned>(cb) * g.C * HW + static_cast<unsigned>((2 * cty - 1) * g.W) + static_cast<unsigned>(16 * ctxc)) * 4u;
    const unsigned w4 = static_cast<unsigned>(g.W) * 4u;
    const unsigned img_g = static_cast<unsigned>(g.K - 1) * static_cast<unsigned>(HW) * 4u, img_x = static_cast<unsigned>(g.C - 1) * static_cast<unsigned>(HW) * 4u;
    auto issue = [&](Regs& R) {              // loads of the cursor's chunk (zeros past the slice's end), then the cursor moves on
        const bool live = cur < ch_end;
        const unsigned go = live ? gv : kWwOob;
        const u32x2 r0 = __builtin_amdgcn_raw_buffer_load_b64(rg, go, sg, 0);
        const u32x2 r1 = __builtin_amdgcn_raw_buffer_load_b64(rg, go, sg + w4, 0);
        R.gy[0] = __uint_as_float(r0.x); R.gy[1] = __uint_as_float(r0.y);
        R.gy[2] = __uint_as_float(r1.x); R.gy[3] = __uint_as_float(r1.y);
        // the outer column exists on the left of every chunk but a row's first, on the right of every chunk but its last (per lane: t8)
        const bool edge_ok = t8 == 0 ? ctxc > 0 : ctxc + 1 < g.TWC;
        const bool top = cty > 0, bottom = cty + 1 < g.TH;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool rok = live && (i == 0 ? top : i == 3 ? bottom : true);  // wave-uniform: input rows 2 ty - 1 .. 2 ty + 2 (H = 2 TH)
            const unsigned srow = sxr + static_cast<unsigned>(i) * w4;
            const u32x2 m = __builtin_amdgcn_raw_buffer_load_b64(rx, rok ? xv : kWwOob, srow, 0);
            // the patch's outer columns are the NEIGHBOURING tiles' inner ones (lanes t8 - 1 / t8 + 1 of the same channel and row):
            // commit() takes them by DPP row shifts; only the chunk's first / last tile gathers one
            R.xp[i * 4 + 0] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rxe, (rok && edge_ok) ? xe : kWwOob, srow, 0));
            R.xp[i * 4 + 1] = __uint_as_float(m.x);
            R.xp[i * 4 + 2] = __uint_as_float(m.y);
        }
        ++cur;
        const bool wrap_x = ctxc + 1 == g.TWC;
        ctxc = wrap_x ? 0 : ctxc + 1;
        const bool wrap_y = wrap_x && cty + 1 == g.TH;
        cty = wrap_x ? (wrap_y ? 0 : cty + 1) : cty;
        cb += wrap_y ? 1 : 0;
        sg += 64u + (wrap_x ? w4 : 0u) + (wrap_y ? img_g : 0u);
        sxr += 64u + (wrap_x ? w4 : 0u) + (wrap_y ? img_x : 0u);
    };
    auto commit = [&](f32x4* buf, const Regs& R) {
        float* A = reinterpret_cast<float*>(buf) + wr_off;
        float* Bv = reinterpret_cast<float*>(buf + kWwOperand) + wr_off;
        // dM = A dY At,  A = [1 0; 1 1; 1 -1; 0 -1]
        {
            const float a = R.gy[0], b = R.gy[1], c = R.gy[2], d = R.gy[3];
            bsum += (a + b) + (c + d);
            const float rp[4] = {a, a + c, a - c, -c}, rq[4] = {b, b + d, b - d, -d};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                A[(i * 4 + 0) * (4 * kWwPos)] = rp[i];
                A[(i * 4 + 1) * (4 * kWwPos)] = rp[i] + rq[i];
                A[(i * 4 + 2) * (4 * kWwPos)] = rp[i] - rq[i];
                A[(i * 4 + 3) * (4 * kWwPos)] = -rq[i];
            }
        }
        // the outer columns of the 4 x 4 patch from the neighbouring lanes (row_shr:1 = lane - 1, row_shl:1 = lane + 1; t8 = lane & 7
        // never crosses a 16-lane DPP row on the lanes that use the shifted value)
        float xq[16];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float from_l = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(R.xp[i * 4 + 2]), 0x111, 0xf, 0xf, true));
            const float from_r = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(R.xp[i * 4 + 1]), 0x101, 0xf, 0xf, true));
            xq[i * 4 + 0] = t8 == 0 ? R.xp[i * 4 + 0] : from_l;          // (xp[4 i]: the gathered outer pixel of tiles 0 and 7)
            xq[i * 4 + 1] = R.xp[i * 4 + 1];
            xq[i * 4 + 2] = R.xp[i * 4 + 2];
            xq[i * 4 + 3] = t8 == 7 ? R.xp[i * 4 + 0] : from_r;
        }
        // V = Bt d B,  Bt = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float r[4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                r[j] = i == 0 ? xq[0 + j] - xq[8 + j] : i == 1 ? xq[4 + j] + xq[8 + j] : i == 2 ? xq[8 + j] - xq[4 + j] : xq[4 + j] - xq[12 + j];
            Bv[(i * 4 + 0) * (4 * kWwPos)] = r[0] - r[2];
            Bv[(i * 4 + 1) * (4 * kWwPos)] = r[1] + r[2];
            Bv[(i * 4 + 2) * (4 * kWwPos)] = r[2] - r[1];
            Bv[(i * 4 + 3) * (4 * kWwPos)] = r[1] - r[3];
        }
    };

    f32x16 acc[8];
#pragma unroll
    for (int p = 0; p < 8; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;

    // step of chunk n (parity P): MFMAs on buffer P; the registers of set Q (chunk n + 1, requested one step ago) are transformed
    // into buffer Q; chunk n + 2 is requested into set P.  Branch-free (a chunk past the slice loads and commits zeros), so that the
    // scheduler can lay the transform, the LDS writes and the loads into the shadow of the 32 MFMAs.
    Regs R[2];
    auto step = [&](auto parity) {
        constexpr int P = decltype(parity)::value, Q = 1 - P;
        const f32x4* ap = smem + P * (2 * kWwOperand) + (ph * 8) * kWwPos + half * kWwHalf + wm * 32 + l31;
        const f32x4* bp = smem + P * (2 * kWwOperand) + kWwOperand + (ph * 8) * kWwPos + half * kWwHalf + wn * 32 + l31;
        f32x4 oa[2][2], ob[2][2];
        oa[0][0] = ap[0]; ob[0][0] = bp[0]; oa[0][1] = ap[kWwPos]; ob[0][1] = bp[kWwPos];
#pragma unroll
        for (int grp = 0; grp < 4; ++grp) {
            const int c2 = grp & 1, n2 = c2 ^ 1;
            if (grp < 3) {
                oa[n2][0] = ap[(2 * grp + 2) * kWwPos]; ob[n2][0] = bp[(2 * grp + 2) * kWwPos];
                oa[n2][1] = ap[(2 * grp + 3) * kWwPos]; ob[n2][1] = bp[(2 * grp + 3) * kWwPos];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[2 * grp] = __builtin_amdgcn_mfma_f32_32x32x2f32(oa[c2][0][j], ob[c2][0][j], acc[2 * grp], 0, 0, 0);
                acc[2 * grp + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(oa[c2][1][j], ob[c2][1][j], acc[2 * grp + 1], 0, 0, 0);
            }
            if (grp == 0) commit(smem + Q * (2 * kWwOperand), R[Q]);
            else if (grp == 2) issue(R[Q]);
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // one MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);      // up to 8 VALU
                __builtin_amdgcn_sched_group_barrier(0x200, 2, 0);      // up to 2 LDS writes
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      // one LDS read
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);      // one VMEM read
            }
        }
        __syncthreads();
    };

    if (ch_begin < ch_end) {
        issue(R[0]);
        issue(R[1]);
        commit(smem, R[0]);
        issue(R[0]);
        __syncthreads();
        // now: buffer 0 = chunk n0, R[1] = chunk n0 + 1 and R[0] = chunk n0 + 2 in flight.  Step n commits R[(n + 1) & 1] (chunk
        // n + 1) into the other buffer and refills that set with chunk n + 3: a load has 1.5-2 steps to land
        const int n = ch_end - ch_begin;
        for (int i = 0; i < n; i += 2) {          // (an odd count ends with a step on a buffer of zeros: no branch around it)
            step(std::integral_constant<int, 0>());
            step(std::integral_constant<int, 1>());
        }
    }
    if (dbias && ct == 0) {
        float v = bsum;
        v += __shfl_xor(v, 1, 64);
        v += __shfl_xor(v, 2, 64);
        v += __shfl_xor(v, 4, 64);
        if (t8 == 0 && k_ok) atomic_add(dbias + k0 + chn, v);
    }

    // ---- epilogue: Gt dU G.  C/D layout: col = lane & 31 (c), row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5) (k).
    // This wave holds rows i = 2 ph, 2 ph + 1 of dU (acc[4 (i - 2 ph) + j]); G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1].
    float* const stage = reinterpret_cast<float*>(smem_raw);                 // [64 k][64 c][9]
    const int cl = wn * 32 + l31;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        if (ph == pass) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kl = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                float z[2][3];
#pragma unroll
                for (int ii = 0; ii < 2; ++ii) {
                    const float u0 = acc[ii * 4 + 0][r], u1 = acc[ii * 4 + 1][r], u2 = acc[ii * 4 + 2][r], u3 = acc[ii * 4 + 3][r];
                    z[ii][0] = u0 + 0.5f * (u1 + u2);
                    z[ii][1] = 0.5f * (u1 - u2);
                    z[ii][2] = 0.5f * (u1 + u2) + u3;
                }
                float* o = stage + (kl * 64 + cl) * 9;
#pragma unroll
                for (int s = 0; s < 3; ++s) {
                    if (pass == 0) {          // rows i = 0, 1:  tap row 0 = z0 + z1 / 2, rows 1 and 2 = z1 / 2
                        o[0 + s] = z[0][s] + 0.5f * z[1][s];
                        o[3 + s] = 0.5f * z[1][s];
                        o[6 + s] = 0.5f * z[1][s];
                    } else {                  // rows i = 2, 3:  tap row 0 += z2 / 2, row 1 -= z2 / 2, row 2 += z2 / 2 + z3
                        o[0 + s] += 0.5f * z[0][s];
                        o[3 + s] -= 0.5f * z[0][s];
                        o[6 + s] += 0.5f * z[0][s] + z[1][s];
                    }
                }
            }
        }
        __syncthreads();
    }
    const int kmax = min(64, g.k_end - k0), cmax = min(64, g.c_end - c0);
    for (int idx = threadIdx.x; idx < 64 * 576; idx += kWwThreads) {
        const int kl = idx / 576, rem = idx - kl * 576;
        if (kl < kmax && rem < cmax * 9) {
            float* dst = dW + (static_cast<size_t>(k0 + kl) * g.C + c0) * 9 + rem;
            const float v = stage[idx];
            if (g.nsplit > 1) atomic_add(dst, v);
            else *dst += v;
        }
    }
}

}  // namespace

// The main (full 64-channel tiles) part of ffwm_conv3x3_wgrad_block on the Winograd-domain kernel: grad_weight[k_begin:k_end,
// c_begin:c_end] += ...  (the caller zero-filled grad_weight).  Returns FFWM_OK after the launch, or a positive value when the
// shape is not served (odd sizes, W not a multiple of 16, too few chunks): the caller then uses the direct kernel.
int launch_wgrad_wino(const float* X, const float* G, float* dW, float* dbias, int64_t B, int64_t C, int64_t K, int64_t H, int64_t W,
                      int64_t k_begin, int64_t k_end, int64_t c_begin, int64_t c_end, hipStream_t st) {
    if (k_begin >= k_end || c_begin >= c_end) return FFWM_OK;
    if ((H & 1) || (W & 15) || B * C * H * W * 4 >= (1LL << 32) - 64 || B * K * H * W * 4 >= (1LL << 32) - 64) return 1;
    WwGeo g;
    g.C = static_cast<int>(C); g.K = static_cast<int>(K); g.H = static_cast<int>(H); g.W = static_cast<int>(W);
    g.k_begin = static_cast<int>(k_begin); g.k_end = static_cast<int>(k_end);
    g.c_begin = static_cast<int>(c_begin); g.c_end = static_cast<int>(c_end);
    g.KT = static_cast<int>((k_end - k_begin + 63) / 64);
    g.CT = static_cast<int>((c_end - c_begin + 63) / 64);
    g.TH = static_cast<int>(H / 2);
    g.TWC = static_cast<int>(W / 16);
    const int64_t chunks = B * g.TH * g.TWC;
    if (chunks >= (1LL << 30)) return 1;
    g.chunks = static_cast<int>(chunks);
    const int64_t tiles = static_cast<int64_t>(g.KT) * g.CT;
    // measured (tools/wgrad_wino_check.py, profiles/r04_wgrad_winograd.txt): 1.3-1.6 x the direct kernel from 64 x 64 planes at batch 8
    // up; [2,64,64,64] -> 64 (256 chunks, one tile: 16 workgroups) loses, 64 vs 50 us
    if (options().conv_wgrad_wino == 0 && chunks * tiles < 4096) return 1;
    int64_t nsplit = tiles >= 256 ? 1 : 256 / tiles;               // one workgroup per CU (147 KB of LDS)
    if (nsplit > chunks / 16) nsplit = chunks / 16;
    if (nsplit < 1) return 1;                                       // fewer than 16 chunks in all: the direct kernel
    g.nsplit = static_cast<int>(nsplit);
    g.remap = options().xcd_remap;
    g.x_bytes = static_cast<unsigned>(B * C * H * W * 4);
    g.g_bytes = static_cast<unsigned>(B * K * H * W * 4);
    const size_t lds = 64 * 576 * sizeof(float);                    // the epilogue's staging = the 2 x 2 padded operand buffers (144 KiB)
    static_assert(4 * kWwOperand * 16 <= 64 * 576 * 4, "operand buffers exceed the LDS request");
    allow_large_lds(reinterpret_cast<const void*>(conv3x3_wgrad_wino_kernel));
    const double kk = static_cast<double>(k_end - k_begin), cc = static_cast<double>(c_end - c_begin);
    // flops = the multiplications the MFMAs execute, as for the forward kernel's scope (the direct sum this call replaces has 2.25 x as many)
    LaunchScope ls("conv3x3_wgrad_winograd", st, 4.0 * (static_cast<double>(B) * H * W * (kk + cc) + 9.0 * kk * cc),
                   2.0 * 16.0 * (static_cast<double>(B) * H * W / 4.0) * kk * cc);
    hipLaunchKernelGGL(conv3x3_wgrad_wino_kernel, dim3(static_cast<unsigned>(tiles * nsplit)), dim3(kWwThreads), lds, st, X, G, dW, dbias, g);
    return check_launch("ffwm_conv3x3_wgrad(winograd)");
}

}  // namespace ffwm
