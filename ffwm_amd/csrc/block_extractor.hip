// block_extractor.hip -- flow-guided bilinear k x k patch extractor for gfx950.
//
// Replaces kernel_block_extractor_update_output / kernel_block_extractor_backward of
// /root/reference/cuda/block_extractor/block_extractor_kernel.cu:21-170 (one thread per OUTPUT
// element, flow + floor + clamp recomputed per element, C*k*k atomics per grad_flow address).
//
// Design here (MI355X-first):
//   * one thread per FLOW PIXEL, looping over a slab of channels.  All per-pixel state -- the k
//     x-taps and k y-taps (clamped indices + weights, formed with the reference's exact
//     operation order) -- is computed once and reused for every channel of the slab.
//   * a wave covers 64 consecutive xf, so its k contiguous outputs per lane form one contiguous
//     64*k-element row segment: stores are full-line coalesced; gathers of neighbouring lanes
//     fall into the same few cache lines.
//   * neighbouring taps share source pixels: in the (overwhelmingly common) "consistent" case
//     xR[j] == xL[j+1], yB[i] == yT[i+1] a pixel reads its (k+1) x (k+1) neighbourhood once,
//     streaming row by row (register use O(k), not O(k^2)).
//   * backward: grad_flow is accumulated in registers over the window and the channel slab
//     (one atomic pair per pixel per slab instead of C*k*k); grad_source contributions are merged
//     per neighbourhood cell in registers before the atomics ((k+1)^2 instead of 4*k^2).
//   * blockIdx -> tile mapping is XCD-aware (common.hpp:xcd_remap).
#include "common.hpp"
#include <type_traits>

namespace ffwm {
namespace {

template <typename T>
__device__ __forceinline__ T fma_t(T a, T b, T c) { return __builtin_fma(a, b, c); }
template <>
__device__ __forceinline__ float fma_t<float>(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

// separately rounded product / sum (what two ATen kernels in a row compute): never contracted to an fma
__device__ __forceinline__ float mul_rn(float a, float b) { return __fmul_rn(a, b); }
__device__ __forceinline__ double mul_rn(double a, double b) { return __dmul_rn(a, b); }
__device__ __forceinline__ float add_rn(float a, float b) { return __fadd_rn(a, b); }
__device__ __forceinline__ double add_rn(double a, double b) { return __dadd_rn(a, b); }

template <typename T, int K>
struct Taps {
    // "consistent" representation: tap (i, j) reads rows row[i], row[i+1] and columns col[j],
    // col[j+1] (unsigned offsets from the wave-uniform plane base -> saddr + voffset addressing).
    unsigned col[K + 1];   // x * sizeof(T)           (BYTE offsets: a 32-bit voffset next to an
    unsigned row[K + 1];   // y * Ws * sizeof(T)        SGPR plane base, no 64-bit VGPR addresses)
    T wxL[K], wxR[K], wyT[K], wyB[K];
    bool consistent;
};

// One tap of one output element, exactly the reference's arithmetic
// (block_extractor_kernel.cu:52-71): flow + offset, + pixel coordinate, floor, clamp, weights
// from the unclamped fraction.
template <typename T>
struct Tap1 {
    unsigned lo, hi;   // clamped index of floor(d), floor(d)+1
    T wlo, whi;        // 1 - frac, frac
};
template <typename T>
__device__ __forceinline__ Tap1<T> make_tap(T flow0, int offset, int coord, int n) {
    const T f = flow0 + static_cast<T>(offset);
    const T d = f + static_cast<T>(coord);
    const T fl = floor_t(d);
    Tap1<T> t;
    t.lo = static_cast<unsigned>(clamp_index(fl, n));
    t.hi = static_cast<unsigned>(clamp_index(fl + 1, n));
    t.whi = d - fl;
    t.wlo = 1 - (d - fl);
    return t;
}

template <typename T, int K>
__device__ __forceinline__ void make_taps(Taps<T, K>& t, T flow_x0, T flow_y0, int xf, int yf,
                                          int Hs, int Ws) {
    bool ok = true;
    unsigned prev_xhi = 0, prev_yhi = 0;
#pragma unroll
    for (int j = 0; j < K; ++j) {
        const Tap1<T> tx = make_tap<T>(flow_x0, j - K / 2, xf, Ws);
        const Tap1<T> ty = make_tap<T>(flow_y0, j - K / 2, yf, Hs);
        if (j > 0) ok = ok && (tx.lo == prev_xhi) && (ty.lo == prev_yhi);
        prev_xhi = tx.hi;
        prev_yhi = ty.hi;
        t.col[j] = tx.lo * static_cast<unsigned>(sizeof(T));
        t.row[j] = ty.lo * static_cast<unsigned>(Ws) * static_cast<unsigned>(sizeof(T));
        t.wxL[j] = tx.wlo;
        t.wxR[j] = tx.whi;
        t.wyT[j] = ty.wlo;
        t.wyB[j] = ty.whi;
    }
    t.col[K] = prev_xhi * static_cast<unsigned>(sizeof(T));
    t.row[K] = prev_yhi * static_cast<unsigned>(Ws) * static_cast<unsigned>(sizeof(T));
    t.consistent = ok;
}

// ------------------------------------------------------------------------------ forward
template <typename T, int K>
__global__ void __launch_bounds__(kBlock)
be_fwd_kernel(const T* __restrict__ src, const T* __restrict__ flow, T* __restrict__ out, int C,
              int Hs, int Ws, int Hf, int Wf, int tiles_x, int tiles_y, int cslabs, int cs,
              int remap) {
    const TileCoord tc = decode_tile(tiles_x, tiles_y, cslabs, remap);
    if (tc.xf >= Wf || tc.yf >= Hf) return;
    const size_t fplane = static_cast<size_t>(Hf) * Wf;
    const T* fl = flow + static_cast<size_t>(tc.b) * 2 * fplane + static_cast<size_t>(tc.yf) * Wf + tc.xf;
    const T fx0 = fl[0], fy0 = fl[fplane];
    Taps<T, K> t;
    make_taps<T, K>(t, fx0, fy0, tc.xf, tc.yf, Hs, Ws);

    const int c0 = tc.slab * cs;
    const int c1 = (c0 + cs < C) ? c0 + cs : C;
    const int W = K * Wf;
    const size_t oplane = static_cast<size_t>(K) * Hf * W;
    const size_t splane = static_cast<size_t>(Hs) * Ws;
    const unsigned sbytes = static_cast<unsigned>(splane * sizeof(T));
    const unsigned obytes = static_cast<unsigned>(oplane * sizeof(T));
    const T* sp = src + (static_cast<size_t>(tc.b) * C + c0) * splane;
    T* op = out + (static_cast<size_t>(tc.b) * C + c0) * oplane;
    const unsigned obase = (static_cast<unsigned>(tc.yf) * K * W + static_cast<unsigned>(tc.xf) * K) *
                           static_cast<unsigned>(sizeof(T));
    const unsigned orow = static_cast<unsigned>(W) * static_cast<unsigned>(sizeof(T));

    if (t.consistent) {
        for (int c = c0; c < c1; ++c, sp += splane, op += oplane) {
            const rsrc_t rs = make_rsrc(sp, sbytes);
            const rsrc_t ro = make_rsrc(op, obytes);
            T prev[K + 1], cur[K + 1];
#pragma unroll
            for (int j = 0; j <= K; ++j) prev[j] = buf_ld<T>(rs, t.row[0] + t.col[j]);
#pragma unroll
            for (int i = 0; i < K; ++i) {
#pragma unroll
                for (int j = 0; j <= K; ++j) cur[j] = buf_ld<T>(rs, t.row[i + 1] + t.col[j]);
                ElemRow<T, K> r;
#pragma unroll
                for (int j = 0; j < K; ++j) {
                    // block_extractor_kernel.cu:73-77, the += contracted to fma as nvcc does by default
                    T s = (t.wxL[j] * t.wyT[i]) * prev[j];
                    s = fma_t<T>(t.wxR[j] * t.wyT[i], prev[j + 1], s);
                    s = fma_t<T>(t.wxL[j] * t.wyB[i], cur[j], s);
                    s = fma_t<T>(t.wxR[j] * t.wyB[i], cur[j + 1], s);
                    r.v[j] = s;
                }
                buf_store_row<T, K>(ro, obase + i * orow, r);
#pragma unroll
                for (int j = 0; j <= K; ++j) prev[j] = cur[j];
            }
        }
    } else {
        // Rare: two neighbouring taps disagree on a floor (fp rounding at an integer boundary).
        // Recompute every tap per element like the reference does; rolled loops keep it small.
        for (int c = c0; c < c1; ++c, sp += splane, op += oplane) {
            const rsrc_t rs = make_rsrc(sp, sbytes);
            const rsrc_t ro = make_rsrc(op, obytes);
#pragma unroll 1
            for (int i = 0; i < K; ++i) {
                const Tap1<T> ty = make_tap<T>(fy0, i - K / 2, tc.yf, Hs);
                const unsigned rT = ty.lo * static_cast<unsigned>(Ws), rB = ty.hi * static_cast<unsigned>(Ws);
#pragma unroll 1
                for (int j = 0; j < K; ++j) {
                    const Tap1<T> tx = make_tap<T>(fx0, j - K / 2, tc.xf, Ws);
                    constexpr unsigned E = sizeof(T);
                    T s = (tx.wlo * ty.wlo) * buf_ld<T>(rs, (rT + tx.lo) * E);
                    s = fma_t<T>(tx.whi * ty.wlo, buf_ld<T>(rs, (rT + tx.hi) * E), s);
                    s = fma_t<T>(tx.wlo * ty.whi, buf_ld<T>(rs, (rB + tx.lo) * E), s);
                    s = fma_t<T>(tx.whi * ty.whi, buf_ld<T>(rs, (rB + tx.hi) * E), s);
                    ElemRow<T, 1> r;
                    r.v[0] = s;
                    buf_store_row<T, 1>(ro, obase + i * orow + j * E, r);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------ forward, LDS-staged
// Same decomposition, but the source window of the whole 64 x 4 pixel tile is staged in LDS:
//   1. the block reduces the bounding box of its taps in UNCLAMPED source coordinates (wave
//      shuffles + one LDS hop);
//   2. if every pixel is "regular" (tap j+1 sits exactly one pixel after tap j, in x and in y) and
//      the box fits kLdsRows x kLdsCols, each channel's box is copied global -> LDS with
//      row-contiguous, fully coalesced loads; border clamping is applied while staging (LDS holds
//      the clamp-extended image), so a pixel's (k+1)^2 neighbourhood is ALWAYS a dense square at
//      one LDS base address + compile-time immediates.  The copy of channel c+1 overlaps the
//      arithmetic of channel c (two buffers, one barrier per channel);
//   3. the neighbourhood reads become ds_read_b32 (128 B/clk/CU) instead of 16 per-lane global
//      gathers per 9 outputs, which is what bounds the direct kernel;
//   otherwise (flow too wide for the tile, or an irregular pixel) the block falls back to direct
//   gathers -- a block-uniform, data-dependent choice.
constexpr int kLdsCols = 128;

// RPT = pixel rows per thread: the block's tile is 64 x (4*RPT) flow pixels.  A taller tile cuts the
// halo re-read of the source ((4*RPT + k + 2*|flow|) / (4*RPT) rows are staged per tile row) and puts
// RPT x more arithmetic and stores between two barriers.
//
// MODE selects what happens to the k x k bilinear samples s_ij of a pixel and channel (the fused
// extractor + attention consumer; see the "block attention" section below):
//   0  block_extractor: the samples are the output                      out [B, C, k Hf, k Wf]
//   1  attention forward: out = (sum_ij s_ij * w_ij) / k^2              out [B, C, Hf, Wf], aux = w [B, k^2, Hf, Wf]
//   2  attention weight gradient: gw_ij += sum_c s_ij * (g_c / k^2)     out = gw [B, k^2, Hf, Wf] (atomic), aux = g [B, C, Hf, Wf]
template <typename T, int K, int RPT, int MODE = 0>
__global__ void __launch_bounds__(kBlock)
be_fwd_lds_kernel(const T* __restrict__ src, const T* __restrict__ flow, T* __restrict__ out, int C,
                  int Hs, int Ws, int Hf, int Wf, int tiles_x, int tiles_y, int cslabs, int cs,
                  int remap, int ablate, int nt, const T* __restrict__ aux = nullptr) {
    constexpr int NW = kBlock / kWave;
    constexpr int LROWS = (RPT == 1) ? 16 : 32;
    constexpr unsigned E = sizeof(T);
    __shared__ T tile[2][LROWS * kLdsCols];
    __shared__ int red[4][NW];
    __shared__ int flag;
    // tile decode (64 x 4*RPT pixels per block)
    unsigned tid = xcd_remap(blockIdx.x, gridDim.x, remap);
    const int tx = tid % tiles_x;
    tid /= tiles_x;
    const int ty = tid % tiles_y;
    tid /= tiles_y;
    const int slab = tid % cslabs;
    const int b = tid / cslabs;
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    const int xf_raw = tx * kTileX + lane;
    const bool inx = xf_raw < Wf;
    const int xf = inx ? xf_raw : Wf - 1;            // out-of-tile lanes shadow a valid pixel
    if (threadIdx.x == 0) flag = 0;

    // ---- per-pixel taps, the reference's arithmetic (block_extractor_kernel.cu:52-71)
    T wxl[RPT][K], wxr[RPT][K], wyt[RPT][K], wyb[RPT][K];
    int u0[RPT], v0[RPT], yfs[RPT];
    bool iny[RPT];
    bool regular = true;
    const size_t fplane = static_cast<size_t>(Hf) * Wf;
    const T* fb = flow + static_cast<size_t>(b) * 2 * fplane;
    int umin = 0x7fffffff, umax = -0x7fffffff, vmin = 0x7fffffff, vmax = -0x7fffffff;
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
        const int yraw = ty * (NW * RPT) + wave + r * NW;
        iny[r] = yraw < Hf;
        const int yf = iny[r] ? yraw : Hf - 1;
        yfs[r] = yf;
        const T fx0 = fb[static_cast<size_t>(yf) * Wf + xf], fy0 = fb[fplane + static_cast<size_t>(yf) * Wf + xf];
        T flx0 = 0, fly0 = 0;
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const T dx = (fx0 + static_cast<T>(j - K / 2)) + static_cast<T>(xf);
            const T dy = (fy0 + static_cast<T>(j - K / 2)) + static_cast<T>(yf);
            const T fxl = floor_t(dx), fyl = floor_t(dy);
            if (j == 0) { flx0 = fxl; fly0 = fyl; }
            // tap j sits j cells after tap 0 -- or its coordinate was rounded up onto the next integer exactly (dx == floor(dx)
            // == that cell + 1: a flow value within an ulp of a cell boundary): then cell + j with the weights (0, 1) is the
            // reference's cell + j + 1 with (1, 0), the same sample bit for bit, and the pixel stays on the LDS path.  (Without
            // this a smooth field whose extrema sit next to integers sent whole blocks to the per-tap fallback.)
            const T cx = flx0 + static_cast<T>(j), cy = fly0 + static_cast<T>(j);
            regular = regular && (fxl == cx || (fxl == cx + 1 && dx == fxl)) && (fyl == cy || (fyl == cy + 1 && dy == fyl));
            wxr[r][j] = dx - cx; wxl[r][j] = 1 - (dx - cx);
            wyb[r][j] = dy - cy; wyt[r][j] = 1 - (dy - cy);
        }
        const T lim = static_cast<T>(1 << 20);
        const bool ok = (flx0 > -lim) && (flx0 < lim) && (fly0 > -lim) && (fly0 < lim);   // also rejects NaN
        regular = regular && ok;
        u0[r] = ok ? static_cast<int>(flx0) : 0;
        v0[r] = ok ? static_cast<int>(fly0) : 0;
        umin = min(umin, u0[r]); umax = max(umax, u0[r] + K);
        vmin = min(vmin, v0[r]); vmax = max(vmax, v0[r] + K);
    }

    // ---- block-wide bounding box of the taps, unclamped coordinates (wave shuffles + one LDS hop)
    umin = wave_min(umin); umax = wave_max(umax); vmin = wave_min(vmin); vmax = wave_max(vmax);
    if (lane == 0) { red[0][wave] = umin; red[1][wave] = umax; red[2][wave] = vmin; red[3][wave] = vmax; }
    __syncthreads();
    if (!regular) flag = 1;            // benign race: every writer stores 1
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        umin = min(umin, red[0][w]); umax = max(umax, red[1][w]);
        vmin = min(vmin, red[2][w]); vmax = max(vmax, red[3][w]);
    }
    __syncthreads();
    const int bw = umax - umin + 1, bh = vmax - vmin + 1;
    const bool use_lds = (flag == 0) && bw <= kLdsCols && bh <= LROWS;

    const int c0 = slab * cs;
    const int c1 = (c0 + cs < C) ? c0 + cs : C;
    const int W = K * Wf;
    const size_t oplane = MODE == 0 ? static_cast<size_t>(K) * Hf * W : fplane;
    const size_t splane = static_cast<size_t>(Hs) * Ws;
    const unsigned sbytes = static_cast<unsigned>(splane * E);
    const unsigned obytes = static_cast<unsigned>(oplane * E);
    const T* sp = src + (static_cast<size_t>(b) * C + c0) * splane;
    T* op = MODE == 2 ? out + static_cast<size_t>(b) * K * K * fplane : out + (static_cast<size_t>(b) * C + c0) * oplane;
    const unsigned orow = static_cast<unsigned>(W) * E;
    // fused modes: per-pixel attention weights (1) / upstream gradient of the current channel (2)
    const T* wb = MODE == 1 ? aux + static_cast<size_t>(b) * K * K * fplane : nullptr;
    const T* gb = MODE == 2 ? aux + (static_cast<size_t>(b) * C + c0) * fplane : nullptr;
    constexpr T kK2 = static_cast<T>(K * K);

    if (use_lds) {
        // staging map: wave w copies box rows w, w+4, ...; lane l copies box columns l, l+64.
        // LDS holds the clamp-extended image: box cell (r, cc) <- src[clamp(vmin+r)][clamp(umin+cc)].
        constexpr int RI = LROWS / NW, CI = kLdsCols / kWave;
        unsigned goff[RI][CI];
#pragma unroll
        for (int ri = 0; ri < RI; ++ri)
#pragma unroll
            for (int ci = 0; ci < CI; ++ci) {
                const int r = wave + ri * NW, cc = lane + ci * kWave;
                const int gy = min(max(vmin + r, 0), Hs - 1), gx = min(max(umin + cc, 0), Ws - 1);
                goff[ri][ci] = (r < bh && cc < bw && !(ablate & 1))
                                   ? (static_cast<unsigned>(gy) * Ws + gx) * E : 0xFFFFFFF0u;   // OOB reads 0
            }
        T stage[RI][CI];
        auto fetch = [&](const T* plane) {
            const rsrc_t rs = make_rsrc(plane, sbytes);
#pragma unroll
            for (int ri = 0; ri < RI; ++ri)
#pragma unroll
                for (int ci = 0; ci < CI; ++ci) stage[ri][ci] = buf_ld<T>(rs, goff[ri][ci]);
        };
        auto commit = [&](T* buf) {     // unconditional: cells outside the box stay inside the buffer
#pragma unroll
            for (int ri = 0; ri < RI; ++ri)
#pragma unroll
                for (int ci = 0; ci < CI; ++ci)
                    buf[(wave + ri * NW) * kLdsCols + lane + ci * kWave] = stage[ri][ci];
        };
        int lbase[RPT];
        unsigned obase[RPT];
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            lbase[r] = (v0[r] - vmin) * kLdsCols + (u0[r] - umin);
            obase[r] = MODE == 0 ? (static_cast<unsigned>(yfs[r]) * K * W + static_cast<unsigned>(xf) * K) * E
                                 : (static_cast<unsigned>(yfs[r]) * Wf + static_cast<unsigned>(xf)) * E;
        }
        if constexpr (MODE != 0) {
            // Fused consumers: the k x k samples are linear in the (k+1)^2 neighbourhood cells n_ab, so
            //   MODE 1  out = sum_ab coef_ab n_ab,  coef_ab = sum_ij (w_ij / k^2) [bilinear weight of cell ab in sample ij]
            //           -- formed ONCE per pixel, then (k+1)^2 LDS reads + fmas per channel instead of 4 k^2 + k^2;
            //   MODE 2  T_ab += (g_c / k^2) n_ab per channel, and gw_ij = sum_ab [weight of ab in ij] T_ab once at the end.
            // (Same sums as the composition, re-associated: ~1e-7 relative.)
            constexpr int NC = (K + 1) * (K + 1);
            T coef[RPT][NC];
#pragma unroll
            for (int r = 0; r < RPT; ++r)
#pragma unroll
                for (int q = 0; q < NC; ++q) coef[r][q] = 0;
            T gcur[RPT], gnxt[RPT];        // MODE 2: g / k^2 of the current / next channel
#pragma unroll
            for (int r = 0; r < RPT; ++r) gcur[r] = gnxt[r] = 0;
            const size_t pixr0 = static_cast<size_t>(xf);
            if constexpr (MODE == 1) {
#pragma unroll
                for (int r = 0; r < RPT; ++r)
#pragma unroll
                    for (int i = 0; i < K; ++i)
#pragma unroll
                        for (int j = 0; j < K; ++j) {
                            const T wij = wb[(i * K + j) * fplane + static_cast<size_t>(yfs[r]) * Wf + pixr0] / kK2;
                            const T wt = wij * wyt[r][i], wbm = wij * wyb[r][i];
                            coef[r][i * (K + 1) + j] = fma_t<T>(wt, wxl[r][j], coef[r][i * (K + 1) + j]);
                            coef[r][i * (K + 1) + j + 1] = fma_t<T>(wt, wxr[r][j], coef[r][i * (K + 1) + j + 1]);
                            coef[r][(i + 1) * (K + 1) + j] = fma_t<T>(wbm, wxl[r][j], coef[r][(i + 1) * (K + 1) + j]);
                            coef[r][(i + 1) * (K + 1) + j + 1] = fma_t<T>(wbm, wxr[r][j], coef[r][(i + 1) * (K + 1) + j + 1]);
                        }
            } else {
#pragma unroll
                for (int r = 0; r < RPT; ++r) gcur[r] = gb[static_cast<size_t>(yfs[r]) * Wf + pixr0] / kK2;
            }
            // Two channels ahead: the box of channel c+2 is requested (into the register set that channel c
            // just left) before channel c is processed, so a fetch has two iterations to land -- one block
            // iteration is only ~(k+1)^2 fmas per pixel, far shorter than the memory latency.
            T stage2[RI][CI];
            auto fetch_to = [&](const T* plane, T (&st)[RI][CI]) {
                const rsrc_t rs = make_rsrc(plane, sbytes);
#pragma unroll
                for (int ri = 0; ri < RI; ++ri)
#pragma unroll
                    for (int ci = 0; ci < CI; ++ci) st[ri][ci] = buf_ld<T>(rs, goff[ri][ci]);
            };
            auto commit_from = [&](T* buf, const T (&st)[RI][CI]) {
#pragma unroll
                for (int ri = 0; ri < RI; ++ri)
#pragma unroll
                    for (int ci = 0; ci < CI; ++ci)
                        buf[(wave + ri * NW) * kLdsCols + lane + ci * kWave] = st[ri][ci];
            };
            fetch_to(sp, stage);
            if (c0 + 1 < c1) fetch_to(sp + splane, stage2);
            commit_from(tile[0], stage);
            __syncthreads();
            int p = 0;
            // `hold` carries channel c+1 (in flight or landed), `spare` is free for channel c+2
            auto iteration = [&](int c, T (&hold)[RI][CI], T (&spare)[RI][CI]) {
                const bool more = c + 1 < c1;
                if constexpr (MODE == 2) {
                    if (more) {
#pragma unroll
                        for (int r = 0; r < RPT; ++r)
                            gnxt[r] = gb[static_cast<size_t>(c + 1 - c0) * fplane + static_cast<size_t>(yfs[r]) * Wf + pixr0] / kK2;
                    }
                }
                if (c + 2 < c1) fetch_to(sp + static_cast<size_t>(c + 2 - c0) * splane, spare);
                const rsrc_t ro = make_rsrc(op, obytes);
#pragma unroll
                for (int r = 0; r < RPT; ++r) {
                    const T* nb = tile[p] + lbase[r];
                    if constexpr (MODE == 1) {
                        T orow[K + 1];
#pragma unroll
                        for (int a = 0; a <= K; ++a) {
                            orow[a] = coef[r][a * (K + 1)] * nb[a * kLdsCols];
#pragma unroll
                            for (int bq = 1; bq <= K; ++bq) orow[a] = fma_t<T>(coef[r][a * (K + 1) + bq], nb[a * kLdsCols + bq], orow[a]);
                        }
                        T o = orow[0];
#pragma unroll
                        for (int a = 1; a <= K; ++a) o += orow[a];
                        if (inx && iny[r]) {
                            ElemRow<T, 1> ov;
                            ov.v[0] = o;
                            buf_store_row<T, 1>(ro, obase[r], ov);
                        }
                    } else {
#pragma unroll
                        for (int a = 0; a <= K; ++a)
#pragma unroll
                            for (int bq = 0; bq <= K; ++bq)
                                coef[r][a * (K + 1) + bq] = fma_t<T>(gcur[r], nb[a * kLdsCols + bq], coef[r][a * (K + 1) + bq]);
                    }
                }
                if constexpr (MODE == 2) {
#pragma unroll
                    for (int r = 0; r < RPT; ++r) gcur[r] = gnxt[r];
                }
                if (more) commit_from(tile[p ^ 1], hold);
                __syncthreads();
                op += (MODE == 2 ? 0 : oplane);
                p ^= 1;
            };
            for (int c = c0; c < c1; c += 2) {
                iteration(c, stage2, stage);
                if (c + 1 < c1) iteration(c + 1, stage, stage2);
            }
            if constexpr (MODE == 2) {
#pragma unroll
                for (int r = 0; r < RPT; ++r)
                    if (inx && iny[r]) {
                        // the tap weights again, from the flow (L2): keeping 4 k RPT of them live across the
                        // channel loop would cost a third of the register file
                        const size_t pix = static_cast<size_t>(yfs[r]) * Wf + pixr0;
                        T fx0 = fb[pix], fy0 = fb[fplane + pix];
                        asm volatile("" : "+v"(fx0), "+v"(fy0));
                        T xr[K], yb2[K];
#pragma unroll
                        for (int j = 0; j < K; ++j) {
                            const T dx = (fx0 + static_cast<T>(j - K / 2)) + static_cast<T>(xf);
                            const T dy = (fy0 + static_cast<T>(j - K / 2)) + static_cast<T>(yfs[r]);
                            xr[j] = dx - floor_t(dx);
                            yb2[j] = dy - floor_t(dy);
                        }
#pragma unroll
                        for (int i = 0; i < K; ++i)
#pragma unroll
                            for (int j = 0; j < K; ++j) {
                                const T* t0 = &coef[r][i * (K + 1) + j];
                                const T xl = 1 - xr[j], yt2 = 1 - yb2[i];
                                T v = (xl * yt2) * t0[0];
                                v = fma_t<T>(xr[j] * yt2, t0[1], v);
                                v = fma_t<T>(xl * yb2[i], t0[K + 1], v);
                                v = fma_t<T>(xr[j] * yb2[i], t0[K + 2], v);
                                atomic_add(op + (i * K + j) * fplane + pix, v);
                            }
                    }
            }
            return;
        }
        fetch(sp);
        commit(tile[0]);
        __syncthreads();
        int p = 0;
        for (int c = c0; c < c1; ++c, op += oplane, p ^= 1) {
            const bool more = c + 1 < c1;
            if (more) fetch(sp + static_cast<size_t>(c + 1 - c0) * splane);   // in flight during the math
            const rsrc_t ro = make_rsrc(op, obytes);
#pragma unroll
            for (int r = 0; r < RPT; ++r) {
                const T* nb = tile[p] + lbase[r];   // dense (K+1) x (K+1) neighbourhood, immediates only
                T prev[K + 1], cur[K + 1];
                T yt[K], yb[K];
#pragma unroll
                for (int i = 0; i < K; ++i) {
                    // opaque copies: stop the compiler from hoisting all 4*K*K weight products out of
                    // the channel loop (it would hold them in 36 * RPT VGPRs and halve the occupancy)
                    yt[i] = wyt[r][i];
                    yb[i] = wyb[r][i];
                    asm volatile("" : "+v"(yt[i]), "+v"(yb[i]));
                }
#pragma unroll
                for (int j = 0; j <= K; ++j) prev[j] = nb[j];
#pragma unroll
                for (int i = 0; i < K; ++i) {
#pragma unroll
                    for (int j = 0; j <= K; ++j) cur[j] = nb[(i + 1) * kLdsCols + j];
                    ElemRow<T, K> row;
#pragma unroll
                    for (int j = 0; j < K; ++j) {
                        // :73-77 -- weight product first, `sample +=` contracted to fma as nvcc does
                        T s = (wxl[r][j] * yt[i]) * prev[j];
                        s = fma_t<T>(wxr[r][j] * yt[i], prev[j + 1], s);
                        s = fma_t<T>(wxl[r][j] * yb[i], cur[j], s);
                        s = fma_t<T>(wxr[r][j] * yb[i], cur[j + 1], s);
                        row.v[j] = s;
                    }
                    if (ablate & 2) {      // ablation: keep the values live, skip the store
#pragma unroll
                        for (int j = 0; j < K; ++j) asm volatile("" ::"v"(row.v[j]));
                    } else if (inx && iny[r]) {
                        if (nt) buf_store_row_nt<T, K>(ro, obase[r] + i * orow, row);
                        else buf_store_row<T, K>(ro, obase[r] + i * orow, row);
                    }
#pragma unroll
                    for (int j = 0; j <= K; ++j) prev[j] = cur[j];
                }
            }
            if (more) commit(tile[p ^ 1]);
            __syncthreads();
        }
        return;
    }

    // ---- fallback: direct global gathers, every tap formed per element like the reference.  Four channels per trip share
    // the taps and keep 16 K loads in flight per lane: ONE block that falls back (a single irregular pixel among its 64 x 4 RPT
    // is enough) used to walk its slab channel by channel with a dependent load group per tap row -- a 340 us tail on a
    // 230 us launch for a smooth flow whose extrema sit next to integers.
    if (!inx) return;
    constexpr int CB = MODE == 0 ? 4 : 2;          // (the attention modes sit at a register-count step: one more VGPR costs a wave)
    for (int c = c0; c < c1; c += CB, sp += CB * splane, op += (MODE == 2 ? 0 : CB * oplane)) {
        const int nb = (c1 - c) < CB ? (c1 - c) : CB;
#pragma unroll 1
        for (int r = 0; r < RPT; ++r) {
            if (!iny[r]) continue;
            const int yf = yfs[r];
            const size_t pix = static_cast<size_t>(yf) * Wf + xf;
            const T fx0 = fb[pix], fy0 = fb[fplane + pix];
            const unsigned ob = (static_cast<unsigned>(yf) * K * W + static_cast<unsigned>(xf) * K) * E;
            T osum[CB], gd[CB];
#pragma unroll
            for (int q = 0; q < CB; ++q) {
                osum[q] = 0;
                gd[q] = 0;
                if constexpr (MODE == 2)
                    if (q < nb) gd[q] = gb[static_cast<size_t>(c - c0 + q) * fplane + pix] / kK2;
            }
            Tap1<T> tx1[K];
#pragma unroll
            for (int j = 0; j < K; ++j) tx1[j] = make_tap<T>(fx0, j - K / 2, xf, Ws);
#pragma unroll 1
            for (int i = 0; i < K; ++i) {
                const Tap1<T> ty1 = make_tap<T>(fy0, i - K / 2, yf, Hs);
                const unsigned rT = ty1.lo * static_cast<unsigned>(Ws), rB = ty1.hi * static_cast<unsigned>(Ws);
                T v[CB][K][4];
#pragma unroll
                for (int q = 0; q < CB; ++q) {
                    if (q >= nb) break;
                    const rsrc_t rs = make_rsrc(sp + static_cast<size_t>(q) * splane, sbytes);
#pragma unroll
                    for (int j = 0; j < K; ++j) {
                        v[q][j][0] = buf_ld<T>(rs, (rT + tx1[j].lo) * E);
                        v[q][j][1] = buf_ld<T>(rs, (rT + tx1[j].hi) * E);
                        v[q][j][2] = buf_ld<T>(rs, (rB + tx1[j].lo) * E);
                        v[q][j][3] = buf_ld<T>(rs, (rB + tx1[j].hi) * E);
                    }
                }
#pragma unroll
                for (int q = 0; q < CB; ++q) {
                    if (q >= nb) break;
                    ElemRow<T, K> row;
#pragma unroll
                    for (int j = 0; j < K; ++j) {
                        T s = (tx1[j].wlo * ty1.wlo) * v[q][j][0];
                        s = fma_t<T>(tx1[j].whi * ty1.wlo, v[q][j][1], s);
                        s = fma_t<T>(tx1[j].wlo * ty1.whi, v[q][j][2], s);
                        s = fma_t<T>(tx1[j].whi * ty1.whi, v[q][j][3], s);
                        row.v[j] = s;
                        if constexpr (MODE == 1) osum[q] = add_rn(osum[q], mul_rn(s, wb[(i * K + j) * fplane + pix]));
                        if constexpr (MODE == 2) atomic_add(op + (i * K + j) * fplane + pix, gd[q] * s);
                    }
                    if constexpr (MODE == 0)
                        buf_store_row<T, K>(make_rsrc(op + static_cast<size_t>(q) * oplane, obytes), ob + i * orow, row);
                }
            }
            if constexpr (MODE == 1) {
#pragma unroll
                for (int q = 0; q < CB; ++q) {
                    if (q >= nb) break;
                    ElemRow<T, 1> o;
                    o.v[0] = osum[q] / kK2;
                    buf_store_row<T, 1>(make_rsrc(op + static_cast<size_t>(q) * oplane, obytes), static_cast<unsigned>(pix) * E, o);
                }
            }
        }
    }
}

// Any kernel_size: one thread per output element (the reference's decomposition, 64-bit safe).
template <typename T>
__global__ void __launch_bounds__(kBlock)
be_fwd_generic(const T* __restrict__ src, const T* __restrict__ flow, T* __restrict__ out,
               int64_t n, int C, int Hs, int Ws, int Hf, int Wf, int k) {
    const int H = k * Hf, W = k * Wf;
    for (int64_t index = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; index < n;
         index += static_cast<int64_t>(gridDim.x) * kBlock) {
        const int x = static_cast<int>(index % W);
        const int y = static_cast<int>((index / W) % H);
        const int64_t bc = index / (static_cast<int64_t>(W) * H);
        const int64_t b = bc / C;
        const int yf = y / k, xf = x / k;
        const size_t fplane = static_cast<size_t>(Hf) * Wf;
        const T* fl = flow + b * 2 * fplane + static_cast<size_t>(yf) * Wf + xf;
        const T dy = (fl[fplane] + static_cast<T>(y % k - k / 2)) + static_cast<T>(yf);
        const T dx = (fl[0] + static_cast<T>(x % k - k / 2)) + static_cast<T>(xf);
        const T flx = floor_t(dx), fly = floor_t(dy);
        const int xL = clamp_index(flx, Ws), xR = clamp_index(flx + 1, Ws);
        const int yT = clamp_index(fly, Hs), yB = clamp_index(fly + 1, Hs);
        const T xLP = 1 - (dx - flx), xRP = dx - flx, yTP = 1 - (dy - fly), yBP = dy - fly;
        const T* sp = src + bc * static_cast<size_t>(Hs) * Ws;
        T s = (xLP * yTP) * sp[static_cast<size_t>(yT) * Ws + xL];
        s = fma_t<T>(xRP * yTP, sp[static_cast<size_t>(yT) * Ws + xR], s);
        s = fma_t<T>(xLP * yBP, sp[static_cast<size_t>(yB) * Ws + xL], s);
        s = fma_t<T>(xRP * yBP, sp[static_cast<size_t>(yB) * Ws + xR], s);
        out[index] = s;
    }
}

// ------------------------------------------------------------------------------ backward
template <typename T, int K>
__global__ void __launch_bounds__(kBlock)
be_bwd_kernel(const T* __restrict__ src, const T* __restrict__ flow, const T* __restrict__ gout,
              T* __restrict__ gsrc, T* __restrict__ gflow, int C, int Hs, int Ws, int Hf, int Wf,
              int tiles_x, int tiles_y, int cslabs, int cs, int remap) {
    const TileCoord tc = decode_tile(tiles_x, tiles_y, cslabs, remap);
    if (tc.xf >= Wf || tc.yf >= Hf) return;
    const size_t fplane = static_cast<size_t>(Hf) * Wf;
    const size_t foff = static_cast<size_t>(tc.b) * 2 * fplane + static_cast<size_t>(tc.yf) * Wf + tc.xf;
    const T fx0 = flow[foff], fy0 = flow[foff + fplane];
    Taps<T, K> t;
    make_taps<T, K>(t, fx0, fy0, tc.xf, tc.yf, Hs, Ws);

    const int c0 = tc.slab * cs;
    const int c1 = (c0 + cs < C) ? c0 + cs : C;
    const int W = K * Wf;
    const size_t oplane = static_cast<size_t>(K) * Hf * W;
    const size_t splane = static_cast<size_t>(Hs) * Ws;
    const unsigned sbytes = static_cast<unsigned>(splane * sizeof(T));
    const unsigned obytes = static_cast<unsigned>(oplane * sizeof(T));
    const size_t soff = (static_cast<size_t>(tc.b) * C + c0) * splane;
    const T* sp = src + soff;
    T* gp = gsrc ? gsrc + soff : nullptr;
    const T* op = gout + (static_cast<size_t>(tc.b) * C + c0) * oplane;
    const unsigned obase = (static_cast<unsigned>(tc.yf) * K * W + static_cast<unsigned>(tc.xf) * K) *
                           static_cast<unsigned>(sizeof(T));
    const unsigned orow = static_cast<unsigned>(W) * static_cast<unsigned>(sizeof(T));
    T gx = 0, gy = 0;

    if (t.consistent) {
        for (int c = c0; c < c1; ++c, sp += splane, op += oplane) {
            const rsrc_t rs = make_rsrc(sp, sbytes);
            const rsrc_t rg = make_rsrc(op, obytes);
            T sprev[K + 1], scur[K + 1], aprev[K + 1], acur[K + 1];
#pragma unroll
            for (int j = 0; j <= K; ++j) {
                sprev[j] = buf_ld<T>(rs, t.row[0] + t.col[j]);
                aprev[j] = 0;
            }
#pragma unroll
            for (int i = 0; i < K; ++i) {
                ElemRow<T, K> g;
                buf_load_row<T, K>(rg, obase + i * orow, g);
#pragma unroll
                for (int j = 0; j <= K; ++j) {
                    scur[j] = buf_ld<T>(rs, t.row[i + 1] + t.col[j]);
                    acur[j] = 0;
                }
#pragma unroll
                for (int j = 0; j < K; ++j) {
                    const T gv = g.v[j];
                    const T xl = t.wxL[j], xr = t.wxR[j], yt = t.wyT[i], yb = t.wyB[i];
                    aprev[j] += gv * xl * yt;          // block_extractor_kernel.cu:158-161
                    aprev[j + 1] += gv * xr * yt;
                    acur[j] += gv * xl * yb;
                    acur[j + 1] += gv * xr * yb;
                    gy += gv * (-xl * sprev[j] - xr * sprev[j + 1] + xl * scur[j] + xr * scur[j + 1]);  // :163
                    gx += gv * (-yt * sprev[j] - yb * scur[j] + yt * sprev[j + 1] + yb * scur[j + 1]);  // :164
                }
                if (gp) {
#pragma unroll
                    for (int j = 0; j <= K; ++j) atomic_add_off(gp, t.row[i] + t.col[j], aprev[j]);
                }
#pragma unroll
                for (int j = 0; j <= K; ++j) {
                    sprev[j] = scur[j];
                    aprev[j] = acur[j];
                }
            }
            if (gp) {
#pragma unroll
                for (int j = 0; j <= K; ++j) atomic_add_off(gp, t.row[K] + t.col[j], aprev[j]);
                gp += splane;
            }
        }
    } else {
        for (int c = c0; c < c1; ++c, sp += splane, op += oplane) {
            const rsrc_t rs = make_rsrc(sp, sbytes);
            const rsrc_t rg = make_rsrc(op, obytes);
#pragma unroll 1
            for (int i = 0; i < K; ++i) {
                constexpr unsigned E = sizeof(T);
                const Tap1<T> ty = make_tap<T>(fy0, i - K / 2, tc.yf, Hs);
                const unsigned rT = ty.lo * static_cast<unsigned>(Ws) * E, rB = ty.hi * static_cast<unsigned>(Ws) * E;
#pragma unroll 1
                for (int j = 0; j < K; ++j) {
                    const Tap1<T> tx = make_tap<T>(fx0, j - K / 2, tc.xf, Ws);
                    const unsigned cL = tx.lo * E, cR = tx.hi * E;
                    const T gv = buf_ld<T>(rg, obase + i * orow + j * E);
                    const T xl = tx.wlo, xr = tx.whi, yt = ty.wlo, yb = ty.whi;
                    const T sTL = buf_ld<T>(rs, rT + cL), sTR = buf_ld<T>(rs, rT + cR);
                    const T sBL = buf_ld<T>(rs, rB + cL), sBR = buf_ld<T>(rs, rB + cR);
                    if (gp) {
                        atomic_add_off(gp, rT + cL, gv * xl * yt);
                        atomic_add_off(gp, rT + cR, gv * xr * yt);
                        atomic_add_off(gp, rB + cL, gv * xl * yb);
                        atomic_add_off(gp, rB + cR, gv * xr * yb);
                    }
                    gy += gv * (-xl * sTL - xr * sTR + xl * sBL + xr * sBR);
                    gx += gv * (-yt * sTL - yb * sBL + yt * sTR + yb * sBR);
                }
            }
            if (gp) gp += splane;
        }
    }
    if (gflow) {
        atomic_add(gflow + foff, gx);            // ch 0 = x   (:168)
        atomic_add(gflow + foff + fplane, gy);   // ch 1 = y   (:167)
    }
}

// d(source) without global atomics for planes that fit LDS (Hs*Ws*sizeof(T) <= 64 KiB -- every
// call the reference itself makes: 1-channel coordinate grids up to 128 x 128, models/losses.py:214-216).
// A block owns `cg` whole (b, c) planes of grad_source in LDS, visits every flow pixel of image b,
// forms each tap exactly as the reference does and accumulates with LDS atomics; the finished planes
// are added to grad_source with plain coalesced stores.  Any kernel_size.
constexpr int kPlaneThreads = 1024;

template <typename T>
__global__ void __launch_bounds__(kPlaneThreads)
be_bwd_src_plane_kernel(const T* __restrict__ flow, const T* __restrict__ gout, T* __restrict__ gsrc, int C,
                        int Hs, int Ws, int Hf, int Wf, int k, int cg, int groups, int nsplit) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    double* acc = reinterpret_cast<double*>(smem_raw);
    unsigned bid = blockIdx.x;
    const int split = bid % nsplit;          // few planes (the reference's 1-channel grids): the pixels are split over nsplit blocks
    bid /= nsplit;
    const int grp = bid % groups;
    const int b = bid / groups;
    const int c0 = grp * cg;
    const int nc = (c0 + cg <= C) ? cg : C - c0;
    const int ncell = Hs * Ws, npix = Hf * Wf;
    const int W = k * Wf;
    const size_t oplane = static_cast<size_t>(k) * Hf * W;
    for (int i = threadIdx.x; i < nc * ncell; i += kPlaneThreads) acc[i] = 0;
    __syncthreads();
    const T* fl = flow + static_cast<size_t>(b) * 2 * npix;
    const T* g0 = gout + (static_cast<size_t>(b) * C + c0) * oplane;
    for (int p = split * kPlaneThreads + threadIdx.x; p < npix; p += kPlaneThreads * nsplit) {
        const int yf = p / Wf, xf = p - yf * Wf;
        const T fx0 = fl[p], fy0 = fl[npix + p];
        for (int i = 0; i < k; ++i) {
            const Tap1<T> ty = make_tap<T>(fy0, i - k / 2, yf, Hs);
            for (int j = 0; j < k; ++j) {
                const Tap1<T> tx = make_tap<T>(fx0, j - k / 2, xf, Ws);
                const size_t go = static_cast<size_t>(yf * k + i) * W + xf * k + j;
                const unsigned oTL = ty.lo * Ws + tx.lo, oTR = ty.lo * Ws + tx.hi;
                const unsigned oBL = ty.hi * Ws + tx.lo, oBR = ty.hi * Ws + tx.hi;
                for (int c = 0; c < nc; ++c) {
                    const T gv = g0[static_cast<size_t>(c) * oplane + go];
                    double* a = acc + c * ncell;
                    lds_add(a + oTL, gv * tx.wlo * ty.wlo);   // block_extractor_kernel.cu:158-161
                    lds_add(a + oTR, gv * tx.whi * ty.wlo);
                    lds_add(a + oBL, gv * tx.wlo * ty.whi);
                    lds_add(a + oBR, gv * tx.whi * ty.whi);
                }
            }
        }
    }
    __syncthreads();
    T* dst = gsrc + (static_cast<size_t>(b) * C + c0) * ncell;
    if (nsplit == 1) {
        for (int i = threadIdx.x; i < nc * ncell; i += kPlaneThreads) dst[i] += static_cast<T>(acc[i]);
    } else {                                 // several blocks share the planes: one global atomic per non-zero cell
        for (int i = threadIdx.x; i < nc * ncell; i += kPlaneThreads) {
            const T v = static_cast<T>(acc[i]);
            if (v != 0) atomic_add(dst + i, v);
        }
    }
}

// ------------------------------------------------------------------------------ backward, owned tiles
// d(source) + d(flow) for planes too large for LDS, without device-scope atomics on grad_source
// (a global float atomic on a multi-XCD part executes at the memory side: the all-atomic kernel
// above runs at ~40 G atomics/s = 114 GB/s on cfg-5).
//
// The source plane is cut into TW x TH tiles (TW = 64 - 2h, TH = RH - 2h).  A block OWNS one tile of
// grad_source for a slab of channels and keeps it in LDS.  It visits the 64 x RH flow pixels of the
// tile grown by a halo of h pixels -- every pixel whose taps can land in the tile when |tap offset|
// <= h -- recomputes their taps, and adds the contributions that fall inside its own tile with LDS
// atomics (ds_add_f32); what falls outside is the neighbouring block's job (it revisits the same
// pixel as part of ITS halo).  The finished tile is added to grad_source with plain coalesced
// read-modify-write rows.  d(flow) of the tile's own pixels accumulates in registers over the channel
// slab (the clamp-extended source box is staged in LDS next to the accumulator, so the (k+1)^2
// neighbourhood is a dense square at one LDS address + immediates, as in the forward kernel).
//
// Exactness for ANY flow: a contribution pixel p -> cell q is taken by the tile kernel iff p lies in
// the 64 x RH region of q's tile; be_bwd_far_kernel (launched first) takes exactly the complement
// with global atomics.  For |flow| < h - 1 the complement is empty and that kernel only reads the
// flow field.
constexpr int kTileRW = 64;

struct TileGeo {
    int TW, TH, h, RH;
};

// Is flow pixel (xf, yf) inside the region of the tile that owns clamped source cell (cu, cv)?
__device__ __forceinline__ bool near_x(int cu, int xf, const TileGeo& g) {
    const int r0 = (cu / g.TW) * g.TW - g.h;
    return xf >= r0 && xf < r0 + kTileRW;
}
__device__ __forceinline__ bool near_y(int cv, int yf, const TileGeo& g) {
    const int r0 = (cv / g.TH) * g.TH - g.h;
    return yf >= r0 && yf < r0 + g.RH;
}

template <int K, int RH, int H>
__global__ void __launch_bounds__(kBlock, (RH == 32 && K <= 3 ? 4 : 2))
be_bwd_tile_kernel(const float* __restrict__ src, const float* __restrict__ flow, const float* __restrict__ gout,
                   float* __restrict__ gsrc, float* __restrict__ gflow, int C, int Hs, int Ws, int Hf, int Wf,
                   int ntx, int nty, int cslabs, int cs, int remap) {
    using T = float;
    constexpr int RW = kTileRW, NW = kBlock / kWave, PPT = RH / NW;
    constexpr int TW = RW - 2 * H, TH = RH - 2 * H;       // owned tile
    constexpr int AP = RW + 2 * H, AH = RH + 2 * H;       // accumulator box = region grown by H again
    constexpr int NA = AP * AH;
    constexpr unsigned E = sizeof(T);
    __shared__ T S[RH * RW];      // clamp-extended source box of the current channel (region coordinates)
    // grad_source accumulator, UNCLAMPED coordinates, origin (x0 - H, y0 - H): every tap of a region
    // pixel whose offset is within +-H lands inside it, so the hot path needs no ownership masks and
    // no clamps -- cells outside the owned TW x TH centre are simply never flushed (the neighbouring
    // block computes them), and cells outside the image are folded onto the border before the flush.
    // DOUBLE on purpose: on gfx950 ds_add_f64 retires a wave in ~9 clk, ds_add_f32 needs ~190
    // (measured, tools/ubench/atomics.hip) -- and the sum is rounded to float once, at the flush.
    __shared__ double A[NA];
    unsigned t = xcd_remap(blockIdx.x, gridDim.x, remap);
    const int tx = t % ntx;
    t /= ntx;
    const int ty = t % nty;
    t /= nty;
    const int slab = t % cslabs;
    const int b = t / cslabs;
    const int x0 = tx * TW - H, y0 = ty * TH - H;          // region / S-box origin (source == flow coordinates)
    const int ax0 = x0 - H, ay0 = y0 - H;                  // A-box origin
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    const int xf = x0 + lane;
    const bool xin = xf >= 0 && xf < Wf;
    const bool xown = lane >= H && lane < H + TW;

    const int c0 = slab * cs;
    const int c1 = (c0 + cs < C) ? c0 + cs : C;
    const int W = K * Wf;
    const size_t oplane = static_cast<size_t>(K) * Hf * W;
    const size_t splane = static_cast<size_t>(Hs) * Ws;
    const size_t fplane = static_cast<size_t>(Hf) * Wf;
    const unsigned sbytes = static_cast<unsigned>(splane * E);
    const unsigned obytes = static_cast<unsigned>(oplane * E);
    const unsigned orow = static_cast<unsigned>(W) * E;
    const T* sp = src + (static_cast<size_t>(b) * C + c0) * splane;
    T* gp = gsrc + (static_cast<size_t>(b) * C + c0) * splane;
    const T* op = gout + (static_cast<size_t>(b) * C + c0) * oplane;
    const rsrc_t rfl = make_rsrc(flow + static_cast<size_t>(b) * 2 * fplane, static_cast<unsigned>(2 * fplane * E));

    // out-of-image accumulator cells fold onto the border cell they clamp to (block-uniform)
    const bool foldL = tx == 0, foldR = tx == (Ws - 1) / TW && Ws - ax0 < AP;
    const bool foldT = ty == 0, foldB = ty == (Hs - 1) / TH && Hs - ay0 < AH;

    // stage one channel's clamp-extended source box (the other resident blocks of the CU cover its
    // latency).  y0w is laundered through an empty asm so the row offsets are recomputed -- one clamp
    // + one multiply-add each -- instead of being hoisted into loop-invariant VGPRs.
    const int gxs = min(max(x0 + lane, 0), Ws - 1);
    auto stage = [&](const T* plane) {
        const rsrc_t rs = make_rsrc(plane, sbytes);
        int y0w = y0 + wave;
        asm volatile("" : "+v"(y0w));
        T st[PPT];
#pragma unroll
        for (int q = 0; q < PPT; ++q) {
            const int gy = min(max(y0w + q * NW, 0), Hs - 1);
            st[q] = buf_ld<T>(rs, (static_cast<unsigned>(gy) * Ws + gxs) * E);
        }
#pragma unroll
        for (int q = 0; q < PPT; ++q) S[(wave + q * NW) * RW + lane] = st[q];
    };
    for (int i = threadIdx.x; i < NA; i += kBlock) A[i] = 0;
    stage(sp);
    __syncthreads();

    T gxa[PPT], gya[PPT];
#pragma unroll
    for (int r = 0; r < PPT; ++r) gxa[r] = gya[r] = 0;

    for (int c = c0; c < c1; ++c, op += oplane) {
        const bool more = c + 1 < c1;
        const rsrc_t rg = make_rsrc(op, obytes);
        const rsrc_t rs = make_rsrc(sp + static_cast<size_t>(c - c0) * splane, sbytes);
        // software pipeline: the flow vector and the k x k grad_output window of pixel row r+1 are
        // requested before row r is processed (their addresses do not depend on the flow), so the
        // ~300 instructions of one row cover the latency of the next one's loads
        struct PixLoad {
            T fx, fy;
            ElemRow<T, K> g[K];
        };
        const int xfc = min(max(xf, 0), Wf - 1);
        auto request = [&](int r, PixLoad& d) {
            int yfc = y0 + wave + r * NW;
            yfc = min(max(yfc, 0), Hf - 1);                    // rows outside the flow image shadow a valid one
            const unsigned fo = (static_cast<unsigned>(yfc) * Wf + xfc) * E;
            d.fx = buf_ld<T>(rfl, fo);
            d.fy = buf_ld<T>(rfl, fo + static_cast<unsigned>(fplane * E));
            const unsigned ob = (static_cast<unsigned>(yfc) * K * W + static_cast<unsigned>(xfc) * K) * E;
#pragma unroll
            for (int i = 0; i < K; ++i) buf_load_row<T, K>(rg, ob + i * orow, d.g[i]);
        };
        PixLoad nxt;
        request(0, nxt);
#pragma unroll 1
        for (int r = 0; r < PPT; ++r) {
            const int row = wave + r * NW;
            const int yf = y0 + row;
            const PixLoad cur = nxt;
            if (r + 1 < PPT) request(r + 1, nxt);
            const bool row_owned = row >= H && row < H + TH && gflow != nullptr;      // wave-uniform
            T gx = 0, gy = 0;
            if (xin && yf >= 0 && yf < Hf) {
                const bool owned = xown && row_owned;
                const T fx0 = cur.fx, fy0 = cur.fy;
                // taps, the reference's arithmetic (block_extractor_kernel.cu:117-135)
                T wxr[K], wyb[K];
                T flx0 = 0, fly0 = 0;
                bool regular = true;
#pragma unroll
                for (int j = 0; j < K; ++j) {
                    const T dx = (fx0 + static_cast<T>(j - K / 2)) + static_cast<T>(xf);
                    const T dy = (fy0 + static_cast<T>(j - K / 2)) + static_cast<T>(yf);
                    const T fxl = floor_t(dx), fyl = floor_t(dy);
                    if (j == 0) { flx0 = fxl; fly0 = fyl; }
                    regular = regular & (fxl == flx0 + static_cast<T>(j)) & (fyl == fly0 + static_cast<T>(j));
                    wxr[j] = dx - fxl;
                    wyb[j] = dy - fyl;
                }
                const T lim = static_cast<T>(1 << 20);
                regular = regular & (flx0 > -lim) & (flx0 < lim) & (fly0 > -lim) & (fly0 < lim);   // rejects NaN too
                const int u0 = regular ? static_cast<int>(flx0) : 0, v0 = regular ? static_cast<int>(fly0) : 0;
                const int au = u0 - ax0, av = v0 - ay0;          // neighbourhood origin in the accumulator box
                const int su = u0 - x0, sv = v0 - y0;            // ... and in the source box
                const bool fit = regular & (static_cast<unsigned>(au) <= static_cast<unsigned>(AP - 1 - K)) &
                                 (static_cast<unsigned>(av) <= static_cast<unsigned>(AH - 1 - K));
                const bool sfit = (static_cast<unsigned>(su) <= static_cast<unsigned>(RW - 1 - K)) &
                                  (static_cast<unsigned>(sv) <= static_cast<unsigned>(RH - 1 - K));
                const unsigned ob = (static_cast<unsigned>(yf) * K * W + static_cast<unsigned>(xf) * K) * E;
                if (fit & (!owned | sfit)) {
                    // hot path: dense (K+1)^2 neighbourhood at one LDS address + immediates, no masks.
                    // A pixel that is not owned reads an arbitrary valid source neighbourhood: its d(flow)
                    // is never written.
                    double* ap = A + av * AP + au;
                    // d(source): the (K+1)^2 contributions are the separable product Wy^T G Wx of the K x K window
                    // (block_extractor_kernel.cu:158-161 summed over the window): columns first (tx[i][c]),
                    // then rows, one accumulator row at a time -> 4K(K+1)/... fmas instead of 6 K^2 operations
                    T xl[K], yt[K];
#pragma unroll
                    for (int j = 0; j < K; ++j) {
                        xl[j] = 1 - wxr[j];
                        yt[j] = 1 - wyb[j];
                    }
                    T tx[K][K + 1];
#pragma unroll
                    for (int i = 0; i < K; ++i) {
#pragma unroll
                        for (int c2 = 0; c2 <= K; ++c2) {
                            T v = 0;
                            if (c2 < K) v = cur.g[i].v[c2] * xl[c2];
                            if (c2 > 0) v = (c2 < K) ? fma_t<T>(cur.g[i].v[c2 - 1], wxr[c2 - 1], v) : cur.g[i].v[c2 - 1] * wxr[c2 - 1];
                            tx[i][c2] = v;
                        }
                    }
#pragma unroll
                    for (int r2 = 0; r2 <= K; ++r2) {
#pragma unroll
                        for (int c2 = 0; c2 <= K; ++c2) {
                            T v = 0;
                            if (r2 < K) v = tx[r2][c2] * yt[r2];
                            if (r2 > 0) v = (r2 < K) ? fma_t<T>(tx[r2 - 1][c2], wyb[r2 - 1], v) : tx[r2 - 1][c2] * wyb[r2 - 1];
                            __hip_atomic_fetch_add(ap + r2 * AP + c2, static_cast<double>(v), __ATOMIC_RELAXED,
                                                   __HIP_MEMORY_SCOPE_WORKGROUP);
                        }
                    }
                    // d(flow) (:163-164), only where this wave's row belongs to the tile (wave-uniform) and for
                    // the lanes that own their pixel: products regrouped into source differences, summed along
                    // the rows (gx) / columns (gy) of the window first
                    if (row_owned) {
                        const T* nb = S + (owned ? sv * RW + su : 0);
                        T sp[K + 1], sc[K + 1];
#pragma unroll
                        for (int j = 0; j <= K; ++j) sp[j] = nb[j];
#pragma unroll
                        for (int i = 0; i < K; ++i) {
#pragma unroll
                            for (int j = 0; j <= K; ++j) sc[j] = nb[(i + 1) * RW + j];
                            T hx = 0, hy = 0;             // sum_j g_ij * (horizontal difference on the top / bottom row)
#pragma unroll
                            for (int j = 0; j < K; ++j) {
                                hx = fma_t<T>(cur.g[i].v[j], sp[j + 1] - sp[j], hx);
                                hy = fma_t<T>(cur.g[i].v[j], sc[j + 1] - sc[j], hy);
                            }
                            gx = fma_t<T>(yt[i], hx, fma_t<T>(wyb[i], hy, gx));
#pragma unroll
                            for (int c2 = 0; c2 <= K; ++c2) gy = fma_t<T>(tx[i][c2], sc[c2] - sp[c2], gy);   // vertical differences
#pragma unroll
                            for (int j = 0; j <= K; ++j) sp[j] = sc[j];
                        }
                    }
                } else {
                    // a tap outside the accumulator box (flow wider than the halo), a floor that disagrees
                    // between neighbouring taps (fp rounding on an integer boundary), NaN or huge flow:
                    // every tap on its own like the reference, clamped cells, ownership tested per cell
#pragma unroll 1
                    for (int i = 0; i < K; ++i) {
                        const Tap1<T> ty1 = make_tap<T>(fy0, i - K / 2, yf, Hs);
#pragma unroll 1
                        for (int j = 0; j < K; ++j) {
                            const Tap1<T> tx1 = make_tap<T>(fx0, j - K / 2, xf, Ws);
                            const T gv = buf_ld<T>(rg, ob + i * orow + j * E);
                            const int cxs[2] = {static_cast<int>(tx1.lo), static_cast<int>(tx1.hi)};
                            const int cys[2] = {static_cast<int>(ty1.lo), static_cast<int>(ty1.hi)};
                            const T wxs[2] = {tx1.wlo, tx1.whi}, wys[2] = {ty1.wlo, ty1.whi};
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const int cx = cxs[q & 1], cy = cys[q >> 1];
                                if (static_cast<unsigned>(cx - tx * TW) < static_cast<unsigned>(TW) &&
                                    static_cast<unsigned>(cy - ty * TH) < static_cast<unsigned>(TH))
                                    __hip_atomic_fetch_add(&A[(cy - ay0) * AP + (cx - ax0)],
                                                           static_cast<double>(gv * wxs[q & 1] * wys[q >> 1]), __ATOMIC_RELAXED,
                                                           __HIP_MEMORY_SCOPE_WORKGROUP);
                            }
                            if (owned) {
                                const unsigned rT = ty1.lo * static_cast<unsigned>(Ws) * E, rB = ty1.hi * static_cast<unsigned>(Ws) * E;
                                const T sTL = buf_ld<T>(rs, rT + tx1.lo * E), sTR = buf_ld<T>(rs, rT + tx1.hi * E);
                                const T sBL = buf_ld<T>(rs, rB + tx1.lo * E), sBR = buf_ld<T>(rs, rB + tx1.hi * E);
                                gy += gv * (-tx1.wlo * sTL - tx1.whi * sTR + tx1.wlo * sBL + tx1.whi * sBR);
                                gx += gv * (-ty1.wlo * sTL - ty1.whi * sBL + ty1.wlo * sTR + ty1.whi * sBR);
                            }
                        }
                    }
                }
            }
            gxa[r] += gx;       // r is wave-uniform: indexed VGPR access (s_set_gpr_idx), no scratch
            gya[r] += gy;
        }
        __syncthreads();                       // every contribution of channel c is in A
        // border tiles: fold the out-of-image cells onto the border cell they clamp to -- columns
        // first (one thread per accumulator row), then rows (one thread per column)
        if (foldL || foldR) {
            if (threadIdx.x < AH) {
                double* arow = A + threadIdx.x * AP;
                if (foldL) {
                    double s = 0;
                    for (int u = 0; u < -ax0; ++u) s += arow[u];
                    arow[-ax0] += s;
                }
                if (foldR) {
                    double s = 0;
                    for (int u = Ws - ax0; u < AP; ++u) s += arow[u];
                    arow[Ws - 1 - ax0] += s;
                }
            }
            __syncthreads();
        }
        if (foldT || foldB) {
            if (threadIdx.x < AP) {
                double* acol = A + threadIdx.x;
                if (foldT) {
                    double s = 0;
                    for (int v = 0; v < -ay0; ++v) s += acol[v * AP];
                    acol[-ay0 * AP] += s;
                }
                if (foldB) {
                    double s = 0;
                    for (int v = Hs - ay0; v < AH; ++v) s += acol[v * AP];
                    acol[(Hs - 1 - ay0) * AP] += s;
                }
            }
            __syncthreads();
        }
        {
            // flush the owned in-image cells (read-modify-write rows of grad_source) and clear the box
            const rsrc_t rq = make_rsrc(gp + static_cast<size_t>(c - c0) * splane, sbytes);
            int tid = threadIdx.x;
            asm volatile("" : "+v"(tid));
#pragma unroll 1
            for (int i0 = 0; i0 < NA; i0 += 4 * kBlock) {
                T old[4];
                unsigned off[4];
                int idx[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    idx[q] = i0 + q * kBlock + tid;
                    const int arow = idx[q] / AP, acol = idx[q] - arow * AP;
                    const int cx = ax0 + acol, cy = ay0 + arow;
                    const bool mine = idx[q] < NA && static_cast<unsigned>(acol - 2 * H) < static_cast<unsigned>(TW) &&
                                      static_cast<unsigned>(arow - 2 * H) < static_cast<unsigned>(TH) && cx >= 0 && cx < Ws &&
                                      cy >= 0 && cy < Hs;
                    off[q] = mine ? (static_cast<unsigned>(cy) * Ws + static_cast<unsigned>(cx)) * E : 0xFFFFFFF0u;
                    old[q] = buf_ld<T>(rq, off[q]);                 // out of range: reads 0
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (idx[q] < NA) {
                        ElemRow<T, 1> v;
                        v.v[0] = old[q] + static_cast<T>(A[idx[q]]);
                        buf_store_row<T, 1>(rq, off[q], v);         // out of range: dropped
                        A[idx[q]] = 0;
                    }
                }
            }
        }
        if (more) stage(sp + static_cast<size_t>(c + 1 - c0) * splane);
        __syncthreads();
    }
    if (gflow) {
#pragma unroll
        for (int r = 0; r < PPT; ++r) {
            const int row = wave + r * NW;
            const int yf = y0 + row;
            if (xin && xown && row >= H && row < H + TH && yf >= 0 && yf < Hf) {
                const size_t fo = static_cast<size_t>(b) * 2 * fplane + static_cast<size_t>(yf) * Wf + xf;
                atomic_add(gflow + fo, gxa[r]);
                atomic_add(gflow + fo + fplane, gya[r]);
            }
        }
    }
}

// FUSED (block attention backward): gout is the gradient of the attention output, [B, C, Hf, Wf], and the
// k x k grad_output window of a pixel is (g / k^2) * w_ij with the attention weights w [B, k^2, Hf, Wf]
// (what avg_pool2d's and the product's backward hand to the extractor) -- formed in registers, never stored.
// FIXED (round 5): the accumulator box holds 32-bit FIXED-POINT cells instead of doubles.  ds_add_u32 retires in half the LDS time of
// ds_add_f64 (4.3 vs 8.6 clk per conflict-free wave instruction, tools/ubench/atomics.hip), a 4-byte cell spreads a wave's lanes over
// twice as many banks, and the box is half the size (more resident blocks).  What makes it safe:
//   * scale: per block and channel, 2^e with |v| 2^e < 2^23 for every contribution below thr = 4 x (the maximum of a 1-in-K^2 sample
//     of the tile's grad_output window values, reduced over the block before the channel's pixels are visited);
//   * a pixel whose own window holds a value >= thr ("big": the sample missed the tail, or NaN / Inf) does not use the box at all:
//     its taps go to grad_source with global float atomics, exactly like a pixel of be_bwd_far2_kernel -- so a bad estimate costs
//     time, never correctness, and non-finite gradients propagate as in the reference;
//   * overflow: every contribution is a convex combination of window values (|v| <= max|g| < thr), i.e. below 2^bits units, and a
//     cell (X, Y) is reached exactly by the `fit` pixels whose neighbourhood origin (u0, v0) lies in [X - K, X] x [Y - K, Y]: (K + 1)^2
//     origins.  `fit` bounds where a pixel's taps land, NOT how far the flow carried it there, so ANY number of the tile's 64 x 32
//     pixels can share an origin (a contracting flow): round 5 assumed <= 144 per cell and wrapped silently beyond.  The
//     block therefore COUNTS, once (the flow is the same for all its channels), the pixels per origin in the accumulator box itself
//     (one ds_add_u32 per pixel), takes the maximum cmax, and sizes the scale for the real population:
//     bits = min(23, 31 - ceil_log2((K + 1)^2) - bitlength(cmax)), so that (K + 1)^2 cmax 2^bits < 2^31.  K = 3: 23 bits up to 15 pixels
//     per origin (a random U[-2, 2) flow reaches 6-8), 22 up to 31, ... 15 when all 2048 pixels collapse onto one cell -- the sum is
//     then that much larger, so its relative error is unchanged.  The border folds (out-of-image cells onto the clamped cell) add up
//     to a whole box in 64-bit and send what does not fit a cell straight to grad_source;
//   * rounding: one unit = 2^-e <= thr / 2^(bits - 1), i.e. <= 1e-6 of the tile's largest gradient per contribution at 23 bits (the
//     reference's own float atomics round each partial sum to 6e-8 of ITS magnitude -- the same order once a cell has a few
//     contributions).
template <int K, int RH, int H, bool FUSED = false, bool ABL = false, int FIXED = 0>          // FIXED: 0 double cells, 1 fixed-point, 2 fixed-point at 5 waves per SIMD
__global__ void __launch_bounds__(kBlock, (RH == 32 && K <= 3 && H <= 4 ? (FIXED == 2 ? 5 : 4) : 2))
be_bwd_tile2_kernel(const float* __restrict__ src, const float* __restrict__ flow, const float* __restrict__ gout,
                   float* __restrict__ gsrc, float* __restrict__ gflow, int C, int Hs, int Ws, int Hf, int Wf,
                   int ntx, int nty, int cslabs, int cs, int remap, const float* __restrict__ attn = nullptr, int ablate_arg = 0,
                   int flush_rmw = 0) {
    using T = float;
    // bench-only ablation (tools/be_bwd_ablate.py; profiles/r04_be_bwd_ablation.txt): 1 = no LDS atomics, 2 = no flush atomics,
    // 4 = no d(flow) arithmetic.  A compile-time zero in the product instantiations.
    const int ablate = ABL ? ablate_arg : 0;
    constexpr int RW = kTileRW, NW = kBlock / kWave, PPT = RH / NW;
    constexpr int TW = RW, TH = RH;                       // the block's flow pixels: no overlap with its neighbours
    constexpr int AP = RW + 2 * H, AH = RH + 2 * H;       // accumulator / source box = tile grown by H
    constexpr int NA = AP * AH;
    constexpr unsigned E = sizeof(T);
    __shared__ T S[NA];           // clamp-extended source box of the current channel (same box as A)
    // grad_source accumulator, UNCLAMPED coordinates, origin (x0 - H, y0 - H): every tap of a tile pixel
    // whose offset is within +-H lands inside it (no masks, no clamps in the hot path); after the channel's
    // pixels are done the out-of-image cells are folded onto the border and every non-zero cell is added
    // to grad_source with ONE global atomic (neighbouring blocks' boxes overlap: their sums meet in memory).
    // Compared with the owned-tile kernel no pixel is visited twice (x1.0 instead of x1.52 pixel visits),
    // at the price of ~1.4 coalesced global atomics per pixel and channel.
    // DOUBLE on purpose: ds_add_f64 ~9 clk per wave, ds_add_f32 ~190 on gfx950 (tools/ubench/atomics.hip).
    using AccT = typename std::conditional<FIXED != 0, int, double>::type;          // (FIXED: int cells)
    __shared__ AccT A[NA];
    __shared__ float red[NW];                              // FIXED: the waves' sampled maxima of the current channel
    unsigned t = xcd_remap(blockIdx.x, gridDim.x, remap);
    const int tx = t % ntx;
    t /= ntx;
    const int ty = t % nty;
    t /= nty;
    const int slab = t % cslabs;
    const int b = t / cslabs;
    const int x0 = tx * TW, y0 = ty * TH;                  // tile origin (source == flow coordinates)
    const int ax0 = x0 - H, ay0 = y0 - H;                  // box origin
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    const int xf = x0 + lane;
    const bool xin = xf >= 0 && xf < Wf;
    const bool xown = true;

    const int c0 = slab * cs;
    const int c1 = (c0 + cs < C) ? c0 + cs : C;
    const int W = K * Wf;
    const size_t splane = static_cast<size_t>(Hs) * Ws;
    const size_t fplane = static_cast<size_t>(Hf) * Wf;
    const size_t oplane = FUSED ? fplane : static_cast<size_t>(K) * Hf * W;
    const unsigned sbytes = static_cast<unsigned>(splane * E);
    const unsigned obytes = static_cast<unsigned>(oplane * E);
    const unsigned orow = static_cast<unsigned>(W) * E;
    const T* sp = src + (static_cast<size_t>(b) * C + c0) * splane;
    T* gp = gsrc + (static_cast<size_t>(b) * C + c0) * splane;
    const T* op = gout + (static_cast<size_t>(b) * C + c0) * oplane;
    const rsrc_t rfl = make_rsrc(flow + static_cast<size_t>(b) * 2 * fplane, static_cast<unsigned>(2 * fplane * E));
    const rsrc_t ratt = make_rsrc(FUSED ? attn + static_cast<size_t>(b) * K * K * fplane : src,
                                  FUSED ? static_cast<unsigned>(K * K * fplane * E) : 0u);

    // out-of-image accumulator cells fold onto the border cell they clamp to (block-uniform)
    const bool inside = ax0 <= Ws - 1 && ay0 <= Hs - 1;   // the box meets the image (else: be_bwd_far2_kernel's job)
    const bool foldL = ax0 < 0, foldR = Ws - ax0 < AP;
    const bool foldT = ay0 < 0, foldB = Hs - ay0 < AH;

    // stage one channel's clamp-extended source box (the other resident blocks of the CU cover its
    // latency).  y0w is laundered through an empty asm so the row offsets are recomputed -- one clamp
    // + one multiply-add each -- instead of being hoisted into loop-invariant VGPRs.
    auto stage = [&](const T* plane) {
        const rsrc_t rs = make_rsrc(plane, sbytes);
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
#pragma unroll 1
        for (int i0 = 0; i0 < NA; i0 += 4 * kBlock) {
            T st[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int idx = i0 + q * kBlock + tid;
                const int arow = idx / AP, acol = idx - arow * AP;
                const int gy = min(max(ay0 + arow, 0), Hs - 1), gx = min(max(ax0 + acol, 0), Ws - 1);
                st[q] = buf_ld<T>(rs, idx < NA ? (static_cast<unsigned>(gy) * Ws + gx) * E : 0xFFFFFFF0u);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int idx = i0 + q * kBlock + tid;
                if (idx < NA) S[idx] = st[q];
            }
        }
    };
    for (int i = threadIdx.x; i < NA; i += kBlock) A[i] = 0;
    stage(sp);
    __syncthreads();

    // FIXED: the population bound (kernel head).  Every `fit` pixel of the tile adds 1 to the box cell of its neighbourhood origin; the
    // block maximum of those counts sizes the fixed-point scale of every channel.  Same `regular` / `fit` arithmetic as the hot loop.
    int fx_bits = 23;
    if constexpr (FIXED != 0) {
        __shared__ int cred[NW];
        if (inside && xin) {
#pragma unroll 1
            for (int r = 0; r < PPT; ++r) {
                const int yf = y0 + wave + r * NW;
                if (yf >= Hf) break;
                const unsigned fo = (static_cast<unsigned>(yf) * Wf + xf) * E;
                const T fx0 = buf_ld<T>(rfl, fo), fy0 = buf_ld<T>(rfl, fo + static_cast<unsigned>(fplane * E));
                T flx0 = 0, fly0 = 0;
                bool regular = true;
#pragma unroll
                for (int j = 0; j < K; ++j) {
                    const T dx = (fx0 + static_cast<T>(j - K / 2)) + static_cast<T>(xf);
                    const T dy = (fy0 + static_cast<T>(j - K / 2)) + static_cast<T>(yf);
                    const T fxl = floor_t(dx), fyl = floor_t(dy);
                    if (j == 0) { flx0 = fxl; fly0 = fyl; }
                    regular = regular & (fxl == flx0 + static_cast<T>(j)) & (fyl == fly0 + static_cast<T>(j));
                }
                const T lim = static_cast<T>(1 << 20);
                regular = regular & (flx0 > -lim) & (flx0 < lim) & (fly0 > -lim) & (fly0 < lim);
                const int au = (regular ? static_cast<int>(flx0) : 0) - ax0, av = (regular ? static_cast<int>(fly0) : 0) - ay0;
                if (regular && static_cast<unsigned>(au) <= static_cast<unsigned>(AP - 1 - K) && static_cast<unsigned>(av) <= static_cast<unsigned>(AH - 1 - K))
                    __hip_atomic_fetch_add(A + av * AP + au, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        __syncthreads();
        int cm = 0;
        for (int i = threadIdx.x; i < NA; i += kBlock) {
            cm = max(cm, static_cast<int>(A[i]));
            A[i] = 0;
        }
        cm = wave_max(cm);
        if (lane == 0) cred[wave] = cm;
        __syncthreads();
        cm = cred[0];
#pragma unroll
        for (int w2 = 1; w2 < NW; ++w2) cm = max(cm, cred[w2]);
        constexpr int KB = (K + 1) * (K + 1) <= 4 ? 2 : ((K + 1) * (K + 1) <= 16 ? 4 : ((K + 1) * (K + 1) <= 32 ? 5 : 6));   // ceil_log2((K + 1)^2), K <= 7
        const int blen = 32 - __clz(cm);                     // cm < 2^blen (cm = 0: 0)
        fx_bits = __builtin_amdgcn_readfirstlane(min(23, 31 - KB - blen));        // block-uniform: a scalar register
    }

    T gxa[PPT], gya[PPT];
#pragma unroll
    for (int r = 0; r < PPT; ++r) gxa[r] = gya[r] = 0;
    // FIXED: one grad_output window value per pixel of this lane's rows (the centre element), as loads in flight: sample_issue() requests
    // them for channel plane `opc`, sample_max() folds them -- called around the flush of the previous channel, which hides their latency
    constexpr int SSTEP = 2;          // every second row of the tile: 1024 samples per tile and channel (FUSED holds two loads per sample)
    T smp[FIXED != 0 ? PPT : 1], smw[(FIXED != 0 && FUSED) ? PPT : 1];
    T m_pre = 0;
    auto sample_issue = [&](const T* opc) {
        if constexpr (FIXED != 0) {
            const rsrc_t rgs = make_rsrc(opc, obytes);
            const int xs = min(max(xf, 0), Wf - 1);
#pragma unroll
            for (int r = 0; r < PPT; r += SSTEP) {
                const int ys = min(max(y0 + wave + r * NW, 0), Hf - 1);
                if constexpr (FUSED) {
                    const unsigned fo = (static_cast<unsigned>(ys) * Wf + xs) * E;
                    smp[r] = buf_ld<T>(rgs, fo);
                    smw[r] = buf_ld<T>(ratt, fo + static_cast<unsigned>((K / 2 * K + K / 2) * fplane * E));
                } else {
                    smp[r] = buf_ld<T>(rgs, (static_cast<unsigned>(ys) * K * W + static_cast<unsigned>(xs) * K) * E + (K / 2) * orow + (K / 2) * E);
                }
            }
        }
    };
    auto sample_max = [&]() {
        if constexpr (FIXED != 0) {
            T m = 0;
#pragma unroll
            for (int r = 0; r < PPT; r += SSTEP) {
                const T v = FUSED ? (smp[r] / static_cast<T>(K * K)) * smw[r] : smp[r];
                m = fmaxf(m, fabsf(v));                     // (fmaxf drops a NaN sample: such a pixel is "big" below)
            }
            m_pre = m;
        }
    };
    sample_issue(op);
    sample_max();

    for (int c = c0; c < c1; ++c, op += oplane) {
        const bool more = c + 1 < c1;
        const rsrc_t rg = make_rsrc(op, obytes);
        const rsrc_t rs = make_rsrc(sp + static_cast<size_t>(c - c0) * splane, sbytes);
        // FIXED: this channel's scale (see the kernel's head) from one window value per pixel -- requested while the previous channel was
        // flushed (`m_pre`), reduced over the block here
        T thr = 0, fx_scale = 0, fx_inv = 0;
        if constexpr (FIXED != 0) {
            T m = wave_max(m_pre);
            if (lane == 0) red[wave] = m;
            __syncthreads();
            T mm = red[0];
#pragma unroll
            for (int w2 = 1; w2 < NW; ++w2) mm = fmaxf(mm, red[w2]);
            thr = 4 * mm;
            int ex = 0;
            (void)frexpf(thr, &ex);                          // thr = f 2^ex, f in [0.5, 1)
            if (!(thr > 0) || !(thr < 1e37f) || ex < -100) {
                thr = 0;                                    // nothing usable (all zeros, Inf): every pixel takes the exact per-tap path
                fx_scale = fx_inv = 1;
            } else {
                fx_scale = ldexpf(1.f, fx_bits - ex);        // |v| < thr = f 2^ex  ->  |v| fx_scale < 2^fx_bits
                fx_inv = ldexpf(1.f, ex - fx_bits);
            }
        }
        // software pipeline: the flow vector and the k x k grad_output window of pixel row r+1 are
        // requested before row r is processed (their addresses do not depend on the flow), so the
        // ~300 instructions of one row cover the latency of the next one's loads
        struct PixLoad {
            T fx, fy, gs;              // gs: FUSED only, the pixel's upstream gradient
            ElemRow<T, K> g[K];        // grad_output window (FUSED: the attention weights until `cur` is formed)
        };
        const int xfc = min(max(xf, 0), Wf - 1);
        auto request = [&](int r, PixLoad& d) {
            int yfc = y0 + wave + r * NW;
            yfc = min(max(yfc, 0), Hf - 1);                    // rows outside the flow image shadow a valid one
            const unsigned fo = (static_cast<unsigned>(yfc) * Wf + xfc) * E;
            d.fx = buf_ld<T>(rfl, fo);
            d.fy = buf_ld<T>(rfl, fo + static_cast<unsigned>(fplane * E));
            if constexpr (FUSED) {
                ElemRow<T, 1> gv;
                buf_load_row_nt<T, 1>(rg, fo, gv);
                d.gs = gv.v[0];
#pragma unroll
                for (int i = 0; i < K; ++i)
#pragma unroll
                    for (int j = 0; j < K; ++j)
                        d.g[i].v[j] = buf_ld<T>(ratt, fo + static_cast<unsigned>((i * K + j) * fplane * E));   // L2-resident
            } else {
                d.gs = 0;
                const unsigned ob = (static_cast<unsigned>(yfc) * K * W + static_cast<unsigned>(xfc) * K) * E;
#pragma unroll
                for (int i = 0; i < K; ++i) buf_load_row_nt<T, K>(rg, ob + i * orow, d.g[i]);   // read exactly once: streaming (nt) loads, -3 %
            }
        };
        PixLoad nxt;
        request(0, nxt);
#pragma unroll 1
        for (int r = 0; r < PPT; ++r) {
            const int row = wave + r * NW;
            const int yf = y0 + row;
            PixLoad cur = nxt;
            if (r + 1 < PPT) request(r + 1, nxt);
            if constexpr (FUSED) {
                const T gd = cur.gs / static_cast<T>(K * K);
#pragma unroll
                for (int i = 0; i < K; ++i)
#pragma unroll
                    for (int j = 0; j < K; ++j) cur.g[i].v[j] = gd * cur.g[i].v[j];
            }
            const bool row_owned = gflow != nullptr;
            T gx = 0, gy = 0;
            if (xin && yf >= 0 && yf < Hf) {
                const bool owned = xown && row_owned;
                const T fx0 = cur.fx, fy0 = cur.fy;
                // taps, the reference's arithmetic (block_extractor_kernel.cu:117-135)
                T wxr[K], wyb[K];
                T flx0 = 0, fly0 = 0;
                bool regular = true;
#pragma unroll
                for (int j = 0; j < K; ++j) {
                    const T dx = (fx0 + static_cast<T>(j - K / 2)) + static_cast<T>(xf);
                    const T dy = (fy0 + static_cast<T>(j - K / 2)) + static_cast<T>(yf);
                    const T fxl = floor_t(dx), fyl = floor_t(dy);
                    if (j == 0) { flx0 = fxl; fly0 = fyl; }
                    regular = regular & (fxl == flx0 + static_cast<T>(j)) & (fyl == fly0 + static_cast<T>(j));
                    wxr[j] = dx - fxl;
                    wyb[j] = dy - fyl;
                }
                const T lim = static_cast<T>(1 << 20);
                regular = regular & (flx0 > -lim) & (flx0 < lim) & (fly0 > -lim) & (fly0 < lim);   // rejects NaN too
                const int u0 = regular ? static_cast<int>(flx0) : 0, v0 = regular ? static_cast<int>(fly0) : 0;
                const int au = u0 - ax0, av = v0 - ay0;          // neighbourhood origin in the accumulator box
                const int su = au, sv = av;                      // ... which is also the source box
                const bool fit = inside & regular & (static_cast<unsigned>(au) <= static_cast<unsigned>(AP - 1 - K)) &
                                 (static_cast<unsigned>(av) <= static_cast<unsigned>(AH - 1 - K));
                const bool sfit = fit;
                const unsigned ob = (static_cast<unsigned>(yf) * K * W + static_cast<unsigned>(xf) * K) * E;
                bool big = false;
                T inv_pix = 1;
                if constexpr (FIXED != 0) {
                    // "big": a window value >= thr, a NaN or an Inf -- one unsigned maximum over the values' magnitude bits (for
                    // non-negative floats the bit patterns order like the values, NaN / Inf patterns lie above every finite one;
                    // thr == 0 makes every pixel big).  A small pixel's window is pre-scaled by 2^e (exact): the contributions come
                    // out as integers-to-be, d(flow) is scaled back once per pixel.
                    unsigned mb = 0;
#pragma unroll
                    for (int i = 0; i < K; ++i)
#pragma unroll
                        for (int j = 0; j < K; ++j) mb = max(mb, __float_as_uint(cur.g[i].v[j]) & 0x7FFFFFFFu);
                    big = mb >= __float_as_uint(thr);
                    const T sc_pix = big ? static_cast<T>(1) : fx_scale;
                    inv_pix = big ? static_cast<T>(1) : fx_inv;
#pragma unroll
                    for (int i = 0; i < K; ++i)
#pragma unroll
                        for (int j = 0; j < K; ++j) cur.g[i].v[j] *= sc_pix;
                }
                if (fit & (!owned | sfit)) {
                    // hot path: dense (K+1)^2 neighbourhood at one LDS address + immediates, no masks.
                    // A pixel that is not owned reads an arbitrary valid source neighbourhood: its d(flow)
                    // is never written.
                    AccT* ap = A + av * AP + au;
                    // d(source): the (K+1)^2 contributions are the separable product Wy^T G Wx of the K x K window
                    // (block_extractor_kernel.cu:158-161 summed over the window): columns first (tx[i][c]),
                    // then rows, one accumulator row at a time -> 4K(K+1)/... fmas instead of 6 K^2 operations
                    T xl[K], yt[K];
#pragma unroll
                    for (int j = 0; j < K; ++j) {
                        xl[j] = 1 - wxr[j];
                        yt[j] = 1 - wyb[j];
                    }
                    T tx[K][K + 1];
#pragma unroll
                    for (int i = 0; i < K; ++i) {
#pragma unroll
                        for (int c2 = 0; c2 <= K; ++c2) {
                            T v = 0;
                            if (c2 < K) v = cur.g[i].v[c2] * xl[c2];
                            if (c2 > 0) v = (c2 < K) ? fma_t<T>(cur.g[i].v[c2 - 1], wxr[c2 - 1], v) : cur.g[i].v[c2 - 1] * wxr[c2 - 1];
                            tx[i][c2] = v;
                        }
                    }
#pragma unroll
                    for (int r2 = 0; r2 <= K; ++r2) {
#pragma unroll
                        for (int c2 = 0; c2 <= K; ++c2) {
                            T v = 0;
                            if (r2 < K) v = tx[r2][c2] * yt[r2];
                            if (r2 > 0) v = (r2 < K) ? fma_t<T>(tx[r2 - 1][c2], wyb[r2 - 1], v) : tx[r2 - 1][c2] * wyb[r2 - 1];
                            if (ablate & 1) { if (v == 12345.f) A[0] = 1; continue; }            // bench-only: no LDS atomics
                            if constexpr (FIXED != 0) {
                                // (a big pixel scatters below instead: the box never sees a value it cannot hold)
                                if (!big)
                                    __hip_atomic_fetch_add(ap + r2 * AP + c2, __float2int_rn(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            } else {
                                __hip_atomic_fetch_add(ap + r2 * AP + c2, static_cast<double>(v), __ATOMIC_RELAXED,
                                                       __HIP_MEMORY_SCOPE_WORKGROUP);
                            }
                        }
                    }
                    if constexpr (FIXED != 0) {
                        if (big) {
                            // the exact per-tap scatter of be_bwd_far2_kernel for this pixel and channel (rare: the sampled maximum
                            // missed this window by more than 4 x, or a non-finite gradient)
                            T* gplane_c = gp + static_cast<size_t>(c - c0) * splane;
#pragma unroll 1
                            for (int i = 0; i < K; ++i) {
                                const Tap1<T> ty1 = make_tap<T>(fy0, i - K / 2, yf, Hs);
#pragma unroll 1
                                for (int j = 0; j < K; ++j) {
                                    const Tap1<T> tx1 = make_tap<T>(fx0, j - K / 2, xf, Ws);
                                    T gv = cur.g[0].v[0];
#pragma unroll
                                    for (int i2 = 0; i2 < K; ++i2)
#pragma unroll
                                        for (int j2 = 0; j2 < K; ++j2)
                                            if (i2 == i && j2 == j) gv = cur.g[i2].v[j2];
                                    const unsigned rT = ty1.lo * static_cast<unsigned>(Ws) * E, rB = ty1.hi * static_cast<unsigned>(Ws) * E;
                                    const unsigned cL = tx1.lo * E, cR = tx1.hi * E;
                                    atomic_add_off(gplane_c, rT + cL, gv * tx1.wlo * ty1.wlo);
                                    atomic_add_off(gplane_c, rT + cR, gv * tx1.whi * ty1.wlo);
                                    atomic_add_off(gplane_c, rB + cL, gv * tx1.wlo * ty1.whi);
                                    atomic_add_off(gplane_c, rB + cR, gv * tx1.whi * ty1.whi);
                                }
                            }
                        }
                    }
                    // d(flow) (:163-164), only where this wave's row belongs to the tile (wave-uniform) and for
                    // the lanes that own their pixel: products regrouped into source differences, summed along
                    // the rows (gx) / columns (gy) of the window first
                    if (row_owned && !(ablate & 4)) {
                        const T* nb = S + (owned ? sv * AP + su : 0);
                        T sp[K + 1], sc[K + 1];
#pragma unroll
                        for (int j = 0; j <= K; ++j) sp[j] = nb[j];
#pragma unroll
                        for (int i = 0; i < K; ++i) {
#pragma unroll
                            for (int j = 0; j <= K; ++j) sc[j] = nb[(i + 1) * AP + j];
                            T hx = 0, hy = 0;             // sum_j g_ij * (horizontal difference on the top / bottom row)
#pragma unroll
                            for (int j = 0; j < K; ++j) {
                                hx = fma_t<T>(cur.g[i].v[j], sp[j + 1] - sp[j], hx);
                                hy = fma_t<T>(cur.g[i].v[j], sc[j + 1] - sc[j], hy);
                            }
                            gx = fma_t<T>(yt[i], hx, fma_t<T>(wyb[i], hy, gx));
#pragma unroll
                            for (int c2 = 0; c2 <= K; ++c2) gy = fma_t<T>(tx[i][c2], sc[c2] - sp[c2], gy);   // vertical differences
#pragma unroll
                            for (int j = 0; j <= K; ++j) sp[j] = sc[j];
                        }
                        if constexpr (FIXED != 0) {
                            gx *= inv_pix;
                            gy *= inv_pix;
                        }
                    }
                } else {
                    // a tap outside the accumulator box (flow wider than the halo), a floor that disagrees
                    // between neighbouring taps (fp rounding on an integer boundary), NaN or huge flow:
                    // every tap on its own like the reference, clamped cells, ownership tested per cell
#pragma unroll 1
                    for (int i = 0; i < K; ++i) {
                        const Tap1<T> ty1 = make_tap<T>(fy0, i - K / 2, yf, Hs);
#pragma unroll 1
                        for (int j = 0; j < K; ++j) {
                            const Tap1<T> tx1 = make_tap<T>(fx0, j - K / 2, xf, Ws);
                            const T gv = FUSED ? (cur.gs / static_cast<T>(K * K)) *
                                                     buf_ld<T>(ratt, (static_cast<unsigned>(yf) * Wf + xf) * E +
                                                                         static_cast<unsigned>((i * K + j) * fplane * E))
                                               : buf_ld<T>(rg, ob + i * orow + j * E);
                            const int cxs[2] = {static_cast<int>(tx1.lo), static_cast<int>(tx1.hi)};
                            const int cys[2] = {static_cast<int>(ty1.lo), static_cast<int>(ty1.hi)};
                            const T wxs[2] = {tx1.wlo, tx1.whi}, wys[2] = {ty1.wlo, ty1.whi};
                            (void)cxs; (void)cys; (void)wxs; (void)wys;      // scattered by be_bwd_far2_kernel
                            if (owned) {
                                const unsigned rT = ty1.lo * static_cast<unsigned>(Ws) * E, rB = ty1.hi * static_cast<unsigned>(Ws) * E;
                                const T sTL = buf_ld<T>(rs, rT + tx1.lo * E), sTR = buf_ld<T>(rs, rT + tx1.hi * E);
                                const T sBL = buf_ld<T>(rs, rB + tx1.lo * E), sBR = buf_ld<T>(rs, rB + tx1.hi * E);
                                gy += gv * (-tx1.wlo * sTL - tx1.whi * sTR + tx1.wlo * sBL + tx1.whi * sBR);
                                gx += gv * (-ty1.wlo * sTL - ty1.whi * sBL + ty1.wlo * sTR + ty1.whi * sBR);
                            }
                        }
                    }
                }
            }
            gxa[r] += gx;       // r is wave-uniform: indexed VGPR access (s_set_gpr_idx), no scratch
            gya[r] += gy;
        }
        if (flush_rmw) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // a "big" pixel's per-tap atomics have reached L2 before any interior cell is read back
        __syncthreads();                       // every contribution of channel c is in A
        // border tiles: fold the out-of-image cells onto the border cell they clamp to -- columns
        // first (one thread per accumulator row), then rows (one thread per column)
        // (FIXED: the sums run in 64-bit -- a fold can collect a large part of the box -- and a total that does not fit the 32-bit cell
        // goes straight to the clamped cell of grad_source)
        using FoldT = typename std::conditional<FIXED != 0, long long, double>::type;
        T* gfold = gp + static_cast<size_t>(c - c0) * splane;
        auto fold_into = [&](AccT& cell, FoldT s, int cy, int cx) {
            if constexpr (FIXED != 0) {
                const long long t = static_cast<long long>(cell) + s;
                if (t >= -2147483647LL && t <= 2147483647LL) {
                    cell = static_cast<int>(t);
                } else {
                    cell = 0;
                    const int gy = min(max(cy, 0), Hs - 1), gx = min(max(cx, 0), Ws - 1);
                    atomic_add(gfold + static_cast<size_t>(gy) * Ws + gx, static_cast<T>(t) * fx_inv);
                }
            } else {
                cell += s;
            }
        };
        if (foldL || foldR) {
            if (threadIdx.x < AH) {
                AccT* arow = A + threadIdx.x * AP;
                const int cy = ay0 + static_cast<int>(threadIdx.x);
                if (foldL) {
                    FoldT s = 0;
                    for (int u = 0; u < -ax0; ++u) s += arow[u];
                    fold_into(arow[-ax0], s, cy, 0);
                }
                if (foldR) {
                    FoldT s = 0;
                    for (int u = Ws - ax0; u < AP; ++u) s += arow[u];
                    fold_into(arow[Ws - 1 - ax0], s, cy, Ws - 1);
                }
            }
            __syncthreads();
        }
        if (foldT || foldB) {
            const int cx = ax0 + static_cast<int>(threadIdx.x);
            if (threadIdx.x < AP && cx >= 0 && cx < Ws) {       // (the out-of-image columns are already part of the border columns)
                AccT* acol = A + threadIdx.x;
                if (foldT) {
                    FoldT s = 0;
                    for (int v = 0; v < -ay0; ++v) s += acol[v * AP];
                    fold_into(acol[-ay0 * AP], s, 0, cx);
                }
                if (foldB) {
                    FoldT s = 0;
                    for (int v = Hs - ay0; v < AH; ++v) s += acol[v * AP];
                    fold_into(acol[(Hs - 1 - ay0) * AP], s, Hs - 1, cx);
                }
            }
            __syncthreads();
        }
        {
            // flush + restage in one sweep over the box: one global atomic per non-zero in-image cell of
            // channel c (then the cell is cleared), and the same cell of channel c+1's clamp-extended
            // source goes into S -- an in-image cell has the same plane offset in both.
            // Round 6 experiment (flush_rmw, option be_bwd_flush = 1, OFF): the INTERIOR cells -- box columns [2H, TW) x rows [2H, TH), i.e.
            // the tile shrunk by H: no other block's box reaches them, the far kernel's atomics are a finished earlier launch -- added by a
            // plain read-modify-write, with ALL old values of a thread's cells requested up front through L2 (sc0 sc1).  Measured on one
            // box, cfg-5 (profiles/r06_be_flush_rmw_negative.txt): 432 -> 577 us random, 414 -> 565 us smooth; block attention 401 -> 538.
            // Like round 4's attempt (loads issued where needed: 597 -> 679): a return-less atomic is ONE write transaction resolved at
            // L2, the read-modify-write a round trip per cell that the four resident blocks do not cover.  The atomics stay.
            T* gplane = gp + static_cast<size_t>(c - c0) * splane;
            const rsrc_t rn = make_rsrc(sp + static_cast<size_t>(more ? c + 1 - c0 : c - c0) * splane, more ? sbytes : 0u);
            const rsrc_t rq = make_rsrc(gplane, sbytes);
            if (more) sample_issue(op + oplane);
            int tid = threadIdx.x;
            asm volatile("" : "+v"(tid));
            constexpr int NCH = (NA + kBlock - 1) / kBlock;
            if (flush_rmw) {
                T old[NCH];
#pragma unroll
                for (int q = 0; q < NCH; ++q) {
                    const int idx = q * kBlock + tid;
                    const int arow = idx / AP, acol = idx - arow * AP;
                    const int cx = ax0 + acol, cy = ay0 + arow;
                    const bool interior = static_cast<unsigned>(acol - 2 * H) < static_cast<unsigned>(TW - 2 * H) &&
                                          static_cast<unsigned>(arow - 2 * H) < static_cast<unsigned>(TH - 2 * H) && cx < Ws && cy < Hs;
                    old[q] = __builtin_bit_cast(T, __builtin_amdgcn_raw_buffer_load_b32(
                        rq, interior ? (static_cast<unsigned>(cy) * Ws + cx) * E : 0xFFFFFFF0u, 0, 17));
                }
#pragma unroll
                for (int q = 0; q < NCH; ++q) {
                    const int idx = q * kBlock + tid;
                    if (idx < NA) {
                        const int arow = idx / AP, acol = idx - arow * AP;
                        const int cx = ax0 + acol, cy = ay0 + arow;
                        const int gy = min(max(cy, 0), Hs - 1), gx = min(max(cx, 0), Ws - 1);
                        const unsigned off = static_cast<unsigned>(gy) * Ws + gx;
                        const bool in_img = gy == cy && gx == cx;
                        const bool interior = static_cast<unsigned>(acol - 2 * H) < static_cast<unsigned>(TW - 2 * H) &&
                                              static_cast<unsigned>(arow - 2 * H) < static_cast<unsigned>(TH - 2 * H);
                        const T nx = buf_ld<T>(rn, off * E);
                        const T v = FIXED != 0 ? static_cast<T>(A[idx]) * fx_inv : static_cast<T>(A[idx]);
                        A[idx] = 0;
                        if (v != 0 && in_img && !(ablate & 2)) {
                            if (interior) gplane[off] = old[q] + v;
                            else atomic_add(gplane + off, v);
                        }
                        if (more) S[idx] = nx;
                    }
                }
            } else {
#pragma unroll 1
            for (int i0 = 0; i0 < NA; i0 += 4 * kBlock) {
                T st[4];
                unsigned off[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int idx = i0 + q * kBlock + tid;
                    const int arow = idx / AP, acol = idx - arow * AP;
                    const int cx = ax0 + acol, cy = ay0 + arow;
                    const int gy = min(max(cy, 0), Hs - 1), gx = min(max(cx, 0), Ws - 1);
                    off[q] = static_cast<unsigned>(gy) * Ws + gx;
                    st[q] = buf_ld<T>(rn, idx < NA ? off[q] * E : 0xFFFFFFF0u);
                    if (gy != cy || gx != cx) off[q] = 0xFFFFFFFFu;          // outside the image: nothing to flush
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int idx = i0 + q * kBlock + tid;
                    if (idx < NA) {
                        const T v = FIXED != 0 ? static_cast<T>(A[idx]) * fx_inv : static_cast<T>(A[idx]);
                        A[idx] = 0;
                        if (v != 0 && off[q] != 0xFFFFFFFFu && !(ablate & 2)) atomic_add(gplane + off[q], v);
                        if (more) S[idx] = st[q];
                    }
                }
            }
            }
            if (more) sample_max();
        }
        __syncthreads();
    }
    if (gflow) {
#pragma unroll
        for (int r = 0; r < PPT; ++r) {
            const int row = wave + r * NW;
            const int yf = y0 + row;
            if (xin && yf >= 0 && yf < Hf) {
                const size_t fo = static_cast<size_t>(b) * 2 * fplane + static_cast<size_t>(yf) * Wf + xf;
                atomic_add(gflow + fo, gxa[r]);
                atomic_add(gflow + fo + fplane, gya[r]);
            }
        }
    }
}

// Complement of be_bwd_tile2_kernel: flow pixels with a tap outside their own tile's box (flow wider than
// the halo, integer-boundary rounding, NaN / huge flow) scatter ALL their taps with global atomics here.
template <typename T, int K, bool FUSED = false>
__global__ void __launch_bounds__(kBlock)
be_bwd_far2_kernel(const T* __restrict__ flow, const T* __restrict__ gout, T* __restrict__ gsrc, int C, int Hs,
                   int Ws, int Hf, int Wf, int tiles_x, int tiles_y, int cslabs, int cs, TileGeo geo,
                   const T* __restrict__ attn = nullptr) {
    const TileCoord tc = decode_tile(tiles_x, tiles_y, cslabs, 0);
    if (tc.yf >= Hf) return;                            // (wave-uniform: a wave is one row)
    const bool live = tc.xf < Wf;                       // lanes past the row's end stay: they share the unfit pixels' work below
    constexpr unsigned E = sizeof(T);
    const size_t fplane = static_cast<size_t>(Hf) * Wf;
    const size_t foff = static_cast<size_t>(tc.b) * 2 * fplane + static_cast<size_t>(tc.yf) * Wf + (live ? tc.xf : 0);
    const T fx0 = flow[foff], fy0 = flow[foff + fplane];
    // the tile kernel's `fit` predicate, same arithmetic
    T flx0 = 0, fly0 = 0;
    bool regular = true;
#pragma unroll
    for (int j = 0; j < K; ++j) {
        const T dx = (fx0 + static_cast<T>(j - K / 2)) + static_cast<T>(tc.xf);
        const T dy = (fy0 + static_cast<T>(j - K / 2)) + static_cast<T>(tc.yf);
        const T fxl = floor_t(dx), fyl = floor_t(dy);
        if (j == 0) { flx0 = fxl; fly0 = fyl; }
        regular = regular & (fxl == flx0 + static_cast<T>(j)) & (fyl == fly0 + static_cast<T>(j));
    }
    const T lim = static_cast<T>(1 << 20);
    regular = regular & (flx0 > -lim) & (flx0 < lim) & (fly0 > -lim) & (fly0 < lim);
    const int u0 = regular ? static_cast<int>(flx0) : 0, v0 = regular ? static_cast<int>(fly0) : 0;
    const int ax0 = (tc.xf / geo.TW) * geo.TW - geo.h, ay0 = (tc.yf / geo.TH) * geo.TH - geo.h;
    const int AP = geo.TW + 2 * geo.h, AH = geo.TH + 2 * geo.h;
    const bool inside = ax0 <= Ws - 1 && ay0 <= Hs - 1;
    const bool fit = inside & regular & (static_cast<unsigned>(u0 - ax0) <= static_cast<unsigned>(AP - 1 - K)) &
                     (static_cast<unsigned>(v0 - ay0) <= static_cast<unsigned>(AH - 1 - K));
    // Unfit pixels are rare (a flow value within an ulp of a cell boundary, a NaN, a wide flow) -- but ONE thread walking its
    // slab's channels x K x K taps with a dependent load each is a 200 us tail on a launch that otherwise ends in 7 us (seen
    // with a smooth field whose extrema sit next to integers).  The wave shares them instead: for every unfit lane in turn, the
    // 64 lanes split that pixel's (channel, tap) pairs, one load + four atomics per pair, all independent.
    const int c0 = tc.slab * cs;
    const int c1 = (c0 + cs < C) ? c0 + cs : C;
    const int W = K * Wf;
    const size_t oplane = FUSED ? fplane : static_cast<size_t>(K) * Hf * W;
    const size_t splane = static_cast<size_t>(Hs) * Ws;
    const unsigned obytes = static_cast<unsigned>(oplane * E);
    const unsigned orow = static_cast<unsigned>(W) * E;
    const int lane = threadIdx.x & (kWave - 1);
    unsigned long long todo = __ballot(live && !fit);
    while (todo) {
        const int src = __ffsll(static_cast<long long>(todo)) - 1;
        todo &= todo - 1;
        const T pfx = __shfl(fx0, src, kWave), pfy = __shfl(fy0, src, kWave);
        const int px = __shfl(tc.xf, src, kWave), py = tc.yf;          // (a wave is 64 consecutive x of one row)
        const unsigned obase = (static_cast<unsigned>(py) * K * W + static_cast<unsigned>(px) * K) * E;
        const unsigned pix = (static_cast<unsigned>(py) * Wf + px) * E;
        for (int e = lane; e < (c1 - c0) * K * K; e += kWave) {
            const int cc = e / (K * K), t = e - cc * (K * K);
            const int i = t / K, j = t - i * K;
            T* gp = gsrc + (static_cast<size_t>(tc.b) * C + c0 + cc) * splane;
            const T* op = gout + (static_cast<size_t>(tc.b) * C + c0 + cc) * oplane;
            const rsrc_t rg = make_rsrc(op, obytes);
            const Tap1<T> ty = make_tap<T>(pfy, i - K / 2, py, Hs);
            const Tap1<T> tx = make_tap<T>(pfx, j - K / 2, px, Ws);
            const T gv = FUSED ? (buf_ld<T>(rg, pix) / static_cast<T>(K * K)) *
                                     attn[(static_cast<size_t>(tc.b) * K * K + i * K + j) * fplane + pix / E]
                               : buf_ld<T>(rg, obase + i * orow + j * E);
            const unsigned rT = ty.lo * static_cast<unsigned>(Ws) * E, rB = ty.hi * static_cast<unsigned>(Ws) * E;
            const unsigned cL = tx.lo * E, cR = tx.hi * E;
            atomic_add_off(gp, rT + cL, gv * tx.wlo * ty.wlo);
            atomic_add_off(gp, rT + cR, gv * tx.whi * ty.wlo);
            atomic_add_off(gp, rB + cL, gv * tx.wlo * ty.whi);
            atomic_add_off(gp, rB + cR, gv * tx.whi * ty.whi);
        }
    }
}

// The complement of the tile kernel: contributions whose flow pixel lies outside the region of the
// destination cell's tile (flow wider than the halo).  One thread per flow pixel; a pixel with no
// such tap leaves after reading its flow vector.
template <typename T, int K>
__global__ void __launch_bounds__(kBlock)
be_bwd_far_kernel(const T* __restrict__ flow, const T* __restrict__ gout, T* __restrict__ gsrc, int C, int Hs,
                  int Ws, int Hf, int Wf, int tiles_x, int tiles_y, int cslabs, int cs, TileGeo geo) {
    const TileCoord tc = decode_tile(tiles_x, tiles_y, cslabs, 0);
    if (tc.xf >= Wf || tc.yf >= Hf) return;
    constexpr unsigned E = sizeof(T);
    const size_t fplane = static_cast<size_t>(Hf) * Wf;
    const size_t foff = static_cast<size_t>(tc.b) * 2 * fplane + static_cast<size_t>(tc.yf) * Wf + tc.xf;
    const T fx0 = flow[foff], fy0 = flow[foff + fplane];
    // per tap: clamped lo/hi cells; a cell is "far" when this pixel is outside its tile's region
    unsigned farx = 0, fary = 0;       // bit 2j = lo cell of tap j, bit 2j+1 = hi cell
#pragma unroll
    for (int j = 0; j < K; ++j) {
        const Tap1<T> tx = make_tap<T>(fx0, j - K / 2, tc.xf, Ws);
        const Tap1<T> ty = make_tap<T>(fy0, j - K / 2, tc.yf, Hs);
        farx |= (near_x(static_cast<int>(tx.lo), tc.xf, geo) ? 0u : 1u) << (2 * j);
        farx |= (near_x(static_cast<int>(tx.hi), tc.xf, geo) ? 0u : 1u) << (2 * j + 1);
        fary |= (near_y(static_cast<int>(ty.lo), tc.yf, geo) ? 0u : 1u) << (2 * j);
        fary |= (near_y(static_cast<int>(ty.hi), tc.yf, geo) ? 0u : 1u) << (2 * j + 1);
    }
    if ((farx | fary) == 0) return;
    const int c0 = tc.slab * cs;
    const int c1 = (c0 + cs < C) ? c0 + cs : C;
    const int W = K * Wf;
    const size_t oplane = static_cast<size_t>(K) * Hf * W;
    const size_t splane = static_cast<size_t>(Hs) * Ws;
    const unsigned obytes = static_cast<unsigned>(oplane * E);
    T* gp = gsrc + (static_cast<size_t>(tc.b) * C + c0) * splane;
    const T* op = gout + (static_cast<size_t>(tc.b) * C + c0) * oplane;
    const unsigned obase = (static_cast<unsigned>(tc.yf) * K * W + static_cast<unsigned>(tc.xf) * K) * E;
    const unsigned orow = static_cast<unsigned>(W) * E;
    for (int c = c0; c < c1; ++c, op += oplane, gp += splane) {
        const rsrc_t rg = make_rsrc(op, obytes);
#pragma unroll 1
        for (int i = 0; i < K; ++i) {
            const Tap1<T> ty = make_tap<T>(fy0, i - K / 2, tc.yf, Hs);
            const unsigned fy2 = (fary >> (2 * i)) & 3u;
#pragma unroll 1
            for (int j = 0; j < K; ++j) {
                const unsigned fx2 = (farx >> (2 * j)) & 3u;
                if ((fx2 | fy2) == 0) continue;
                const Tap1<T> tx = make_tap<T>(fx0, j - K / 2, tc.xf, Ws);
                const T gv = buf_ld<T>(rg, obase + i * orow + j * E);
                const unsigned rT = ty.lo * static_cast<unsigned>(Ws) * E, rB = ty.hi * static_cast<unsigned>(Ws) * E;
                const unsigned cL = tx.lo * E, cR = tx.hi * E;
                if ((fx2 & 1u) | (fy2 & 1u)) atomic_add_off(gp, rT + cL, gv * tx.wlo * ty.wlo);
                if ((fx2 & 2u) | (fy2 & 1u)) atomic_add_off(gp, rT + cR, gv * tx.whi * ty.wlo);
                if ((fx2 & 1u) | (fy2 & 2u)) atomic_add_off(gp, rB + cL, gv * tx.wlo * ty.whi);
                if ((fx2 & 2u) | (fy2 & 2u)) atomic_add_off(gp, rB + cR, gv * tx.whi * ty.whi);
            }
        }
    }
}

template <typename T>
__global__ void __launch_bounds__(kBlock)
be_bwd_generic(const T* __restrict__ src, const T* __restrict__ flow, const T* __restrict__ gout,
               T* __restrict__ gsrc, T* __restrict__ gflow, int64_t n, int C, int Hs, int Ws,
               int Hf, int Wf, int k, const GoStrides gs = GoStrides{0, 0, 0, 0}) {
    const int H = k * Hf, W = k * Wf;
    for (int64_t index = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; index < n;
         index += static_cast<int64_t>(gridDim.x) * kBlock) {
        const int x = static_cast<int>(index % W);
        const int y = static_cast<int>((index / W) % H);
        const int64_t bc = index / (static_cast<int64_t>(W) * H);
        const int64_t b = bc / C;
        const int yf = y / k, xf = x / k;
        const size_t fplane = static_cast<size_t>(Hf) * Wf;
        const size_t foff = b * 2 * fplane + static_cast<size_t>(yf) * Wf + xf;
        const T dy = (flow[foff + fplane] + static_cast<T>(y % k - k / 2)) + static_cast<T>(yf);
        const T dx = (flow[foff] + static_cast<T>(x % k - k / 2)) + static_cast<T>(xf);
        const T flx = floor_t(dx), fly = floor_t(dy);
        const int xL = clamp_index(flx, Ws), xR = clamp_index(flx + 1, Ws);
        const size_t yT = static_cast<size_t>(clamp_index(fly, Hs)) * Ws;
        const size_t yB = static_cast<size_t>(clamp_index(fly + 1, Hs)) * Ws;
        const T xLP = 1 - (dx - flx), xRP = dx - flx, yTP = 1 - (dy - fly), yBP = dy - fly;
        const size_t soff = bc * static_cast<size_t>(Hs) * Ws;
        const T* sp = src + soff;
        const T sTL = sp[yT + xL], sTR = sp[yT + xR], sBL = sp[yB + xL], sBR = sp[yB + xR];
        // (grad_output through its strides when the caller handed them over: all zero = contiguous)
        const T g = gs.x == 0 ? gout[index] : gout[b * gs.b + (bc - b * C) * gs.c + static_cast<long long>(y) * gs.y + static_cast<long long>(x) * gs.x];
        if (gsrc) {
            T* gp = gsrc + soff;
            atomic_add(gp + yT + xL, g * xLP * yTP);
            atomic_add(gp + yT + xR, g * xRP * yTP);
            atomic_add(gp + yB + xL, g * xLP * yBP);
            atomic_add(gp + yB + xR, g * xRP * yBP);
        }
        if (gflow) {
            atomic_add(gflow + foff + fplane, g * (-xLP * sTL - xRP * sTR + xLP * sBL + xRP * sBR));
            atomic_add(gflow + foff, g * (-yTP * sTL - yBP * sBL + yTP * sTR + yBP * sBR));
        }
    }
}

// ------------------------------------------------------------------------------ host
template <typename T>
int launch_fwd(const T* src, const T* flow, T* out, int64_t B, int64_t C, int64_t Hs, int64_t Ws,
               int64_t Hf, int64_t Wf, int k, hipStream_t st) {
    const double bytes = sizeof(T) * static_cast<double>(B) * (C * Hs * Ws + 2.0 * Hf * Wf + static_cast<double>(C) * k * k * Hf * Wf);
    const Geometry g = plan(B, C, Hf, Wf, 16);
    const int remap = options().xcd_remap;
    // an output that cannot stay in L2 / MALL anyway is written with streaming (nt) stores: 4.9 -> 5.7 TB/s on cfg-5
    const int nt = static_cast<double>(sizeof(T)) * B * C * k * k * Hf * Wf >= 64.0 * 1024 * 1024 ? 1 : 0;
    // variant: 0 = auto (LDS-staged for k <= 4, direct gather above), 1 = direct gather, 2 = LDS-staged
    const int variant = options().be_fwd_variant;
#define FFWM_BE_FWD(KK)                                                                            \
    case KK: {                                                                                     \
        const bool lds = KK <= 4 && (variant == 2 || variant == 0);                              \
        LaunchScope ls(lds ? "block_extractor_fwd_lds" : "block_extractor_fwd", st, bytes);        \
        if (lds) {                                                                                 \
            const int rpt = sizeof(T) == 8 ? 1 : (options().rows_per_thread > 0 ? options().rows_per_thread : (Hf >= 64 ? 4 : 1)); \
            const int th = (kBlock / kWave) * (rpt >= 4 ? 4 : (rpt >= 2 ? 2 : 1));                  \
            const int tyl = static_cast<int>((Hf + th - 1) / th);                                  \
            const unsigned gridl = static_cast<unsigned>(B * g.tiles_x * tyl * g.cslabs);           \
            if (rpt >= 4)                                                                          \
                hipLaunchKernelGGL((be_fwd_lds_kernel<float, (KK <= 4 ? KK : 1), 4>), dim3(gridl),  \
                                   dim3(kBlock), 0, st, (const float*)src, (const float*)flow,     \
                                   (float*)out, (int)C, (int)Hs, (int)Ws,                          \
                                   (int)Hf, (int)Wf, g.tiles_x, tyl, g.cslabs, g.cs, remap,        \
                                   options().ablate, nt);                                          \
            else if (rpt >= 2)                                                                     \
                hipLaunchKernelGGL((be_fwd_lds_kernel<float, (KK <= 4 ? KK : 1), 2>), dim3(gridl),  \
                                   dim3(kBlock), 0, st, (const float*)src, (const float*)flow,     \
                                   (float*)out, (int)C, (int)Hs, (int)Ws,                          \
                                   (int)Hf, (int)Wf, g.tiles_x, tyl, g.cslabs, g.cs, remap,        \
                                   options().ablate, nt);                                          \
            else                                                                                   \
                hipLaunchKernelGGL((be_fwd_lds_kernel<T, (KK <= 4 ? KK : 1), 1>), dim3(gridl),      \
                                   dim3(kBlock), 0, st, src, flow, out, (int)C, (int)Hs, (int)Ws,  \
                                   (int)Hf, (int)Wf, g.tiles_x, tyl, g.cslabs, g.cs, remap,        \
                                   options().ablate, nt);                                          \
        } else                                                                                       \
            hipLaunchKernelGGL((be_fwd_kernel<T, KK>), dim3(g.grid), dim3(kBlock), 0, st, src,     \
                               flow, out, (int)C, (int)Hs, (int)Ws, (int)Hf, (int)Wf, g.tiles_x,   \
                               g.tiles_y, g.cslabs, g.cs, remap);                                  \
    } break;
    const bool generic = options().be_fwd_variant == 9;
    switch (generic ? 0 : k) {
        FFWM_BE_FWD(1) FFWM_BE_FWD(2) FFWM_BE_FWD(3) FFWM_BE_FWD(4) FFWM_BE_FWD(5) FFWM_BE_FWD(6)
        FFWM_BE_FWD(7)
        default: {
            const int64_t n = B * C * k * Hf * k * Wf;
            const unsigned grid = static_cast<unsigned>(n / kBlock + 1 < 16384 ? n / kBlock + 1 : 16384);
            LaunchScope ls("block_extractor_fwd_generic", st, bytes);
            hipLaunchKernelGGL((be_fwd_generic<T>), dim3(grid), dim3(kBlock), 0, st, src, flow, out,
                               n, (int)C, (int)Hs, (int)Ws, (int)Hf, (int)Wf, k);
        }
    }
#undef FFWM_BE_FWD
    return check_launch("ffwm_block_extractor_forward");
}

template <typename T>
int launch_bwd(const T* src, const T* flow, const T* gout, T* gsrc, T* gflow, int64_t B, int64_t C,
               int64_t Hs, int64_t Ws, int64_t Hf, int64_t Wf, int k, hipStream_t st) {
    const double bytes = sizeof(T) * static_cast<double>(B) * (static_cast<double>(C) * k * k * Hf * Wf + 2.0 * C * Hs * Ws + 4.0 * Hf * Wf);
    const size_t plane_lds = static_cast<size_t>(Hs) * Ws * sizeof(double);     // the LDS accumulator is double
    if (gsrc && plane_lds <= 131072 && options().scatter_variant != 1 && options().be_bwd_variant != 9) {
        int cg = static_cast<int>(131072 / plane_lds);
        if (cg > C) cg = static_cast<int>(C);
        while (cg > 1 && B * ((C + cg - 1) / cg) < 512) cg = (cg + 1) / 2;
        const int groups = static_cast<int>((C + cg - 1) / cg);
        int nsplit = 1;
        while (B * groups * nsplit < 256 && nsplit * 2 * kPlaneThreads <= Hf * Wf) nsplit *= 2;
        {
            LaunchScope ls("block_extractor_bwd_src_plane", st,
                           sizeof(T) * static_cast<double>(B) * (static_cast<double>(C) * k * k * Hf * Wf + 2.0 * C * Hs * Ws + 2.0 * Hf * Wf));
            allow_large_lds(reinterpret_cast<const void*>(be_bwd_src_plane_kernel<T>));
            hipLaunchKernelGGL((be_bwd_src_plane_kernel<T>), dim3(static_cast<unsigned>(B * groups * nsplit)), dim3(kPlaneThreads),
                               static_cast<size_t>(cg) * plane_lds, st, flow, gout, gsrc, (int)C, (int)Hs, (int)Ws,
                               (int)Hf, (int)Wf, k, cg, groups, nsplit);
        }
        if (int rc = check_launch("ffwm_block_extractor_backward(source, plane)")) return rc;
        if (!gflow) return FFWM_OK;
        gsrc = nullptr;    // the pixel-major kernel below now only produces d(flow)
    }
    const int remap = options().xcd_remap;
    // owned-tile path: float, k <= 4, grad_source wanted, plane too large for the LDS-plane kernel
    // be_bwd_variant: 0 = auto (3), 1 = all-atomic pixel kernel, 2 = owned tiles (halo revisits, plain stores),
    //                 3 = shared-cell tiles (no revisits, atomic flush), 9 = generic
    if constexpr (sizeof(T) == 4) {
        const int variant = options().be_bwd_variant;
        if (gsrc && k >= 1 && k <= 4 && (variant == 0 || variant == 2 || variant == 3)) {
            // halo 4 (|tap offset| <= 4 stays on the fast path) or 8; region height 32 or 64 rows
            const int h = options().be_bwd_halo > 4 ? 8 : 4;
            const bool shared_cells = variant == 0 || variant == 3;      // tiles without halo revisits + atomic flush
            const int RH = (!shared_cells && h == 4 && k == 3 && options().be_bwd_rows == 64) ? 64 : 32;
            const TileGeo geo = shared_cells ? TileGeo{kTileRW, RH, h, RH} : TileGeo{kTileRW - 2 * h, RH - 2 * h, h, RH};
            const int ntx = static_cast<int>(((Ws > Wf ? Ws : Wf) + geo.TW - 1) / geo.TW);
            const int nty = static_cast<int>(((Hs > Hf ? Hs : Hf) + geo.TH - 1) / geo.TH);
            int cs = options().channel_slab > 0 ? options().channel_slab : 4;
            if (cs > C) cs = static_cast<int>(C);
            while (cs > 4 && B * ntx * nty * ((C + cs - 1) / cs) < 1536) cs = (cs + 1) / 2;   // >= 6 blocks per CU
            const int cslabs = static_cast<int>((C + cs - 1) / cs);
            const Geometry gf = plan(B, C, Hf, Wf, 32);
            {
                LaunchScope ls("block_extractor_bwd_far", st, sizeof(T) * 2.0 * B * Hf * Wf);
                switch (k) {
#define FFWM_BE_FAR(KK)                                                                                       \
    case KK:                                                                                                  \
        if (shared_cells)                                                                                     \
            hipLaunchKernelGGL((be_bwd_far2_kernel<float, KK>), dim3(gf.grid), dim3(kBlock), 0, st, (const float*)flow, \
                               (const float*)gout, (float*)gsrc, (int)C, (int)Hs, (int)Ws, (int)Hf, (int)Wf,  \
                               gf.tiles_x, gf.tiles_y, gf.cslabs, gf.cs, geo);                                \
        else                                                                                                  \
            hipLaunchKernelGGL((be_bwd_far_kernel<float, KK>), dim3(gf.grid), dim3(kBlock), 0, st, (const float*)flow, \
                               (const float*)gout, (float*)gsrc, (int)C, (int)Hs, (int)Ws, (int)Hf, (int)Wf,  \
                               gf.tiles_x, gf.tiles_y, gf.cslabs, gf.cs, geo);                                \
        break;
                    FFWM_BE_FAR(1) FFWM_BE_FAR(2) FFWM_BE_FAR(3) FFWM_BE_FAR(4)
#undef FFWM_BE_FAR
                }
            }
            if (int rc = check_launch("ffwm_block_extractor_backward(far)")) return rc;
            {
                LaunchScope ls(shared_cells ? "block_extractor_bwd_tile2" : "block_extractor_bwd_tile", st, bytes);
                const unsigned grid = static_cast<unsigned>(B * ntx * nty * cslabs);
                const bool fixed_cells = options().be_bwd_fixed != 2;        // 32-bit fixed-point accumulator cells (round 5); 2 = the double cells of rounds 2-4
                const int flush_rmw = options().be_bwd_flush == 1 ? 1 : 0;   // 1 = interior cells by read-modify-write (round 6 experiment: SLOWER, profiles/r06_be_flush_rmw_negative.txt); 0 = every cell by a global atomic
#define FFWM_BE_TILE(KERNEL, KK, RR, HH)                                                                      \
    hipLaunchKernelGGL((KERNEL<KK, RR, HH>), dim3(grid), dim3(kBlock), 0, st, (const float*)src,              \
                       (const float*)flow, (const float*)gout, (float*)gsrc, (float*)gflow, (int)C, (int)Hs,  \
                       (int)Ws, (int)Hf, (int)Wf, ntx, nty, cslabs, cs, remap)
#define FFWM_BE_TILE2(KK, RR, HH)                                                                             \
    hipLaunchKernelGGL((be_bwd_tile2_kernel<KK, RR, HH>), dim3(grid), dim3(kBlock), 0, st, (const float*)src, \
                       (const float*)flow, (const float*)gout, (float*)gsrc, (float*)gflow, (int)C, (int)Hs,  \
                       (int)Ws, (int)Hf, (int)Wf, ntx, nty, cslabs, cs, remap, (const float*)nullptr, 0, flush_rmw)
#define FFWM_BE_TILE2F(KK, RR, HH)                                                                            \
    hipLaunchKernelGGL((be_bwd_tile2_kernel<KK, RR, HH, false, false, 1>), dim3(grid), dim3(kBlock), 0, st, (const float*)src, \
                       (const float*)flow, (const float*)gout, (float*)gsrc, (float*)gflow, (int)C, (int)Hs,  \
                       (int)Ws, (int)Hf, (int)Wf, ntx, nty, cslabs, cs, remap, (const float*)nullptr, 0, flush_rmw)
#define FFWM_BE_TILE_K(KK)                                                                                    \
    case KK:                                                                                                  \
        if (shared_cells) {                                                                                   \
            if (h == 8) FFWM_BE_TILE2(KK, 32, 8);                                                             \
            else if (fixed_cells) FFWM_BE_TILE2F(KK, 32, 4);                                                  \
            else FFWM_BE_TILE2(KK, 32, 4);                                                                    \
        } else {                                                                                              \
            if (h == 8) FFWM_BE_TILE(be_bwd_tile_kernel, KK, 32, 8);                                          \
            else FFWM_BE_TILE(be_bwd_tile_kernel, KK, 32, 4);                                                 \
        }                                                                                                     \
        break;
                if (RH == 64) {
                    FFWM_BE_TILE(be_bwd_tile_kernel, 3, 64, 4);
                } else if (shared_cells && k == 3 && h == 4 && options().ablate != 0) {
                    if (fixed_cells)
                        hipLaunchKernelGGL((be_bwd_tile2_kernel<3, 32, 4, false, true, 1>), dim3(grid), dim3(kBlock), 0, st, (const float*)src,
                                           (const float*)flow, (const float*)gout, (float*)gsrc, (float*)gflow, (int)C, (int)Hs, (int)Ws, (int)Hf,
                                           (int)Wf, ntx, nty, cslabs, cs, remap, (const float*)nullptr, options().ablate, flush_rmw);
                    else
                        hipLaunchKernelGGL((be_bwd_tile2_kernel<3, 32, 4, false, true, 0>), dim3(grid), dim3(kBlock), 0, st, (const float*)src,
                                           (const float*)flow, (const float*)gout, (float*)gsrc, (float*)gflow, (int)C, (int)Hs, (int)Ws, (int)Hf,
                                           (int)Wf, ntx, nty, cslabs, cs, remap, (const float*)nullptr, options().ablate, flush_rmw);
                } else {
                    switch (k) { FFWM_BE_TILE_K(1) FFWM_BE_TILE_K(2) FFWM_BE_TILE_K(3) FFWM_BE_TILE_K(4) }
                }
#undef FFWM_BE_TILE_K
#undef FFWM_BE_TILE2F
#undef FFWM_BE_TILE2
#undef FFWM_BE_TILE
            }
            return check_launch("ffwm_block_extractor_backward(tile)");
        }
    }
    const Geometry g = plan(B, C, Hf, Wf, 32);
#define FFWM_BE_BWD(KK)                                                                            \
    case KK: {                                                                                     \
        LaunchScope ls("block_extractor_bwd", st, bytes);                                          \
        hipLaunchKernelGGL((be_bwd_kernel<T, KK>), dim3(g.grid), dim3(kBlock), 0, st, src, flow,   \
                           gout, gsrc, gflow, (int)C, (int)Hs, (int)Ws, (int)Hf, (int)Wf,          \
                           g.tiles_x, g.tiles_y, g.cslabs, g.cs, remap);                           \
    } break;
    const bool generic = options().be_bwd_variant == 9;
    switch (generic ? 0 : k) {
        FFWM_BE_BWD(1) FFWM_BE_BWD(2) FFWM_BE_BWD(3) FFWM_BE_BWD(4) FFWM_BE_BWD(5) FFWM_BE_BWD(6)
        FFWM_BE_BWD(7)
        default: {
            const int64_t n = B * C * k * Hf * k * Wf;
            const unsigned grid = static_cast<unsigned>(n / kBlock + 1 < 16384 ? n / kBlock + 1 : 16384);
            LaunchScope ls("block_extractor_bwd_generic", st, bytes);
            hipLaunchKernelGGL((be_bwd_generic<T>), dim3(grid), dim3(kBlock), 0, st, src, flow, gout,
                               gsrc, gflow, n, (int)C, (int)Hs, (int)Ws, (int)Hf, (int)Wf, k);
        }
    }
#undef FFWM_BE_BWD
    return check_launch("ffwm_block_extractor_backward");
}

int check_dims(const char* fn, int64_t B, int64_t C, int64_t Hs, int64_t Ws, int64_t Hf, int64_t Wf,
               int k, int dtype) {
    FFWM_REQUIRE(dtype_ok(dtype), FFWM_ERR_DTYPE, "%s: dtype %d is not FFWM_F32/FFWM_F64", fn, dtype);
    FFWM_REQUIRE(B > 0 && C > 0 && Hs > 0 && Ws > 0 && Hf > 0 && Wf > 0 && k >= 1, FFWM_ERR_ARG,
                 "%s: sizes must be positive (B=%lld C=%lld Hs=%lld Ws=%lld Hf=%lld Wf=%lld k=%d)", fn,
                 (long long)B, (long long)C, (long long)Hs, (long long)Ws, (long long)Hf, (long long)Wf, k);
    FFWM_REQUIRE(Hs * Ws < (1LL << 29) && static_cast<int64_t>(k) * Hf * k * Wf < (1LL << 29), FFWM_ERR_SIZE,
                 "%s: a single H*W plane must stay below 2^29 elements (32-bit byte offsets)", fn);
    const int64_t spatial = B * ((Wf + kTileX - 1) / kTileX) * ((Hf + kTileY - 1) / kTileY);
    FFWM_REQUIRE(spatial * C < (1LL << 31), FFWM_ERR_SIZE, "%s: grid too large", fn);
    return FFWM_OK;
}


// ------------------------------------------------------------------------------ block attention
// The fused extractor + attention consumer: for the GFLA-style local attention the
// cfg-5 shape models,
//     out = avg_pool2d(BlockExtractor(source, flow, k) * LocalAttnReshape(weights, k), k, k)
// never needs the k^2-fold expanded tensors: out[b,c,y,x] = (sum_ij s_ij(c) * w_ij) / k^2 with the k x k
// bilinear samples s_ij of pixel (y, x) and its k^2 attention weights w[b, i k + j, y, x].
// Round 6 (fp32, k = 3): by linearity in the channel-independent window (section further down)
//   forward              ba_fwd_pix_kernel: coefficients Wy^T (w / k^2) Wx per pixel, boxes channel-innermost, one ds_read_b128 per cell
//   d(source)            be_bwd_far2_kernel<.., FUSED> (pixels outside their box) + ba_bwd_src_kernel
//   d(flow), d(weights)  ba_bwd_pix_kernel
// Rounds 3-5's kernels: be_fwd_lds_kernel<.., MODE 1> is still the forward behind option ba_fwd_pix = 0; be_bwd_tile2_kernel<.., FUSED> and
// be_fwd_lds_kernel<.., MODE 2> (d(source) + d(flow); d(weights)) are no longer instantiated -- the branches stay in the templates.
// Everything else (float64, k != 3) runs the literal per-pixel kernels below.
template <typename T>
__global__ void __launch_bounds__(kBlock)
ba_fwd_generic(const T* __restrict__ src, const T* __restrict__ flow, const T* __restrict__ wts, T* __restrict__ out,
               int64_t n, int C, int Hs, int Ws, int Hf, int Wf, int k) {
    const size_t fplane = static_cast<size_t>(Hf) * Wf;
    for (int64_t index = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; index < n;
         index += static_cast<int64_t>(gridDim.x) * kBlock) {
        const int xf = static_cast<int>(index % Wf);
        const int yf = static_cast<int>((index / Wf) % Hf);
        const int64_t bc = index / static_cast<int64_t>(fplane);
        const int64_t b = bc / C;
        const size_t pix = static_cast<size_t>(yf) * Wf + xf;
        const T fx0 = flow[b * 2 * fplane + pix], fy0 = flow[b * 2 * fplane + fplane + pix];
        const T* sp = src + bc * static_cast<size_t>(Hs) * Ws;
        const T* wp = wts + b * k * k * fplane + pix;
        T osum = 0;
        for (int i = 0; i < k; ++i) {
            const Tap1<T> ty = make_tap<T>(fy0, i - k / 2, yf, Hs);
            const size_t rT = static_cast<size_t>(ty.lo) * Ws, rB = static_cast<size_t>(ty.hi) * Ws;
            for (int j = 0; j < k; ++j) {
                const Tap1<T> tx = make_tap<T>(fx0, j - k / 2, xf, Ws);
                T s = (tx.wlo * ty.wlo) * sp[rT + tx.lo];
                s = fma_t<T>(tx.whi * ty.wlo, sp[rT + tx.hi], s);
                s = fma_t<T>(tx.wlo * ty.whi, sp[rB + tx.lo], s);
                s = fma_t<T>(tx.whi * ty.whi, sp[rB + tx.hi], s);
                osum = add_rn(osum, mul_rn(s, wp[(i * k + j) * fplane]));
            }
        }
        out[index] = osum / static_cast<T>(k * k);
    }
}

template <typename T>
__global__ void __launch_bounds__(kBlock)
ba_bwd_generic(const T* __restrict__ src, const T* __restrict__ flow, const T* __restrict__ wts,
               const T* __restrict__ gout, T* __restrict__ gsrc, T* __restrict__ gflow, T* __restrict__ gw, int64_t n,
               int C, int Hs, int Ws, int Hf, int Wf, int k) {
    const size_t fplane = static_cast<size_t>(Hf) * Wf;
    for (int64_t index = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; index < n;
         index += static_cast<int64_t>(gridDim.x) * kBlock) {
        const int xf = static_cast<int>(index % Wf);
        const int yf = static_cast<int>((index / Wf) % Hf);
        const int64_t bc = index / static_cast<int64_t>(fplane);
        const int64_t b = bc / C;
        const size_t pix = static_cast<size_t>(yf) * Wf + xf;
        const size_t foff = b * 2 * fplane + pix;
        const T fx0 = flow[foff], fy0 = flow[foff + fplane];
        const size_t soff = bc * static_cast<size_t>(Hs) * Ws;
        const T* sp = src + soff;
        const T* wp = wts + b * k * k * fplane + pix;
        const T gd = gout[index] / static_cast<T>(k * k);
        T gx = 0, gy = 0;
        for (int i = 0; i < k; ++i) {
            const Tap1<T> ty = make_tap<T>(fy0, i - k / 2, yf, Hs);
            const size_t rT = static_cast<size_t>(ty.lo) * Ws, rB = static_cast<size_t>(ty.hi) * Ws;
            for (int j = 0; j < k; ++j) {
                const Tap1<T> tx = make_tap<T>(fx0, j - k / 2, xf, Ws);
                const T sTL = sp[rT + tx.lo], sTR = sp[rT + tx.hi], sBL = sp[rB + tx.lo], sBR = sp[rB + tx.hi];
                const T g = gd * wp[(i * k + j) * fplane];
                if (gsrc) {
                    T* gp = gsrc + soff;
                    atomic_add(gp + rT + tx.lo, g * tx.wlo * ty.wlo);
                    atomic_add(gp + rT + tx.hi, g * tx.whi * ty.wlo);
                    atomic_add(gp + rB + tx.lo, g * tx.wlo * ty.whi);
                    atomic_add(gp + rB + tx.hi, g * tx.whi * ty.whi);
                }
                gy += g * (-tx.wlo * sTL - tx.whi * sTR + tx.wlo * sBL + tx.whi * sBR);
                gx += g * (-ty.wlo * sTL - ty.whi * sBL + ty.wlo * sTR + ty.whi * sBR);
                if (gw) {
                    T s = (tx.wlo * ty.wlo) * sTL;
                    s = fma_t<T>(tx.whi * ty.wlo, sTR, s);
                    s = fma_t<T>(tx.wlo * ty.whi, sBL, s);
                    s = fma_t<T>(tx.whi * ty.whi, sBR, s);
                    atomic_add(gw + b * k * k * fplane + (i * k + j) * fplane + pix, gd * s);
                }
            }
        }
        if (gflow) {
            atomic_add(gflow + foff, gx);
            atomic_add(gflow + foff + fplane, gy);
        }
    }
}

// ------------------------------------------------------------------------------ block attention backward by linearity (round 6)
// be_bwd_tile2_kernel<.., FUSED> + be_fwd_lds_kernel<.., MODE 2> spend ~350 VALU instructions per (pixel, channel) -- VALU-bound at 0.12 of
// the HBM roofline -- although the k x k window of the fused operator is (g_c / k^2) w_ij with a channel-INDEPENDENT w.  By linearity
// everything channel-independent moves out of the channel loop:
//   * d(flow), d(weights) (ba_bwd_pix_kernel): with  P = sum_c (g_c / k^2) S_c  over the (K+1)^2 source neighbourhood of a pixel -- 16 LDS
//     reads + 16 fmas per channel --   d(w_ij) = bilinear_ij(P),   d(flow_x) = sum_ij w_ij d/dx bilinear_ij(P),   d(flow_y) likewise, once
//     per pixel and channel slab (11 global atomics).  A block keeps P of its 64 x TH pixels in registers while it walks its slab of
//     channels in LDS-staged groups of CG clamp-extended source boxes; no scatter, no third launch (the samples again for d(weights)).
//   * d(source) (ba_bwd_src_kernel): the (K+1)^2 cell coefficients  Kc = Wy^T w Wx  are formed once per pixel; a channel adds
//     (g_c / k^2) Kc to its fixed-point cells.  A block owns the 64 x TH flow pixels of a tile and CS channels, LDS box [CS][AH][AP] (tile
//     grown by the halo H = 4), flushed like be_bwd_tile2_kernel's (border folds in 64 bits, one global atomic per non-zero in-image
//     cell).  The fixed-point scale of a channel comes from the EXACT block maximum of |g_c / k^2| max_ij |w_ij| over the tile's pixels
//     (a pass over the block's g and w in front of the scatter), one bit of margin for the rounding of the coefficient sums; the
//     population bound is counted as in be_bwd_tile2_kernel.  A channel whose maximum is not finite (NaN / Inf gradient or weight) or
//     zero takes the per-tap global atomics of the reference for every pixel of the block.
// Pixels that do not fit the box (flow wider than the halo, a floor that disagrees between taps, NaN flow): d(source) by
// be_bwd_far2_kernel<.., FUSED> as before, d(flow) / d(weights) in ba_bwd_pix_kernel from global loads, tap by tap.
// (First cut, one kernel for all three with P per slab of 4 channels: 594 us -- 216 of them the 11 atomics per pixel and 4-channel slab,
// 144 the flush atomics; tools/r06/ba_bwd_time.py, profiles/r06_ba_bwd_linearity.txt.)
struct BaTaps {
    float wxr[3], wyb[3];
    int au, av;
    bool fit;
};
template <int AP, int AH>
__device__ __forceinline__ BaTaps ba_taps(float fx0, float fy0, int xf, int yf, int ax0, int ay0, bool inside) {
    constexpr int K = 3;
    BaTaps t;
    float flx0 = 0, fly0 = 0;
    bool regular = true;
#pragma unroll
    for (int j = 0; j < K; ++j) {
        const float dx = (fx0 + static_cast<float>(j - K / 2)) + static_cast<float>(xf);
        const float dy = (fy0 + static_cast<float>(j - K / 2)) + static_cast<float>(yf);
        const float fxl = floor_t(dx), fyl = floor_t(dy);
        if (j == 0) { flx0 = fxl; fly0 = fyl; }
        regular = regular & (fxl == flx0 + static_cast<float>(j)) & (fyl == fly0 + static_cast<float>(j));
        t.wxr[j] = dx - fxl;
        t.wyb[j] = dy - fyl;
    }
    const float lim = static_cast<float>(1 << 20);
    regular = regular & (flx0 > -lim) & (flx0 < lim) & (fly0 > -lim) & (fly0 < lim);   // rejects NaN too
    t.au = (regular ? static_cast<int>(flx0) : 0) - ax0;
    t.av = (regular ? static_cast<int>(fly0) : 0) - ay0;
    t.fit = inside & regular & (static_cast<unsigned>(t.au) <= static_cast<unsigned>(AP - 1 - K)) &
            (static_cast<unsigned>(t.av) <= static_cast<unsigned>(AH - 1 - K));
    return t;
}

#ifndef FFWM_BA_ABLATE
#define FFWM_BA_ABLATE 0      // bench-only (tools/r06/ba_bwd_time.py): 1 no LDS atomics, 2 no flush atomics (ba_bwd_src_kernel); 4 no d(flow) / d(weights) atomics, 8 no P accumulation (ba_bwd_pix_kernel)
#endif
template <int TH, int NT, int CS>
__global__ void __launch_bounds__(NT, 4)
ba_bwd_src_kernel(const float* __restrict__ flow, const float* __restrict__ attn, const float* __restrict__ gout, float* __restrict__ gsrc,
                  int C, int Hs, int Ws, int Hf, int Wf, int ntx, int nty, int cslabs, int remap) {
    using T = float;
    constexpr int K = 3, H = 4, RW = kTileRW, NW = NT / kWave, PPT = TH / NW;
    constexpr int AP = RW + 2 * H, AH = TH + 2 * H, NA = AP * AH;
    constexpr unsigned E = sizeof(T);
    constexpr float kInvKK = 1.f / static_cast<float>(K * K);
    static_assert(TH % NW == 0, "rows per wave");
    __shared__ unsigned box[CS * NA];
    __shared__ unsigned red[NW][CS];
    __shared__ int cred[NW];
    int* const A = reinterpret_cast<int*>(box);
    unsigned t = xcd_remap(blockIdx.x, gridDim.x, remap);
    const int tx = t % ntx;
    t /= ntx;
    const int ty = t % nty;
    t /= nty;
    const int slab = t % cslabs;
    const int b = t / cslabs;
    const int x0 = tx * RW, y0 = ty * TH;
    const int ax0 = x0 - H, ay0 = y0 - H;
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    const int xf = x0 + lane;
    const bool xin = xf < Wf;
    const int c0 = slab * CS;
    const int nc = (c0 + CS < C) ? CS : C - c0;
    const size_t splane = static_cast<size_t>(Hs) * Ws;
    const size_t fplane = static_cast<size_t>(Hf) * Wf;
    const unsigned sbytes = static_cast<unsigned>(splane * E);
    const unsigned fpb = static_cast<unsigned>(fplane * E);
    T* gp = gsrc + (static_cast<size_t>(b) * C + c0) * splane;
    const rsrc_t rfl = make_rsrc(flow + static_cast<size_t>(b) * 2 * fplane, 2 * fpb);
    const rsrc_t ratt = make_rsrc(attn + static_cast<size_t>(b) * K * K * fplane, K * K * fpb);
    const rsrc_t rg = make_rsrc(gout + (static_cast<size_t>(b) * C + c0) * fplane, static_cast<unsigned>(nc) * fpb);   // channels >= nc read 0
    const bool inside = ax0 <= Ws - 1 && ay0 <= Hs - 1;
    const bool foldL = ax0 < 0, foldR = Ws - ax0 < AP;
    const bool foldT = ay0 < 0, foldB = Hs - ay0 < AH;

    for (int i = threadIdx.x; i < NA; i += NT) A[i] = 0;
    __syncthreads();
    struct PixLoad {
        T fx, fy;
        T w[K * K];
        T g[CS];
    };
    const int xfc = min(xf, Wf - 1);
    auto request = [&](int r, PixLoad& d) {
        const int yfc = min(y0 + wave + r * NW, Hf - 1);       // rows outside the flow image shadow a valid one
        const unsigned fo = (static_cast<unsigned>(yfc) * Wf + xfc) * E;
        d.fx = buf_ld<T>(rfl, fo);
        d.fy = buf_ld<T>(rfl, fo + fpb);
#pragma unroll
        for (int q = 0; q < K * K; ++q) d.w[q] = buf_ld<T>(ratt, fo + static_cast<unsigned>(q) * fpb);
#pragma unroll
        for (int c = 0; c < CS; ++c) d.g[c] = buf_ld<T>(rg, fo + static_cast<unsigned>(c) * fpb);
    };

    // ---- one pass over the tile's pixels: the channels' maxima and the population count
    unsigned mb[CS];
#pragma unroll
    for (int c = 0; c < CS; ++c) mb[c] = 0;
    {
        PixLoad nxt;
        request(0, nxt);
#pragma unroll 1
        for (int r = 0; r < PPT; ++r) {
            const int yf = y0 + wave + r * NW;
            PixLoad cur = nxt;
            if (r + 1 < PPT) request(r + 1, nxt);
            if (!(xin && yf < Hf)) continue;
            {
                // the population bound: pixels per neighbourhood origin, counted in the first plane of the box (be_bwd_tile2_kernel's head)
                const BaTaps tp = ba_taps<AP, AH>(cur.fx, cur.fy, xf, yf, ax0, ay0, inside);
                if (tp.fit) __hip_atomic_fetch_add(A + tp.av * AP + tp.au, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            unsigned wm = 0;
#pragma unroll
            for (int q = 0; q < K * K; ++q) wm = max(wm, __builtin_bit_cast(unsigned, cur.w[q]) & 0x7FFFFFFFu);
            const T wmax = __builtin_bit_cast(T, wm);
#pragma unroll
            for (int c = 0; c < CS; ++c)
                mb[c] = max(mb[c], __builtin_bit_cast(unsigned, (cur.g[c] * kInvKK) * wmax) & 0x7FFFFFFFu);   // NaN / Inf patterns rank above every finite one
        }
    }
    __syncthreads();
    int fx_bits;
    {
        int cm = 0;
        for (int i = threadIdx.x; i < NA; i += NT) cm = max(cm, A[i]);
        cm = wave_max(cm);
        if (lane == 0) cred[wave] = cm;
        __syncthreads();                                    // (also: every count has been read before the box is zeroed)
        cm = cred[0];
#pragma unroll
        for (int w2 = 1; w2 < NW; ++w2) cm = max(cm, cred[w2]);
        const int blen = 32 - __clz(cm);
        fx_bits = __builtin_amdgcn_readfirstlane(min(22, 31 - 4 - blen));          // 16 cmax 2^bits < 2^31; 22: one bit of margin on the bound
    }
    // the channels' scales: block maximum of |g_c / k^2| max|w|, 2 x margin
#pragma unroll
    for (int c = 0; c < CS; ++c) {
        const unsigned m = wave_max(mb[c]);
        if (lane == 0) red[wave][c] = m;
    }
    __syncthreads();
    T fx_scale[CS], fx_inv[CS];
    bool exact[CS], nothing[CS];
#pragma unroll
    for (int c = 0; c < CS; ++c) {
        unsigned m = red[0][c];
#pragma unroll
        for (int w2 = 1; w2 < NW; ++w2) m = max(m, red[w2][c]);
        m = __builtin_amdgcn_readfirstlane(m);
        const T thr = 2 * __builtin_bit_cast(T, m);
        int ex = 0;
        (void)frexpf(thr, &ex);
        exact[c] = !(thr >= 0) || !(thr < 1e37f) || (m != 0 && ex < -100);   // NaN, Inf, denormal range: the exact per-tap path
        nothing[c] = m == 0;                                                // every (g_c / k^2) w of the tile is +-0: nothing to add
        fx_scale[c] = exact[c] ? 1.f : ldexpf(1.f, fx_bits - ex);
        fx_inv[c] = exact[c] ? 1.f : ldexpf(1.f, ex - fx_bits);
    }
    for (int i = threadIdx.x; i < CS * NA; i += NT) A[i] = 0;
    __syncthreads();

    // ---- the scatter: the cell coefficients Kc = Wy^T w Wx once per pixel, (g_c / k^2) Kc into the channel's fixed-point cells
    if (inside) {
        PixLoad nxt;
        request(0, nxt);
#pragma unroll 1
        for (int r = 0; r < PPT; ++r) {
            const int yf = y0 + wave + r * NW;
            PixLoad cur = nxt;
            if (r + 1 < PPT) request(r + 1, nxt);
            if (!(xin && yf < Hf)) continue;
            const BaTaps tp = ba_taps<AP, AH>(cur.fx, cur.fy, xf, yf, ax0, ay0, inside);
            if (!tp.fit) continue;                           // be_bwd_far2_kernel's pixel
            T xl[K], yt[K];
#pragma unroll
            for (int j = 0; j < K; ++j) {
                xl[j] = 1 - tp.wxr[j];
                yt[j] = 1 - tp.wyb[j];
            }
            T txc[K][K + 1];
#pragma unroll
            for (int i = 0; i < K; ++i) {
#pragma unroll
                for (int c2 = 0; c2 <= K; ++c2) {
                    T v = 0;
                    if (c2 < K) v = cur.w[i * K + c2] * xl[c2];
                    if (c2 > 0) v = (c2 < K) ? fma_t<T>(cur.w[i * K + c2 - 1], tp.wxr[c2 - 1], v) : cur.w[i * K + c2 - 1] * tp.wxr[c2 - 1];
                    txc[i][c2] = v;
                }
            }
            T Kc[(K + 1) * (K + 1)];
#pragma unroll
            for (int r2 = 0; r2 <= K; ++r2) {
#pragma unroll
                for (int c2 = 0; c2 <= K; ++c2) {
                    T v = 0;
                    if (r2 < K) v = txc[r2][c2] * yt[r2];
                    if (r2 > 0) v = (r2 < K) ? fma_t<T>(txc[r2 - 1][c2], tp.wyb[r2 - 1], v) : txc[r2 - 1][c2] * tp.wyb[r2 - 1];
                    Kc[r2 * (K + 1) + c2] = v;
                }
            }
            int* ap = A + tp.av * AP + tp.au;
#pragma unroll
            for (int c = 0; c < CS; ++c) {
                if (c >= nc) break;
                if (nothing[c]) continue;
                if (!exact[c]) {
                    const T gs = (cur.g[c] * kInvKK) * fx_scale[c];
#pragma unroll
                    for (int q = 0; q < ((FFWM_BA_ABLATE & 1) ? 1 : (K + 1) * (K + 1)); ++q)
                        __hip_atomic_fetch_add(ap + c * NA + (q / (K + 1)) * AP + (q % (K + 1)), __float2int_rn(gs * Kc[q]), __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_WORKGROUP);
                } else {
                    // the reference's per-tap scatter for this pixel and channel (non-finite or all-zero gradients: rare)
                    const T gd = cur.g[c] * kInvKK;
                    T* gplane_c = gp + static_cast<size_t>(c) * splane;
                    const unsigned pixb = (static_cast<unsigned>(yf) * Wf + xf) * E;
#pragma unroll 1
                    for (int i = 0; i < K; ++i) {
                        const Tap1<T> ty1 = make_tap<T>(cur.fy, i - K / 2, yf, Hs);
#pragma unroll 1
                        for (int j = 0; j < K; ++j) {
                            const Tap1<T> tx1 = make_tap<T>(cur.fx, j - K / 2, xf, Ws);
                            const T gv = gd * buf_ld<T>(ratt, pixb + static_cast<unsigned>(i * K + j) * fpb);
                            const unsigned rT = ty1.lo * static_cast<unsigned>(Ws) * E, rB = ty1.hi * static_cast<unsigned>(Ws) * E;
                            const unsigned cL = tx1.lo * E, cR = tx1.hi * E;
                            atomic_add_off(gplane_c, rT + cL, gv * tx1.wlo * ty1.wlo);
                            atomic_add_off(gplane_c, rT + cR, gv * tx1.whi * ty1.wlo);
                            atomic_add_off(gplane_c, rB + cL, gv * tx1.wlo * ty1.whi);
                            atomic_add_off(gplane_c, rB + cR, gv * tx1.whi * ty1.whi);
                        }
                    }
                }
            }
        }
    }
    __syncthreads();
    if (!inside) return;
    // ---- border folds (64-bit sums; a total that does not fit a cell goes straight to grad_source) and the flush
    if (foldL || foldR) {
        for (int e = threadIdx.x; e < nc * AH; e += NT) {
            const int c = e / AH, row = e - c * AH;
            int* arow = A + c * NA + row * AP;
            const int cy = ay0 + row;
            if (foldL) {
                long long s = 0;
                for (int u = 0; u < -ax0; ++u) s += arow[u];
                // (fx_inv is indexed by a run-time channel below: selected without a scratch array)
                const long long tot = static_cast<long long>(arow[-ax0]) + s;
                if (tot >= -2147483647LL && tot <= 2147483647LL) arow[-ax0] = static_cast<int>(tot);
                else {
                    arow[-ax0] = 0;
                    T iv = fx_inv[0];
#pragma unroll
                    for (int q = 1; q < CS; ++q) iv = (c == q) ? fx_inv[q] : iv;
                    atomic_add(gp + static_cast<size_t>(c) * splane + static_cast<size_t>(min(max(cy, 0), Hs - 1)) * Ws, static_cast<T>(tot) * iv);
                }
            }
            if (foldR) {
                long long s = 0;
                for (int u = Ws - ax0; u < AP; ++u) s += arow[u];
                const long long tot = static_cast<long long>(arow[Ws - 1 - ax0]) + s;
                if (tot >= -2147483647LL && tot <= 2147483647LL) arow[Ws - 1 - ax0] = static_cast<int>(tot);
                else {
                    arow[Ws - 1 - ax0] = 0;
                    T iv = fx_inv[0];
#pragma unroll
                    for (int q = 1; q < CS; ++q) iv = (c == q) ? fx_inv[q] : iv;
                    atomic_add(gp + static_cast<size_t>(c) * splane + static_cast<size_t>(min(max(cy, 0), Hs - 1)) * Ws + (Ws - 1), static_cast<T>(tot) * iv);
                }
            }
        }
        __syncthreads();
    }
    if (foldT || foldB) {
        for (int e = threadIdx.x; e < nc * AP; e += NT) {
            const int c = e / AP, col = e - c * AP;
            const int cx = ax0 + col;
            if (cx < 0 || cx >= Ws) continue;               // (the out-of-image columns are already part of the border columns)
            int* acol = A + c * NA + col;
            T iv = fx_inv[0];
#pragma unroll
            for (int q = 1; q < CS; ++q) iv = (c == q) ? fx_inv[q] : iv;
            if (foldT) {
                long long s = 0;
                for (int v = 0; v < -ay0; ++v) s += acol[v * AP];
                const long long tot = static_cast<long long>(acol[-ay0 * AP]) + s;
                if (tot >= -2147483647LL && tot <= 2147483647LL) acol[-ay0 * AP] = static_cast<int>(tot);
                else {
                    acol[-ay0 * AP] = 0;
                    atomic_add(gp + static_cast<size_t>(c) * splane + cx, static_cast<T>(tot) * iv);
                }
            }
            if (foldB) {
                long long s = 0;
                for (int v = Hs - ay0; v < AH; ++v) s += acol[v * AP];
                const long long tot = static_cast<long long>(acol[(Hs - 1 - ay0) * AP]) + s;
                if (tot >= -2147483647LL && tot <= 2147483647LL) acol[(Hs - 1 - ay0) * AP] = static_cast<int>(tot);
                else {
                    acol[(Hs - 1 - ay0) * AP] = 0;
                    atomic_add(gp + static_cast<size_t>(c) * splane + static_cast<size_t>(Hs - 1) * Ws + cx, static_cast<T>(tot) * iv);
                }
            }
        }
        __syncthreads();
    }
    {
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
#pragma unroll
        for (int c = 0; c < CS; ++c) {
            if (c >= nc) break;
            T* gplane = gp + static_cast<size_t>(c) * splane;
            const T iv = fx_inv[c];
            // (every in-image cell by a global atomic: the interior cells by read-modify-write, old values requested in front of the
            // scatter, measured 263 -> 304 us -- profiles/r06_ba_bwd_linearity.txt)
            // wave w flushes box rows w, w + NW, ..: lane -> column (row tests are scalar), the 2 H columns past the 64th afterwards
            const int cxl = ax0 + lane;
            const bool cxin = cxl >= 0 && cxl < Ws;
#pragma unroll 1
            for (int arow = wave; arow < AH; arow += NW) {
                const int cy = ay0 + arow;
                if (cy < 0 || cy >= Hs) continue;
                const int a = A[c * NA + arow * AP + lane];
                if (a != 0 && cxin && !(FFWM_BA_ABLATE & 2)) atomic_add(gplane + static_cast<size_t>(cy) * Ws + cxl, static_cast<T>(a) * iv);
            }
#pragma unroll 1
            for (int e = tid; e < AH * 2 * H; e += NT) {
                const int arow = e / (2 * H), acol = RW + (e & (2 * H - 1));
                const int cx = ax0 + acol, cy = ay0 + arow;
                const int a = A[c * NA + arow * AP + acol];
                if (a != 0 && cx >= 0 && cx < Ws && cy >= 0 && cy < Hs && !(FFWM_BA_ABLATE & 2))
                    atomic_add(gplane + static_cast<size_t>(cy) * Ws + cx, static_cast<T>(a) * iv);
            }
        }
    }
}


// d(flow), d(weights) of ONE pixel over `nc` channels, every tap on its own from global memory (ba_bwd_generic's arithmetic).
__device__ __attribute__((noinline)) void ba_pixel_by_taps(const float* sp, rsrc_t rga, rsrc_t ratt, float* gwb, int nc, int Hs, int Ws,
                                                           size_t splane, unsigned fpb, float fx0, float fy0, int xf, int yf, unsigned pix,
                                                           float& gx_out, float& gy_out) {
    using T = float;
    constexpr int K = 3;
    constexpr unsigned E = sizeof(T);
    constexpr float kInvKK = 1.f / static_cast<float>(K * K);
    T gx = 0, gy = 0;
#pragma unroll 1
    for (int c = 0; c < nc; ++c) {
        const T gd = buf_ld<T>(rga, pix * E + static_cast<unsigned>(c) * fpb) * kInvKK;
        const T* spc = sp + static_cast<size_t>(c) * splane;
#pragma unroll 1
        for (int i = 0; i < K; ++i) {
            const Tap1<T> ty1 = make_tap<T>(fy0, i - K / 2, yf, Hs);
#pragma unroll 1
            for (int j = 0; j < K; ++j) {
                const Tap1<T> tx1 = make_tap<T>(fx0, j - K / 2, xf, Ws);
                const size_t rT = static_cast<size_t>(ty1.lo) * Ws, rB = static_cast<size_t>(ty1.hi) * Ws;
                const T sTL = spc[rT + tx1.lo], sTR = spc[rT + tx1.hi], sBL = spc[rB + tx1.lo], sBR = spc[rB + tx1.hi];
                const T wij = buf_ld<T>(ratt, pix * E + static_cast<unsigned>(i * K + j) * fpb);
                const T g1 = gd * wij;
                gy += g1 * (-tx1.wlo * sTL - tx1.whi * sTR + tx1.wlo * sBL + tx1.whi * sBR);
                gx += g1 * (-ty1.wlo * sTL - ty1.whi * sBL + ty1.wlo * sTR + ty1.whi * sBR);
                if (gwb) {
                    T s1 = (tx1.wlo * ty1.wlo) * sTL;
                    s1 = fma_t<T>(tx1.whi * ty1.wlo, sTR, s1);
                    s1 = fma_t<T>(tx1.wlo * ty1.whi, sBL, s1);
                    s1 = fma_t<T>(tx1.whi * ty1.whi, sBR, s1);
                    atomic_add_off(gwb, pix * E + static_cast<unsigned>(i * K + j) * fpb, gd * s1);
                }
            }
        }
    }
    gx_out = gx;
    gy_out = gy;
}

template <int TH, int CG, int WPE>
__global__ void __launch_bounds__(kBlock, WPE)
ba_bwd_pix_kernel(const float* __restrict__ src, const float* __restrict__ flow, const float* __restrict__ attn,
                  const float* __restrict__ gout, float* __restrict__ gflow, float* __restrict__ gw, int C, int Hs, int Ws, int Hf, int Wf,
                  int ntx, int nty, int cslabs, int cs, int remap) {
    using T = float;
    constexpr int K = 3, H = 4, RW = kTileRW, NW = kBlock / kWave, PPT = TH / NW, NP = (K + 1) * (K + 1);
    constexpr int AP = RW + 2 * H, AH = TH + 2 * H, NA = AP * AH;
    constexpr unsigned E = sizeof(T);
    constexpr float kInvKK = 1.f / static_cast<float>(K * K);
    static_assert(CG % 4 == 0, "channels per group: whole float4s");
    // the staged boxes, CHANNEL-INNERMOST: S[cell][CG] -- one ds_read_b128 brings four channels of a cell (the wide-read rate: twice the
    // bytes per clock of four ds_read_b32, a quarter of the instructions)
    __shared__ __attribute__((aligned(16))) T S[CG * NA];
    unsigned t = xcd_remap(blockIdx.x, gridDim.x, remap);
    const int tx = t % ntx;
    t /= ntx;
    const int ty = t % nty;
    t /= nty;
    const int slab = t % cslabs;
    const int b = t / cslabs;
    const int x0 = tx * RW, y0 = ty * TH;
    const int ax0 = x0 - H, ay0 = y0 - H;
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    const int xf = x0 + lane;
    const bool xin = xf < Wf;
    const int c0 = slab * cs;
    const int nc = (c0 + cs < C) ? cs : C - c0;
    const size_t splane = static_cast<size_t>(Hs) * Ws;
    const size_t fplane = static_cast<size_t>(Hf) * Wf;
    const unsigned sbytes = static_cast<unsigned>(splane * E);
    const unsigned fpb = static_cast<unsigned>(fplane * E);
    const T* sp = src + (static_cast<size_t>(b) * C + c0) * splane;
    const T* gop = gout + (static_cast<size_t>(b) * C + c0) * fplane;
    const rsrc_t rfl = make_rsrc(flow + static_cast<size_t>(b) * 2 * fplane, 2 * fpb);
    const rsrc_t ratt = make_rsrc(attn + static_cast<size_t>(b) * K * K * fplane, K * K * fpb);
    const rsrc_t rga = make_rsrc(gop, static_cast<unsigned>(nc) * fpb);      // the slab's gradients: channels past it read 0 (host: cs fplane 4 < 2^31)
    const bool inside = ax0 <= Ws - 1 && ay0 <= Hs - 1;

    // the pixels of this thread: where their neighbourhood starts in a staged box (< 0: no pixel / not in the box)
    int nbo[PPT];
    unsigned fo[PPT];
#pragma unroll
    for (int r = 0; r < PPT; ++r) {
        const int yf = y0 + wave + r * NW;
        const bool live = xin && yf < Hf;
        fo[r] = (static_cast<unsigned>(min(yf, Hf - 1)) * Wf + min(xf, Wf - 1)) * E;
        const BaTaps tp = ba_taps<AP, AH>(buf_ld<T>(rfl, fo[r]), buf_ld<T>(rfl, fo[r] + fpb), xf, yf, ax0, ay0, inside);
        nbo[r] = live ? (tp.fit ? tp.av * AP + tp.au : -1) : -2;
    }
    T P[PPT][NP];
#pragma unroll
    for (int r = 0; r < PPT; ++r)
#pragma unroll
        for (int q = 0; q < NP; ++q) P[r][q] = 0;

    static_assert(AH % NW == 0 && AH * 2 * H <= kBlock, "staging layout");
    const int tid = threadIdx.x;
    const unsigned gxa = static_cast<unsigned>(min(max(ax0 + lane, 0), Ws - 1)) * E;
    const unsigned xoff = tid < AH * 2 * H ? (static_cast<unsigned>(min(max(ay0 + tid / (2 * H), 0), Hs - 1)) * Ws +
                                              static_cast<unsigned>(min(max(ax0 + RW + (tid & (2 * H - 1)), 0), Ws - 1))) * E
                                           : 0xFFFFFFF0u;
#pragma unroll 1
    for (int cg = 0; cg < nc; cg += CG) {
        // this group's gradients (channels past the slab read 0) -- requested before the boxes are staged
        // (ONE resource for the slab and a loop-VARIANT channel offset: per-group resources make the PPT x CG offsets loop invariants
        // that hipcc keeps in as many registers)
        const unsigned goff = static_cast<unsigned>(cg) * fpb;
        T g[PPT][CG];
#pragma unroll
        for (int r = 0; r < PPT; ++r)
#pragma unroll
            for (int c = 0; c < CG; ++c) g[r][c] = buf_ld<T>(rga, nbo[r] >= 0 ? fo[r] + goff + static_cast<unsigned>(c) * fpb : 0xFFFFFFF0u);
        __syncthreads();                                   // the previous group's boxes have been read
        // wave w stages box rows w, w + NW, ..: lane -> column (one row offset per load, a scalar); a thread loads the CG channels of its
        // cell and writes them with ds_write_b128; the 2 H columns past the 64th by the first AH * 2 H threads
        typedef float f32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int h4 = 0; h4 < CG / 4; ++h4) {
            rsrc_t rs[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const bool have = cg + h4 * 4 + c < nc;
                rs[c] = make_rsrc(sp + static_cast<size_t>(have ? cg + h4 * 4 + c : 0) * splane, have ? sbytes : 0u);
            }
            constexpr int RPW = AH / NW, HALF = (RPW + 1) / 2;       // rows per wave, in two batches (registers)
#pragma unroll
            for (int r0 = 0; r0 < RPW; r0 += HALF) {
                f32x4 st[HALF];
#pragma unroll
                for (int rr = 0; rr < HALF; ++rr) {
                    if (r0 + rr >= RPW) break;
                    const int gy = min(max(ay0 + wave + (r0 + rr) * NW, 0), Hs - 1);
                    const unsigned off = (static_cast<unsigned>(gy) * Ws) * E + gxa;
#pragma unroll
                    for (int c = 0; c < 4; ++c) st[rr][c] = buf_ld<T>(rs[c], off);
                }
#pragma unroll
                for (int rr = 0; rr < HALF; ++rr) {
                    if (r0 + rr >= RPW) break;
                    *reinterpret_cast<f32x4*>(S + (((wave + (r0 + rr) * NW) * AP + lane) * CG + h4 * 4)) = st[rr];
                }
            }
            f32x4 sx;
#pragma unroll
            for (int c = 0; c < 4; ++c) sx[c] = buf_ld<T>(rs[c], xoff);
            if (tid < AH * 2 * H) *reinterpret_cast<f32x4*>(S + (((tid / (2 * H)) * AP + RW + (tid & (2 * H - 1))) * CG + h4 * 4)) = sx;
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < PPT; ++r) {
            if (nbo[r] < 0) continue;
            const T* nb = S + nbo[r] * CG;
            T gd[CG];
#pragma unroll
            for (int c = 0; c < CG; ++c) gd[c] = ((FFWM_BA_ABLATE & 8) && c > 0) ? 0.f : g[r][c] * kInvKK;
#pragma unroll
            for (int q = 0; q < NP; ++q) {
#pragma unroll
                for (int h4 = 0; h4 < CG / 4; ++h4) {
                    if ((FFWM_BA_ABLATE & 8) && (q & 3)) continue;
                    const f32x4 v = *reinterpret_cast<const f32x4*>(nb + ((q / (K + 1)) * AP + (q % (K + 1))) * CG + h4 * 4);
#pragma unroll
                    for (int c = 0; c < 4; ++c) P[r][q] = fma_t<T>(gd[h4 * 4 + c], v[c], P[r][q]);
                }
                if ((q & 3) == 3) __builtin_amdgcn_sched_barrier(0);          // four cells' reads in flight, not sixteen (registers)
            }
        }
    }
    // d(weights), d(flow) of the thread's pixels
#pragma unroll
    for (int r = 0; r < PPT; ++r) {
        if (nbo[r] == -2) continue;
        const int yf = y0 + wave + r * NW;
        const unsigned pix = static_cast<unsigned>(yf) * Wf + xf;
        const T fx0 = buf_ld<T>(rfl, fo[r]), fy0 = buf_ld<T>(rfl, fo[r] + fpb);
        T gx = 0, gy = 0;
        if (nbo[r] >= 0) {
            const BaTaps tp = ba_taps<AP, AH>(fx0, fy0, xf, yf, ax0, ay0, inside);
            T w[K * K];
#pragma unroll
            for (int q = 0; q < K * K; ++q) w[q] = buf_ld<T>(ratt, fo[r] + static_cast<unsigned>(q) * fpb);
            T xl[K], yt[K];
#pragma unroll
            for (int j = 0; j < K; ++j) {
                xl[j] = 1 - tp.wxr[j];
                yt[j] = 1 - tp.wyb[j];
            }
            T* gwp = gw ? gw + static_cast<size_t>(b) * K * K * fplane : nullptr;      // (block-uniform base + 32-bit byte offsets)
#pragma unroll
            for (int i = 0; i < K; ++i) {
#pragma unroll
                for (int j = 0; j < K; ++j) {
                    const T TL = P[r][i * (K + 1) + j], TR = P[r][i * (K + 1) + j + 1];
                    const T BL = P[r][(i + 1) * (K + 1) + j], BR = P[r][(i + 1) * (K + 1) + j + 1];
                    const T top = fma_t<T>(tp.wxr[j], TR, xl[j] * TL), bot = fma_t<T>(tp.wxr[j], BR, xl[j] * BL);
                    const T lef = fma_t<T>(tp.wyb[i], BL, yt[i] * TL), rig = fma_t<T>(tp.wyb[i], BR, yt[i] * TR);
                    gy = fma_t<T>(w[i * K + j], bot - top, gy);
                    gx = fma_t<T>(w[i * K + j], rig - lef, gx);
                    if (gwp && !(FFWM_BA_ABLATE & 4)) atomic_add_off(gwp, fo[r] + static_cast<unsigned>(i * K + j) * fpb, fma_t<T>(tp.wyb[i], bot, yt[i] * top));
                }
            }
        } else {
            // every tap on its own from global memory, like the reference (rare; a called function: its registers are not the hot path's)
            ba_pixel_by_taps(sp, rga, ratt, gw ? gw + static_cast<size_t>(b) * K * K * fplane : nullptr, nc, Hs, Ws, splane, fpb, fx0, fy0, xf, yf, pix, gx, gy);
        }
        if (gflow && !(FFWM_BA_ABLATE & 4)) {
            atomic_add_off(gflow + static_cast<size_t>(b) * 2 * fplane, fo[r], gx);
            atomic_add_off(gflow + static_cast<size_t>(b) * 2 * fplane, fo[r] + fpb, gy);
        }
        __builtin_amdgcn_sched_barrier(0);                  // one pixel at a time (registers)
    }
}

// The fused forward of ONE pixel over `nc` channels, every tap on its own from global memory (ba_fwd_generic's arithmetic).
__device__ __attribute__((noinline)) void ba_pixel_forward_by_taps(const float* sp, rsrc_t ratt, float* op, int nc, int Hs, int Ws, size_t splane,
                                                                   unsigned fpb, float fx0, float fy0, int xf, int yf, unsigned pix) {
    using T = float;
    constexpr int K = 3;
    constexpr unsigned E = sizeof(T);
#pragma unroll 1
    for (int c = 0; c < nc; ++c) {
        const T* spc = sp + static_cast<size_t>(c) * splane;
        T osum = 0;
#pragma unroll 1
        for (int i = 0; i < K; ++i) {
            const Tap1<T> ty1 = make_tap<T>(fy0, i - K / 2, yf, Hs);
            const size_t rT = static_cast<size_t>(ty1.lo) * Ws, rB = static_cast<size_t>(ty1.hi) * Ws;
#pragma unroll 1
            for (int j = 0; j < K; ++j) {
                const Tap1<T> tx1 = make_tap<T>(fx0, j - K / 2, xf, Ws);
                T s1 = (tx1.wlo * ty1.wlo) * spc[rT + tx1.lo];
                s1 = fma_t<T>(tx1.whi * ty1.wlo, spc[rT + tx1.hi], s1);
                s1 = fma_t<T>(tx1.wlo * ty1.whi, spc[rB + tx1.lo], s1);
                s1 = fma_t<T>(tx1.whi * ty1.whi, spc[rB + tx1.hi], s1);
                osum = add_rn(osum, mul_rn(s1, buf_ld<T>(ratt, pix * E + static_cast<unsigned>(i * K + j) * fpb)));
            }
        }
        op[static_cast<size_t>(c) * (fpb / E) + pix] = osum / static_cast<T>(K * K);
    }
}

// The fused FORWARD on the same layout (round 6): out_c = sum over the (K+1)^2 neighbourhood of coef S_c with the channel-independent
// coefficients  coef = Wy^T (w / k^2) Wx  kept in registers; boxes staged channel-innermost, one ds_read_b128 per cell and four channels.
template <int TH, int CG, int WPE>
__global__ void __launch_bounds__(kBlock, WPE)
ba_fwd_pix_kernel(const float* __restrict__ src, const float* __restrict__ flow, const float* __restrict__ attn,
                  float* __restrict__ out, int C, int Hs, int Ws, int Hf, int Wf,
                  int ntx, int nty, int cslabs, int cs, int remap) {
    using T = float;
    constexpr int K = 3, H = 4, RW = kTileRW, NW = kBlock / kWave, PPT = TH / NW, NP = (K + 1) * (K + 1);
    constexpr int AP = RW + 2 * H, AH = TH + 2 * H, NA = AP * AH;
    constexpr unsigned E = sizeof(T);
    constexpr float kInvKK = 1.f / static_cast<float>(K * K);
    static_assert(CG % 4 == 0, "channels per group: whole float4s");
    // the staged boxes, CHANNEL-INNERMOST: S[cell][CG] -- one ds_read_b128 brings four channels of a cell (the wide-read rate: twice the
    // bytes per clock of four ds_read_b32, a quarter of the instructions)
    __shared__ __attribute__((aligned(16))) T S[CG * NA];
    unsigned t = xcd_remap(blockIdx.x, gridDim.x, remap);
    const int tx = t % ntx;
    t /= ntx;
    const int ty = t % nty;
    t /= nty;
    const int slab = t % cslabs;
    const int b = t / cslabs;
    const int x0 = tx * RW, y0 = ty * TH;
    const int ax0 = x0 - H, ay0 = y0 - H;
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    const int xf = x0 + lane;
    const bool xin = xf < Wf;
    const int c0 = slab * cs;
    const int nc = (c0 + cs < C) ? cs : C - c0;
    const size_t splane = static_cast<size_t>(Hs) * Ws;
    const size_t fplane = static_cast<size_t>(Hf) * Wf;
    const unsigned sbytes = static_cast<unsigned>(splane * E);
    const unsigned fpb = static_cast<unsigned>(fplane * E);
    const T* sp = src + (static_cast<size_t>(b) * C + c0) * splane;
    T* op = out + (static_cast<size_t>(b) * C + c0) * fplane;
    const rsrc_t rfl = make_rsrc(flow + static_cast<size_t>(b) * 2 * fplane, 2 * fpb);
    const rsrc_t ratt = make_rsrc(attn + static_cast<size_t>(b) * K * K * fplane, K * K * fpb);
    const bool inside = ax0 <= Ws - 1 && ay0 <= Hs - 1;

    // the pixels of this thread: where their neighbourhood starts in a staged box (< 0: no pixel / not in the box) and the (K+1)^2 cell
    // coefficients  coef = Wy^T (w / k^2) Wx  of the fused operator (channel-independent: formed once)
    int nbo[PPT];
    unsigned fo[PPT];
    T coef[PPT][NP];
#pragma unroll
    for (int r = 0; r < PPT; ++r) {
        const int yf = y0 + wave + r * NW;
        const bool live = xin && yf < Hf;
        fo[r] = (static_cast<unsigned>(min(yf, Hf - 1)) * Wf + min(xf, Wf - 1)) * E;
        const BaTaps tp = ba_taps<AP, AH>(buf_ld<T>(rfl, fo[r]), buf_ld<T>(rfl, fo[r] + fpb), xf, yf, ax0, ay0, inside);
        nbo[r] = live ? (tp.fit ? tp.av * AP + tp.au : -1) : -2;
        T w[K * K];
#pragma unroll
        for (int q = 0; q < K * K; ++q) w[q] = buf_ld<T>(ratt, fo[r] + static_cast<unsigned>(q) * fpb) * kInvKK;
        T txc[K][K + 1];
#pragma unroll
        for (int i = 0; i < K; ++i) {
#pragma unroll
            for (int c2 = 0; c2 <= K; ++c2) {
                T v = 0;
                if (c2 < K) v = w[i * K + c2] * (1 - tp.wxr[c2]);
                if (c2 > 0) v = (c2 < K) ? fma_t<T>(w[i * K + c2 - 1], tp.wxr[c2 - 1], v) : w[i * K + c2 - 1] * tp.wxr[c2 - 1];
                txc[i][c2] = v;
            }
        }
#pragma unroll
        for (int r2 = 0; r2 <= K; ++r2) {
#pragma unroll
            for (int c2 = 0; c2 <= K; ++c2) {
                T v = 0;
                if (r2 < K) v = txc[r2][c2] * (1 - tp.wyb[r2]);
                if (r2 > 0) v = (r2 < K) ? fma_t<T>(txc[r2 - 1][c2], tp.wyb[r2 - 1], v) : txc[r2 - 1][c2] * tp.wyb[r2 - 1];
                coef[r][r2 * (K + 1) + c2] = v;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    static_assert(AH % NW == 0 && AH * 2 * H <= kBlock, "staging layout");
    const int tid = threadIdx.x;
    const unsigned gxa = static_cast<unsigned>(min(max(ax0 + lane, 0), Ws - 1)) * E;
    const unsigned xoff = tid < AH * 2 * H ? (static_cast<unsigned>(min(max(ay0 + tid / (2 * H), 0), Hs - 1)) * Ws +
                                              static_cast<unsigned>(min(max(ax0 + RW + (tid & (2 * H - 1)), 0), Ws - 1))) * E
                                           : 0xFFFFFFF0u;
#pragma unroll 1
    for (int cg = 0; cg < nc; cg += CG) {
        __syncthreads();                                   // the previous group's boxes have been read
        // wave w stages box rows w, w + NW, ..: lane -> column (one row offset per load, a scalar); a thread loads the CG channels of its
        // cell and writes them with ds_write_b128; the 2 H columns past the 64th by the first AH * 2 H threads
        typedef float f32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int h4 = 0; h4 < CG / 4; ++h4) {
            rsrc_t rs[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const bool have = cg + h4 * 4 + c < nc;
                rs[c] = make_rsrc(sp + static_cast<size_t>(have ? cg + h4 * 4 + c : 0) * splane, have ? sbytes : 0u);
            }
            constexpr int RPW = AH / NW, HALF = (RPW + 1) / 2;       // rows per wave, in two batches (registers)
#pragma unroll
            for (int r0 = 0; r0 < RPW; r0 += HALF) {
                f32x4 st[HALF];
#pragma unroll
                for (int rr = 0; rr < HALF; ++rr) {
                    if (r0 + rr >= RPW) break;
                    const int gy = min(max(ay0 + wave + (r0 + rr) * NW, 0), Hs - 1);
                    const unsigned off = (static_cast<unsigned>(gy) * Ws) * E + gxa;
#pragma unroll
                    for (int c = 0; c < 4; ++c) st[rr][c] = buf_ld<T>(rs[c], off);
                }
#pragma unroll
                for (int rr = 0; rr < HALF; ++rr) {
                    if (r0 + rr >= RPW) break;
                    *reinterpret_cast<f32x4*>(S + (((wave + (r0 + rr) * NW) * AP + lane) * CG + h4 * 4)) = st[rr];
                }
            }
            f32x4 sx;
#pragma unroll
            for (int c = 0; c < 4; ++c) sx[c] = buf_ld<T>(rs[c], xoff);
            if (tid < AH * 2 * H) *reinterpret_cast<f32x4*>(S + (((tid / (2 * H)) * AP + RW + (tid & (2 * H - 1))) * CG + h4 * 4)) = sx;
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < PPT; ++r) {
            if (nbo[r] < 0) continue;
            const T* nb = S + nbo[r] * CG;
            T acc[CG];
#pragma unroll
            for (int c = 0; c < CG; ++c) acc[c] = 0;
#pragma unroll
            for (int q = 0; q < NP; ++q) {
#pragma unroll
                for (int h4 = 0; h4 < CG / 4; ++h4) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(nb + ((q / (K + 1)) * AP + (q % (K + 1))) * CG + h4 * 4);
#pragma unroll
                    for (int c = 0; c < 4; ++c) acc[h4 * 4 + c] = fma_t<T>(coef[r][q], v[c], acc[h4 * 4 + c]);
                }
                if ((q & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
            const rsrc_t ro = make_rsrc(op, static_cast<unsigned>(nc) * fpb);      // channels past the slab: stores dropped
#pragma unroll
            for (int c = 0; c < CG; ++c) {
                ElemRow<T, 1> v1;
                v1.v[0] = acc[c];
                buf_store_row<T, 1>(ro, fo[r] + static_cast<unsigned>(cg + c) * fpb, v1);
            }
        }
    }
    // the pixels outside the box: every tap on its own from global memory (rare)
#pragma unroll
    for (int r = 0; r < PPT; ++r) {
        if (nbo[r] != -1) continue;
        const int yf = y0 + wave + r * NW;
        ba_pixel_forward_by_taps(sp, ratt, op, nc, Hs, Ws, splane, fpb, buf_ld<T>(rfl, fo[r]), buf_ld<T>(rfl, fo[r] + fpb), xf, yf,
                                 static_cast<unsigned>(yf) * Wf + xf);
    }
}

template <typename T>
int launch_attn_fwd(const T* src, const T* flow, const T* wts, T* out, int64_t B, int64_t C, int64_t Hs, int64_t Ws,
                    int64_t Hf, int64_t Wf, int k, hipStream_t st) {
    const double bytes = sizeof(T) * static_cast<double>(B) * (C * Hs * Ws + (2.0 + k * k) * Hf * Wf + static_cast<double>(C) * Hf * Wf);
    if constexpr (sizeof(T) == 4) {
        if (k == 3 && options().be_fwd_variant != 9 && options().ba_fwd_pix != 0 && Hs * Ws < (1LL << 29)) {
            const int fm = options().ba_fwd_pix;                                       // 1: 64 x 8 pixels, 4 channels per group; 2: 64 x 16, 4; 3: 64 x 8, 8; 4: 64 x 8, 4, 6 blocks per CU
            const int tha = fm == 2 ? 16 : 8, cga = fm == 3 ? 8 : 4, wpe = fm == 4 ? 6 : 4;
            const int ntx = static_cast<int>((Wf + kTileRW - 1) / kTileRW), ntya = static_cast<int>((Hf + tha - 1) / tha);
            const int64_t tiles = B * ntx * ntya;
            int64_t want = (static_cast<int64_t>(wpe) * device_cus() + tiles - 1) / tiles;                   // slabs: one resident round of blocks
            if (want < 1) want = 1;
            int csa = static_cast<int>((C + want - 1) / want);
            csa = (csa + cga - 1) / cga * cga;
            while (csa > cga && static_cast<int64_t>(csa) * Hf * Wf * 4 >= (1LL << 31)) csa -= cga;       // 32-bit byte offsets over a slab of the output
            const int slabsa = static_cast<int>((C + csa - 1) / csa);
            FFWM_REQUIRE(tiles * slabsa < (1LL << 31), FFWM_ERR_SIZE, "ffwm_block_attention_forward: grid too large");
            LaunchScope ls("block_attention_fwd_lds", st, bytes);
#define FFWM_BA_FWD(TH_, CG_, WPE_)                                                                                                             \
    hipLaunchKernelGGL((ba_fwd_pix_kernel<TH_, CG_, WPE_>), dim3(static_cast<unsigned>(tiles * slabsa)), dim3(kBlock), 0, st, src, flow, wts, out, \
                       (int)C, (int)Hs, (int)Ws, (int)Hf, (int)Wf, ntx, ntya, slabsa, csa, options().xcd_remap)
            if (fm == 2) FFWM_BA_FWD(16, 4, 4);
            else if (fm == 3) FFWM_BA_FWD(8, 8, 4);
            else if (fm == 4) FFWM_BA_FWD(8, 4, 6);
            else FFWM_BA_FWD(8, 4, 4);
#undef FFWM_BA_FWD
            return check_launch("ffwm_block_attention_forward");
        }
        if (k == 3 && options().be_fwd_variant != 9) {
            const Geometry g = plan(B, C, Hf, Wf, 16);
            const int rpt = Hf >= 64 ? 4 : 1;
            const int th = (kBlock / kWave) * rpt;
            const int tyl = static_cast<int>((Hf + th - 1) / th);
            const unsigned gridl = static_cast<unsigned>(B * g.tiles_x * tyl * g.cslabs);
            LaunchScope ls("block_attention_fwd_lds", st, bytes);
            if (rpt == 4)
                hipLaunchKernelGGL((be_fwd_lds_kernel<float, 3, 4, 1>), dim3(gridl), dim3(kBlock), 0, st, src, flow, out, (int)C,
                                   (int)Hs, (int)Ws, (int)Hf, (int)Wf, g.tiles_x, tyl, g.cslabs, g.cs, options().xcd_remap, 0, 0, wts);
            else
                hipLaunchKernelGGL((be_fwd_lds_kernel<float, 3, 1, 1>), dim3(gridl), dim3(kBlock), 0, st, src, flow, out, (int)C,
                                   (int)Hs, (int)Ws, (int)Hf, (int)Wf, g.tiles_x, tyl, g.cslabs, g.cs, options().xcd_remap, 0, 0, wts);
            return check_launch("ffwm_block_attention_forward");
        }
    }
    const int64_t n = B * C * Hf * Wf;
    const unsigned grid = static_cast<unsigned>(n / kBlock + 1 < 65536 ? n / kBlock + 1 : 65536);
    LaunchScope ls("block_attention_fwd_generic", st, bytes);
    hipLaunchKernelGGL((ba_fwd_generic<T>), dim3(grid), dim3(kBlock), 0, st, src, flow, wts, out, n, (int)C, (int)Hs, (int)Ws,
                       (int)Hf, (int)Wf, k);
    return check_launch("ffwm_block_attention_forward");
}

template <typename T>
int launch_attn_bwd(const T* src, const T* flow, const T* wts, const T* gout, T* gsrc, T* gflow, T* gw, int64_t B,
                    int64_t C, int64_t Hs, int64_t Ws, int64_t Hf, int64_t Wf, int k, hipStream_t st) {
    const double bytes = sizeof(T) * static_cast<double>(B) * (static_cast<double>(C) * Hf * Wf + 2.0 * C * Hs * Ws + (4.0 + k * k) * Hf * Wf);
    if constexpr (sizeof(T) == 4) {
        // (rounds 4-5's route -- be_bwd_tile2_kernel<.., FUSED> + be_fwd_lds_kernel<.., MODE 2> -- is gone: 1.5 x slower, and round 6's wide-flow
        // test found its d(weights) launch wrong where flows leave the forward kernel's LDS window)
        if (k == 3 && options().be_bwd_variant != 9 && Hs * Ws < (1LL << 29)) {
            const int mode = options().ba_bwd_fused;
            const int th = mode == 2 ? 16 : 32;
            const int nts = mode == 3 ? 512 : 256;
            constexpr int csf = 4;
            const TileGeo geo{kTileRW, th, 4, th};
            const int ntx = static_cast<int>((Wf + kTileRW - 1) / kTileRW), nty = static_cast<int>((Hf + th - 1) / th);
            const int cslabs = static_cast<int>((C + csf - 1) / csf);
            const Geometry gf = plan(B, C, Hf, Wf, 32);
            if (gsrc) {                          // (the two halves are independent: a call that wants only d(flow) / d(weights) runs the pixel kernel alone)
            {
                LaunchScope ls("block_attention_bwd_far", st, sizeof(T) * 2.0 * B * Hf * Wf);
                hipLaunchKernelGGL((be_bwd_far2_kernel<float, 3, true>), dim3(gf.grid), dim3(kBlock), 0, st, flow, gout, gsrc,
                                   (int)C, (int)Hs, (int)Ws, (int)Hf, (int)Wf, gf.tiles_x, gf.tiles_y, gf.cslabs, gf.cs, geo, wts);
            }
            if (int rc = check_launch("ffwm_block_attention_backward(far)")) return rc;
            FFWM_REQUIRE(B * ntx * nty * static_cast<int64_t>(cslabs) < (1LL << 31), FFWM_ERR_SIZE, "ffwm_block_attention_backward: grid too large");
            {
                const unsigned grid = static_cast<unsigned>(B * ntx * nty * cslabs);
                LaunchScope ls("block_attention_bwd_src", st, bytes);        // (the OPERATOR's algorithmic bytes: bench.py prices the sum of the three scopes' times against them)
#define FFWM_BA_SRC(TH_, NT_)                                                                                                  \
    hipLaunchKernelGGL((ba_bwd_src_kernel<TH_, NT_, csf>), dim3(grid), dim3(NT_), 0, st, flow, wts, gout, gsrc, (int)C, (int)Hs, \
                       (int)Ws, (int)Hf, (int)Wf, ntx, nty, cslabs, options().xcd_remap)
                if (th == 16) FFWM_BA_SRC(16, 256);
                else if (nts == 512) FFWM_BA_SRC(32, 512);
                else FFWM_BA_SRC(32, 256);
#undef FFWM_BA_SRC
            }
            if (int rc = check_launch("ffwm_block_attention_backward(source)")) return rc;
            }
            if (gflow || gw) {
                // d(flow), d(weights): as many channels per block as still give every CU its resident blocks
                const int cga = options().ba_bwd_pix == 1 || options().ba_bwd_pix == 3 ? 8 : 4;
                const int pm = options().ba_bwd_pix;                                   // (common.hpp)
                const int tha = pm == 1 || pm >= 4 ? 8 : 16;
                const int ntya = static_cast<int>((Hf + tha - 1) / tha);
                const int64_t tiles = B * ntx * ntya;
                const int wpe = pm == 5 ? 6 : ((pm == 0 || pm == 3) ? 3 : 4);   // resident blocks per CU
                int64_t want = (static_cast<int64_t>(wpe) * device_cus() + tiles - 1) / tiles;               // slabs: one resident round of blocks
                if (want < 1) want = 1;
                int csa = static_cast<int>((C + want - 1) / want);
                csa = (csa + cga - 1) / cga * cga;
                while (csa > cga && static_cast<int64_t>(csa) * Hf * Wf * 4 >= (1LL << 31)) csa -= cga;       // 32-bit byte offsets over a slab of grad_output
                const int slabsa = static_cast<int>((C + csa - 1) / csa);
                const unsigned grid = static_cast<unsigned>(tiles * slabsa);
                LaunchScope ls("block_attention_bwd_pix", st, sizeof(T) * static_cast<double>(B) * (static_cast<double>(C) * Hf * Wf + 1.0 * C * Hs * Ws + (4.0 + 2.0 * k * k) * Hf * Wf));
#define FFWM_BA_PIX(TH_, CG_, WPE_)                                                                                                     \
    hipLaunchKernelGGL((ba_bwd_pix_kernel<TH_, CG_, WPE_>), dim3(grid), dim3(kBlock), 0, st, src, flow, wts, gout, gflow, gw, (int)C, \
                       (int)Hs, (int)Ws, (int)Hf, (int)Wf, ntx, ntya, slabsa, csa, options().xcd_remap)
                if (pm == 1) FFWM_BA_PIX(8, 8, 4);
                else if (pm == 2) FFWM_BA_PIX(16, 4, 4);
                else if (pm == 3) FFWM_BA_PIX(16, 8, 3);
                else if (pm == 4) FFWM_BA_PIX(8, 4, 4);
                else if (pm == 5) FFWM_BA_PIX(8, 4, 6);
                else FFWM_BA_PIX(16, 4, 3);
#undef FFWM_BA_PIX
                return check_launch("ffwm_block_attention_backward(pixels)");
            }
            return FFWM_OK;
        }
    }
    const int64_t n = B * C * Hf * Wf;
    const unsigned grid = static_cast<unsigned>(n / kBlock + 1 < 65536 ? n / kBlock + 1 : 65536);
    LaunchScope ls("block_attention_bwd_generic", st, bytes);
    hipLaunchKernelGGL((ba_bwd_generic<T>), dim3(grid), dim3(kBlock), 0, st, src, flow, wts, gout, gsrc, gflow, gw, n, (int)C,
                       (int)Hs, (int)Ws, (int)Hf, (int)Wf, k);
    return check_launch("ffwm_block_attention_backward");
}

}  // namespace
}  // namespace ffwm

using namespace ffwm;

extern "C" int ffwm_block_extractor_forward(const void* source, const void* flow_field, void* output,
                                            int64_t B, int64_t C, int64_t Hs, int64_t Ws, int64_t Hf,
                                            int64_t Wf, int kernel_size, int dtype, void* stream) {
    const char* fn = "ffwm_block_extractor_forward";
    FFWM_REQUIRE(source && flow_field && output, FFWM_ERR_ARG, "%s: NULL tensor pointer", fn);
    if (int rc = check_dims(fn, B, C, Hs, Ws, Hf, Wf, kernel_size, dtype)) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dtype == FFWM_F32)
        return launch_fwd<float>((const float*)source, (const float*)flow_field, (float*)output, B, C,
                                 Hs, Ws, Hf, Wf, kernel_size, st);
    return launch_fwd<double>((const double*)source, (const double*)flow_field, (double*)output, B, C,
                              Hs, Ws, Hf, Wf, kernel_size, st);
}

extern "C" int ffwm_block_extractor_backward(const void* source, const void* flow_field,
                                             const void* grad_output, void* grad_source,
                                             void* grad_flow_field, int64_t B, int64_t C, int64_t Hs,
                                             int64_t Ws, int64_t Hf, int64_t Wf, int kernel_size,
                                             int dtype, void* stream) {
    const char* fn = "ffwm_block_extractor_backward";
    FFWM_REQUIRE(source && flow_field && grad_output, FFWM_ERR_ARG, "%s: NULL tensor pointer", fn);
    if (int rc = check_dims(fn, B, C, Hs, Ws, Hf, Wf, kernel_size, dtype)) return rc;
    if (!grad_source && !grad_flow_field) return FFWM_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dtype == FFWM_F32)
        return launch_bwd<float>((const float*)source, (const float*)flow_field, (const float*)grad_output,
                                 (float*)grad_source, (float*)grad_flow_field, B, C, Hs, Ws, Hf, Wf,
                                 kernel_size, st);
    return launch_bwd<double>((const double*)source, (const double*)flow_field, (const double*)grad_output,
                              (double*)grad_source, (double*)grad_flow_field, B, C, Hs, Ws, Hf, Wf,
                              kernel_size, st);
}

// grad_output read through its element strides (NULL / contiguous strides: the entry point above).  A strided grad_output -- autograd
// hands one over whenever the gradient is an expanded or permuted view, and the reference's Function drops the result of its
// .contiguous() call (models/external_function.py:46-47), so its kernels really read it that way (DIM3_INDEX,
// block_extractor_kernel.cu:8-15) -- takes the per-element kernel: correct for any layout, not a tuned path.
template <typename T>
int launch_bwd_strided(const T* src, const T* flow, const T* gout, T* gsrc, T* gflow, int64_t B, int64_t C, int64_t Hs, int64_t Ws,
                       int64_t Hf, int64_t Wf, int k, const int64_t* st4, hipStream_t st) {
    const int64_t n = B * C * k * Hf * k * Wf;
    const unsigned grid = static_cast<unsigned>(n / kBlock + 1 < 16384 ? n / kBlock + 1 : 16384);
    const double bytes = sizeof(T) * static_cast<double>(B) * (static_cast<double>(C) * k * k * Hf * Wf + 2.0 * C * Hs * Ws + 4.0 * Hf * Wf);
    LaunchScope ls("block_extractor_bwd_strided", st, bytes);
    // (x stride 0 is the kernel's "contiguous" mark: an expanded last dimension passes through a stride struct with x = 0 replaced below)
    FFWM_REQUIRE(st4[3] != 0 || k * Wf == 1, FFWM_ERR_ARG, "ffwm_block_extractor_backward_strided: a grad_output expanded along its last dimension (stride 0) is not supported");
    hipLaunchKernelGGL((be_bwd_generic<T>), dim3(grid), dim3(kBlock), 0, st, src, flow, gout, gsrc, gflow, n, (int)C, (int)Hs, (int)Ws,
                       (int)Hf, (int)Wf, k, GoStrides{st4[0], st4[1], st4[2], st4[3] != 0 ? st4[3] : 1});
    return check_launch("ffwm_block_extractor_backward_strided");
}

extern "C" int ffwm_block_extractor_backward_strided(const void* source, const void* flow_field, const void* grad_output,
                                                     const int64_t* grad_output_strides, void* grad_source, void* grad_flow_field,
                                                     int64_t B, int64_t C, int64_t Hs, int64_t Ws, int64_t Hf, int64_t Wf,
                                                     int kernel_size, int dtype, void* stream) {
    const char* fn = "ffwm_block_extractor_backward_strided";
    if (go_contiguous(grad_output_strides, C, kernel_size * Hf, kernel_size * Wf))
        return ffwm_block_extractor_backward(source, flow_field, grad_output, grad_source, grad_flow_field, B, C, Hs, Ws, Hf, Wf,
                                             kernel_size, dtype, stream);
    FFWM_REQUIRE(source && flow_field && grad_output, FFWM_ERR_ARG, "%s: NULL tensor pointer", fn);
    if (int rc = check_dims(fn, B, C, Hs, Ws, Hf, Wf, kernel_size, dtype)) return rc;
    for (int d = 0; d < 4; ++d)
        FFWM_REQUIRE(grad_output_strides[d] >= 0, FFWM_ERR_ARG, "%s: negative strides are not supported", fn);
    if (!grad_source && !grad_flow_field) return FFWM_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dtype == FFWM_F32)
        return launch_bwd_strided<float>((const float*)source, (const float*)flow_field, (const float*)grad_output, (float*)grad_source,
                                         (float*)grad_flow_field, B, C, Hs, Ws, Hf, Wf, kernel_size, grad_output_strides, st);
    return launch_bwd_strided<double>((const double*)source, (const double*)flow_field, (const double*)grad_output, (double*)grad_source,
                                      (double*)grad_flow_field, B, C, Hs, Ws, Hf, Wf, kernel_size, grad_output_strides, st);
}

extern "C" int ffwm_block_attention_forward(const void* source, const void* flow_field, const void* weights, void* output,
                                            int64_t B, int64_t C, int64_t Hs, int64_t Ws, int64_t Hf, int64_t Wf,
                                            int kernel_size, int dtype, void* stream) {
    const char* fn = "ffwm_block_attention_forward";
    FFWM_REQUIRE(source && flow_field && weights && output, FFWM_ERR_ARG, "%s: NULL tensor pointer", fn);
    if (int rc = check_dims(fn, B, C, Hs, Ws, Hf, Wf, kernel_size, dtype)) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dtype == FFWM_F32)
        return launch_attn_fwd<float>((const float*)source, (const float*)flow_field, (const float*)weights, (float*)output,
                                      B, C, Hs, Ws, Hf, Wf, kernel_size, st);
    return launch_attn_fwd<double>((const double*)source, (const double*)flow_field, (const double*)weights,
                                   (double*)output, B, C, Hs, Ws, Hf, Wf, kernel_size, st);
}

extern "C" int ffwm_block_attention_backward(const void* source, const void* flow_field, const void* weights,
                                             const void* grad_output, void* grad_source, void* grad_flow_field,
                                             void* grad_weights, int64_t B, int64_t C, int64_t Hs, int64_t Ws,
                                             int64_t Hf, int64_t Wf, int kernel_size, int dtype, void* stream) {
    const char* fn = "ffwm_block_attention_backward";
    FFWM_REQUIRE(source && flow_field && weights && grad_output, FFWM_ERR_ARG, "%s: NULL tensor pointer", fn);
    if (int rc = check_dims(fn, B, C, Hs, Ws, Hf, Wf, kernel_size, dtype)) return rc;
    if (!grad_source && !grad_flow_field && !grad_weights) return FFWM_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dtype == FFWM_F32)
        return launch_attn_bwd<float>((const float*)source, (const float*)flow_field, (const float*)weights,
                                      (const float*)grad_output, (float*)grad_source, (float*)grad_flow_field,
                                      (float*)grad_weights, B, C, Hs, Ws, Hf, Wf, kernel_size, st);
    return launch_attn_bwd<double>((const double*)source, (const double*)flow_field, (const double*)weights,
                                   (const double*)grad_output, (double*)grad_source, (double*)grad_flow_field,
                                   (double*)grad_weights, B, C, Hs, Ws, Hf, Wf, kernel_size, st);
}
