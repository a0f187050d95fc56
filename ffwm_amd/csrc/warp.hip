// warp.hip -- WarpNet (flow-guided bilinear feature warp) for gfx950, optionally fused with the
// flip + concat that FFWM's warp-attention module applies right after it.
//
// Reference: models/base_networks.py:168-173 -- F.grid_sample(images, flow.permute(0,2,3,1),
// mode='bilinear') with padding_mode='zeros', align_corners=False -- followed in FFWM.forward
// (:326-329) by torch.flip(w, (3,)) and torch.cat((w, flipped), 1): three full-tensor passes
// (3 reads, 3 writes).  Here: 1 read of the features, 2 writes, one kernel.
//
//   * one thread per OUTPUT PIXEL looping over a channel slab: the unnormalised coordinate, the
//     four corner weights and the four (validity-masked) byte offsets are computed once;
//   * zeros padding is done by the hardware: an out-of-range corner gets the byte offset
//     0xFFFFFFF0, the buffer range check returns 0 for it, and its weight is forced to 0 so
//     0 * w never produces NaN;
//   * a wave covers 64 consecutive x: the direct store is one contiguous run, the flipped store
//     is the mirrored contiguous run;
//   * backward: d(flow) accumulates in registers over the channel slab (2 atomics per pixel per
//     slab); d(feat) is the 4-corner scatter.
#include <algorithm>
#include <vector>

#include "common.hpp"
#include <type_traits>

namespace ffwm {
namespace {

constexpr unsigned kOob = 0xFFFFFFF0u;

template <typename T>
struct Corners {
    unsigned off[4];   // nw, ne, sw, se byte offsets (kOob when outside the image)
    T w[4];            // matching weights (0 when outside)
    T dxw[2], dyw[2];  // (x1 - ix), (ix - x0), (y1 - iy), (iy - y0): for d(flow)
    bool valid[4];
};

// ATen grid_sampler_2d, bilinear / zeros / align_corners=False:
//   ix = ((gx + 1) * W_in - 1) / 2, corner weights from the opposite corner.
template <typename T>
__device__ __forceinline__ void make_corners(Corners<T>& c, T gx, T gy, int Hi, int Wi) {
    const T ix = ((gx + 1) * static_cast<T>(Wi) - 1) / 2;
    const T iy = ((gy + 1) * static_cast<T>(Hi) - 1) / 2;
    const T fx = floor_t(ix), fy = floor_t(iy);
    // corner indices as floats are exact for |f| < 2^24 (2^53); anything outside [-1, size] is
    // out of range on both corners anyway, so clamp before converting (NaN -> out of range).
    const T lim_x = static_cast<T>(Wi), lim_y = static_cast<T>(Hi);
    const bool okx = (fx >= static_cast<T>(-1)) && (fx <= lim_x);
    const bool oky = (fy >= static_cast<T>(-1)) && (fy <= lim_y);
    const int x0 = okx ? static_cast<int>(fx) : -2;
    const int y0 = oky ? static_cast<int>(fy) : -2;
    const int x1 = x0 + 1, y1 = y0 + 1;
    const T x1f = fx + 1, y1f = fy + 1;     // == (T)x1, (T)y1 whenever a corner is valid
    const bool finite = okx && oky;         // otherwise every corner is outside: all terms drop out
    c.dxw[0] = finite ? x1f - ix : static_cast<T>(0);
    c.dxw[1] = finite ? ix - fx : static_cast<T>(0);
    c.dyw[0] = finite ? y1f - iy : static_cast<T>(0);
    c.dyw[1] = finite ? iy - fy : static_cast<T>(0);
    const bool vx0 = x0 >= 0 && x0 < Wi, vx1 = x1 >= 0 && x1 < Wi;
    const bool vy0 = y0 >= 0 && y0 < Hi, vy1 = y1 >= 0 && y1 < Hi;
    c.valid[0] = vy0 && vx0;
    c.valid[1] = vy0 && vx1;
    c.valid[2] = vy1 && vx0;
    c.valid[3] = vy1 && vx1;
    const T w[4] = {c.dxw[0] * c.dyw[0], c.dxw[1] * c.dyw[0], c.dxw[0] * c.dyw[1], c.dxw[1] * c.dyw[1]};
    const int xs[4] = {x0, x1, x0, x1};
    const int ys[4] = {y0, y0, y1, y1};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        c.off[q] = c.valid[q] ? static_cast<unsigned>(ys[q] * Wi + xs[q]) * static_cast<unsigned>(sizeof(T)) : kOob;
        c.w[q] = c.valid[q] ? w[q] : static_cast<T>(0);
    }
}

template <typename T, bool FLIP>
__device__ __forceinline__ void warp_fwd_body(const T* __restrict__ feat, const T* __restrict__ flow, T* __restrict__ out, int C,
                                              int Hi, int Wi, int H, int W, const TileCoord tc, int cs, int nt = 0) {
    const int x = tc.xf, y = tc.yf;
    if (x >= W || y >= H) return;
    const size_t plane = static_cast<size_t>(H) * W;
    const size_t foff = static_cast<size_t>(tc.b) * 2 * plane + static_cast<size_t>(y) * W + x;
    Corners<T> cn;
    make_corners<T>(cn, flow[foff], flow[foff + plane], Hi, Wi);

    const int c0 = tc.slab * cs;
    const int c1 = (c0 + cs < C) ? c0 + cs : C;
    const int Co = FLIP ? 2 * C : C;
    const size_t iplane = static_cast<size_t>(Hi) * Wi;
    const unsigned ibytes = static_cast<unsigned>(iplane * sizeof(T));
    const unsigned obytes = static_cast<unsigned>(plane * sizeof(T));
    const T* fp = feat + (static_cast<size_t>(tc.b) * C + c0) * iplane;
    T* op = out + (static_cast<size_t>(tc.b) * Co + c0) * plane;
    const unsigned o_direct = static_cast<unsigned>(y * W + x) * static_cast<unsigned>(sizeof(T));
    const unsigned o_flip = static_cast<unsigned>(y * W + (W - 1 - x)) * static_cast<unsigned>(sizeof(T));
    const size_t flip_planes = static_cast<size_t>(C) * plane;

    // four channels per trip: their sixteen gathers are all in flight before the first product is formed (one channel per trip
    // left the memory pipeline with four loads per thread between waits)
    constexpr int U = 4;
    int c = c0;
    for (; c + U <= c1; c += U, fp += U * iplane, op += U * plane) {
        T s[U][4];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const rsrc_t rf = make_rsrc(fp + u * iplane, ibytes);
#pragma unroll
            for (int q = 0; q < 4; ++q) s[u][q] = buf_ld<T>(rf, cn.off[q]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            T v = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) v += s[u][q] * cn.w[q];
            ElemRow<T, 1> r;
            r.v[0] = v;
            if (nt) {            // (wave-uniform) streaming stores: the output is read once, by a later kernel
                buf_store_row_nt<T, 1>(make_rsrc(op + u * plane, obytes), o_direct, r);
                if (FLIP) buf_store_row_nt<T, 1>(make_rsrc(op + u * plane + flip_planes, obytes), o_flip, r);
            } else {
                buf_store_row<T, 1>(make_rsrc(op + u * plane, obytes), o_direct, r);
                if (FLIP) buf_store_row<T, 1>(make_rsrc(op + u * plane + flip_planes, obytes), o_flip, r);
            }
        }
    }
    for (; c < c1; ++c, fp += iplane, op += plane) {
        const rsrc_t rf = make_rsrc(fp, ibytes);
        T v = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) v += buf_ld<T>(rf, cn.off[q]) * cn.w[q];
        ElemRow<T, 1> r;
        r.v[0] = v;
        buf_store_row<T, 1>(make_rsrc(op, obytes), o_direct, r);
        if (FLIP) buf_store_row<T, 1>(make_rsrc(op + flip_planes, obytes), o_flip, r);
    }
}

template <typename T, bool FLIP>
__global__ void __launch_bounds__(kBlock)
warp_fwd_kernel(const T* __restrict__ feat, const T* __restrict__ flow, T* __restrict__ out, int C,
                int Hi, int Wi, int H, int W, int tiles_x, int tiles_y, int cslabs, int cs,
                int remap) {
    warp_fwd_body<T, FLIP>(feat, flow, out, C, Hi, Wi, H, W, decode_tile(tiles_x, tiles_y, cslabs, remap), cs);
}

// ---- several warps in ONE launch.  A train step of FFWM issues its warps in small groups whose members are
// independent and individually launch-bound (models/ffwm_model.py:74-88: eight 32 x 32 part crops; losses.py:149: three
// illumination warps; base_networks.py:323-333: the three warp-attention levels, 1.6 - 12.7 MB per image): the problem
// table travels in the kernel arguments and a workgroup finds its problem from its index.
constexpr int kMaxWarpProblems = 8;
struct WarpProblem {
    const void* feat;
    const void* flow;
    void* out;           // forward: output; backward: grad_flow (may be NULL)
    const void* gout;    // backward: grad_output
    int C, Hi, Wi, H, W;
    int tiles_x, tiles_y, cslabs, cs;
    int lds;             // 1: LDS-staged 64 x 16 tiles (warp_fwd_lds_body / warp_bwd_flow_lds_body), 0: direct gathers on 64 x 4 tiles
    unsigned begin;      // first workgroup of this problem (a multiple of 8: workgroup b runs on XCD b % 8)
    unsigned nblk;       // its real workgroups; the ids up to the next multiple of 8 exit at once
};
struct WarpTable {
    int n;
    int nt;              // streaming stores (options().warp_nt)
    WarpProblem p[kMaxWarpProblems];
};

__device__ __forceinline__ TileCoord decode_tile_local(unsigned t, int tiles_x, int tiles_y, int cslabs) {
    TileCoord tc;
    const int tx = t % tiles_x;
    t /= tiles_x;
    const int ty = t % tiles_y;
    t /= tiles_y;
    tc.slab = t % cslabs;
    tc.b = t / cslabs;
    tc.xf = tx * kTileX + (threadIdx.x & (kTileX - 1));
    tc.yf = ty * kTileY + (threadIdx.x / kTileX);
    return tc;
}

// ------------------------------------------------------------------------------ forward, LDS-staged
// The direct kernel above issues 4 dword gathers + 1-2 dword stores per pixel and channel: with 4 bytes per
// lane every memory instruction costs a full address-processing slot, so it is instruction-rate-bound
// (3.4 TB/s on a smooth flow, 1.2 TB/s on a random one) long before HBM is.  Here a block owns a 64 x 16
// output tile; it reduces the bounding box of the tile's sampling positions (wave shuffles + one LDS hop),
// and if the box fits 128 x 32 source cells it copies the ZERO-PADDED box of each channel global -> LDS with
// 16-byte row-contiguous loads (the copy of channel c+1 overlaps the arithmetic of channel c), reads the
// four corners from LDS (one address + immediates), and stores 4 consecutive pixels per lane as one
// dwordx4 (direct) + one dwordx4 (mirrored, for the fused flip + cat).  A tile whose box is too large
// (flow zooming out by more than ~2x, or random) falls back to direct gathers -- a block-uniform choice.
constexpr int kWlTileX = 64, kWlTileY = 16, kWlPix = 4, kWlRows = 32, kWlCols = 128;

struct WlShared {
    float tile[2][kWlRows * kWlCols];       // two staged boxes (the copy of channel c+1 overlaps the arithmetic of channel c)
    int red[4][kBlock / kWave];
};

template <bool FLIP>
__device__ __forceinline__ void warp_fwd_lds_body(const float* __restrict__ feat, const float* __restrict__ flow, float* __restrict__ out,
                                                  int C, int Hi, int Wi, int H, int W, int tx, int ty, int slab, int b, int cs, int nt,
                                                  WlShared& sh) {
    using T = float;
    constexpr int NW = kBlock / kWave;
    constexpr unsigned E = sizeof(T);
    T (&tile)[2][kWlRows * kWlCols] = sh.tile;
    int (&red)[4][NW] = sh.red;
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    const int y = ty * kWlTileY + (threadIdx.x >> 4);
    const int xb = tx * kWlTileX + (threadIdx.x & 15) * kWlPix;
    const bool yin = y < H;
    const size_t plane = static_cast<size_t>(H) * W;
    const rsrc_t rfl = make_rsrc(flow + static_cast<size_t>(b) * 2 * plane, static_cast<unsigned>(2 * plane * E));

    // ---- per-pixel corners (ATen grid_sampler_2d, bilinear / zeros / align_corners=False)
    int x0[kWlPix], y0[kWlPix];
    T w[kWlPix][4];
    bool live[kWlPix];
    int umin = 0x7fffffff, umax = -0x7fffffff, vmin = 0x7fffffff, vmax = -0x7fffffff;
#pragma unroll
    for (int k = 0; k < kWlPix; ++k) {
        const int x = xb + k;
        const bool pin = yin && x < W;
        const unsigned fo = pin ? (static_cast<unsigned>(y) * W + x) * E : kOob;
        const T gx = buf_ld<T>(rfl, fo), gy = buf_ld<T>(rfl, pin ? fo + static_cast<unsigned>(plane * E) : kOob);
        Corners<T> cn;
        make_corners<T>(cn, gx, gy, Hi, Wi);
        const bool any = pin && (cn.valid[0] || cn.valid[1] || cn.valid[2] || cn.valid[3]);
        live[k] = any;
        // a pixel with a valid corner has floor(ix) in [-1, Wi-1], floor(iy) in [-1, Hi-1]
        const T ix = ((gx + 1) * static_cast<T>(Wi) - 1) / 2, iy = ((gy + 1) * static_cast<T>(Hi) - 1) / 2;
        x0[k] = any ? static_cast<int>(floor_t(ix)) : 0;
        y0[k] = any ? static_cast<int>(floor_t(iy)) : 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) w[k][q] = any ? ((q & 1) ? cn.dxw[1] : cn.dxw[0]) * ((q >> 1) ? cn.dyw[1] : cn.dyw[0]) : static_cast<T>(0);
        if (any) {
            umin = min(umin, x0[k]); umax = max(umax, x0[k] + 1);
            vmin = min(vmin, y0[k]); vmax = max(vmax, y0[k] + 1);
        }
    }
    umin = wave_min(umin); umax = wave_max(umax); vmin = wave_min(vmin); vmax = wave_max(vmax);
    if (lane == 0) { red[0][wave] = umin; red[1][wave] = umax; red[2][wave] = vmin; red[3][wave] = vmax; }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NW; ++k) {
        umin = min(umin, red[0][k]); umax = max(umax, red[1][k]);
        vmin = min(vmin, red[2][k]); vmax = max(vmax, red[3][k]);
    }
    const bool empty = umin > umax;                    // no pixel of the tile samples inside the image
    const int bx0 = empty ? 0 : (umin & ~3), by0 = empty ? 0 : vmin;          // 16-byte aligned box origin
    const int bw = empty ? 0 : umax - bx0 + 1, bh = empty ? 0 : vmax - by0 + 1;
    const bool use_lds = bw <= kWlCols && bh <= kWlRows;

    const int c0 = slab * cs;
    const int c1 = (c0 + cs < C) ? c0 + cs : C;
    const int Co = FLIP ? 2 * C : C;
    const size_t iplane = static_cast<size_t>(Hi) * Wi;
    const unsigned ibytes = static_cast<unsigned>(iplane * E);
    const unsigned obytes = static_cast<unsigned>(plane * E);
    const T* fp = feat + (static_cast<size_t>(b) * C + c0) * iplane;
    T* op = out + (static_cast<size_t>(b) * Co + c0) * plane;
    const size_t flip_planes = static_cast<size_t>(C) * plane;
    const bool full = yin && xb + kWlPix <= W;         // all 4 pixels inside the row: one dwordx4 store
    const unsigned o_direct = (static_cast<unsigned>(y) * W + xb) * E;
    const unsigned o_flip = (static_cast<unsigned>(y) * W + (W - kWlPix - xb)) * E;

    auto store4 = [&](T* plane_out, const T (&v)[kWlPix]) {
        const rsrc_t ro = make_rsrc(plane_out, obytes);
        if (full) {
            ElemRow<T, kWlPix> r;
#pragma unroll
            for (int k = 0; k < kWlPix; ++k) r.v[k] = v[k];
            if (nt) buf_store_row_nt<T, kWlPix>(ro, o_direct, r);
            else buf_store_row<T, kWlPix>(ro, o_direct, r);
            if (FLIP) {
                ElemRow<T, kWlPix> m;
#pragma unroll
                for (int k = 0; k < kWlPix; ++k) m.v[k] = v[kWlPix - 1 - k];
                if (nt) buf_store_row_nt<T, kWlPix>(make_rsrc(plane_out + flip_planes, obytes), o_flip, m);
                else buf_store_row<T, kWlPix>(make_rsrc(plane_out + flip_planes, obytes), o_flip, m);
            }
        } else if (yin) {
#pragma unroll
            for (int k = 0; k < kWlPix; ++k) {
                if (xb + k < W) {
                    ElemRow<T, 1> r;
                    r.v[0] = v[k];
                    buf_store_row<T, 1>(ro, (static_cast<unsigned>(y) * W + xb + k) * E, r);
                    if (FLIP)
                        buf_store_row<T, 1>(make_rsrc(plane_out + flip_planes, obytes),
                                            (static_cast<unsigned>(y) * W + (W - 1 - xb - k)) * E, r);
                }
            }
        }
    };

    if (use_lds) {
        // staging map: chunk = 4 consecutive columns; thread handles box row (tid >> 5) + 8 i, columns (tid & 31) * 4
        constexpr int NCH = kWlRows * kWlCols / 4 / kBlock;       // 4 chunks per thread
        unsigned goff[NCH];
        unsigned keep[NCH];                                       // per-element validity bits of the chunk
        const int col4 = (threadIdx.x & 31) * 4;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int r = (threadIdx.x >> 5) + 8 * i;
            const int gy = by0 + r, gx = bx0 + col4;
            const bool rowok = r < bh && col4 < bw && gy >= 0 && gy < Hi && gx >= 0;     // gx is a multiple of 4
            unsigned m = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) m |= (rowok && gx + e < Wi) ? (1u << e) : 0u;
            keep[i] = m;
            goff[i] = m ? (static_cast<unsigned>(gy) * Wi + gx) * E : kOob;
        }
        u32x4 stage[NCH];
        auto fetch = [&](const T* pl) {
            const rsrc_t rs = make_rsrc(pl, ibytes);
#pragma unroll
            for (int i = 0; i < NCH; ++i) stage[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, goff[i], 0, 0);
        };
        auto commit = [&](T* buf) {
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                u32x4 v = stage[i];
                v.x = (keep[i] & 1u) ? v.x : 0u;      // zero padding lives in LDS: the corner loop needs no masks
                v.y = (keep[i] & 2u) ? v.y : 0u;
                v.z = (keep[i] & 4u) ? v.z : 0u;
                v.w = (keep[i] & 8u) ? v.w : 0u;
                *reinterpret_cast<u32x4*>(buf + ((threadIdx.x >> 5) + 8 * i) * kWlCols + col4) = v;
            }
        };
        int lbase[kWlPix];
#pragma unroll
        for (int k = 0; k < kWlPix; ++k) lbase[k] = live[k] ? (y0[k] - by0) * kWlCols + (x0[k] - bx0) : 0;
        fetch(fp);
        commit(tile[0]);
        __syncthreads();
        int p = 0;
        for (int c = c0; c < c1; ++c, op += plane, p ^= 1) {
            const bool more = c + 1 < c1;
            if (more) fetch(fp + static_cast<size_t>(c + 1 - c0) * iplane);      // in flight during the math
            T v[kWlPix];
#pragma unroll
            for (int k = 0; k < kWlPix; ++k) {
                const T* nb = tile[p] + lbase[k];
                T s = 0;                                // same summation order as the direct kernel / the oracle
                s += nb[0] * w[k][0];
                s += nb[1] * w[k][1];
                s += nb[kWlCols] * w[k][2];
                s += nb[kWlCols + 1] * w[k][3];
                v[k] = s;
            }
            store4(op, v);
            if (more) commit(tile[p ^ 1]);
            __syncthreads();
        }
        return;
    }

    // ---- fallback: direct gathers with hardware zero padding (kOob offsets), 4 pixels per thread
    unsigned off[kWlPix][4];
    T wm[kWlPix][4];
#pragma unroll
    for (int k = 0; k < kWlPix; ++k) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int cx = x0[k] + (q & 1), cy = y0[k] + (q >> 1);
            const bool ok = live[k] && cx >= 0 && cx < Wi && cy >= 0 && cy < Hi;
            off[k][q] = ok ? (static_cast<unsigned>(cy) * Wi + cx) * E : kOob;
            wm[k][q] = ok ? w[k][q] : static_cast<T>(0);
        }
    }
    for (int c = c0; c < c1; ++c, fp += iplane, op += plane) {
        const rsrc_t rf = make_rsrc(fp, ibytes);
        T v[kWlPix];
#pragma unroll
        for (int k = 0; k < kWlPix; ++k) {
            T s = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) s += buf_ld<T>(rf, off[k][q]) * wm[k][q];
            v[k] = s;
        }
        store4(op, v);
    }
}

template <bool FLIP>
__global__ void __launch_bounds__(kBlock)
warp_fwd_lds_kernel(const float* __restrict__ feat, const float* __restrict__ flow, float* __restrict__ out, int C,
                    int Hi, int Wi, int H, int W, int tiles_x, int tiles_y, int cslabs, int cs, int remap, int nt) {
    __shared__ __attribute__((aligned(16))) WlShared sh;
    unsigned t = xcd_remap(blockIdx.x, gridDim.x, remap);
    const int tx = t % tiles_x;
    t /= tiles_x;
    const int ty = t % tiles_y;
    t /= tiles_y;
    const int slab = t % cslabs;
    const int b = t / cslabs;
    warp_fwd_lds_body<FLIP>(feat, flow, out, C, Hi, Wi, H, W, tx, ty, slab, b, cs, nt, sh);
}

// ------------------------------------------------------------------------------ d(flow), LDS-staged
// The same tile decomposition for the gradient of the flow field: a block owns 64 x 16 output pixels (4 consecutive pixels per
// lane) and a slab of channels; per channel it stages the zero-padded source box through LDS with 16-byte loads (the next
// channel's copy in flight during the arithmetic), reads grad_output as one dwordx4 (direct half) + one mirrored dwordx4 (flipped
// half of the fused flip + cat) per lane, takes the four corners from LDS and accumulates d(ix), d(iy) in registers over the
// slab: 2 atomics per pixel and slab at the end.  Per pixel and channel: 0.5 + 1 vector memory instructions instead of the
// direct kernel's 2 + 4 dword ones.  Same arithmetic, same order over the channels of a slab as warp_bwd_body.
template <bool FLIP>
__device__ __forceinline__ void warp_bwd_flow_lds_body(const float* __restrict__ feat, const float* __restrict__ flow,
                                                       const float* __restrict__ gout, float* __restrict__ gflow, int C, int Hi, int Wi,
                                                       int H, int W, int tx, int ty, int slab, int b, int cs, WlShared& sh) {
    using T = float;
    constexpr int NW = kBlock / kWave;
    constexpr unsigned E = sizeof(T);
    T (&tile)[2][kWlRows * kWlCols] = sh.tile;
    int (&red)[4][NW] = sh.red;
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    const int y = ty * kWlTileY + (threadIdx.x >> 4);
    const int xb = tx * kWlTileX + (threadIdx.x & 15) * kWlPix;
    const bool yin = y < H;
    const size_t plane = static_cast<size_t>(H) * W;
    const rsrc_t rfl = make_rsrc(flow + static_cast<size_t>(b) * 2 * plane, static_cast<unsigned>(2 * plane * E));

    int x0[kWlPix], y0[kWlPix];
    T dxw[kWlPix][2], dyw[kWlPix][2];
    bool live[kWlPix], pin[kWlPix];
    int umin = 0x7fffffff, umax = -0x7fffffff, vmin = 0x7fffffff, vmax = -0x7fffffff;
#pragma unroll
    for (int k = 0; k < kWlPix; ++k) {
        const int x = xb + k;
        pin[k] = yin && x < W;
        const unsigned fo = pin[k] ? (static_cast<unsigned>(y) * W + x) * E : kOob;
        const T gx = buf_ld<T>(rfl, fo), gy = buf_ld<T>(rfl, pin[k] ? fo + static_cast<unsigned>(plane * E) : kOob);
        Corners<T> cn;
        make_corners<T>(cn, gx, gy, Hi, Wi);
        const bool any = pin[k] && (cn.valid[0] || cn.valid[1] || cn.valid[2] || cn.valid[3]);
        live[k] = any;
        const T ix = ((gx + 1) * static_cast<T>(Wi) - 1) / 2, iy = ((gy + 1) * static_cast<T>(Hi) - 1) / 2;
        x0[k] = any ? static_cast<int>(floor_t(ix)) : 0;
        y0[k] = any ? static_cast<int>(floor_t(iy)) : 0;
        // a pixel without a valid corner contributes nothing (every sample it would read is padding)
        dxw[k][0] = any ? cn.dxw[0] : static_cast<T>(0);
        dxw[k][1] = any ? cn.dxw[1] : static_cast<T>(0);
        dyw[k][0] = any ? cn.dyw[0] : static_cast<T>(0);
        dyw[k][1] = any ? cn.dyw[1] : static_cast<T>(0);
        if (any) {
            umin = min(umin, x0[k]); umax = max(umax, x0[k] + 1);
            vmin = min(vmin, y0[k]); vmax = max(vmax, y0[k] + 1);
        }
    }
    umin = wave_min(umin); umax = wave_max(umax); vmin = wave_min(vmin); vmax = wave_max(vmax);
    if (lane == 0) { red[0][wave] = umin; red[1][wave] = umax; red[2][wave] = vmin; red[3][wave] = vmax; }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NW; ++k) {
        umin = min(umin, red[0][k]); umax = max(umax, red[1][k]);
        vmin = min(vmin, red[2][k]); vmax = max(vmax, red[3][k]);
    }
    const bool empty = umin > umax;
    const int bx0 = empty ? 0 : (umin & ~3), by0 = empty ? 0 : vmin;
    const int bw = empty ? 0 : umax - bx0 + 1, bh = empty ? 0 : vmax - by0 + 1;
    const bool use_lds = bw <= kWlCols && bh <= kWlRows;

    const int c0 = slab * cs;
    const int c1 = (c0 + cs < C) ? c0 + cs : C;
    const int Co = FLIP ? 2 * C : C;
    const size_t iplane = static_cast<size_t>(Hi) * Wi;
    const unsigned ibytes = static_cast<unsigned>(iplane * E);
    const unsigned obytes = static_cast<unsigned>(plane * E);
    const T* fp = feat + (static_cast<size_t>(b) * C + c0) * iplane;
    const T* op = gout + (static_cast<size_t>(b) * Co + c0) * plane;
    const size_t flip_planes = static_cast<size_t>(C) * plane;
    const bool full = yin && xb + kWlPix <= W;
    const unsigned o_direct = full ? (static_cast<unsigned>(y) * W + xb) * E : kOob;
    const unsigned o_flip = full ? (static_cast<unsigned>(y) * W + (W - kWlPix - xb)) * E : kOob;

    // grad_output of the lane's 4 pixels: direct half + mirrored flipped half
    auto load_g = [&](const T* plane_g, T (&g)[kWlPix]) {
        if (full || !yin) {        // (rows outside the image: kOob offsets read 0)
            const u32x4 d = __builtin_amdgcn_raw_buffer_load_b128(make_rsrc(plane_g, obytes), o_direct, 0, 0);
            g[0] = __uint_as_float(d.x); g[1] = __uint_as_float(d.y); g[2] = __uint_as_float(d.z); g[3] = __uint_as_float(d.w);
            if (FLIP) {
                const u32x4 m = __builtin_amdgcn_raw_buffer_load_b128(make_rsrc(plane_g + flip_planes, obytes), o_flip, 0, 0);
                g[0] += __uint_as_float(m.w); g[1] += __uint_as_float(m.z); g[2] += __uint_as_float(m.y); g[3] += __uint_as_float(m.x);
            }
        } else {
#pragma unroll
            for (int k = 0; k < kWlPix; ++k) {
                const unsigned od = pin[k] ? (static_cast<unsigned>(y) * W + xb + k) * E : kOob;
                g[k] = buf_ld<T>(make_rsrc(plane_g, obytes), od);
                if (FLIP) g[k] += buf_ld<T>(make_rsrc(plane_g + flip_planes, obytes),
                                            pin[k] ? (static_cast<unsigned>(y) * W + (W - 1 - xb - k)) * E : kOob);
            }
        }
    };
    T gix[kWlPix], giy[kWlPix];
#pragma unroll
    for (int k = 0; k < kWlPix; ++k) gix[k] = giy[k] = 0;
    auto one = [&](int k, const T g, const T s0, const T s1, const T s2, const T s3) {
        // ATen grid_sampler_2d_backward, the operation order of warp_bwd_body
        gix[k] -= s0 * dyw[k][0] * g;
        giy[k] -= s0 * dxw[k][0] * g;
        gix[k] += s1 * dyw[k][0] * g;
        giy[k] -= s1 * dxw[k][1] * g;
        gix[k] -= s2 * dyw[k][1] * g;
        giy[k] += s2 * dxw[k][0] * g;
        gix[k] += s3 * dyw[k][1] * g;
        giy[k] += s3 * dxw[k][1] * g;
    };

    if (use_lds) {
        constexpr int NCH = kWlRows * kWlCols / 4 / kBlock;
        unsigned goff[NCH];
        unsigned keep[NCH];
        const int col4 = (threadIdx.x & 31) * 4;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int r = (threadIdx.x >> 5) + 8 * i;
            const int gy = by0 + r, gx = bx0 + col4;
            const bool rowok = r < bh && col4 < bw && gy >= 0 && gy < Hi && gx >= 0;
            unsigned m = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) m |= (rowok && gx + e < Wi) ? (1u << e) : 0u;
            keep[i] = m;
            goff[i] = m ? (static_cast<unsigned>(gy) * Wi + gx) * E : kOob;
        }
        u32x4 stage[NCH];
        auto fetch = [&](const T* pl) {
            const rsrc_t rs = make_rsrc(pl, ibytes);
#pragma unroll
            for (int i = 0; i < NCH; ++i) stage[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, goff[i], 0, 0);
        };
        auto commit = [&](T* buf) {
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                u32x4 v = stage[i];
                v.x = (keep[i] & 1u) ? v.x : 0u;
                v.y = (keep[i] & 2u) ? v.y : 0u;
                v.z = (keep[i] & 4u) ? v.z : 0u;
                v.w = (keep[i] & 8u) ? v.w : 0u;
                *reinterpret_cast<u32x4*>(buf + ((threadIdx.x >> 5) + 8 * i) * kWlCols + col4) = v;
            }
        };
        int lbase[kWlPix];
#pragma unroll
        for (int k = 0; k < kWlPix; ++k) lbase[k] = live[k] ? (y0[k] - by0) * kWlCols + (x0[k] - bx0) : 0;
        fetch(fp);
        T gn[kWlPix];
        load_g(op, gn);                        // grad_output of a channel is requested one channel ahead, like the source box
        commit(tile[0]);
        __syncthreads();
        int p = 0;
        for (int c = c0; c < c1; ++c, op += plane, p ^= 1) {
            const bool more = c + 1 < c1;
            T g[kWlPix];
#pragma unroll
            for (int k = 0; k < kWlPix; ++k) g[k] = gn[k];
            if (more) {
                fetch(fp + static_cast<size_t>(c + 1 - c0) * iplane);
                load_g(op + plane, gn);
            }
#pragma unroll
            for (int k = 0; k < kWlPix; ++k) {
                const T* nb = tile[p] + lbase[k];
                one(k, g[k], nb[0], nb[1], nb[kWlCols], nb[kWlCols + 1]);
            }
            if (more) commit(tile[p ^ 1]);
            __syncthreads();
        }
    } else {
        // fallback (box too large for LDS): direct gathers with hardware zero padding
        unsigned off[kWlPix][4];
#pragma unroll
        for (int k = 0; k < kWlPix; ++k) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int cx = x0[k] + (q & 1), cy = y0[k] + (q >> 1);
                const bool ok = live[k] && cx >= 0 && cx < Wi && cy >= 0 && cy < Hi;
                off[k][q] = ok ? (static_cast<unsigned>(cy) * Wi + cx) * E : kOob;
            }
        }
        for (int c = c0; c < c1; ++c, fp += iplane, op += plane) {
            const rsrc_t rf = make_rsrc(fp, ibytes);
            T g[kWlPix];
            load_g(op, g);
#pragma unroll
            for (int k = 0; k < kWlPix; ++k)
                one(k, g[k], buf_ld<T>(rf, off[k][0]), buf_ld<T>(rf, off[k][1]), buf_ld<T>(rf, off[k][2]), buf_ld<T>(rf, off[k][3]));
        }
    }
    T* gf = gflow + static_cast<size_t>(b) * 2 * plane;
#pragma unroll
    for (int k = 0; k < kWlPix; ++k) {
        if (pin[k]) {
            const size_t fo = static_cast<size_t>(y) * W + xb + k;
            atomic_add(gf + fo, (static_cast<T>(Wi) / 2) * gix[k]);
            atomic_add(gf + fo + plane, (static_cast<T>(Hi) / 2) * giy[k]);
        }
    }
}

template <bool FLIP>
__global__ void __launch_bounds__(kBlock)
warp_bwd_flow_lds_kernel(const float* __restrict__ feat, const float* __restrict__ flow, const float* __restrict__ gout,
                         float* __restrict__ gflow, int C, int Hi, int Wi, int H, int W, int tiles_x, int tiles_y, int cslabs, int cs,
                         int remap) {
    __shared__ __attribute__((aligned(16))) WlShared sh;
    unsigned t = xcd_remap(blockIdx.x, gridDim.x, remap);
    const int tx = t % tiles_x;
    t /= tiles_x;
    const int ty = t % tiles_y;
    t /= tiles_y;
    const int slab = t % cslabs;
    const int b = t / cslabs;
    warp_bwd_flow_lds_body<FLIP>(feat, flow, gout, gflow, C, Hi, Wi, H, W, tx, ty, slab, b, cs, sh);
}

template <typename T, bool FLIP>
__device__ __forceinline__ void warp_bwd_body(const T* __restrict__ feat, const T* __restrict__ flow, const T* __restrict__ gout,
                                              T* __restrict__ gfeat, T* __restrict__ gflow, int C, int Hi, int Wi, int H, int W,
                                              const TileCoord tc, int cs, bool pair_loads = false) {
    const int x = tc.xf, y = tc.yf;
    if (x >= W || y >= H) return;
    const size_t plane = static_cast<size_t>(H) * W;
    const size_t foff = static_cast<size_t>(tc.b) * 2 * plane + static_cast<size_t>(y) * W + x;
    Corners<T> cn;
    make_corners<T>(cn, flow[foff], flow[foff + plane], Hi, Wi);

    const int c0 = tc.slab * cs;
    const int c1 = (c0 + cs < C) ? c0 + cs : C;
    const int Co = FLIP ? 2 * C : C;
    const size_t iplane = static_cast<size_t>(Hi) * Wi;
    const unsigned ibytes = static_cast<unsigned>(iplane * sizeof(T));
    const unsigned obytes = static_cast<unsigned>(plane * sizeof(T));
    const size_t ioff = (static_cast<size_t>(tc.b) * C + c0) * iplane;
    const T* fp = feat + ioff;
    T* gp = gfeat ? gfeat + ioff : nullptr;
    const T* op = gout + (static_cast<size_t>(tc.b) * Co + c0) * plane;
    const unsigned o_direct = static_cast<unsigned>(y * W + x) * static_cast<unsigned>(sizeof(T));
    const unsigned o_flip = static_cast<unsigned>(y * W + (W - 1 - x)) * static_cast<unsigned>(sizeof(T));
    const size_t flip_planes = static_cast<size_t>(C) * plane;
    const bool options_pair_loads = pair_loads;
    T gix = 0, giy = 0;

    auto one = [&](const T g, const T s0, const T s1, const T s2, const T s3) {
        // ATen grid_sampler_2d_backward; invalid corners read 0 and drop out.
        gix -= s0 * cn.dyw[0] * g;
        giy -= s0 * cn.dxw[0] * g;
        gix += s1 * cn.dyw[0] * g;
        giy -= s1 * cn.dxw[1] * g;
        gix -= s2 * cn.dyw[1] * g;
        giy += s2 * cn.dxw[0] * g;
        gix += s3 * cn.dyw[1] * g;
        giy += s3 * cn.dxw[1] * g;
    };
    int c = c0;
    if constexpr (sizeof(T) == 4) {
        if (!gp && gflow && options_pair_loads) {
            // d(flow) alone, fp32: the two corners of a row are neighbours in memory, so ONE 8-byte load per row replaces two dword
            // gathers (4 instead of 6 vector memory instructions per pixel and channel) wherever, for every lane of the wave, a row
            // is either inside the image with both corners or outside with both (a lane at the left / right border has half a
            // row: its wave takes the dword path below)
            const bool top_pair = cn.valid[0] && cn.valid[1], bot_pair = cn.valid[2] && cn.valid[3];
            const bool whole = (top_pair || !(cn.valid[0] || cn.valid[1])) && (bot_pair || !(cn.valid[2] || cn.valid[3]));
            if (__all(whole)) {
                const unsigned ot = top_pair ? cn.off[0] : kOob, ob = bot_pair ? cn.off[2] : kOob;
                constexpr int U = 4;
                for (; c + U <= c1; c += U, fp += U * iplane, op += U * plane) {
                    T g[U];
                    u32x2 st[U], sb[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        g[u] = buf_ld<T>(make_rsrc(op + u * plane, obytes), o_direct);
                        if (FLIP) g[u] += buf_ld<T>(make_rsrc(op + u * plane + flip_planes, obytes), o_flip);
                        const rsrc_t rf = make_rsrc(fp + u * iplane, ibytes);
                        st[u] = __builtin_amdgcn_raw_buffer_load_b64(rf, ot, 0, 0);
                        sb[u] = __builtin_amdgcn_raw_buffer_load_b64(rf, ob, 0, 0);
                    }
#pragma unroll
                    for (int u = 0; u < U; ++u)
                        one(g[u], __uint_as_float(st[u].x), __uint_as_float(st[u].y), __uint_as_float(sb[u].x), __uint_as_float(sb[u].y));
                }
            }
        }
    }
    if (!gp && gflow) {
        // d(flow) alone (the plane / tile kernels took d(feat)): four channels per trip, their 24 loads in flight together
        constexpr int U = 4;
        for (; c + U <= c1; c += U, fp += U * iplane, op += U * plane) {
            T g[U], s[U][4];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                g[u] = buf_ld<T>(make_rsrc(op + u * plane, obytes), o_direct);
                if (FLIP) g[u] += buf_ld<T>(make_rsrc(op + u * plane + flip_planes, obytes), o_flip);
                const rsrc_t rf = make_rsrc(fp + u * iplane, ibytes);
#pragma unroll
                for (int q = 0; q < 4; ++q) s[u][q] = buf_ld<T>(rf, cn.off[q]);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) one(g[u], s[u][0], s[u][1], s[u][2], s[u][3]);
        }
    }
    for (; c < c1; ++c, fp += iplane, op += plane) {
        T g = buf_ld<T>(make_rsrc(op, obytes), o_direct);
        if (FLIP) g += buf_ld<T>(make_rsrc(op + flip_planes, obytes), o_flip);
        if (gp) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (cn.valid[q]) atomic_add_off(gp, cn.off[q], cn.w[q] * g);
            gp += iplane;
        }
        if (gflow) {
            const rsrc_t rf = make_rsrc(fp, ibytes);
            const T s0 = buf_ld<T>(rf, cn.off[0]), s1 = buf_ld<T>(rf, cn.off[1]);
            const T s2 = buf_ld<T>(rf, cn.off[2]), s3 = buf_ld<T>(rf, cn.off[3]);
            one(g, s0, s1, s2, s3);
        }
    }
    if (gflow) {
        atomic_add(gflow + foff, (static_cast<T>(Wi) / 2) * gix);
        atomic_add(gflow + foff + plane, (static_cast<T>(Hi) / 2) * giy);
    }
}

template <typename T, bool FLIP>
__global__ void __launch_bounds__(kBlock)
warp_bwd_kernel(const T* __restrict__ feat, const T* __restrict__ flow, const T* __restrict__ gout,
                T* __restrict__ gfeat, T* __restrict__ gflow, int C, int Hi, int Wi, int H, int W,
                int tiles_x, int tiles_y, int cslabs, int cs, int remap) {
    warp_bwd_body<T, FLIP>(feat, flow, gout, gfeat, gflow, C, Hi, Wi, H, W, decode_tile(tiles_x, tiles_y, cslabs, remap), cs);
}

template <typename T, bool FLIP>
__global__ void __launch_bounds__(kBlock)
warp_fwd_multi_kernel(const WarpTable tab) {
    __shared__ __attribute__((aligned(16))) WlShared sh;
    int i = 0;
#pragma unroll
    for (int k = 1; k < kMaxWarpProblems; ++k)
        if (k < tab.n && blockIdx.x >= tab.p[k].begin) i = k;
    const WarpProblem& q = tab.p[i];
    // every XCD gets a contiguous range of the problem's tiles (neighbouring tiles share source rows: one L2 fetches them)
    const unsigned t = xcd_remap(blockIdx.x - q.begin, (q.nblk + 7u) & ~7u, 1);
    if (t >= q.nblk) return;
    if constexpr (sizeof(T) == 4) {
        if (q.lds) {           // (block-uniform: a problem is served by one kind of tile)
            unsigned u = t;
            const int tx = u % q.tiles_x;
            u /= q.tiles_x;
            const int ty = u % q.tiles_y;
            u /= q.tiles_y;
            warp_fwd_lds_body<FLIP>(static_cast<const float*>(q.feat), static_cast<const float*>(q.flow), static_cast<float*>(q.out), q.C,
                                    q.Hi, q.Wi, q.H, q.W, tx, ty, static_cast<int>(u % q.cslabs), static_cast<int>(u / q.cslabs), q.cs, tab.nt, sh);
            return;
        }
    }
    warp_fwd_body<T, FLIP>(static_cast<const T*>(q.feat), static_cast<const T*>(q.flow), static_cast<T*>(q.out), q.C, q.Hi, q.Wi,
                           q.H, q.W, decode_tile_local(t, q.tiles_x, q.tiles_y, q.cslabs), q.cs, tab.nt);
}

// d(flow) of several warps in one launch (the pixel-major kernel with grad_feat == NULL).  The LDS-staged body is NOT used here:
// measured on the step's three levels (profiles/r04_warp_multi_lds_sweep.txt) it loses -- 66 vs 25 us on [8,64,128,128]: with so few
// tiles a block keeps 4-8 channels, and the 2 atomics per pixel and slab of 16 slabs meet on the same addresses -- and its 33 KB of
// static LDS alone cost the direct body a third of its waves (45 vs 37 us).  It wins from ~2^24 pixel-channels up (single launches).
template <typename T, bool FLIP>
__global__ void __launch_bounds__(kBlock)
warp_bwd_flow_multi_kernel(const WarpTable tab) {
    int i = 0;
#pragma unroll
    for (int k = 1; k < kMaxWarpProblems; ++k)
        if (k < tab.n && blockIdx.x >= tab.p[k].begin) i = k;
    const WarpProblem& q = tab.p[i];
    const unsigned t = xcd_remap(blockIdx.x - q.begin, (q.nblk + 7u) & ~7u, 1);
    if (t >= q.nblk) return;
    warp_bwd_body<T, FLIP>(static_cast<const T*>(q.feat), static_cast<const T*>(q.flow), static_cast<const T*>(q.gout), nullptr,
                           static_cast<T*>(q.out), q.C, q.Hi, q.Wi, q.H, q.W, decode_tile_local(t, q.tiles_x, q.tiles_y, q.cslabs), q.cs, tab.nt != 0);
}

// d(feat) without contended global atomics: a block owns `cg` whole (b, c) planes of grad_feat in LDS.
// It visits the output pixels of image b (all of them, or one of `nsplit` interleaved row groups when
// there are too few planes to fill the chip), forms the four corners ONCE per pixel and adds the
// contributions of its cg channels with LDS atomics; the finished planes are added to grad_feat with
// plain coalesced read-modify-write rows (nsplit == 1) or one global atomic per non-zero cell.
//   * The LDS accumulator is DOUBLE: on gfx950 ds_add_f64 retires a wave in ~9 clk where ds_add_f32
//     needs ~190 (tools/ubench/atomics.hip); the sum is rounded to float once, at the flush.
//   * Robust to the degenerate flows of an untrained FlowNet (every pixel samples the same cell):
//     the collisions are resolved inside the CU instead of at the memory-side atomic unit.
// Used whenever one double plane fits LDS (Hi*Wi <= 20480 cells: every warp in FFWM, <= 128 x 128).
constexpr int kPlaneThreads = 1024;
constexpr int kPlaneLdsBytes = kMaxLdsBytes;
constexpr int kPlaneUnroll = 4;

template <typename T, bool FLIP, int CG>
__device__ __forceinline__ void warp_bwd_feat_plane_body(const T* __restrict__ flow, const T* __restrict__ gout, T* __restrict__ gfeat,
                                                         int C, int Hi, int Wi, int H, int W, int groups, int nsplit, unsigned t,
                                                         double* acc) {
    const int split = t % nsplit;
    t /= nsplit;
    const int grp = t % groups;
    const int b = t / groups;
    const int c0 = grp * CG;
    const int nc = (c0 + CG <= C) ? CG : C - c0;
    const int ncell = Hi * Wi, npix = H * W;
    for (int i = threadIdx.x; i < nc * ncell; i += kPlaneThreads) acc[i] = 0;
    __syncthreads();
    const int Co = FLIP ? 2 * C : C;
    const T* fl = flow + static_cast<size_t>(b) * 2 * npix;
    const T* g0 = gout + (static_cast<size_t>(b) * Co + c0) * npix;
    const T* g1 = g0 + static_cast<size_t>(C) * npix;
    // pixel p = (it * nsplit + split) * kPlaneThreads + tid: kPlaneUnroll pixels per trip so that all
    // their loads are in flight together
    const int stride = kPlaneThreads * nsplit;
    for (int p0 = split * kPlaneThreads + threadIdx.x; p0 < npix; p0 += stride * kPlaneUnroll) {
        T fx[kPlaneUnroll], fy[kPlaneUnroll], g[kPlaneUnroll][CG];
        int mir[kPlaneUnroll];
#pragma unroll
        for (int u = 0; u < kPlaneUnroll; ++u) {
            const int p = p0 + u * stride;
            const int pc = p < npix ? p : npix - 1;
            fx[u] = fl[pc];
            fy[u] = fl[npix + pc];
            const int y = pc / W, x = pc - y * W;
            mir[u] = y * W + (W - 1 - x);
#pragma unroll
            for (int c = 0; c < CG; ++c) {
                const int cc = c < nc ? c : 0;
                g[u][c] = g0[static_cast<size_t>(cc) * npix + pc];
                if (FLIP) g[u][c] += g1[static_cast<size_t>(cc) * npix + mir[u]];
            }
        }
#pragma unroll
        for (int u = 0; u < kPlaneUnroll; ++u) {
            if (p0 + u * stride >= npix) break;
            Corners<T> cn;
            make_corners<T>(cn, fx[u], fy[u], Hi, Wi);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (!cn.valid[q]) continue;
                double* cell = acc + cn.off[q] / static_cast<unsigned>(sizeof(T));
#pragma unroll
                for (int c = 0; c < CG; ++c)
                    if (c < nc)
                        __hip_atomic_fetch_add(cell + c * ncell, static_cast<double>(cn.w[q] * g[u][c]), __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
    }
    __syncthreads();
    T* dst = gfeat + (static_cast<size_t>(b) * C + c0) * ncell;
    if (nsplit == 1) {
        for (int i = threadIdx.x; i < nc * ncell; i += kPlaneThreads) dst[i] += static_cast<T>(acc[i]);
    } else {
        for (int i = threadIdx.x; i < nc * ncell; i += kPlaneThreads) {
            const T v = static_cast<T>(acc[i]);
            if (v != 0) atomic_add(dst + i, v);
        }
    }
}

template <typename T, bool FLIP, int CG>
__global__ void __launch_bounds__(kPlaneThreads)
warp_bwd_feat_plane_kernel(const T* __restrict__ flow, const T* __restrict__ gout, T* __restrict__ gfeat,
                           int C, int Hi, int Wi, int H, int W, int groups, int nsplit) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    warp_bwd_feat_plane_body<T, FLIP, CG>(flow, gout, gfeat, C, Hi, Wi, H, W, groups, nsplit, blockIdx.x,
                                          reinterpret_cast<double*>(smem_raw));
}

// d(feat) of several warps in ONE launch: the step's image warps (three illumination warps, the part crops) each need the plane
// kernel for a 3-channel tensor -- five to seven launches of ~10 us per step, one per problem.  Problems that share the channels
// per block (CG) share a launch; the dynamic LDS is the largest plane group's.
struct PlaneProblem {
    const void* flow;
    const void* gout;
    void* gfeat;
    int C, Hi, Wi, H, W, groups, nsplit;
    unsigned begin;
};
struct PlaneTable {
    int n;
    PlaneProblem p[kMaxWarpProblems];
};

template <typename T, bool FLIP, int CG>
__global__ void __launch_bounds__(kPlaneThreads)
warp_bwd_feat_plane_multi_kernel(const PlaneTable tab) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    int i = 0;
#pragma unroll
    for (int k = 1; k < kMaxWarpProblems; ++k)
        if (k < tab.n && blockIdx.x >= tab.p[k].begin) i = k;
    const PlaneProblem& q = tab.p[i];
    warp_bwd_feat_plane_body<T, FLIP, CG>(static_cast<const T*>(q.flow), static_cast<const T*>(q.gout), static_cast<T*>(q.gfeat), q.C, q.Hi,
                                          q.Wi, q.H, q.W, q.groups, q.nsplit, blockIdx.x - q.begin, reinterpret_cast<double*>(smem_raw));
}

// d(feat) for planes BEYOND LDS (Hi * Wi > 20480 cells) when the warp keeps the resolution (Hi == H, Wi == W) and the
// flow is a displaced identity grid -- what a flow net produces: OWNED TILES, no global atomics on the hot path.
//   * A 512-thread block owns a 56 x 56 tile of grad_feat for CG channels at a time (LDS accumulators, DOUBLE: ds_add_f64
//     ~9 clk per wave) and visits the 64 x 64 output pixels of the tile grown by a halo of 4 (a wave per pixel row, eight
//     rows per wave); the four corners of a pixel are formed ONCE and kept in registers for all channels.  A corner is
//     added iff its cell lies in the block's own tile -- the neighbouring blocks visit the same halo pixels and keep what
//     is theirs (x1.31 pixel visits, the re-read grad_output rows come from L2).  The finished tile goes to grad_feat with
//     plain coalesced read-modify-write rows.
//   * warp_bwd_feat_far_kernel is the exact complement: a (pixel, corner) pair whose pixel lies OUTSIDE the 64 x 64 region
//     of the tile that owns the corner's cell (displacement beyond the halo) is scattered with global atomics over all
//     channels.  For a flow net's field it only reads the flow.
// Together they replace 4 contended global atomics per pixel and channel (3.2 ms at [32,64,256,256]: 9 % of the HBM peak).
constexpr int kWtThreads = 512;
constexpr int kWtWaves = kWtThreads / kWave;
constexpr int kWtHalo = 4;
constexpr int kWtRegion = 64;                       // visited pixels per side
constexpr int kWtTile = kWtRegion - 2 * kWtHalo;    // owned cells per side: 56
constexpr int kWtRows = kWtRegion / kWtWaves;       // pixel rows per wave: 8

// OVW (round 5): the finished tile is STORED instead of added -- every in-image cell of grad_feat has exactly one owner, so a caller
// that hands over an uninitialised buffer (flipcat bit 1 of ffwm_warp_backward) saves its zero-fill and this kernel the read of it.
// FIX (round 6): 32-bit fixed-point cells instead of doubles, the contribution formed by ONE fused multiply-add,
// fma(w, g 2^s, 1.5 2^23): its bit pattern is 0x4B400000 + k (k = the contribution rounded to nearest, |k| < 2^22) and ds_add_u32 of
// the patterns leaves n 0x4B400000 + sum k in a cell that took n corners.  n per cell comes from a count pass over the block's corners
// (once: the flow is the same for every channel) and stays in a plane of its own; the scale of a channel group is exact -- the group's
// gradients sit in registers before the first add, max|g| < 2^e is one block reduction -- and bits = min(22, 31 - bitlength(max n))
// keeps n 2^bits below 2^31 for any flow.  ds_add_u32 retires in half the LDS time of ds_add_f64 (4.3 vs 8.6 clk per wave), the
// v_cvt_f64_f32 per contribution is gone, the box is half the size.  A group with a NaN / Inf gradient adds its own-tile corners with
// global atomics behind a flush of zeros.
template <bool FLIP, int CG, bool OVW = false, bool FIX = false>
__global__ void __launch_bounds__(kWtThreads, 4)          // two 8-wave blocks per CU: 128 registers
warp_bwd_feat_tile_kernel(const float* __restrict__ flow, const float* __restrict__ gout, float* __restrict__ gfeat, int C, int H,
                          int W, int ntx, int nty, int groups_per_slab, int cslabs) {
    using AccT = typename std::conditional<FIX, unsigned, double>::type;
    constexpr int NC = kWtTile * kWtTile;
    constexpr unsigned kMagicBits = 0x4B400000u;
    __shared__ AccT acc[(CG + (FIX ? 1 : 0)) * NC];           // FIX: plane CG = corners per cell
    __shared__ unsigned redm[(CG > 1 ? CG : 1) * kWtWaves];
    unsigned t = xcd_remap(blockIdx.x, gridDim.x, 1);
    const int tx = t % ntx;
    t /= ntx;
    const int ty = t % nty;
    t /= nty;
    const int slab = t % cslabs;
    const int b = t / cslabs;
    const int X0 = tx * kWtTile, Y0 = ty * kWtTile;
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    const int x = X0 - kWtHalo + lane;
    const size_t plane = static_cast<size_t>(H) * W;
    const float* fl = flow + static_cast<size_t>(b) * 2 * plane;

    // corners of this lane's eight pixels: cell index inside the own tile (or -1) and weight per corner
    int cell[kWtRows][4];
    float wgt[kWtRows][4];
    unsigned poff[kWtRows], moff[kWtRows];           // byte offsets of the pixel and of its mirror in a grad_output plane (OOB: none)
#pragma unroll
    for (int r = 0; r < kWtRows; ++r) {
        const int y = Y0 - kWtHalo + wave + r * kWtWaves;
        const bool pin = x >= 0 && x < W && y >= 0 && y < H;
        poff[r] = pin ? static_cast<unsigned>(y * W + x) * 4u : kOob;
        moff[r] = pin ? static_cast<unsigned>(y * W + (W - 1 - x)) * 4u : kOob;
        Corners<float> cn;
        const size_t fo = static_cast<size_t>(pin ? y : 0) * W + (pin ? x : 0);
        make_corners<float>(cn, fl[fo], fl[plane + fo], H, W);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int ci = static_cast<int>(cn.off[q] / 4u);          // cy * W + cx when valid
            const int cy = ci / W, cx = ci - cy * W;
            const bool mine = pin && cn.valid[q] && cx >= X0 && cx < X0 + kWtTile && cy >= Y0 && cy < Y0 + kWtTile;
            cell[r][q] = mine ? ((cy - Y0) * kWtTile + (cx - X0)) * static_cast<int>(sizeof(AccT)) : -1;      // BYTE offset in a plane: the only form kept (an index AND its address cost 32 more registers)
            wgt[r][q] = cn.w[q];
        }
    }
    auto at = [](AccT* plane0, int byte_off) { return reinterpret_cast<AccT*>(reinterpret_cast<char*>(plane0) + byte_off); };
    for (int i = threadIdx.x; i < (CG + (FIX ? 1 : 0)) * NC; i += kWtThreads) acc[i] = 0;
    __syncthreads();
    int bits = 22;
    if constexpr (FIX) {
        AccT* cnt = acc + CG * NC;
#pragma unroll
        for (int r = 0; r < kWtRows; ++r)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (cell[r][q] >= 0) __hip_atomic_fetch_add(at(cnt, cell[r][q]), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __syncthreads();
        unsigned pop = 0;
        for (int i = threadIdx.x; i < NC; i += kWtThreads) pop = max(pop, static_cast<unsigned>(cnt[i]));
        pop = wave_max(pop);
        if (lane == 0) redm[wave] = pop;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kWtWaves; ++k) pop = max(pop, redm[k]);
        bits = __builtin_amdgcn_readfirstlane(min(22, 31 - (32 - __clz(static_cast<int>(pop)))));
        __syncthreads();                             // (redm is rewritten by the first group)
    }

    const int Co = FLIP ? 2 * C : C;
    const unsigned obytes = static_cast<unsigned>(plane * 4u);
    int ex_assumed[CG], next_ex[CG];
#pragma unroll
    for (int c = 0; c < CG; ++c) ex_assumed[c] = next_ex[c] = 0;
    for (int gi = 0; gi < groups_per_slab; ++gi) {
        const int c0 = (slab * groups_per_slab + gi) * CG;
        if (c0 >= C) break;
        const float* g0 = gout + (static_cast<size_t>(b) * Co + c0) * plane;
        float g[kWtRows][CG];
#pragma unroll
        for (int c = 0; c < CG; ++c) {
            const unsigned nb = c0 + c < C ? obytes : 0u;            // a channel past C reads zeros
            const rsrc_t rd = make_rsrc(g0 + static_cast<size_t>(c) * plane, nb);
            const rsrc_t rm = make_rsrc(g0 + (static_cast<size_t>(C) + c) * plane, FLIP ? nb : 0u);
#pragma unroll
            for (int r = 0; r < kWtRows; ++r) {
                g[r][c] = buf_ld<float>(rd, poff[r]);
                if (FLIP) g[r][c] += buf_ld<float>(rm, moff[r]);
            }
        }
        float fx_inv[CG];
#pragma unroll
        for (int c = 0; c < CG; ++c) fx_inv[c] = 0.f;
        bool exact_path = false;
        if constexpr (FIX) {
            // The scale: 2^(bits - e) with max|g| < 2^e over the block's pixels.  Reducing that maximum over the block BEFORE the first add
            // costs a barrier behind the group's loads -- every add of a two-channel group then waits for the slowest load of the block
            // (measured: 492 -> 805 us).  So a group ASSUMES the exponent of the previous group's maximum + 1, adds while its loads arrive
            // (as the double cells did), and the block's true maximum -- its gradients are in registers -- is checked behind the barrier
            // that ends the adds anyway: above the assumed range (a contribution left the magic-number binade) or more than 3 bits below
            // it (precision), the planes are cleared and the group is added again with its own exponent.  The first group of a block
            // has no predecessor: it pays the reduction.
            // (per CHANNEL: a channel's precision is 2^-bits of ITS largest gradient, whatever its neighbour in the group holds)
            // (the maxima are folded BEHIND the adds: folding them first would make the first add wait for the last load of the group)
            auto lane_max = [&](unsigned (&m)[CG]) {
#pragma unroll
                for (int c = 0; c < CG; ++c) {
                    m[c] = 0;
#pragma unroll
                    for (int r = 0; r < kWtRows; ++r) m[c] = max(m[c], __float_as_uint(g[r][c]) & 0x7FFFFFFFu);
                    m[c] = wave_max(m[c]);
                }
            };
            auto block_max = [&](unsigned (&m)[CG]) {          // redm[c][wave] -> every thread holds the block's maxima
                if (lane == 0) {
#pragma unroll
                    for (int c = 0; c < CG; ++c) redm[c * kWtWaves + wave] = m[c];
                }
                __syncthreads();
#pragma unroll
                for (int c = 0; c < CG; ++c)
#pragma unroll
                    for (int k = 0; k < kWtWaves; ++k) m[c] = max(m[c], redm[c * kWtWaves + k]);
            };
            if (gi == 0) {
                unsigned m0[CG];
                lane_max(m0);
                block_max(m0);
#pragma unroll
                for (int c = 0; c < CG; ++c) {
                    ex_assumed[c] = 0;
                    if (m0[c] != 0u && m0[c] < 0x7F800000u) (void)frexpf(__uint_as_float(m0[c]), &ex_assumed[c]);
                }
                __syncthreads();
            }
            for (int attempt = 0; attempt < 2; ++attempt) {
                float sc[CG];
#pragma unroll
                for (int c = 0; c < CG; ++c) {
                    const int ex = __builtin_amdgcn_readfirstlane(min(max(ex_assumed[c], -80), 120));
                    ex_assumed[c] = ex;
                    sc[c] = ldexpf(1.f, bits - ex);
                    fx_inv[c] = ldexpf(1.f, ex - bits);
                }
#pragma unroll
                for (int r = 0; r < kWtRows; ++r) {
                    float gs[CG];
#pragma unroll
                    for (int c = 0; c < CG; ++c) gs[c] = g[r][c] * sc[c];        // (exact: a power of two)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (cell[r][q] >= 0) {
#pragma unroll
                            for (int c = 0; c < CG; ++c)
                                __hip_atomic_fetch_add(at(acc + c * NC, cell[r][q]), __float_as_uint(__builtin_fmaf(wgt[r][q], gs[c], 12582912.f)),
                                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        }
                }
                unsigned mb[CG];
                lane_max(mb);
                block_max(mb);                        // (its barrier: the adds of the attempt are in)
                bool fits = true;
                exact_path = false;
#pragma unroll
                for (int c = 0; c < CG; ++c) {
                    const bool bad = mb[c] >= 0x7F800000u;       // a NaN / Inf gradient among the channel's pixels
                    exact_path = exact_path || bad;
                    int ex_true = ex_assumed[c];
                    if (mb[c] != 0u && !bad) (void)frexpf(__uint_as_float(mb[c]), &ex_true);
                    ex_true = __builtin_amdgcn_readfirstlane(ex_true);
                    fits = fits && (mb[c] == 0u || bad || (ex_true <= ex_assumed[c] && ex_true >= ex_assumed[c] - 3));
                    next_ex[c] = (mb[c] != 0u && !bad) ? ex_true : ex_assumed[c];
                }
                if (exact_path) {
#pragma unroll
                    for (int c = 0; c < CG; ++c) fx_inv[c] = 0.f;
                }
                if (fits || exact_path || attempt == 1) break;
#pragma unroll
                for (int c = 0; c < CG; ++c) ex_assumed[c] = next_ex[c];
                for (int i = threadIdx.x; i < CG * NC; i += kWtThreads) acc[i] = 0;
                __syncthreads();
            }
#pragma unroll
            for (int c = 0; c < CG; ++c) ex_assumed[c] = next_ex[c] + 1;        // the next group's assumption: this group's exponents + 1
        } else {
#pragma unroll
            for (int r = 0; r < kWtRows; ++r)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (cell[r][q] >= 0) {
#pragma unroll
                        for (int c = 0; c < CG; ++c)
                            __hip_atomic_fetch_add(at(acc + c * NC, cell[r][q]), static_cast<AccT>(wgt[r][q] * g[r][c]),
                                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
            __syncthreads();
        }
        // flush the own tile: coalesced rows of 56 cells, plain read-modify-write (every cell has exactly one owner)
        for (int i = threadIdx.x; i < CG * NC; i += kWtThreads) {
            const int c = i / NC, rem = i - c * NC;
            const int cy = Y0 + rem / kWtTile, cx = X0 + rem % kWtTile;
            float v;
            if constexpr (FIX) {
                float inv = fx_inv[0];
#pragma unroll
                for (int cc = 1; cc < CG; ++cc) inv = c == cc ? fx_inv[cc] : inv;
                v = inv != 0.f ? static_cast<float>(static_cast<int>(acc[i] - static_cast<unsigned>(acc[CG * NC + rem]) * kMagicBits)) * inv : 0.f;
            }
            else v = static_cast<float>(acc[i]);
            acc[i] = 0;
            if (c0 + c < C && cx < W && cy < H) {
                float* d = gfeat + (static_cast<size_t>(b) * C + c0 + c) * plane + static_cast<size_t>(cy) * W + cx;
                if (OVW) *d = v;
                else *d += v;
            }
        }
        if constexpr (FIX) {
            if (exact_path) {
                // own-tile corners of the group by global atomics, behind this block's own stores (no other block writes these cells).
                // Re-derived from the flow in a ROLLED loop: unrolled over the register arrays of the hot path it costs 67 more registers.
                __builtin_amdgcn_s_waitcnt(0);
                __syncthreads();
#pragma unroll 1
                for (int r = 0; r < kWtRows; ++r) {
                    const int y = Y0 - kWtHalo + wave + r * kWtWaves;
                    if (!(x >= 0 && x < W && y >= 0 && y < H)) continue;
                    Corners<float> cn;
                    const size_t fo = static_cast<size_t>(y) * W + x;
                    make_corners<float>(cn, fl[fo], fl[plane + fo], H, W);
                    float gq[CG];
#pragma unroll
                    for (int c = 0; c < CG; ++c) {
                        gq[c] = c0 + c < C ? g0[static_cast<size_t>(c) * plane + fo] : 0.f;
                        if (FLIP && c0 + c < C) gq[c] += g0[(static_cast<size_t>(C) + c) * plane + static_cast<size_t>(y) * W + (W - 1 - x)];
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int ci = static_cast<int>(cn.off[q] / 4u);
                        const int cy = ci / W, cx = ci - cy * W;
                        if (!(cn.valid[q] && cx >= X0 && cx < X0 + kWtTile && cy >= Y0 && cy < Y0 + kWtTile)) continue;
#pragma unroll
                        for (int c = 0; c < CG; ++c)
                            if (c0 + c < C)
                                atomic_add(gfeat + (static_cast<size_t>(b) * C + c0 + c) * plane + static_cast<size_t>(cy) * W + cx, cn.w[q] * gq[c]);
                    }
                }
            }
        }
        __syncthreads();
    }
}

// Round 6, second form of the fixed-point owned-tile kernel: the per-pixel state is 3 registers instead of 10.  The first form kept, for
// each of a lane's eight pixels, four cell indices, four weights and two load offsets (80 registers); with the fixed-point machinery on
// top hipcc spilled into the add loop (495 -> 900 us).  Here a pixel keeps ONE packed word -- the tile-relative position of its
// north-west corner, which of the four corners land in the own tile, whether the pixel exists -- and its two fractions (tx, ty); the four
// cells are one LDS address + the immediates {0, 4, 224, 228}, the weights four products formed where they are used, the load offsets
// come from the row index.  24 registers of state leave room for FOUR channels per group (half the barriers and flush passes per byte).
// Cells / scale / count plane / retry: as in warp_bwd_feat_tile_kernel<.., FIX>.
template <bool FLIP, int CG, bool OVW>
__global__ void __launch_bounds__(kWtThreads, 4)
warp_bwd_feat_tile2_kernel(const float* __restrict__ flow, const float* __restrict__ gout, float* __restrict__ gfeat, int C, int H,
                           int W, int ntx, int nty, int groups_per_slab, int cslabs) {
    constexpr int NC = kWtTile * kWtTile;
    constexpr int PITCH = kWtTile;
    constexpr unsigned kMagicBits = 0x4B400000u;
    // box rows / columns carry a one-cell apron on the north and west (the NW corner of a pixel may sit at -1 while its SE corner is the
    // tile's cell 0): the address of the NW corner is formed unconditionally, only `mine` corners are touched
    __shared__ unsigned acc[(CG + 1) * NC + PITCH + 8];
    __shared__ unsigned redm[CG * kWtWaves];
    unsigned* const box = acc + PITCH + 4;                   // cell (0, 0) of plane 0; (-1, -1) is still inside the array
    unsigned t = xcd_remap(blockIdx.x, gridDim.x, 1);
    const int tx = t % ntx;
    t /= ntx;
    const int ty = t % nty;
    t /= nty;
    const int slab = t % cslabs;
    const int b = t / cslabs;
    const int X0 = tx * kWtTile, Y0 = ty * kWtTile;
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    const int x = X0 - kWtHalo + lane;
    const size_t plane = static_cast<size_t>(H) * W;
    const float* fl = flow + static_cast<size_t>(b) * 2 * plane;
    const bool xin = x >= 0 && x < W;

    // per pixel: packed = (cy0 + 1) << 8 | (cx0 + 1) | mine bits << 16 | exists << 20, with (cx0, cy0) the NW corner relative to the tile
    int packed[kWtRows];
    float fxs[kWtRows], fys[kWtRows];
#pragma unroll
    for (int r = 0; r < kWtRows; ++r) {
        const int y = Y0 - kWtHalo + wave + r * kWtWaves;
        const bool pin = xin && y >= 0 && y < H;
        Corners<float> cn;
        const size_t fo = static_cast<size_t>(pin ? y : 0) * W + (pin ? x : 0);
        make_corners<float>(cn, fl[fo], fl[plane + fo], H, W);
        int mine = 0, cx0 = 0, cy0 = 0;
        bool have = false;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int ci = static_cast<int>(cn.off[q] / 4u);          // cy * W + cx when valid
            const int cy = ci / W, cx = ci - cy * W;
            const bool m = pin && cn.valid[q] && cx >= X0 && cx < X0 + kWtTile && cy >= Y0 && cy < Y0 + kWtTile;
            if (m) {
                mine |= 1 << q;
                if (!have) { cx0 = cx - X0 - (q & 1); cy0 = cy - Y0 - (q >> 1); have = true; }       // -> the NW corner's position
            }
        }
        packed[r] = ((cy0 + 1) << 8) | (cx0 + 1) | (mine << 16) | (pin ? 1 << 20 : 0);
        fxs[r] = cn.dxw[1];                                  // ix - x0: the east corners' factor; the west ones take 1 - it
        fys[r] = cn.dyw[1];
    }
    for (int i = threadIdx.x; i < (CG + 1) * NC + PITCH + 8; i += kWtThreads) acc[i] = 0;
    __syncthreads();
    unsigned* const cnt = box + CG * NC;
    constexpr int kOffs[4] = {0, 1, PITCH, PITCH + 1};
#pragma unroll
    for (int r = 0; r < kWtRows; ++r) {
        const int pk = packed[r];
        unsigned* nw = cnt + (((pk >> 8) & 0xff) - 1) * PITCH + ((pk & 0xff) - 1);
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if ((pk >> (16 + q)) & 1) __hip_atomic_fetch_add(nw + kOffs[q], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __syncthreads();
    unsigned pop = 0;
    for (int i = threadIdx.x; i < NC; i += kWtThreads) pop = max(pop, cnt[i]);
    pop = wave_max(pop);
    if (lane == 0) redm[wave] = pop;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kWtWaves; ++k) pop = max(pop, redm[k]);
    const int bits = __builtin_amdgcn_readfirstlane(min(22, 31 - (32 - __clz(static_cast<int>(pop)))));
    __syncthreads();

    const int Co = FLIP ? 2 * C : C;
    const unsigned obytes = static_cast<unsigned>(plane * 4u);
    const int ybase = Y0 - kWtHalo + wave;
    int ex_assumed[CG], next_ex[CG];
#pragma unroll
    for (int c = 0; c < CG; ++c) ex_assumed[c] = next_ex[c] = 0;
    for (int gi = 0; gi < groups_per_slab; ++gi) {
        const int c0 = (slab * groups_per_slab + gi) * CG;
        if (c0 >= C) break;
        const float* g0 = gout + (static_cast<size_t>(b) * Co + c0) * plane;
        // (requesting a group's gradients one group ahead was tried: 32 more registers of loads in flight, 144-388 bytes of scratch, 498 -> 955 us)
        float g[kWtRows][CG];
#pragma unroll
        for (int c = 0; c < CG; ++c) {
            const unsigned nb = c0 + c < C ? obytes : 0u;            // a channel past C reads zeros
            const rsrc_t rd = make_rsrc(g0 + static_cast<size_t>(c) * plane, nb);
            const rsrc_t rm = make_rsrc(g0 + (static_cast<size_t>(C) + c) * plane, FLIP ? nb : 0u);
#pragma unroll
            for (int r = 0; r < kWtRows; ++r) {
                const bool pin = (packed[r] >> 20) & 1;
                const int y = ybase + r * kWtWaves;
                const unsigned po = pin ? static_cast<unsigned>(y * W + x) * 4u : kOob;
                const unsigned mo = pin ? static_cast<unsigned>(y * W + (W - 1 - x)) * 4u : kOob;
                g[r][c] = buf_ld<float>(rd, po);
                if (FLIP) g[r][c] += buf_ld<float>(rm, mo);
            }
        }
        float fx_inv[CG];
        bool exact_path = false;
        auto lane_max = [&](unsigned (&m)[CG]) {
#pragma unroll
            for (int c = 0; c < CG; ++c) {
                m[c] = 0;
#pragma unroll
                for (int r = 0; r < kWtRows; ++r) m[c] = max(m[c], __float_as_uint(g[r][c]) & 0x7FFFFFFFu);
                m[c] = wave_max(m[c]);
            }
        };
        auto block_max = [&](unsigned (&m)[CG]) {
            if (lane == 0) {
#pragma unroll
                for (int c = 0; c < CG; ++c) redm[c * kWtWaves + wave] = m[c];
            }
            __syncthreads();
#pragma unroll
            for (int c = 0; c < CG; ++c)
#pragma unroll
                for (int k = 0; k < kWtWaves; ++k) m[c] = max(m[c], redm[c * kWtWaves + k]);
        };
        if (gi == 0) {
            unsigned m0[CG];
            lane_max(m0);
            block_max(m0);
#pragma unroll
            for (int c = 0; c < CG; ++c) {
                ex_assumed[c] = 0;
                if (m0[c] != 0u && m0[c] < 0x7F800000u) (void)frexpf(__uint_as_float(m0[c]), &ex_assumed[c]);
            }
            __syncthreads();
        }
        for (int attempt = 0; attempt < 2; ++attempt) {
            float sc[CG];
#pragma unroll
            for (int c = 0; c < CG; ++c) {
                const int ex = __builtin_amdgcn_readfirstlane(min(max(ex_assumed[c], -80), 120));
                ex_assumed[c] = ex;
                sc[c] = ldexpf(1.f, bits - ex);
                fx_inv[c] = ldexpf(1.f, ex - bits);
            }
#pragma unroll
            for (int r = 0; r < kWtRows; ++r) {
                int pk = packed[r];
                asm volatile("" : "+v"(pk));           // (the address and the four weights are channel-group invariant: not to be hoisted)
                if (!((pk >> 16) & 15)) continue;
                unsigned* nw = box + (((pk >> 8) & 0xff) - 1) * PITCH + ((pk & 0xff) - 1);
                const float ex_ = fxs[r], ey_ = fys[r], wx_ = 1.f - ex_, wy_ = 1.f - ey_;
                const float wq[4] = {wx_ * wy_, ex_ * wy_, wx_ * ey_, ex_ * ey_};
                float gs[CG];
#pragma unroll
                for (int c = 0; c < CG; ++c) gs[c] = g[r][c] * sc[c];        // (exact: a power of two)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if ((pk >> (16 + q)) & 1) {
#pragma unroll
                        for (int c = 0; c < CG; ++c)
                            __hip_atomic_fetch_add(nw + c * NC + kOffs[q], __float_as_uint(__builtin_fmaf(wq[q], gs[c], 12582912.f)),
                                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
            }
            unsigned mb[CG];
            lane_max(mb);
            block_max(mb);                        // (its barrier: the adds of the attempt are in)
            bool fits = true;
            exact_path = false;
#pragma unroll
            for (int c = 0; c < CG; ++c) {
                const bool bad = mb[c] >= 0x7F800000u;       // a NaN / Inf gradient among the channel's pixels
                exact_path = exact_path || bad;
                int ex_true = ex_assumed[c];
                if (mb[c] != 0u && !bad) (void)frexpf(__uint_as_float(mb[c]), &ex_true);
                ex_true = __builtin_amdgcn_readfirstlane(ex_true);
                fits = fits && (mb[c] == 0u || bad || (ex_true <= ex_assumed[c] && ex_true >= ex_assumed[c] - 3));
                next_ex[c] = (mb[c] != 0u && !bad) ? ex_true : ex_assumed[c];
            }
            if (exact_path) {
#pragma unroll
                for (int c = 0; c < CG; ++c) fx_inv[c] = 0.f;
            }
            if (fits || exact_path || attempt == 1) break;
#pragma unroll
            for (int c = 0; c < CG; ++c) ex_assumed[c] = next_ex[c];
            for (int i = threadIdx.x; i < CG * NC; i += kWtThreads) box[i] = 0;
            __syncthreads();
        }
#pragma unroll
        for (int c = 0; c < CG; ++c) ex_assumed[c] = next_ex[c] + 1;        // the next group's assumption: this group's exponents + 1
        // flush the own tile: coalesced rows of 56 cells (every cell has exactly one owner)
        for (int i = threadIdx.x; i < CG * NC; i += kWtThreads) {
            const int c = i / NC, rem = i - c * NC;
            const int cy = Y0 + rem / kWtTile, cx = X0 + rem % kWtTile;
            float inv = fx_inv[0];
#pragma unroll
            for (int cc = 1; cc < CG; ++cc) inv = c == cc ? fx_inv[cc] : inv;
            const float v = inv != 0.f ? static_cast<float>(static_cast<int>(box[i] - cnt[rem] * kMagicBits)) * inv : 0.f;
            box[i] = 0;
            if (c0 + c < C && cx < W && cy < H) {
                float* d = gfeat + (static_cast<size_t>(b) * C + c0 + c) * plane + static_cast<size_t>(cy) * W + cx;
                if (OVW) *d = v;
                else *d += v;
            }
        }
        if (exact_path) {
            __builtin_amdgcn_s_waitcnt(0);
            __syncthreads();
#pragma unroll 1
            for (int r = 0; r < kWtRows; ++r) {
                const int y = ybase + r * kWtWaves;
                if (!(xin && y >= 0 && y < H)) continue;
                Corners<float> cn;
                const size_t fo = static_cast<size_t>(y) * W + x;
                make_corners<float>(cn, fl[fo], fl[plane + fo], H, W);
                float gq[CG];
#pragma unroll
                for (int c = 0; c < CG; ++c) {
                    gq[c] = c0 + c < C ? g0[static_cast<size_t>(c) * plane + fo] : 0.f;
                    if (FLIP && c0 + c < C) gq[c] += g0[(static_cast<size_t>(C) + c) * plane + static_cast<size_t>(y) * W + (W - 1 - x)];
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int ci = static_cast<int>(cn.off[q] / 4u);
                    const int cy = ci / W, cx = ci - cy * W;
                    if (!(cn.valid[q] && cx >= X0 && cx < X0 + kWtTile && cy >= Y0 && cy < Y0 + kWtTile)) continue;
#pragma unroll
                    for (int c = 0; c < CG; ++c)
                        if (c0 + c < C)
                            atomic_add(gfeat + (static_cast<size_t>(b) * C + c0 + c) * plane + static_cast<size_t>(cy) * W + cx, cn.w[q] * gq[c]);
                }
            }
        }
        __syncthreads();
    }
}

template <bool FLIP>
__global__ void __launch_bounds__(kBlock)
warp_bwd_feat_far_kernel(const float* __restrict__ flow, const float* __restrict__ gout, float* __restrict__ gfeat, int C, int H, int W,
                         int tiles_x, int tiles_y) {
    unsigned t = blockIdx.x;
    const int txb = t % tiles_x;
    t /= tiles_x;
    const int tyb = t % tiles_y;
    const int b = t / tiles_y;
    const int x = txb * kTileX + (threadIdx.x & (kTileX - 1)), y = tyb * kTileY + threadIdx.x / kTileX;
    if (x >= W || y >= H) return;
    const size_t plane = static_cast<size_t>(H) * W;
    const size_t fo = static_cast<size_t>(y) * W + x;
    Corners<float> cn;
    make_corners<float>(cn, flow[static_cast<size_t>(b) * 2 * plane + fo], flow[static_cast<size_t>(b) * 2 * plane + plane + fo], H, W);
    bool far[4];
    bool any = false;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int ci = static_cast<int>(cn.off[q] / 4u);
        const int cy = ci / W, cx = ci - cy * W;
        const int rx0 = (cx / kWtTile) * kWtTile - kWtHalo, ry0 = (cy / kWtTile) * kWtTile - kWtHalo;      // region of the cell's owner
        far[q] = cn.valid[q] && !(x >= rx0 && x < rx0 + kWtRegion && y >= ry0 && y < ry0 + kWtRegion);
        any = any || far[q];
    }
    if (!any) return;
    const int Co = FLIP ? 2 * C : C;
    const float* g0 = gout + static_cast<size_t>(b) * Co * plane;
    float* gp = gfeat + static_cast<size_t>(b) * C * plane;
    for (int c = 0; c < C; ++c) {
        float g = g0[static_cast<size_t>(c) * plane + fo];
        if (FLIP) g += g0[(static_cast<size_t>(C) + c) * plane + static_cast<size_t>(y) * W + (W - 1 - x)];
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (far[q]) atomic_add_off(gp + static_cast<size_t>(c) * plane, cn.off[q], cn.w[q] * g);
    }
}

// Geometry of a plane launch: channels per block (cg in {1,2,4,8}) and pixel split.
struct PlanePlan {
    int cg, groups, nsplit;
    size_t lds;
    bool ok;
};
inline PlanePlan plan_planes(int64_t B, int64_t C, int64_t cells, int64_t npix, int max_cg) {
    PlanePlan p{1, 0, 1, 0, false};
    const size_t plane = static_cast<size_t>(cells) * sizeof(double);
    if (plane > static_cast<size_t>(kPlaneLdsBytes)) return p;
    p.ok = true;
    int cg = max_cg;
    // as many channels per block as LDS holds twice over (2 blocks per CU), but keep >= 512 blocks
    while (cg > 1 && (cg * plane * 2 > static_cast<size_t>(kPlaneLdsBytes) || cg > C || B * ((C + cg - 1) / cg) < 512)) cg >>= 1;
    p.cg = cg;
    p.groups = static_cast<int>((C + cg - 1) / cg);
    p.lds = cg * plane;
    int ns = 1;
    const int64_t trips = (npix + kPlaneThreads * kPlaneUnroll - 1) / (kPlaneThreads * kPlaneUnroll);
    while (B * p.groups * ns < 256 && ns * 2 <= trips) ns *= 2;   // too few planes: split the pixels
    p.nsplit = ns;
    return p;
}

int check_dims(const char* fn, int64_t B, int64_t C, int64_t Hi, int64_t Wi, int64_t H, int64_t W,
               int dtype) {
    FFWM_REQUIRE(dtype_ok(dtype), FFWM_ERR_DTYPE, "%s: dtype %d is not FFWM_F32/FFWM_F64", fn, dtype);
    FFWM_REQUIRE(B > 0 && C > 0 && Hi > 0 && Wi > 0 && H > 0 && W > 0, FFWM_ERR_ARG,
                 "%s: sizes must be positive (B=%lld C=%lld Hi=%lld Wi=%lld H=%lld W=%lld)", fn, (long long)B,
                 (long long)C, (long long)Hi, (long long)Wi, (long long)H, (long long)W);
    FFWM_REQUIRE(Hi * Wi < (1LL << 28) && H * W < (1LL << 28), FFWM_ERR_SIZE,
                 "%s: a single H*W plane must stay below 2^28 elements (32-bit byte offsets)", fn);
    const int64_t spatial = B * ((W + kTileX - 1) / kTileX) * ((H + kTileY - 1) / kTileY);
    FFWM_REQUIRE(spatial * C < (1LL << 31), FFWM_ERR_SIZE, "%s: grid too large", fn);
    return FFWM_OK;
}

template <typename T>
int launch_fwd(const T* feat, const T* flow, T* out, int64_t B, int64_t C, int64_t Hi, int64_t Wi,
               int64_t H, int64_t W, int flip, hipStream_t st) {
    const double bytes = sizeof(T) * static_cast<double>(B) *
                         (static_cast<double>(C) * Hi * Wi + 2.0 * H * W + (flip ? 2.0 : 1.0) * C * H * W);
    const Geometry g = plan(B, C, H, W, 16);
    const int remap = options().xcd_remap;
    LaunchScope ls(scope_at(flip ? "warp_flipcat_fwd" : "warp_fwd", H), st, bytes);
    // warp_fwd_variant: 0 = auto (LDS-staged tiles for large float outputs), 1 = direct gathers, 2 = LDS-staged
    if constexpr (sizeof(T) == 4) {
        const int variant = options().warp_fwd_variant;
        // (measured: the tile kernel wins once the output no longer sits in L2 -- 4.6 vs 3.4 TB/s on [32,64,256,256];
        //  on the <= 34 MB tensors of netG the direct kernel is as fast or faster)
        if (variant == 2 || (variant == 0 && H >= 64 && W >= 64 && B * C * H * W >= (1LL << 24))) {
            const int txs = static_cast<int>((W + kWlTileX - 1) / kWlTileX), tys = static_cast<int>((H + kWlTileY - 1) / kWlTileY);
            int cs = options().channel_slab > 0 ? options().channel_slab : 32;      // measured at [32,64,256,256]: 8 / 16 / 32 / 64 -> 4.4 / 4.8 / 5.0 / 4.9 TB/s
            if (cs > C) cs = static_cast<int>(C);
            while (cs > 2 && B * txs * tys * ((C + cs - 1) / cs) < 2048) cs = (cs + 1) / 2;
            const int cslabs = static_cast<int>((C + cs - 1) / cs);
            const unsigned grid = static_cast<unsigned>(B * txs * tys * cslabs);
            const int nt = sizeof(T) * static_cast<double>(B) * C * H * W * (flip ? 2 : 1) >= 64.0 * 1024 * 1024 ? 1 : 0;   // streaming stores
            if (flip)
                hipLaunchKernelGGL((warp_fwd_lds_kernel<true>), dim3(grid), dim3(kBlock), 0, st, (const float*)feat,
                                   (const float*)flow, (float*)out, (int)C, (int)Hi, (int)Wi, (int)H, (int)W, txs, tys, cslabs,
                                   cs, remap, nt);
            else
                hipLaunchKernelGGL((warp_fwd_lds_kernel<false>), dim3(grid), dim3(kBlock), 0, st, (const float*)feat,
                                   (const float*)flow, (float*)out, (int)C, (int)Hi, (int)Wi, (int)H, (int)W, txs, tys, cslabs,
                                   cs, remap, nt);
            return check_launch("ffwm_warp_forward(lds)");
        }
    }
    if (flip)
        hipLaunchKernelGGL((warp_fwd_kernel<T, true>), dim3(g.grid), dim3(kBlock), 0, st, feat, flow, out,
                           (int)C, (int)Hi, (int)Wi, (int)H, (int)W, g.tiles_x, g.tiles_y, g.cslabs, g.cs, remap);
    else
        hipLaunchKernelGGL((warp_fwd_kernel<T, false>), dim3(g.grid), dim3(kBlock), 0, st, feat, flow, out,
                           (int)C, (int)Hi, (int)Wi, (int)H, (int)W, g.tiles_x, g.tiles_y, g.cslabs, g.cs, remap);
    return check_launch("ffwm_warp_forward");
}

template <typename T>
int launch_bwd(const T* feat, const T* flow, const T* gout, T* gfeat, T* gflow, int64_t B, int64_t C,
               int64_t Hi, int64_t Wi, int64_t H, int64_t W, int flipcat, hipStream_t st) {
    const int flip = flipcat & 1;
    // flipcat bit 1: grad_feat is UNINITIALISED and must be produced whole (no caller zero-fill).  Only the owned-tile path stores
    // instead of adding; every other path gets the zero-fill it relies on from here.
    bool ovw = (flipcat & 2) != 0 && gfeat != nullptr;
    const double bytes = sizeof(T) * static_cast<double>(B) *
                         (2.0 * C * Hi * Wi + 4.0 * H * W + (flip ? 2.0 : 1.0) * C * H * W);
    const int remap = options().xcd_remap;
    const PlanePlan pp = plan_planes(B, C, Hi * Wi, H * W, sizeof(T) == 8 ? 2 : 8);
    const bool tile_path = sizeof(T) == 4 && gfeat && !(pp.ok && options().scatter_variant != 1) && Hi == H && Wi == W &&
                           options().scatter_variant != 1;
    if (ovw && !tile_path) {
        if (zero_fill(gfeat, sizeof(T) * static_cast<size_t>(B) * C * Hi * Wi, st)) return FFWM_ERR_LAUNCH;
        ovw = false;
    }
    if (gfeat && pp.ok && options().scatter_variant != 1) {
        {   // d(feat): LDS-resident planes, no contended global atomics
            LaunchScope ls(scope_at(flip ? "warp_flipcat_bwd_feat" : "warp_bwd_feat", Hi), st,
                           sizeof(T) * static_cast<double>(B) * (2.0 * C * Hi * Wi + 2.0 * H * W + (flip ? 2.0 : 1.0) * C * H * W));
            const unsigned grid = static_cast<unsigned>(B * pp.groups * pp.nsplit);
#define FFWM_WARP_PLANE(FL, CG)                                                                              \
    do {                                                                                                     \
        auto kfn = warp_bwd_feat_plane_kernel<T, FL, CG>;                                                    \
        allow_large_lds(reinterpret_cast<const void*>(kfn));                                                 \
        hipLaunchKernelGGL(kfn, dim3(grid), dim3(kPlaneThreads), pp.lds, st, flow, gout, gfeat, (int)C,      \
                           (int)Hi, (int)Wi, (int)H, (int)W, pp.groups, pp.nsplit);                          \
    } while (0)
#define FFWM_WARP_PLANE_CG(FL)                                                                               \
    switch (pp.cg) {                                                                                         \
        case 8: FFWM_WARP_PLANE(FL, 8); break;                                                               \
        case 4: FFWM_WARP_PLANE(FL, 4); break;                                                               \
        case 2: FFWM_WARP_PLANE(FL, 2); break;                                                               \
        default: FFWM_WARP_PLANE(FL, 1); break;                                                              \
    }
            if (flip) {
                FFWM_WARP_PLANE_CG(true)
            } else {
                FFWM_WARP_PLANE_CG(false)
            }
#undef FFWM_WARP_PLANE_CG
#undef FFWM_WARP_PLANE
        }
        if (int rc = check_launch("ffwm_warp_backward(feat)")) return rc;
        if (!gflow) return FFWM_OK;
        gfeat = nullptr;   // the pixel-major kernel below now only produces d(flow)
    }
    if constexpr (sizeof(T) == 4) {
        if (gfeat && Hi == H && Wi == W && options().scatter_variant != 1) {
            // planes beyond LDS, resolution kept: owned tiles + the far complement (no contended global atomics)
            constexpr int CG = 2;
            // warp_feat_fixed: 0 = double cells, two channels per group (rounds 3-5, the default); 1 = fixed-point cells on that kernel (900 us:
            // scratch in the add loop); 3 / 4 = fixed-point cells with a compact per-pixel state (3 registers instead of 10), four / two channels
            // per group: 551-588 / 498-528 us against the double cells' 474-535 on the same boxes -- the kernel is not bound by its LDS
            // atomics (profiles/r06_warp_feat_fixed_negative.txt), so all of them stay options
            const int fixmode = options().warp_feat_fixed;
            const int cgx = fixmode == 3 ? 4 : CG;        // (4 = the compact kernel with two channels per group)
            const int ntx = static_cast<int>((W + kWtTile - 1) / kWtTile), nty = static_cast<int>((H + kWtTile - 1) / kWtTile);
            const int groups = static_cast<int>((C + cgx - 1) / cgx);
            int gps = groups;                                       // channel groups per block: split until >= 3 blocks per CU
            while (gps > 1 && B * ntx * nty * ((groups + gps - 1) / gps) < 768) gps = (gps + 1) / 2;
            // ... and on towards ~12 blocks per CU while a block keeps >= 8 groups: two 8-wave blocks are resident per CU (119 registers),
            // so 800 blocks of 32 groups ran as two rounds with the second 44 % empty -- [32,64,256,256]: 897 -> 782 us with 8 groups
            // per block (tools/warp_feat_gps_sweep.py, profiles/r05_warp_feat_gps_sweep.txt; 4: 806, 2: 911)
            while (gps > 8 && B * ntx * nty * ((groups + gps - 1) / gps) < 3072) gps = (gps + 1) / 2;
            if (options().warp_feat_gps > 0) gps = options().warp_feat_gps < groups ? options().warp_feat_gps : groups;
            const int cslabs = (groups + gps - 1) / gps;
            auto launch_far = [&]() {
                LaunchScope ls(scope_at(flip ? "warp_flipcat_bwd_feat_far" : "warp_bwd_feat_far", Hi), st, sizeof(T) * static_cast<double>(B) * 2.0 * H * W);
                const int fx = static_cast<int>((W + kTileX - 1) / kTileX), fy = static_cast<int>((H + kTileY - 1) / kTileY);
                const unsigned fgrid = static_cast<unsigned>(B * fx * fy);
                if (flip) hipLaunchKernelGGL((warp_bwd_feat_far_kernel<true>), dim3(fgrid), dim3(kBlock), 0, st, (const float*)flow, (const float*)gout, (float*)gfeat, (int)C, (int)H, (int)W, fx, fy);
                else hipLaunchKernelGGL((warp_bwd_feat_far_kernel<false>), dim3(fgrid), dim3(kBlock), 0, st, (const float*)flow, (const float*)gout, (float*)gfeat, (int)C, (int)H, (int)W, fx, fy);
                return check_launch("ffwm_warp_backward(feat, far)");
            };
            // (stream order: the far kernel ADDS with global atomics -- in front of the tiles' read-modify-write, behind their plain stores)
            if (!ovw)
                if (int rc = launch_far()) return rc;
            {
                // algorithmic bytes: the overwriting variant does not read grad_feat
                LaunchScope ls(scope_at(flip ? "warp_flipcat_bwd_feat_tile" : "warp_bwd_feat_tile", Hi), st,
                               sizeof(T) * static_cast<double>(B) * ((ovw ? 1.0 : 2.0) * C * Hi * Wi + 2.0 * H * W + (flip ? 2.0 : 1.0) * C * H * W));
                const unsigned grid = static_cast<unsigned>(B * ntx * nty * cslabs);
                // 1 = 32-bit fixed-point cells (round 6 experiment, OFF: correct, but hipcc cannot hold the kernel in 128 registers -- 64-104 bytes of
                // scratch reloaded inside the add loop, each behind an s_waitcnt vmcnt(0): 495 -> 900 us; profiles/r06_warp_feat_fixed_negative.txt)
                const bool fix = fixmode == 1;
#define FFWM_WT(FL, OV) do { if (fix) hipLaunchKernelGGL((warp_bwd_feat_tile_kernel<FL, CG, OV, true>), dim3(grid), dim3(kWtThreads), 0, st, (const float*)flow, (const float*)gout, (float*)gfeat, (int)C, (int)H, (int)W, ntx, nty, gps, cslabs); \
                             else hipLaunchKernelGGL((warp_bwd_feat_tile_kernel<FL, CG, OV, false>), dim3(grid), dim3(kWtThreads), 0, st, (const float*)flow, (const float*)gout, (float*)gfeat, (int)C, (int)H, (int)W, ntx, nty, gps, cslabs); } while (0)
#define FFWM_WT2(FL, OV) hipLaunchKernelGGL((warp_bwd_feat_tile2_kernel<FL, 4, OV>), dim3(grid), dim3(kWtThreads), 0, st, (const float*)flow, (const float*)gout, (float*)gfeat, (int)C, (int)H, (int)W, ntx, nty, gps, cslabs)
                if (fixmode == 3) {
                    if (flip) { if (ovw) FFWM_WT2(true, true); else FFWM_WT2(true, false); }
                    else { if (ovw) FFWM_WT2(false, true); else FFWM_WT2(false, false); }
                } else if (fixmode == 4) {
#define FFWM_WT3(FL, OV) hipLaunchKernelGGL((warp_bwd_feat_tile2_kernel<FL, 2, OV>), dim3(grid), dim3(kWtThreads), 0, st, (const float*)flow, (const float*)gout, (float*)gfeat, (int)C, (int)H, (int)W, ntx, nty, gps, cslabs)
                    if (flip) { if (ovw) FFWM_WT3(true, true); else FFWM_WT3(true, false); }
                    else { if (ovw) FFWM_WT3(false, true); else FFWM_WT3(false, false); }
#undef FFWM_WT3
                } else if (flip) { if (ovw) FFWM_WT(true, true); else FFWM_WT(true, false); }
                else { if (ovw) FFWM_WT(false, true); else FFWM_WT(false, false); }
#undef FFWM_WT2
#undef FFWM_WT
            }
            if (int rc = check_launch("ffwm_warp_backward(feat, tiles)")) return rc;
            if (ovw)
                if (int rc = launch_far()) return rc;
            if (!gflow) return FFWM_OK;
            gfeat = nullptr;
        }
    }
    if constexpr (sizeof(T) == 4) {
        // d(flow) alone on LDS-staged tiles (warp_bwd_flow_lds_body) for the same tensors the forward takes there; warp_multi_lds = 1: never
        const int variant = options().warp_fwd_variant;
        if (!gfeat && gflow && options().warp_multi_lds != 1 &&
            (variant == 2 || (variant == 0 && H >= 64 && W >= 64 && B * C * H * W >= (1LL << 24)))) {
            const int txs = static_cast<int>((W + kWlTileX - 1) / kWlTileX), tys = static_cast<int>((H + kWlTileY - 1) / kWlTileY);
            // measured at [32,64,256,256] (tools/warp_bwd_flow_variants.py): slab 8 / 16 / 32 / 64 -> 492 / 375 / 330 / 310 us (direct: 423)
            int cs = options().channel_slab > 0 ? options().channel_slab : 64;
            if (cs > C) cs = static_cast<int>(C);
            while (cs > 8 && B * txs * tys * ((C + cs - 1) / cs) < 2048) cs = (cs + 1) / 2;
            const int cslabs = static_cast<int>((C + cs - 1) / cs);
            const unsigned grid = static_cast<unsigned>(B * txs * tys * cslabs);
            LaunchScope ls(scope_at(flip ? "warp_flipcat_bwd_flow" : "warp_bwd_flow", H), st,
                           sizeof(T) * static_cast<double>(B) * (static_cast<double>(C) * Hi * Wi + 4.0 * H * W + (flip ? 2.0 : 1.0) * C * H * W));
            if (flip)
                hipLaunchKernelGGL((warp_bwd_flow_lds_kernel<true>), dim3(grid), dim3(kBlock), 0, st, (const float*)feat, (const float*)flow,
                                   (const float*)gout, (float*)gflow, (int)C, (int)Hi, (int)Wi, (int)H, (int)W, txs, tys, cslabs, cs, remap);
            else
                hipLaunchKernelGGL((warp_bwd_flow_lds_kernel<false>), dim3(grid), dim3(kBlock), 0, st, (const float*)feat, (const float*)flow,
                                   (const float*)gout, (float*)gflow, (int)C, (int)Hi, (int)Wi, (int)H, (int)W, txs, tys, cslabs, cs, remap);
            return check_launch("ffwm_warp_backward(flow, lds)");
        }
    }
    const Geometry g = plan(B, C, H, W, 32);
    LaunchScope ls(scope_at(gfeat ? (flip ? "warp_flipcat_bwd" : "warp_bwd") : (flip ? "warp_flipcat_bwd_flow" : "warp_bwd_flow"), H), st,
                   gfeat ? bytes : sizeof(T) * static_cast<double>(B) * (static_cast<double>(C) * Hi * Wi + 4.0 * H * W + (flip ? 2.0 : 1.0) * C * H * W));
    if (flip)
        hipLaunchKernelGGL((warp_bwd_kernel<T, true>), dim3(g.grid), dim3(kBlock), 0, st, feat, flow, gout,
                           gfeat, gflow, (int)C, (int)Hi, (int)Wi, (int)H, (int)W, g.tiles_x, g.tiles_y,
                           g.cslabs, g.cs, remap);
    else
        hipLaunchKernelGGL((warp_bwd_kernel<T, false>), dim3(g.grid), dim3(kBlock), 0, st, feat, flow, gout,
                           gfeat, gflow, (int)C, (int)Hi, (int)Wi, (int)H, (int)W, g.tiles_x, g.tiles_y,
                           g.cslabs, g.cs, remap);
    return check_launch("ffwm_warp_backward");
}


inline bool fwd_wants_lds(int64_t B, int64_t C, int64_t H, int64_t W, size_t esz) {
    const int variant = options().warp_fwd_variant;
    return esz == 4 && (variant == 2 || (variant == 0 && H >= 64 && W >= 64 && B * C * H * W >= (1LL << 24)));
}

// A problem of a multi FORWARD launch on LDS-staged tiles?  warp_multi_lds: 0 = auto (float, planes of >= 32 x 32 output pixels with
// >= 32 channels: netG's three levels -- 31 vs 42 us warm, 49 vs 69 us cold for the launch, profiles/r04_warp_multi_lds_sweep.txt;
// the 3-channel image warps stay on direct gathers), 1 = never, 2 = always (float).
inline bool multi_wants_lds(const ffwm_warp_problem& pr, size_t esz) {
    const int v = options().warp_multi_lds;
    if (esz != 4 || v == 1) return false;
    return v == 2 || (pr.H >= 32 && pr.W >= 32 && pr.C >= 32);
}

inline void fill_problem(WarpProblem& q, const ffwm_warp_problem& pr, int cs_default, unsigned begin, bool lds) {
    q.C = static_cast<int>(pr.C); q.Hi = static_cast<int>(pr.Hi); q.Wi = static_cast<int>(pr.Wi);
    q.H = static_cast<int>(pr.H); q.W = static_cast<int>(pr.W);
    q.begin = begin;
    q.lds = lds ? 1 : 0;
    if (lds) {
        const int txs = static_cast<int>((pr.W + kWlTileX - 1) / kWlTileX), tys = static_cast<int>((pr.H + kWlTileY - 1) / kWlTileY);
        int cs = options().channel_slab > 0 ? options().channel_slab : 16;
        if (cs > pr.C) cs = static_cast<int>(pr.C);
        while (cs > 4 && pr.B * txs * tys * ((pr.C + cs - 1) / cs) < 1024) cs = (cs + 1) / 2;     // >= 4 tiles per CU for the problem
        q.tiles_x = txs; q.tiles_y = tys; q.cs = cs;
        q.cslabs = static_cast<int>((pr.C + cs - 1) / cs);
        q.nblk = static_cast<unsigned>(pr.B * txs * tys * q.cslabs);
        return;
    }
    const Geometry g = plan(pr.B, pr.C, pr.H, pr.W, cs_default);
    q.tiles_x = g.tiles_x; q.tiles_y = g.tiles_y; q.cslabs = g.cslabs; q.cs = g.cs;
    q.nblk = g.grid;
}

// Largest problem first: the workgroups of a launch are dispatched in index order, so the big level's tiles start at once and the
// small levels fill the tail (longest-processing-time order; option warp_multi_order = 1 keeps the caller's order).
inline std::vector<int> multi_order(const ffwm_warp_problem* probs, int n) {
    std::vector<int> idx(n);
    for (int i = 0; i < n; ++i) idx[i] = i;
    if (options().warp_multi_order == 0)
        std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) {
            return probs[a].B * probs[a].C * probs[a].H * probs[a].W > probs[b].B * probs[b].C * probs[b].H * probs[b].W;
        });
    return idx;
}

template <typename T>
int launch_fwd_multi(const ffwm_warp_problem* probs, int n, int flip, hipStream_t st) {
    WarpTable tab;
    tab.n = 0;
    tab.nt = options().warp_nt;
    unsigned blocks = 0;
    double bytes = 0;
    auto flush = [&]() -> int {
        if (tab.n == 0) return FFWM_OK;
        {
            LaunchScope ls(flip ? "warp_flipcat_fwd_multi" : "warp_fwd_multi", st, bytes);
            if (flip) hipLaunchKernelGGL((warp_fwd_multi_kernel<T, true>), dim3(blocks), dim3(kBlock), 0, st, tab);
            else hipLaunchKernelGGL((warp_fwd_multi_kernel<T, false>), dim3(blocks), dim3(kBlock), 0, st, tab);
        }
        tab.n = 0; blocks = 0; bytes = 0;
        return check_launch("ffwm_warp_multi_forward");
    };
    const std::vector<int> order = multi_order(probs, n);
    for (int oi = 0; oi < n; ++oi) {
        const ffwm_warp_problem& pr = probs[order[oi]];
        if (fwd_wants_lds(pr.B, pr.C, pr.H, pr.W, sizeof(T))) {      // HBM-resident output: the tile kernel, by itself
            if (int rc = launch_fwd<T>((const T*)pr.feat, (const T*)pr.flow, (T*)pr.output, pr.B, pr.C, pr.Hi, pr.Wi, pr.H, pr.W, flip, st)) return rc;
            continue;
        }
        WarpProblem& q = tab.p[tab.n];
        q.feat = pr.feat; q.flow = pr.flow; q.out = pr.output; q.gout = nullptr;
        fill_problem(q, pr, 16, blocks, multi_wants_lds(pr, sizeof(T)));
        blocks += (q.nblk + 7u) & ~7u;
        bytes += sizeof(T) * static_cast<double>(pr.B) * (static_cast<double>(pr.C) * pr.Hi * pr.Wi + 2.0 * pr.H * pr.W + (flip ? 2.0 : 1.0) * pr.C * pr.H * pr.W);
        if (++tab.n == kMaxWarpProblems)
            if (int rc = flush()) return rc;
    }
    return flush();
}

template <typename T>
int launch_bwd_multi(const ffwm_warp_problem* probs, int n, int flip, hipStream_t st) {
    // (the d(flow) table below does not use the store policy)
    // d(feat): problems whose planes fit LDS and that share the channels-per-block of their plane plan go out together (one launch per
    // CG value: the step's seven image-warp problems are two launches); everything else one launch per problem (scope names carry the level)
    std::vector<bool> done(n, false);
    if (options().scatter_variant != 1 && options().warp_multi_planes != 1) {
        for (int cg = 1; cg <= 8; cg *= 2) {
            PlaneTable tab;
            tab.n = 0;
            unsigned blocks = 0;
            size_t lds = 0;
            double bytes = 0;
            std::vector<int> members;
            auto flush = [&]() -> int {
                if (tab.n < 2) {                 // a lone problem keeps its own launch (and its per-level scope name)
                    tab.n = 0; blocks = 0; lds = 0; bytes = 0; members.clear();
                    return FFWM_OK;
                }
                {
                    LaunchScope ls(flip ? "warp_flipcat_bwd_feat_multi" : "warp_bwd_feat_multi", st, bytes);
#define FFWM_PLANE_MULTI(FL, CGV)                                                                                  \
    do {                                                                                                           \
        auto kfn = warp_bwd_feat_plane_multi_kernel<T, FL, CGV>;                                                   \
        allow_large_lds(reinterpret_cast<const void*>(kfn));                                                       \
        hipLaunchKernelGGL(kfn, dim3(blocks), dim3(kPlaneThreads), lds, st, tab);                                  \
    } while (0)
                    if (flip) {
                        switch (cg) { case 8: FFWM_PLANE_MULTI(true, 8); break; case 4: FFWM_PLANE_MULTI(true, 4); break;
                                      case 2: FFWM_PLANE_MULTI(true, 2); break; default: FFWM_PLANE_MULTI(true, 1); break; }
                    } else {
                        switch (cg) { case 8: FFWM_PLANE_MULTI(false, 8); break; case 4: FFWM_PLANE_MULTI(false, 4); break;
                                      case 2: FFWM_PLANE_MULTI(false, 2); break; default: FFWM_PLANE_MULTI(false, 1); break; }
                    }
#undef FFWM_PLANE_MULTI
                }
                for (int m : members) done[m] = true;
                tab.n = 0; blocks = 0; lds = 0; bytes = 0; members.clear();
                return check_launch("ffwm_warp_multi_backward(feat, planes)");
            };
            for (int i = 0; i < n; ++i) {
                const ffwm_warp_problem& pr = probs[i];
                if (!pr.grad_feat || done[i]) continue;
                const PlanePlan pp = plan_planes(pr.B, pr.C, pr.Hi * pr.Wi, pr.H * pr.W, sizeof(T) == 8 ? 2 : 8);
                // (only the few-channel image warps: netG's 64 / 128-channel levels gain nothing from sharing a launch -- measured 63 us
                //  for the 128 + 64 px levels together against 51 + 13 us apart)
                if (!pp.ok || pp.cg != cg || pr.C > 8) continue;
                PlaneProblem& q = tab.p[tab.n];
                q.flow = pr.flow; q.gout = pr.grad_output; q.gfeat = pr.grad_feat;
                q.C = static_cast<int>(pr.C); q.Hi = static_cast<int>(pr.Hi); q.Wi = static_cast<int>(pr.Wi);
                q.H = static_cast<int>(pr.H); q.W = static_cast<int>(pr.W);
                q.groups = pp.groups; q.nsplit = pp.nsplit; q.begin = blocks;
                blocks += static_cast<unsigned>(pr.B * pp.groups * pp.nsplit);
                if (pp.lds > lds) lds = pp.lds;
                bytes += sizeof(T) * static_cast<double>(pr.B) * (2.0 * pr.C * pr.Hi * pr.Wi + 2.0 * pr.H * pr.W + (flip ? 2.0 : 1.0) * pr.C * pr.H * pr.W);
                members.push_back(i);
                if (++tab.n == kMaxWarpProblems)
                    if (int rc = flush()) return rc;
            }
            if (int rc = flush()) return rc;
        }
    }
    for (int i = 0; i < n; ++i) {
        const ffwm_warp_problem& pr = probs[i];
        if (!pr.grad_feat || done[i]) continue;
        if (int rc = launch_bwd<T>((const T*)pr.feat, (const T*)pr.flow, (const T*)pr.grad_output, (T*)pr.grad_feat, nullptr, pr.B, pr.C,
                                   pr.Hi, pr.Wi, pr.H, pr.W, flip, st)) return rc;
    }
    // d(flow): every problem that wants it, one launch
    WarpTable tab;
    tab.n = 0;
    tab.nt = options().warp_pair_loads;            // (the backward has no stores to stream: the field carries the pair-load switch)
    unsigned blocks = 0;
    double bytes = 0;
    auto flush = [&]() -> int {
        if (tab.n == 0) return FFWM_OK;
        {
            LaunchScope ls(flip ? "warp_flipcat_bwd_flow_multi" : "warp_bwd_flow_multi", st, bytes);
            if (flip) hipLaunchKernelGGL((warp_bwd_flow_multi_kernel<T, true>), dim3(blocks), dim3(kBlock), 0, st, tab);
            else hipLaunchKernelGGL((warp_bwd_flow_multi_kernel<T, false>), dim3(blocks), dim3(kBlock), 0, st, tab);
        }
        tab.n = 0; blocks = 0; bytes = 0;
        return check_launch("ffwm_warp_multi_backward(flow)");
    };
    const std::vector<int> order = multi_order(probs, n);
    for (int oi = 0; oi < n; ++oi) {
        const ffwm_warp_problem& pr = probs[order[oi]];
        if (!pr.grad_flow) continue;
        WarpProblem& q = tab.p[tab.n];
        q.feat = pr.feat; q.flow = pr.flow; q.out = pr.grad_flow; q.gout = pr.grad_output;
        fill_problem(q, pr, 32, blocks, false);
        blocks += (q.nblk + 7u) & ~7u;
        bytes += sizeof(T) * static_cast<double>(pr.B) * (static_cast<double>(pr.C) * pr.Hi * pr.Wi + 4.0 * pr.H * pr.W + (flip ? 2.0 : 1.0) * pr.C * pr.H * pr.W);
        if (++tab.n == kMaxWarpProblems)
            if (int rc = flush()) return rc;
    }
    return flush();
}

}  // namespace
}  // namespace ffwm

using namespace ffwm;

extern "C" int ffwm_warp_forward(const void* feat, const void* flow, void* output, int64_t B, int64_t C,
                                 int64_t Hi, int64_t Wi, int64_t H, int64_t W, int flipcat, int dtype,
                                 void* stream) {
    const char* fn = "ffwm_warp_forward";
    FFWM_REQUIRE(feat && flow && output, FFWM_ERR_ARG, "%s: NULL tensor pointer", fn);
    if (int rc = check_dims(fn, B, C, Hi, Wi, H, W, dtype)) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dtype == FFWM_F32)
        return launch_fwd<float>((const float*)feat, (const float*)flow, (float*)output, B, C, Hi, Wi, H, W,
                                 flipcat, st);
    return launch_fwd<double>((const double*)feat, (const double*)flow, (double*)output, B, C, Hi, Wi, H, W,
                              flipcat, st);
}

extern "C" int ffwm_warp_backward(const void* feat, const void* flow, const void* grad_output,
                                  void* grad_feat, void* grad_flow, int64_t B, int64_t C, int64_t Hi,
                                  int64_t Wi, int64_t H, int64_t W, int flipcat, int dtype, void* stream) {
    const char* fn = "ffwm_warp_backward";
    FFWM_REQUIRE(feat && flow && grad_output, FFWM_ERR_ARG, "%s: NULL tensor pointer", fn);
    if (int rc = check_dims(fn, B, C, Hi, Wi, H, W, dtype)) return rc;
    if (!grad_feat && !grad_flow) return FFWM_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dtype == FFWM_F32)
        return launch_bwd<float>((const float*)feat, (const float*)flow, (const float*)grad_output,
                                 (float*)grad_feat, (float*)grad_flow, B, C, Hi, Wi, H, W, flipcat, st);
    return launch_bwd<double>((const double*)feat, (const double*)flow, (const double*)grad_output,
                              (double*)grad_feat, (double*)grad_flow, B, C, Hi, Wi, H, W, flipcat, st);
}

extern "C" int ffwm_warp_multi_forward(const ffwm_warp_problem* problems, int n, int flipcat, int dtype, void* stream) {
    const char* fn = "ffwm_warp_multi_forward";
    FFWM_REQUIRE(problems && n > 0, FFWM_ERR_ARG, "%s: empty problem list", fn);
    for (int i = 0; i < n; ++i) {
        FFWM_REQUIRE(problems[i].feat && problems[i].flow && problems[i].output, FFWM_ERR_ARG, "%s: NULL tensor pointer in problem %d", fn, i);
        if (int rc = check_dims(fn, problems[i].B, problems[i].C, problems[i].Hi, problems[i].Wi, problems[i].H, problems[i].W, dtype)) return rc;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    return dtype == FFWM_F32 ? launch_fwd_multi<float>(problems, n, flipcat, st) : launch_fwd_multi<double>(problems, n, flipcat, st);
}

extern "C" int ffwm_warp_multi_backward(const ffwm_warp_problem* problems, int n, int flipcat, int dtype, void* stream) {
    const char* fn = "ffwm_warp_multi_backward";
    FFWM_REQUIRE(problems && n > 0, FFWM_ERR_ARG, "%s: empty problem list", fn);
    for (int i = 0; i < n; ++i) {
        FFWM_REQUIRE(problems[i].feat && problems[i].flow && problems[i].grad_output, FFWM_ERR_ARG, "%s: NULL tensor pointer in problem %d", fn, i);
        if (int rc = check_dims(fn, problems[i].B, problems[i].C, problems[i].Hi, problems[i].Wi, problems[i].H, problems[i].W, dtype)) return rc;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    return dtype == FFWM_F32 ? launch_bwd_multi<float>(problems, n, flipcat, st) : launch_bwd_multi<double>(problems, n, flipcat, st);
}
