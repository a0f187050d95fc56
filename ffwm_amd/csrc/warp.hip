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
#include "common.hpp"

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
__global__ void __launch_bounds__(kBlock)
warp_fwd_kernel(const T* __restrict__ feat, const T* __restrict__ flow, T* __restrict__ out, int C,
                int Hi, int Wi, int H, int W, int tiles_x, int tiles_y, int cslabs, int cs,
                int remap) {
    const TileCoord tc = decode_tile(tiles_x, tiles_y, cslabs, remap);
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

    for (int c = c0; c < c1; ++c, fp += iplane, op += plane) {
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
warp_bwd_kernel(const T* __restrict__ feat, const T* __restrict__ flow, const T* __restrict__ gout,
                T* __restrict__ gfeat, T* __restrict__ gflow, int C, int Hi, int Wi, int H, int W,
                int tiles_x, int tiles_y, int cslabs, int cs, int remap) {
    const TileCoord tc = decode_tile(tiles_x, tiles_y, cslabs, remap);
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
    T gix = 0, giy = 0;

    for (int c = c0; c < c1; ++c, fp += iplane, op += plane) {
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
            // ATen grid_sampler_2d_backward; invalid corners read 0 and drop out.
            gix -= s0 * cn.dyw[0] * g;
            giy -= s0 * cn.dxw[0] * g;
            gix += s1 * cn.dyw[0] * g;
            giy -= s1 * cn.dxw[1] * g;
            gix -= s2 * cn.dyw[1] * g;
            giy += s2 * cn.dxw[0] * g;
            gix += s3 * cn.dyw[1] * g;
            giy += s3 * cn.dxw[1] * g;
        }
    }
    if (gflow) {
        atomic_add(gflow + foff, (static_cast<T>(Wi) / 2) * gix);
        atomic_add(gflow + foff + plane, (static_cast<T>(Hi) / 2) * giy);
    }
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
__global__ void __launch_bounds__(kPlaneThreads)
warp_bwd_feat_plane_kernel(const T* __restrict__ flow, const T* __restrict__ gout, T* __restrict__ gfeat,
                           int C, int Hi, int Wi, int H, int W, int groups, int nsplit) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    double* acc = reinterpret_cast<double*>(smem_raw);
    unsigned t = blockIdx.x;
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
    LaunchScope ls(flip ? "warp_flipcat_fwd" : "warp_fwd", st, bytes);
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
               int64_t Hi, int64_t Wi, int64_t H, int64_t W, int flip, hipStream_t st) {
    const double bytes = sizeof(T) * static_cast<double>(B) *
                         (2.0 * C * Hi * Wi + 4.0 * H * W + (flip ? 2.0 : 1.0) * C * H * W);
    const int remap = options().xcd_remap;
    const PlanePlan pp = plan_planes(B, C, Hi * Wi, H * W, sizeof(T) == 8 ? 2 : 8);
    if (gfeat && pp.ok && options().scatter_variant != 1) {
        {   // d(feat): LDS-resident planes, no contended global atomics
            LaunchScope ls(flip ? "warp_flipcat_bwd_feat" : "warp_bwd_feat", st,
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
    const Geometry g = plan(B, C, H, W, 32);
    LaunchScope ls(gfeat ? (flip ? "warp_flipcat_bwd" : "warp_bwd") : (flip ? "warp_flipcat_bwd_flow" : "warp_bwd_flow"), st,
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
