// common.hpp -- shared host/device helpers of libffwm_hip.so (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ffwm_hip.h"

namespace ffwm {

constexpr int kWave = 64;          // CDNA wavefront
constexpr int kBlock = 256;        // 4 waves, one per SIMD
constexpr int kTileX = 64;         // one wave covers 64 consecutive x: full-line coalescing
constexpr int kTileY = kBlock / kTileX;

// ---------------------------------------------------------------- host side
void set_error(const char* fmt, ...);
int check_launch(const char* what);
// Kernels that ask for more than 64 KiB of dynamic LDS need their limit raised once (160 KiB per CU on gfx950).
void allow_large_lds(const void* kernel);
// A 256-byte device scratch that belongs to (device, stream): counters a launch sequence hands from one kernel to the next without
// a host round trip.  NULL when it cannot be had (allocation failure, or first use on a stream that is being captured).
void* stream_scratch(hipStream_t st);
// Compute units of the CURRENT device (the tensors' device: every entry point runs under the caller's device guard), cached per device.
int device_cus();
// Zero-fill of `bytes` bytes (a multiple of 4, `p` 4-byte aligned) by a KERNEL of this library on `st`.  Not hipMemsetAsync: under
// hipGraph capture that becomes a memset node, and on ROCm 7.0 the fill pattern of such nodes came out corrupted (garbage in fixed
// dwords of every 16 bytes, over the whole buffer) when several graphs with side-stream branches were replayed back to back --
// round 5, profiles/r05_wgrad_nan_root_cause.txt.  Returns a hipError_t-free status: 0 = launched.
int zero_fill(void* p, size_t bytes, hipStream_t st);
constexpr int kMaxLdsBytes = 160 * 1024;

struct Options {
    int be_fwd_variant = 0;   // 0 = auto
    int be_bwd_variant = 0;
    int channel_slab = 0;     // 0 = auto
    int xcd_remap = 1;
    int scatter_variant = 0;  // backward scatters: 0 = auto (LDS-resident plane when it fits), 1 = global atomics
    int rows_per_thread = 0;  // block_extractor LDS kernels: 0 = auto, 1 / 2 / 4
    int warp_fwd_variant = 0; // warp forward: 0 = auto, 1 = direct gathers, 2 = LDS-staged tiles
    int be_bwd_halo = 0;      // block_extractor owned-tile backward: halo in pixels (<= 4 -> 4 (default), else 8)
    int be_bwd_rows = 0;      // ... region height: 32 (default) or 64 rows
    int rs_fwd_variant = 0;   // resample2d forward: 0 = auto, 1 = direct gathers, LDS-staged tiles 64 x 16 / 4 / 8: 2 / 3 / 4 (two buffers), 6 / 7 / 5 (one)
    int rs_bwd1_variant = 0;  // resample2d d_input1: 0 = auto (ks 4: tap-lane kernel, on calls of >= 2^18 pixels the tile kernel instead when a pre-pass finds the flow smooth; else plane kernel when a plane fits LDS, else tile kernel), 1 = round-2 auto (plane / tile), 2 = tile kernel, 5 = tap-lane kernel with 8-wave blocks (16-row tiles)
    int conv_tile_variant = 0; // conv_fwd.hip workgroup tile: 0 auto, 1 = 64 x 64, 2 = 128 x 64, 3 = 64 x 128, 4 = 128 x 128
    int conv_wino_raw = 1;     // conv_winograd.hip: stage the input window through LDS when a workgroup covers whole tile rows
    int conv_fwd_split_target = 0; // conv_fwd.hip: workgroups a launch with < 256 tiles is cut into along the reduction (each slice adds its tile by atomics); 0 = by shape (768 / 384)
    int conv_wgrad_slice_target = 0; // conv_bwd.hip tiled weight gradient: workgroups a launch is cut into along the pixels (0 = 512)
    int conv_wino_ws = 0;      // conv_winograd.hip: the wave-specialised variant (8 MFMA waves + 4 staging waves) for calls without a split reduction
    int conv_wino_split = 1;   // conv_winograd.hip: cut the reduction of a call with few (strip, k tile) pairs over 2 / 4 workgroups (atomics into a zeroed output); 2 = at most two pieces (a two-term float sum does not depend on the order: bit-reproducible); 0 = never
    int conv_thin_tail = 1;    // conv_winograd.hip: 1-4 output channels past a multiple of 64 on the thin direct kernel
    int warp_nt = 0;          // warp forward (direct / multi-problem kernels): 1 = streaming (nt) stores, 2 = nt feature loads too
    int warp_pair_loads = 1;  // warp d(flow), multi-problem launch, fp32: one 8-byte load per corner ROW instead of two dword gathers
    int conv_wgrad_wino = 0;   // conv_wgrad.hip, the full 64-channel tiles on the Winograd-domain kernel (conv_wgrad_wino.hip): 0 = auto (>= 16 chunks per CU), 1 = whenever served, 2 = never
    int warp_multi_planes = 0; // multi-problem warp backward: 0 = d(feat) plane problems of one CG share a launch, 1 = one launch per problem
    int warp_multi_lds = 0;   // multi-problem warp launches: 0 = auto (LDS-staged tiles for float planes >= 64 x 64, C >= 32), 1 = direct gathers, 2 = LDS tiles
    int warp_multi_order = 0; // multi-problem warp launches: 0 = largest problem first, 1 = the caller's order
    int conv_fwd_kfast = 1;      // conv_fwd.hip: workgroup index with the channel tiles fastest + XCD remap (round 5); 0 = pixel tiles fastest
    int warp_feat_gps = 0;       // warp d(feat) owned-tile kernel: channel groups per block, 0 = auto
    int rs_bwd1_rpt = 0;         // resample2d d_input1 tile kernel: pixel rows per thread, 0 = auto (4), 2
    int rs_bwd1_fixed = 0;       // resample2d d_input1 tile kernel: 0 = 32-bit fixed-point box cells (round 5), 2 = double cells
    int be_bwd_fixed = 0;        // block_extractor / block attention shared-cell backward: 0 = 32-bit fixed-point accumulator cells (round 5), 2 = double cells
    int rs_bwd1_owned = 0;       // resample2d d_input1, large calls: 0 = owned tiles + far complement, plain stores (round 6), 2 = the shared-cell tile kernel with its fold atomics (rounds 3-5)
    int rs_bwd1_owned_min_pixels = 0;    // owned tiles for calls of at least this many pixels (B H W); 0 = 2^18
    int rs_bwd1_owned_blocks = 0; // owned tiles: the channel slab is halved until the launch has this many blocks (0 = 2048)
    int warp_feat_fixed = 0;     // warp d(feat) owned-tile kernel: 0 = double cells, 1 = 32-bit fixed-point cells (round 6 experiment: slower -- register spills)
    int conv_thin_variant = 0;   // ffwm_conv_thin_forward (3 x 3): 0 = by shape, 1 / 2 = 8 / 16 output channels per lane, +4 = one input channel per step (no grouped prefetch)
    int ba_fwd_pix = 1;          // block attention forward: 1-4 = ba_fwd_pix_kernel (channel-innermost boxes, coefficients in registers; pixel rows / channels per group / blocks per CU 8/4/4, 16/4/4, 8/8/4, 8/4/6: 65-86 us at cfg-5), 0 = rounds 3-5's be_fwd_lds_kernel<.., MODE 1> (98-117 us)
    int ba_bwd_fused = 3;        // block attention backward, ba_bwd_src_kernel's tile rows / threads: 1 = 32 / 256, 2 = 16 / 256, 3 = 32 / 512 (tools/r06/ba_bwd_time.py)
    int ba_bwd_pix = 4;          // ba_bwd_pix_kernel's pixel rows / channels per group / waves per SIMD: 0 = 16 / 4 / 3, 1 = 8 / 8 / 4, 2 = 16 / 4 / 4, 3 = 16 / 8 / 3, 4 = 8 / 4 / 4, 5 = 8 / 4 / 6 (tools/r06/ba_bwd_time.py)
    int be_bwd_flush = 0;        // block_extractor / block attention shared-cell backward: 0 = every in-image cell by a global atomic, 1 = interior box cells by read-modify-write (round 6 experiment, slower)
    int zero_fill_memset = 0;    // 1 = zero_fill() calls hipMemsetAsync as rounds 1-4 did (diagnosis: reproduces the corrupted memset nodes)
    int conv_wgrad_unsliced = 0; // conv_bwd.hip tiled weight gradient: 1 = never cut the pixel range into slices (no zero-fill, no atomics: a diagnosis switch)
    int conv_wgrad_prezeroed = 0; // conv_bwd.hip tiled weight gradient: 1 = the caller hands over ZEROED grad_weight / grad_bias (a slice of the trainer's gradient arena, cleared by one launch per step): a sliced launch skips its own zero-fill
    int ablate = 0;           // bench-only ablation bits (1 = skip source fetch, 2 = skip stores)
};
Options& options();

// Per-launch timing scope: records HIP events on `stream` around the enclosed launch when
// profiling is enabled (ffwm_prof_enable).  `bytes` = algorithmic bytes of that launch.
class LaunchScope {
  public:
    LaunchScope(const char* name, hipStream_t stream, double bytes, double flops = 0.0);   // flops: MFMA kernels
    ~LaunchScope();

  private:
    int slot_;
    hipStream_t stream_;
};

// "<base>@<size>": an interned launch-scope name that keeps the levels of a multi-scale caller apart in the profile
// (netG warps 32 x 32, 64 x 64 and 128 x 128 planes; only the last one is large enough to be HBM-bound).
const char* scope_at(const char* base, int64_t size);

// Pixel-tile launch geometry shared by the per-pixel kernels: a 64 x 4 pixel tile per block
// (kTileX x kTileY), a slab of `cs` channels per block.
struct Geometry {
    int tiles_x, tiles_y, cslabs, cs;
    unsigned grid;
};
Geometry plan(int64_t B, int64_t C, int64_t H, int64_t W, int cs_default);

inline bool dtype_ok(int dtype) { return dtype == FFWM_F32 || dtype == FFWM_F64; }

// Element strides of a 4-D grad_output as the reference's kernels read it (DIM3_INDEX with the tensor's strides,
// cuda/block_extractor/block_extractor_kernel.cu:8-15): the *_backward_strided entry points (ABI 5).
struct GoStrides {
    long long b, c, y, x;
};
inline bool go_contiguous(const int64_t* st, int64_t C, int64_t H, int64_t W) {
    return st == nullptr || (st[3] == 1 && st[2] == W && st[1] == H * W && st[0] == C * H * W);
}

#define FFWM_REQUIRE(cond, code, ...)  \
    do {                               \
        if (!(cond)) {                 \
            ::ffwm::set_error(__VA_ARGS__); \
            return (code);             \
        }                              \
    } while (0)

// ---------------------------------------------------------------- device side
#ifdef __HIPCC__

// Observed dispatch: workgroup b runs on XCD b % 8, each XCD has its own 4 MiB L2.  Give every
// XCD a CONTIGUOUS range of logical tiles so neighbouring tiles (which share source halo rows)
// hit the same L2.  Bijective for any grid size.  Speed only -- never correctness.
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nblk, int enable) {
    if (!enable) return bid;
    const unsigned xcd = bid & 7u, idx = bid >> 3;
    const unsigned q = nblk >> 3, r = nblk & 7u;
    const unsigned start = xcd < r ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q;
    return start + idx;
}

__device__ __forceinline__ float floor_t(float v) { return floorf(v); }
__device__ __forceinline__ double floor_t(double v) { return floor(v); }

// max(min(int(v), n-1), 0) with the float->int conversion made safe for NaN / huge values
// (the hardware conversion saturates and maps NaN to 0, as CUDA's does; C++ leaves it undefined).
template <typename T>
__device__ __forceinline__ int clamp_index(T v, int n) {
    T lo = static_cast<T>(-1), hi = static_cast<T>(n);
    T c = v < lo ? lo : (v > hi ? hi : v);   // NaN compares false twice -> stays NaN
    int i = (c != c) ? 0 : static_cast<int>(c);
    i = i < n - 1 ? i : n - 1;
    return i > 0 ? i : 0;
}

// int(v): C truncation toward zero, made safe the way the hardware conversion behaves
// (saturating, NaN -> 0).  Used for the reference's `alpha = xf - int(xf)` quirk.
template <typename T>
__device__ __forceinline__ int clamp_index_wide(T v) {
    if (v != v) return 0;
    if (v >= static_cast<T>(2147483647.0)) return 2147483647;
    if (v <= static_cast<T>(-2147483648.0)) return -2147483647 - 1;
    return static_cast<int>(v);
}

// SAFE_DIV(a, b) = (b == 0) ? a / 1e-8 : a / b, evaluated as the reference's macro is:
// the conditional has type double (EPS is a double literal), the a/b arm is a T division.
template <typename T>
__device__ __forceinline__ double safe_div(T a, T b) {
    if (b == static_cast<T>(0)) return static_cast<double>(a) / 1e-8;
    return static_cast<double>(static_cast<T>(a / b));
}

__device__ __forceinline__ void atomic_add(float* p, float v) { unsafeAtomicAdd(p, v); }
__device__ __forceinline__ void atomic_add(double* p, double v) { unsafeAtomicAdd(p, v); }

// LDS accumulation goes through DOUBLE cells: ds_add_f64 ~9 clk per wave, ds_add_f32 ~190 on gfx950.
template <typename T>
__device__ __forceinline__ void lds_add(double* cell, T v) {
    __hip_atomic_fetch_add(cell, static_cast<double>(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// ---- buffer addressing -------------------------------------------------------------------
// Division of a 32-bit unsigned by a run-time divisor that is FIXED for a launch (a plane size, a row length): the host works out a
// multiplier and two shifts (Granlund / Montgomery round-up method, exact for every 32-bit dividend and divisor >= 1), the device
// pays a v_mul_hi and four cheap instructions instead of the ~25 of the compiler's generic sequence.
struct FastDiv {
    unsigned m, s1, s2, d;
};
inline FastDiv make_fast_div(unsigned d) {
    FastDiv f;
    f.d = d;
    unsigned l = 0;
    while ((1ull << l) < d) ++l;                     // ceil(log2 d)
    f.m = static_cast<unsigned>(((1ull << 32) * ((1ull << l) - d)) / d + 1);
    f.s1 = l < 1 ? l : 1;
    f.s2 = l < 1 ? 0 : l - 1;
    return f;
}
__device__ __forceinline__ unsigned fast_div(unsigned n, const FastDiv& f) {
    const unsigned t = __umulhi(f.m, n);
    return (t + ((n - t) >> f.s1)) >> f.s2;
}

// A tensor plane is addressed as (wave-uniform 128-bit buffer resource in SGPRs) + (32-bit
// per-lane BYTE offset in one VGPR): `buffer_load_dword v, v_off, s[rsrc], 0 offen`.  This is the
// CDNA way to do "uniform base + per-lane gather": no 64-bit VGPR address per tap, and the
// hardware range check (offset >= num_records reads 0 / drops the store) comes for free.
using rsrc_t = __amdgpu_buffer_rsrc_t;
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x3 __attribute__((ext_vector_type(3)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ rsrc_t make_rsrc(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, static_cast<int>(bytes), 0x00020000);
}

__device__ __forceinline__ float buf_load(rsrc_t r, unsigned boff, float) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, boff, 0, 0));
}
__device__ __forceinline__ double buf_load(rsrc_t r, unsigned boff, double) {
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, boff, 0, 0));
}
template <typename T>
__device__ __forceinline__ T buf_ld(rsrc_t r, unsigned boff) {
    return buf_load(r, boff, T());
}

// N consecutive dwords starting at dword `POS` of w[], as the widest instructions available.
// AUX = cache-policy bits (0 = default, 2 = nt: a stream that is read once).
template <int N, int POS = 0, int AUX = 0>
__device__ __forceinline__ void buf_load_dwords(rsrc_t r, unsigned boff, unsigned* w) {
    if constexpr (N - POS >= 4) {
        u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, boff + 4u * POS, 0, AUX);
        w[POS] = v.x; w[POS + 1] = v.y; w[POS + 2] = v.z; w[POS + 3] = v.w;
        buf_load_dwords<N, POS + 4, AUX>(r, boff, w);
    } else if constexpr (N - POS == 3) {
        u32x3 v = __builtin_amdgcn_raw_buffer_load_b96(r, boff + 4u * POS, 0, AUX);
        w[POS] = v.x; w[POS + 1] = v.y; w[POS + 2] = v.z;
    } else if constexpr (N - POS == 2) {
        u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r, boff + 4u * POS, 0, AUX);
        w[POS] = v.x; w[POS + 1] = v.y;
    } else if constexpr (N - POS == 1) {
        w[POS] = __builtin_amdgcn_raw_buffer_load_b32(r, boff + 4u * POS, 0, AUX);
    }
}
// AUX = cache-policy bits of the instruction (0 = default, 2 = nt: streaming store that is not kept in L2)
template <int N, int POS = 0, int AUX = 0>
__device__ __forceinline__ void buf_store_dwords(rsrc_t r, unsigned boff, const unsigned* w) {
    if constexpr (N - POS >= 4) {
        u32x4 v = {w[POS], w[POS + 1], w[POS + 2], w[POS + 3]};
        __builtin_amdgcn_raw_buffer_store_b128(v, r, boff + 4u * POS, 0, AUX);
        buf_store_dwords<N, POS + 4, AUX>(r, boff, w);
    } else if constexpr (N - POS == 3) {
        u32x3 v = {w[POS], w[POS + 1], w[POS + 2]};
        __builtin_amdgcn_raw_buffer_store_b96(v, r, boff + 4u * POS, 0, AUX);
    } else if constexpr (N - POS == 2) {
        u32x2 v = {w[POS], w[POS + 1]};
        __builtin_amdgcn_raw_buffer_store_b64(v, r, boff + 4u * POS, 0, AUX);
    } else if constexpr (N - POS == 1) {
        __builtin_amdgcn_raw_buffer_store_b32(w[POS], r, boff + 4u * POS, 0, AUX);
    }
}

// K consecutive elements of T (a row of one k x k window).
template <typename T, int K>
struct ElemRow {
    union {
        T v[K];
        unsigned w[K * sizeof(T) / 4];
    };
};
template <typename T, int K>
__device__ __forceinline__ void buf_load_row(rsrc_t r, unsigned boff, ElemRow<T, K>& row) {
    buf_load_dwords<K * sizeof(T) / 4>(r, boff, row.w);
}
template <typename T, int K>
__device__ __forceinline__ void buf_load_row_nt(rsrc_t r, unsigned boff, ElemRow<T, K>& row) {
    buf_load_dwords<K * sizeof(T) / 4, 0, 2>(r, boff, row.w);
}
template <typename T, int K>
__device__ __forceinline__ void buf_store_row(rsrc_t r, unsigned boff, const ElemRow<T, K>& row) {
    buf_store_dwords<K * sizeof(T) / 4>(r, boff, row.w);
}
template <typename T, int K>
__device__ __forceinline__ void buf_store_row_nt(rsrc_t r, unsigned boff, const ElemRow<T, K>& row) {
    buf_store_dwords<K * sizeof(T) / 4, 0, 2>(r, boff, row.w);
}

struct TileCoord {
    int b, slab, yf, xf;
};

__device__ __forceinline__ TileCoord decode_tile(int tiles_x, int tiles_y, int cslabs, int remap) {
    unsigned t = xcd_remap(blockIdx.x, gridDim.x, remap);
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

template <typename T>
__device__ __forceinline__ void atomic_add_off(T* base, unsigned boff, T v) {
    atomic_add(reinterpret_cast<T*>(reinterpret_cast<char*>(base) + boff), v);
}

template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
template <typename T>
__device__ __forceinline__ T wave_min(T v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        T u = __shfl_xor(v, o, 64);
        v = u < v ? u : v;
    }
    return v;
}
template <typename T>
__device__ __forceinline__ T wave_max(T v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        T u = __shfl_xor(v, o, 64);
        v = u > v ? u : v;
    }
    return v;
}

#endif  // __HIPCC__

}  // namespace ffwm
