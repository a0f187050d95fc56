// correlation.hip -- max_i <source_i, target_j> over all source positions, for every target position,
// on the matrix cores (fp32-in / fp32-accumulate MFMA).
//
// Reference: PerceptualCorrectness.calculate_loss, /root/reference/models/losses.py:347-353 --
//     correction = torch.bmm(source_norm [B,N,C], target_norm [B,C,N])      # [B, N, N]
//     correction_max, _ = torch.max(correction, dim=1)                      # [B, N]
// with N = h*w of a VGG feature map: 16384 at relu1_1, i.e. a 1 GiB matrix PER SAMPLE that is written,
// read back once for the max and (in the reference) kept for a backward pass nobody needs.
//
// This is a genuine contraction (2 N^2 C flop per sample: 34 GFLOP at relu1_1), so it runs on MFMA;
// the N x N matrix never exists.  A workgroup owns 128 target columns of one sample, one 64-lane wave per
// 32 columns.  The wave keeps its B operands (the 32 target columns, all of K = C) in registers for the
// whole kernel, walks the source rows in tiles of 32 (staged through LDS once per workgroup, double
// buffered), issues C/2 v_mfma_f32_32x32x2_f32 per tile -- 64 cycles each, back to back on its SIMD: the
// f32 MFMA peak from one wave per SIMD -- and folds the 32 x 32 products into 16 running maxima per
// lane.  v_mfma_f32_32x32x2_f32 is an exact k-ordered fp32 fma chain, so the values are those of an
// fp32 GEMM (different summation order than rocBLAS: ~1e-7).
//
// K is permuted so that lanes 0-31 take k in [0, 32) of every 64-chunk and lanes 32-63 take [32, 64):
// every lane then loads 32 CONTIGUOUS floats of its row (the sum over k does not care about the order,
// and A and B use the same permutation).
#include "common.hpp"

namespace ffwm {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kCorRows = 32;          // source rows per tile
constexpr int kCorCols = 128;         // target columns per workgroup (4 waves x 32)

template <int KC>                     // KC = C / 64
__global__ void __launch_bounds__(kBlock)
corr_colmax_kernel(const float* __restrict__ src, const float* __restrict__ tgt, float* __restrict__ out, int N,
                   int col_tiles) {
    constexpr int C = KC * 64;
    constexpr int PITCH = C + 4;      // floats; rows 272 B apart (C = 64): conflict-free ds_read_b128 across 32 rows
    constexpr int F4_PER_THREAD = kCorRows * C / 4 / kBlock;      // float4 loads per thread per tile = 2 KC
    extern __shared__ __attribute__((aligned(16))) float tile_mem[];      // 2 x 32 x PITCH floats
    const int b = blockIdx.x / col_tiles;
    const int j0 = (blockIdx.x % col_tiles) * kCorCols;
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    const int half = lane >> 5, l31 = lane & 31;
    const float* sb = src + static_cast<size_t>(b) * N * C;
    const float* tb = tgt + static_cast<size_t>(b) * C * N;

    // B operands: column j0 + 32 wave + l31 (clamped: out-of-range columns are computed and not stored)
    const int col = min(j0 + wave * 32 + l31, N - 1);
    float breg[KC * 32];
#pragma unroll
    for (int ch = 0; ch < KC; ++ch)
#pragma unroll
        for (int kk = 0; kk < 32; ++kk) breg[ch * 32 + kk] = tb[static_cast<size_t>(ch * 64 + half * 32 + kk) * N + col];

    // cooperative staging: float4 index f of the tile -> row f / (C/4), 4 floats at (f % (C/4)) * 4
    f32x4 stage[F4_PER_THREAD];
    auto fetch = [&](int i0) {
#pragma unroll
        for (int q = 0; q < F4_PER_THREAD; ++q) {
            const int f = threadIdx.x + q * kBlock;
            const int r = f / (C / 4), c4 = (f - r * (C / 4)) * 4;
            const int row = min(i0 + r, N - 1);
            stage[q] = *reinterpret_cast<const f32x4*>(sb + static_cast<size_t>(row) * C + c4);
        }
    };
    auto commit = [&](float* buf) {
#pragma unroll
        for (int q = 0; q < F4_PER_THREAD; ++q) {
            const int f = threadIdx.x + q * kBlock;
            const int r = f / (C / 4), c4 = (f - r * (C / 4)) * 4;
            *reinterpret_cast<f32x4*>(buf + r * PITCH + c4) = stage[q];
        }
    };

    float m[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) m[r] = -3.0e38f;
    const int ntiles = (N + kCorRows - 1) / kCorRows;
    fetch(0);
    commit(tile_mem);
    __syncthreads();
    int p = 0;
    for (int t = 0; t < ntiles; ++t, p ^= 1) {
        if (t + 1 < ntiles) fetch((t + 1) * kCorRows);           // lands during the MFMA chain
        const float* arow = tile_mem + p * (kCorRows * PITCH) + l31 * PITCH + half * 32;
        f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int ch = 0; ch < KC; ++ch) {
            float areg[32];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(arow + ch * 64 + q * 4);
                areg[q * 4] = v.x; areg[q * 4 + 1] = v.y; areg[q * 4 + 2] = v.z; areg[q * 4 + 3] = v.w;
            }
#pragma unroll
            for (int kk = 0; kk < 32; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(areg[kk], breg[ch * 32 + kk], acc, 0, 0, 0);
        }
        // (rows past N in a ragged last tile are clamped copies of row N-1: harmless for a max)
#pragma unroll
        for (int r = 0; r < 16; ++r) m[r] = fmaxf(m[r], acc[r]);
        if (t + 1 < ntiles) commit(tile_mem + (p ^ 1) * (kCorRows * PITCH));
        __syncthreads();
    }
    // C/D layout: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5): the two halves hold
    // disjoint rows of the same column
    float best = m[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) best = fmaxf(best, m[r]);
    best = fmaxf(best, __shfl_xor(best, 32, kWave));
    const int jc = j0 + wave * 32 + l31;
    if (half == 0 && jc < N) out[static_cast<size_t>(b) * N + jc] = best;
}

}  // namespace
}  // namespace ffwm

using namespace ffwm;

extern "C" int ffwm_correlation_colmax(const void* source, const void* target, void* out, int64_t B, int64_t N,
                                       int64_t C, int dtype, void* stream) {
    const char* fn = "ffwm_correlation_colmax";
    FFWM_REQUIRE(dtype == FFWM_F32, FFWM_ERR_DTYPE, "%s: float32 only (fp32 MFMA)", fn);
    FFWM_REQUIRE(source && target && out, FFWM_ERR_ARG, "%s: NULL tensor pointer", fn);
    FFWM_REQUIRE(B > 0 && N > 0 && (C == 64 || C == 128 || C == 256), FFWM_ERR_ARG,
                 "%s: need B, N > 0 and C in {64, 128, 256} (VGG relu1_1 / relu2_1 / relu3_1), got B=%lld N=%lld C=%lld", fn,
                 (long long)B, (long long)N, (long long)C);
    const int64_t col_tiles = (N + kCorCols - 1) / kCorCols;
    FFWM_REQUIRE(N < (1LL << 30) && B * col_tiles < (1LL << 31), FFWM_ERR_SIZE, "%s: tensor too large", fn);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const unsigned grid = static_cast<unsigned>(B * col_tiles);
    // "bytes" of this compute-bound kernel: operands read once + result (the roofline that matters is MFMA)
    LaunchScope ls("correlation_colmax", st, 4.0 * B * (2.0 * N * C + N), 2.0 * B * N * N * C);
    const size_t lds = 2 * static_cast<size_t>(kCorRows) * (C + 4) * sizeof(float);
    allow_large_lds(reinterpret_cast<const void*>(corr_colmax_kernel<4>));
    if (C == 64)
        hipLaunchKernelGGL((corr_colmax_kernel<1>), dim3(grid), dim3(kBlock), lds, st, (const float*)source, (const float*)target,
                           (float*)out, (int)N, (int)col_tiles);
    else if (C == 128)
        hipLaunchKernelGGL((corr_colmax_kernel<2>), dim3(grid), dim3(kBlock), lds, st, (const float*)source, (const float*)target,
                           (float*)out, (int)N, (int)col_tiles);
    else
        hipLaunchKernelGGL((corr_colmax_kernel<4>), dim3(grid), dim3(kBlock), lds, st, (const float*)source, (const float*)target,
                           (float*)out, (int)N, (int)col_tiles);
    return check_launch(fn);
}
