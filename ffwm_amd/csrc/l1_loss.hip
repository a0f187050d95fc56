// l1_loss.hip -- the L1 terms of FFWM's generator loss as ONE launch forward and ONE backward (gfx950).
//
// Reference: models/ffwm_model.py:107-139 (backward_G) adds up ~25 terms of the form  w * F.l1_loss(x * m, y * m):
// the multi-scale pixel loss (:112-115), PerceptualLoss (models/losses.py:293-320: five VGG19 feature levels per call,
// three scales + four part crops), the multi-scale illumination loss (MSL1Loss, losses.py:130-157) and the identity loss
// (IdentityLoss, :76-112).  In PyTorch every term is 5-7 element-wise / reduction launches forward (two masked products, a
// difference, abs, mean, a scalar product, a scalar sum) and as many backward: ~250 launches of a few microseconds per
// train step, each followed by a dispatch gap.
//
// Here a "problem" is one term:  out[slot] += scale * sum_i | x[i] * m[mi] - y[i] * m[mi] |  with scale = w / numel and an
// optional mask broadcast over the channels (m [B, 1, H, W] against x [B, C, H, W]; mi = (i / (C H W)) * H W + i % (H W)).
// All problems of a call travel in the kernel arguments; a workgroup finds its problem from its index (as the multi-problem
// warp launches do), reduces 4096 elements (float4 loads, wave shuffles, one LDS hop) and adds ONE float atomically to its
// slot of the zero-filled result vector.  The backward writes d(x) = g[slot] * scale * sign(x m - y m) * m for every
// problem in one launch (y is data or a detached feature in every term of the reference: no gradient).
#include "common.hpp"

namespace ffwm {
namespace {

constexpr int kL1Max = 32;           // problems per launch
constexpr int kL1PerBlock = 4096;    // elements per workgroup: 256 threads x 4 float4

struct L1Problem {
    const float* x;
    const float* y;
    const float* m;       // NULL: no mask
    float* gx;            // backward only
    long long n;          // elements of x
    int chw, hw;          // mask broadcast: C*H*W and H*W of x (hw == chw: the mask has x's shape)
    float scale;
    int slot;
    unsigned begin, nblk;
};
struct L1Table {
    int n;
    L1Problem p[kL1Max];
};

__device__ __forceinline__ float l1_mask(const L1Problem& q, long long i) {
    if (!q.m) return 1.f;
    const long long b = i / q.chw;
    const int r = static_cast<int>(i - b * q.chw);
    return q.m[b * q.hw + r % q.hw];
}

template <bool BWD>
__global__ void __launch_bounds__(kBlock)
l1_multi_kernel(const L1Table tab, float* __restrict__ out, const float* __restrict__ gout) {
    __shared__ float red[kBlock / kWave];
    int pi = 0;
#pragma unroll 1
    for (int k = 1; k < tab.n; ++k)
        if (blockIdx.x >= tab.p[k].begin) pi = k;
    const L1Problem& q = tab.p[pi];
    const long long base = static_cast<long long>(blockIdx.x - q.begin) * kL1PerBlock;
    const float gs = BWD ? gout[q.slot] * q.scale : 0.f;
    float acc = 0.f;
    // float4 path: x, y (and gx) 16-byte aligned, and a float4 never straddles two mask rows (hw % 4 == 0)
    const bool vec = ((reinterpret_cast<uintptr_t>(q.x) | reinterpret_cast<uintptr_t>(q.y) | reinterpret_cast<uintptr_t>(q.gx)) & 15) == 0 &&
                     (!q.m || ((q.hw & 3) == 0 && (reinterpret_cast<uintptr_t>(q.m) & 15) == 0));
    if (vec) {
#pragma unroll
        for (int u = 0; u < kL1PerBlock / (4 * kBlock); ++u) {
            const long long i = base + (static_cast<long long>(u) * kBlock + threadIdx.x) * 4;
            if (i + 3 < q.n) {
                const float4 a = *reinterpret_cast<const float4*>(q.x + i);
                const float4 b = *reinterpret_cast<const float4*>(q.y + i);
                float4 m = {1.f, 1.f, 1.f, 1.f};
                if (q.m) {
                    const long long bb = i / q.chw;
                    const int r = static_cast<int>(i - bb * q.chw);
                    m = *reinterpret_cast<const float4*>(q.m + bb * q.hw + r % q.hw);
                }
                const float d0 = a.x * m.x - b.x * m.x, d1 = a.y * m.y - b.y * m.y, d2 = a.z * m.z - b.z * m.z, d3 = a.w * m.w - b.w * m.w;
                if constexpr (BWD) {
                    auto sg = [](float d) { return d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f); };
                    *reinterpret_cast<float4*>(q.gx + i) = float4{gs * sg(d0) * m.x, gs * sg(d1) * m.y, gs * sg(d2) * m.z, gs * sg(d3) * m.w};
                } else {
                    acc += (fabsf(d0) + fabsf(d1)) + (fabsf(d2) + fabsf(d3));
                }
            } else {
                for (long long j = i; j < q.n; ++j) {
                    const float m = l1_mask(q, j);
                    const float d = q.x[j] * m - q.y[j] * m;
                    if constexpr (BWD) q.gx[j] = gs * (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) * m;
                    else acc += fabsf(d);
                }
            }
        }
    } else {
        for (int u = 0; u < kL1PerBlock / kBlock; ++u) {
            const long long j = base + static_cast<long long>(u) * kBlock + threadIdx.x;
            if (j < q.n) {
                const float m = l1_mask(q, j);
                const float d = q.x[j] * m - q.y[j] * m;
                if constexpr (BWD) q.gx[j] = gs * (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) * m;
                else acc += fabsf(d);
            }
        }
    }
    if constexpr (!BWD) {
        acc = wave_sum(acc);
        const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
        if (lane == 0) red[wave] = acc;
        __syncthreads();
        if (threadIdx.x == 0) {
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < kBlock / kWave; ++k) s += red[k];
            atomic_add(out + q.slot, s * q.scale);
        }
    }
}

}  // namespace
}  // namespace ffwm

using namespace ffwm;

// problems: host array of n ffwm_l1_problem (include/ffwm_hip.h).  Forward: out[slots] must be ZERO-FILLED (or hold the value to
// add to); backward (grad_out != NULL): every problem's gx is overwritten.
extern "C" int ffwm_l1_multi(const ffwm_l1_problem* problems, int n, void* out, const void* grad_out, int n_slots, int dtype, void* stream) {
    const char* fn = "ffwm_l1_multi";
    FFWM_REQUIRE(dtype == FFWM_F32, FFWM_ERR_DTYPE, "%s: float32 only", fn);
    FFWM_REQUIRE(problems && n > 0 && n_slots > 0, FFWM_ERR_ARG, "%s: empty problem list", fn);
    FFWM_REQUIRE(grad_out || out, FFWM_ERR_ARG, "%s: neither an output vector nor a gradient", fn);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const bool bwd = grad_out != nullptr;
    int done = 0;
    while (done < n) {
        L1Table tab;
        tab.n = 0;
        unsigned blocks = 0;
        double bytes = 0;
        for (; done < n && tab.n < kL1Max; ++done) {
            const ffwm_l1_problem& pr = problems[done];
            FFWM_REQUIRE(pr.x && pr.y && pr.n >= 0 && pr.slot >= 0 && pr.slot < n_slots, FFWM_ERR_ARG, "%s: bad problem %d", fn, done);
            FFWM_REQUIRE(!bwd || pr.grad_x, FFWM_ERR_ARG, "%s: problem %d has no gradient buffer", fn, done);
            FFWM_REQUIRE(!pr.mask || (pr.chw > 0 && pr.hw > 0 && pr.chw % pr.hw == 0 && pr.n % pr.chw == 0), FFWM_ERR_ARG,
                         "%s: problem %d: the mask's H*W must divide C*H*W, which must divide numel", fn, done);
            if (pr.n == 0) continue;
            L1Problem& q = tab.p[tab.n++];
            q.x = static_cast<const float*>(pr.x); q.y = static_cast<const float*>(pr.y); q.m = static_cast<const float*>(pr.mask);
            q.gx = static_cast<float*>(pr.grad_x);
            q.n = pr.n; q.chw = pr.mask ? static_cast<int>(pr.chw) : 1; q.hw = pr.mask ? static_cast<int>(pr.hw) : 1;
            q.scale = static_cast<float>(pr.scale); q.slot = pr.slot;
            q.begin = blocks;
            q.nblk = static_cast<unsigned>((pr.n + kL1PerBlock - 1) / kL1PerBlock);
            blocks += q.nblk;
            bytes += 4.0 * pr.n * (bwd ? 3.0 : 2.0);
        }
        if (tab.n == 0) continue;
        LaunchScope ls(bwd ? "l1_multi_bwd" : "l1_multi_fwd", st, bytes);
        if (bwd)
            hipLaunchKernelGGL((l1_multi_kernel<true>), dim3(blocks), dim3(kBlock), 0, st, tab, static_cast<float*>(out),
                               static_cast<const float*>(grad_out));
        else
            hipLaunchKernelGGL((l1_multi_kernel<false>), dim3(blocks), dim3(kBlock), 0, st, tab, static_cast<float*>(out),
                               static_cast<const float*>(grad_out));
        if (int rc = check_launch(fn)) return rc;
    }
    return FFWM_OK;
}
