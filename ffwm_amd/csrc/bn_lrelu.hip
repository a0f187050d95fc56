// bn_lrelu.hip -- training-mode BatchNorm2d fused with the LeakyReLU that follows it.
//
// Reference: every conv block of FlowNet (models/base_networks.py:12-31: conv -> BatchNorm2d -> LeakyReLU(0.2)),
// of FFWM's encoder / decoder / residual blocks (:207-264) and of the discriminators (:381-413) is
// nn.BatchNorm2d in training mode followed by nn.LeakyReLU: ~110 pairs per train step, each two kernels forward
// (statistics + normalise, then the activation's read-modify-write) and two backward.  Here a pair is ONE kernel
// per direction:
//   forward : per channel, mean and biased variance over (B, H, W) -- double accumulators, one pass --, running
//             statistics updated in place exactly like F.batch_norm (momentum, unbiased variance), then
//             y = lrelu(gamma (x - mean) invstd + beta); the second read of x comes from L2.
//   backward: the activation's mask is recomputed from x (pre = gamma xhat + beta), so the forward output is not
//             needed: g = dy * (pre > 0 ? 1 : slope); sum(g), sum(g xhat) per channel; then
//             dx = gamma invstd (g - mean(g) - xhat mean(g xhat)), dgamma = sum(g xhat), dbeta = sum(g).
// Decomposition: a workgroup per channel streams that channel's B planes (float4, coalesced); channels with few
// elements (the 8x8 ... 2x2 layers with up to 1024 channels) take one WAVE per channel instead, four to a
// workgroup.  A channel's statistics never leave the workgroup: no atomics, no second launch.
#include "common.hpp"

namespace ffwm {
namespace {

constexpr int kBnThreads = 1024;

struct BnGeo {
    int B, C, HW;
    float eps, momentum, slope;
    // a channel split over S workgroups (blockIdx.y): phase 1 adds the slice's two sums to scratch[2 c], [2 c + 1]
    // (doubles, zero-filled by the caller), phase 2 reads them and applies; phase 0 = one workgroup does both
    int S, phase;
    int act;                 // residual variant only: 0 = LeakyReLU(slope), 1 = sigmoid
};

__device__ __forceinline__ float lrelu(float v, float slope) { return v > 0.f ? v : v * slope; }
__device__ __forceinline__ float act_of(float v, float slope, int act) { return act == 1 ? 1.f / (1.f + __expf(-v)) : lrelu(v, slope); }
// d(act)/d(pre-activation) from the OUTPUT: LeakyReLU keeps the sign (slope > 0), sigmoid' = y (1 - y)
__device__ __forceinline__ float dact_of(float y, float slope, int act) { return act == 1 ? y * (1.f - y) : (y > 0.f ? 1.f : slope); }

// sum of two doubles over the group (a whole 1024-thread block through LDS, or one wave)
template <int THREADS>
__device__ __forceinline__ void group_sum2(double& a, double& b, double* red) {
    a = wave_sum(a);
    b = wave_sum(b);
    if constexpr (THREADS > kWave) {
        constexpr int NW = THREADS / kWave;
        const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
        __syncthreads();                       // red may still be read from the previous use
        if (lane == 0) {
            red[wave] = a;
            red[NW + wave] = b;
        }
        __syncthreads();
        double sa = 0, sb = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            sa += red[w];
            sb += red[NW + w];
        }
        a = sa;
        b = sb;
    }
}

// THREADS = threads that share one channel (1024: block per channel; 64: wave per channel)
// RES: the tail of a residual block (base_networks.py:207-233) in the same pass: y = act(BatchNorm(x) + res + rbias[c]) with res the
// shortcut convolution's output WITHOUT its bias (the library's GEMM path would add it in a separate read-modify-write pass).
template <int THREADS, bool RES = false>
__global__ void __launch_bounds__(kBnThreads)
bn_lrelu_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                    float* __restrict__ run_mean, float* __restrict__ run_var, float* __restrict__ y,
                    float* __restrict__ save_mean, float* __restrict__ save_invstd, BnGeo g, double* __restrict__ scratch,
                    const float* __restrict__ res = nullptr, const float* __restrict__ rbias = nullptr) {
    __shared__ double red[2 * (kBnThreads / kWave)];
    constexpr int CPB = kBnThreads / THREADS;              // channels per block
    const int sub = threadIdx.x / THREADS, t = threadIdx.x % THREADS;
    const int c = blockIdx.x * CPB + sub;
    const bool live = c < g.C;
    const int cc = live ? c : g.C - 1;                     // dead sub-groups shadow a valid channel, store nothing
    const size_t cstride = static_cast<size_t>(g.C) * g.HW;
    const float* xc = x + static_cast<size_t>(cc) * g.HW;
    const int hw4 = (g.HW % 4 == 0) ? g.HW / 4 : 0;
    double s = 0, ss = 0;
    // flat index j over (plane b, float4 i): four independent loads in flight per thread
    const int all4 = g.B * hw4;
    const int chunk = (all4 + g.S - 1) / g.S;
    const int first4 = blockIdx.y * chunk;                 // this workgroup's slice of the flat (plane, float4) index
    const int total4 = min(all4, first4 + chunk);
    auto addr4 = [&](const float* base, int j) {
        const int b = j / hw4, i = j - b * hw4;
        return reinterpret_cast<const float4*>(base + b * cstride) + i;
    };
    if (g.phase == 2) {
        // statistics already reduced over the slices
    } else if (hw4) {
        for (int j0 = first4 + t; j0 < total4; j0 += 4 * THREADS) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = j0 + u * THREADS;
                v[u] = j < total4 ? *addr4(xc, j) : float4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                s += (static_cast<double>(v[u].x) + static_cast<double>(v[u].y)) + (static_cast<double>(v[u].z) + static_cast<double>(v[u].w));
                ss += (static_cast<double>(v[u].x) * v[u].x + static_cast<double>(v[u].y) * v[u].y) +
                      (static_cast<double>(v[u].z) * v[u].z + static_cast<double>(v[u].w) * v[u].w);
            }
        }
    } else {
        for (int b = 0; b < g.B; ++b) {
            const float* p = xc + b * cstride;
            for (int i = t; i < g.HW; i += THREADS) {
                const double v = p[i];
                s += v;
                ss += v * v;
            }
        }
    }
    if (g.phase != 2) group_sum2<THREADS>(s, ss, red);
    if (g.phase == 1) {
        if (live && t == 0) {
            atomic_add(scratch + 2 * c, s);
            atomic_add(scratch + 2 * c + 1, ss);
        }
        return;
    }
    if (g.phase == 2) {
        // the channel's sums over the slices; the LAST of its S workgroups to have read them clears the two cells and the arrival
        // counter behind the sums (scratch[2 C ...] as 32-bit counters), so the buffer is zero again for the next call: a
        // caller keeps ONE zero-filled scratch instead of filling a fresh one per call (60 fill launches per train step)
        if (threadIdx.x == 0) {
            red[0] = scratch[2 * cc];
            red[1] = scratch[2 * cc + 1];
        }
        __syncthreads();
        s = red[0];
        ss = red[1];
        if (threadIdx.x == 0) {
            unsigned* cnt = reinterpret_cast<unsigned*>(scratch + 2 * g.C) + cc;
            if (atomicAdd(cnt, 1u) == static_cast<unsigned>(g.S) - 1u) {
                scratch[2 * cc] = 0.0;
                scratch[2 * cc + 1] = 0.0;
                *cnt = 0u;
            }
        }
    }
    const double n = static_cast<double>(g.B) * g.HW;
    const double mean = s / n;
    double var = ss / n - mean * mean;
    var = var > 0 ? var : 0;
    const float invstd = static_cast<float>(1.0 / sqrt(var + static_cast<double>(g.eps)));
    const float meanf = static_cast<float>(mean);
    if (live && t == 0 && blockIdx.y == 0) {
        save_mean[c] = meanf;
        save_invstd[c] = invstd;
        if (run_mean) {        // F.batch_norm: running = (1 - momentum) running + momentum batch; variance unbiased
            const double unbiased = n > 1 ? var * n / (n - 1) : var;
            run_mean[c] = (1.f - g.momentum) * run_mean[c] + g.momentum * meanf;
            run_var[c] = (1.f - g.momentum) * run_var[c] + g.momentum * static_cast<float>(unbiased);
        }
    }
    const float ga = gamma ? gamma[cc] : 1.f, be = beta ? beta[cc] : 0.f;
    const float sc = ga * invstd;
    float sh = be - meanf * ga * invstd;
    if (!live) return;
    float* yc = y + static_cast<size_t>(c) * g.HW;
    const float* rc = RES ? res + static_cast<size_t>(c) * g.HW : nullptr;
    if (RES && rbias) sh += rbias[c];
    if (hw4) {
        for (int j0 = first4 + t; j0 < total4; j0 += 4 * THREADS) {
            float4 v[4], r[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = j0 + u * THREADS;
                v[u] = j < total4 ? *addr4(xc, j) : float4{0.f, 0.f, 0.f, 0.f};
                if (RES) r[u] = j < total4 ? *addr4(rc, j) : float4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = j0 + u * THREADS;
                if (j < total4) {
                    float4 o;
                    if (RES) {
                        o.x = act_of(v[u].x * sc + sh + r[u].x, g.slope, g.act);
                        o.y = act_of(v[u].y * sc + sh + r[u].y, g.slope, g.act);
                        o.z = act_of(v[u].z * sc + sh + r[u].z, g.slope, g.act);
                        o.w = act_of(v[u].w * sc + sh + r[u].w, g.slope, g.act);
                    } else {
                        o.x = lrelu(v[u].x * sc + sh, g.slope);
                        o.y = lrelu(v[u].y * sc + sh, g.slope);
                        o.z = lrelu(v[u].z * sc + sh, g.slope);
                        o.w = lrelu(v[u].w * sc + sh, g.slope);
                    }
                    *const_cast<float4*>(addr4(yc, j)) = o;
                }
            }
        }
    } else {
        for (int b = 0; b < g.B; ++b) {
            const float* p = xc + b * cstride;
            float* q = yc + b * cstride;
            if (RES) {
                const float* rr = rc + b * cstride;
                for (int i = t; i < g.HW; i += THREADS) q[i] = act_of(p[i] * sc + sh + rr[i], g.slope, g.act);
            } else {
                for (int i = t; i < g.HW; i += THREADS) q[i] = lrelu(p[i] * sc + sh, g.slope);
            }
        }
    }
}

// RES: backward of y = act(BatchNorm(x) + res + rbias): g = dy * act'(.) from the saved OUTPUT `out` (not recomputed: the
// pre-activation would need res again), written out as d(res) (`dz`; its per-channel sum, dbeta, is d(rbias) as well), then the
// BatchNorm backward on g as before.
template <int THREADS, bool RES = false>
__global__ void __launch_bounds__(kBnThreads)
bn_lrelu_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ gamma,
                    const float* __restrict__ beta, const float* __restrict__ save_mean,
                    const float* __restrict__ save_invstd, float* __restrict__ dx, float* __restrict__ dgamma,
                    float* __restrict__ dbeta, BnGeo g, double* __restrict__ scratch,
                    const float* __restrict__ out = nullptr, float* __restrict__ dz = nullptr) {
    __shared__ double red[2 * (kBnThreads / kWave)];
    constexpr int CPB = kBnThreads / THREADS;
    const int sub = threadIdx.x / THREADS, t = threadIdx.x % THREADS;
    const int c = blockIdx.x * CPB + sub;
    const bool live = c < g.C;
    const int cc = live ? c : g.C - 1;
    const size_t cstride = static_cast<size_t>(g.C) * g.HW;
    const float* xc = x + static_cast<size_t>(cc) * g.HW;
    const float* dyc = dy + static_cast<size_t>(cc) * g.HW;
    const float mean = save_mean[cc], invstd = save_invstd[cc];
    const float ga = gamma ? gamma[cc] : 1.f, be = beta ? beta[cc] : 0.f;
    const int hw4 = (g.HW % 4 == 0) ? g.HW / 4 : 0;
    const float* oc = RES ? out + static_cast<size_t>(cc) * g.HW : nullptr;
    // g = dy * lrelu'(gamma xhat + beta); RES: ov = the forward output at this element, g = dy * act'(.) from it
    auto gval = [&](float xv, float dv, float& xhat, float ov = 0.f) {
        xhat = (xv - mean) * invstd;
        if (RES) return dv * dact_of(ov, g.slope, g.act);
        return (ga * xhat + be) > 0.f ? dv : dv * g.slope;
    };
    double s = 0, sx = 0;
    const int all4 = g.B * hw4;
    const int chunk = (all4 + g.S - 1) / g.S;
    const int first4 = blockIdx.y * chunk;                 // this workgroup's slice of the flat (plane, float4) index
    const int total4 = min(all4, first4 + chunk);
    auto addr4 = [&](const float* base, int j) {
        const int b = j / hw4, i = j - b * hw4;
        return reinterpret_cast<const float4*>(base + b * cstride) + i;
    };
    if (g.phase == 2) {
        // sums already reduced over the slices
    } else if (hw4) {
        for (int j0 = first4 + t; j0 < total4; j0 += 2 * THREADS) {
            float4 v[2], e[2], o4[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int j = j0 + u * THREADS;
                v[u] = j < total4 ? *addr4(xc, j) : float4{0.f, 0.f, 0.f, 0.f};
                e[u] = j < total4 ? *addr4(dyc, j) : float4{0.f, 0.f, 0.f, 0.f};
                o4[u] = (RES && j < total4) ? *addr4(oc, j) : float4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                float xh;
                float gv = gval(v[u].x, e[u].x, xh, o4[u].x); s += gv; sx += static_cast<double>(gv) * xh;
                gv = gval(v[u].y, e[u].y, xh, o4[u].y); s += gv; sx += static_cast<double>(gv) * xh;
                gv = gval(v[u].z, e[u].z, xh, o4[u].z); s += gv; sx += static_cast<double>(gv) * xh;
                gv = gval(v[u].w, e[u].w, xh, o4[u].w); s += gv; sx += static_cast<double>(gv) * xh;
            }
        }
    } else {
        for (int b = 0; b < g.B; ++b) {
            const float* p = xc + b * cstride;
            const float* d = dyc + b * cstride;
            const float* oo = RES ? oc + b * cstride : p;
            for (int i = t; i < g.HW; i += THREADS) {
                float xh;
                const float gv = gval(p[i], d[i], xh, oo[i]);
                s += gv;
                sx += static_cast<double>(gv) * xh;
            }
        }
    }
    if (g.phase != 2) group_sum2<THREADS>(s, sx, red);
    if (g.phase == 1) {
        if (live && t == 0) {
            atomic_add(scratch + 2 * c, s);
            atomic_add(scratch + 2 * c + 1, sx);
        }
        return;
    }
    if (g.phase == 2) {
        if (threadIdx.x == 0) {               // as in the forward: read, then the last reader clears for the next call
            red[0] = scratch[2 * cc];
            red[1] = scratch[2 * cc + 1];
        }
        __syncthreads();
        s = red[0];
        sx = red[1];
        if (threadIdx.x == 0) {
            unsigned* cnt = reinterpret_cast<unsigned*>(scratch + 2 * g.C) + cc;
            if (atomicAdd(cnt, 1u) == static_cast<unsigned>(g.S) - 1u) {
                scratch[2 * cc] = 0.0;
                scratch[2 * cc + 1] = 0.0;
                *cnt = 0u;
            }
        }
    }
    if (live && t == 0 && blockIdx.y == 0) {
        if (dgamma) dgamma[c] = static_cast<float>(sx);
        if (dbeta) dbeta[c] = static_cast<float>(s);
    }
    if (!live || !(dx || (RES && dz))) return;
    const double n = static_cast<double>(g.B) * g.HW;
    const float mg = static_cast<float>(s / n), mgx = static_cast<float>(sx / n);
    const float k = ga * invstd;
    float* dxc = dx ? dx + static_cast<size_t>(c) * g.HW : nullptr;
    float* dzc = (RES && dz) ? dz + static_cast<size_t>(c) * g.HW : nullptr;
    if (hw4) {
        for (int j0 = first4 + t; j0 < total4; j0 += 2 * THREADS) {
            float4 v[2], e[2], o4[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int j = j0 + u * THREADS;
                v[u] = j < total4 ? *addr4(xc, j) : float4{0.f, 0.f, 0.f, 0.f};
                e[u] = j < total4 ? *addr4(dyc, j) : float4{0.f, 0.f, 0.f, 0.f};
                o4[u] = (RES && j < total4) ? *addr4(oc, j) : float4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int j = j0 + u * THREADS;
                if (j < total4) {
                    float4 o, z;
                    float xh;
                    float gv = gval(v[u].x, e[u].x, xh, o4[u].x); o.x = k * (gv - mg - xh * mgx); z.x = gv;
                    gv = gval(v[u].y, e[u].y, xh, o4[u].y); o.y = k * (gv - mg - xh * mgx); z.y = gv;
                    gv = gval(v[u].z, e[u].z, xh, o4[u].z); o.z = k * (gv - mg - xh * mgx); z.z = gv;
                    gv = gval(v[u].w, e[u].w, xh, o4[u].w); o.w = k * (gv - mg - xh * mgx); z.w = gv;
                    if (dxc) *const_cast<float4*>(addr4(dxc, j)) = o;
                    if (RES && dzc) *const_cast<float4*>(addr4(dzc, j)) = z;
                }
            }
        }
    } else {
        for (int b = 0; b < g.B; ++b) {
            const float* p = xc + b * cstride;
            const float* d = dyc + b * cstride;
            const float* oo = RES ? oc + b * cstride : p;
            for (int i = t; i < g.HW; i += THREADS) {
                float xh;
                const float gv = gval(p[i], d[i], xh, oo[i]);
                if (dxc) dxc[b * cstride + i] = k * (gv - mg - xh * mgx);
                if (RES && dzc) dzc[b * cstride + i] = gv;
            }
        }
    }
}

int check_bn(const char* fn, int64_t B, int64_t C, int64_t HW, int dtype) {
    FFWM_REQUIRE(dtype == FFWM_F32, FFWM_ERR_DTYPE, "%s: float32 only", fn);
    FFWM_REQUIRE(B > 0 && C > 0 && HW > 0 && B * HW > 1, FFWM_ERR_ARG, "%s: need B, C, H*W > 0 and more than one value per channel", fn);
    FFWM_REQUIRE(B < (1LL << 20) && C < (1LL << 24) && HW < (1LL << 30), FFWM_ERR_SIZE, "%s: tensor too large", fn);
    return FFWM_OK;
}

}  // namespace
}  // namespace ffwm

using namespace ffwm;

// slices per channel when the caller provides scratch: enough workgroups for the chip, slices of >= 16 K elements
static int bn_slices(int64_t B, int64_t C, int64_t HW, const void* scratch) {
    if (!scratch || HW % 4 != 0 || B * HW < 2048) return 1;
    int64_t S = (1024 + C - 1) / C;
    const int64_t cap = B * HW / 16384;
    if (S > cap) S = cap;
    if (S > 32) S = 32;
    return S < 2 ? 1 : static_cast<int>(S);
}

extern "C" int ffwm_bn_lrelu_forward(const void* x, const void* weight, const void* bias, void* running_mean,
                                     void* running_var, void* y, void* save_mean, void* save_invstd, void* scratch, int64_t B,
                                     int64_t C, int64_t HW, double eps, double momentum, double negative_slope, int dtype,
                                     void* stream) {
    const char* fn = "ffwm_bn_lrelu_forward";
    if (int rc = check_bn(fn, B, C, HW, dtype)) return rc;
    FFWM_REQUIRE(x && y && save_mean && save_invstd, FFWM_ERR_ARG, "%s: NULL tensor pointer", fn);
    FFWM_REQUIRE((running_mean == nullptr) == (running_var == nullptr), FFWM_ERR_ARG, "%s: running_mean and running_var go together", fn);
    BnGeo g{(int)B, (int)C, (int)HW, (float)eps, (float)momentum, (float)negative_slope, 1, 0};
    hipStream_t st = static_cast<hipStream_t>(stream);
    LaunchScope ls("bn_lrelu_fwd", st, 4.0 * 2.0 * B * C * HW);
#define FFWM_BN_FWD(THR, GRID)                                                                                              \
    hipLaunchKernelGGL((bn_lrelu_fwd_kernel<THR>), GRID, dim3(kBnThreads), 0, st, (const float*)x, (const float*)weight,    \
                       (const float*)bias, (float*)running_mean, (float*)running_var, (float*)y, (float*)save_mean,         \
                       (float*)save_invstd, g, (double*)scratch)
    const int S = bn_slices(B, C, HW, scratch);
    if (S > 1) {
        g.S = S;
        g.phase = 1;
        FFWM_BN_FWD(kBnThreads, dim3((unsigned)C, (unsigned)S));
        g.phase = 2;
        FFWM_BN_FWD(kBnThreads, dim3((unsigned)C, (unsigned)S));
    } else if (B * HW >= 2048) {
        FFWM_BN_FWD(kBnThreads, dim3((unsigned)C));
    } else {
        FFWM_BN_FWD(kWave, dim3((unsigned)((C + 15) / 16)));
    }
#undef FFWM_BN_FWD
    return check_launch(fn);
}

extern "C" int ffwm_bn_lrelu_backward(const void* x, const void* grad_out, const void* weight, const void* bias,
                                      const void* save_mean, const void* save_invstd, void* grad_x, void* grad_weight,
                                      void* grad_bias, void* scratch, int64_t B, int64_t C, int64_t HW, double negative_slope,
                                      int dtype, void* stream) {
    const char* fn = "ffwm_bn_lrelu_backward";
    if (int rc = check_bn(fn, B, C, HW, dtype)) return rc;
    FFWM_REQUIRE(x && grad_out && save_mean && save_invstd, FFWM_ERR_ARG, "%s: NULL tensor pointer", fn);
    BnGeo g{(int)B, (int)C, (int)HW, 0.f, 0.f, (float)negative_slope, 1, 0};
    hipStream_t st = static_cast<hipStream_t>(stream);
    LaunchScope ls("bn_lrelu_bwd", st, 4.0 * 3.0 * B * C * HW);
#define FFWM_BN_BWD(THR, GRID)                                                                                              \
    hipLaunchKernelGGL((bn_lrelu_bwd_kernel<THR>), GRID, dim3(kBnThreads), 0, st, (const float*)x, (const float*)grad_out,  \
                       (const float*)weight, (const float*)bias, (const float*)save_mean, (const float*)save_invstd,        \
                       (float*)grad_x, (float*)grad_weight, (float*)grad_bias, g, (double*)scratch)
    const int S = bn_slices(B, C, HW, scratch);
    if (S > 1) {
        g.S = S;
        g.phase = 1;
        FFWM_BN_BWD(kBnThreads, dim3((unsigned)C, (unsigned)S));
        g.phase = 2;
        FFWM_BN_BWD(kBnThreads, dim3((unsigned)C, (unsigned)S));
    } else if (B * HW >= 2048) {
        FFWM_BN_BWD(kBnThreads, dim3((unsigned)C));
    } else {
        FFWM_BN_BWD(kWave, dim3((unsigned)((C + 15) / 16)));
    }
#undef FFWM_BN_BWD
    return check_launch(fn);
}

// ---- the tail of a residual block: y = act(BatchNorm(x) + res + rbias[c]) (base_networks.py:207-233: activ(blocks(x) + input(x)) with
// blocks ending in a BatchNorm2d and input = a 1x1 convolution whose bias is rbias); act 0 = LeakyReLU(negative_slope), 1 = sigmoid.
extern "C" int ffwm_bn_res_act_forward(const void* x, const void* weight, const void* bias, void* running_mean, void* running_var,
                                       const void* res, const void* rbias, void* y, void* save_mean, void* save_invstd, void* scratch,
                                       int64_t B, int64_t C, int64_t HW, double eps, double momentum, double negative_slope, int act,
                                       int dtype, void* stream) {
    const char* fn = "ffwm_bn_res_act_forward";
    if (int rc = check_bn(fn, B, C, HW, dtype)) return rc;
    FFWM_REQUIRE(x && y && res && save_mean && save_invstd, FFWM_ERR_ARG, "%s: NULL tensor pointer", fn);
    FFWM_REQUIRE(act == 0 || act == 1, FFWM_ERR_ARG, "%s: act must be 0 (leaky relu) or 1 (sigmoid)", fn);
    FFWM_REQUIRE((running_mean == nullptr) == (running_var == nullptr), FFWM_ERR_ARG, "%s: running_mean and running_var go together", fn);
    BnGeo g{(int)B, (int)C, (int)HW, (float)eps, (float)momentum, (float)negative_slope, 1, 0, act};
    hipStream_t st = static_cast<hipStream_t>(stream);
    LaunchScope ls("bn_res_act_fwd", st, 4.0 * 3.0 * B * C * HW);
#define FFWM_BNR_FWD(THR, GRID)                                                                                                  \
    hipLaunchKernelGGL((bn_lrelu_fwd_kernel<THR, true>), GRID, dim3(kBnThreads), 0, st, (const float*)x, (const float*)weight,     \
                       (const float*)bias, (float*)running_mean, (float*)running_var, (float*)y, (float*)save_mean,               \
                       (float*)save_invstd, g, (double*)scratch, (const float*)res, (const float*)rbias)
    const int S = bn_slices(B, C, HW, scratch);
    if (S > 1) {
        g.S = S;
        g.phase = 1;
        FFWM_BNR_FWD(kBnThreads, dim3((unsigned)C, (unsigned)S));
        g.phase = 2;
        FFWM_BNR_FWD(kBnThreads, dim3((unsigned)C, (unsigned)S));
    } else if (B * HW >= 2048) {
        FFWM_BNR_FWD(kBnThreads, dim3((unsigned)C));
    } else {
        FFWM_BNR_FWD(kWave, dim3((unsigned)((C + 15) / 16)));
    }
#undef FFWM_BNR_FWD
    return check_launch(fn);
}

// grad_res = grad_out * act'(.) (from the saved output y), grad_x = the BatchNorm backward of it, grad_weight / grad_bias the
// BatchNorm's (grad_bias is the gradient of rbias as well).  grad_x or grad_res may be NULL.
extern "C" int ffwm_bn_res_act_backward(const void* x, const void* y, const void* grad_out, const void* weight, const void* save_mean,
                                        const void* save_invstd, void* grad_x, void* grad_res, void* grad_weight, void* grad_bias,
                                        void* scratch, int64_t B, int64_t C, int64_t HW, double negative_slope, int act, int dtype,
                                        void* stream) {
    const char* fn = "ffwm_bn_res_act_backward";
    if (int rc = check_bn(fn, B, C, HW, dtype)) return rc;
    FFWM_REQUIRE(x && y && grad_out && save_mean && save_invstd, FFWM_ERR_ARG, "%s: NULL tensor pointer", fn);
    FFWM_REQUIRE(act == 0 || act == 1, FFWM_ERR_ARG, "%s: act must be 0 (leaky relu) or 1 (sigmoid)", fn);
    BnGeo g{(int)B, (int)C, (int)HW, 0.f, 0.f, (float)negative_slope, 1, 0, act};
    hipStream_t st = static_cast<hipStream_t>(stream);
    LaunchScope ls("bn_res_act_bwd", st, 4.0 * 5.0 * B * C * HW);
#define FFWM_BNR_BWD(THR, GRID)                                                                                                  \
    hipLaunchKernelGGL((bn_lrelu_bwd_kernel<THR, true>), GRID, dim3(kBnThreads), 0, st, (const float*)x, (const float*)grad_out,   \
                       (const float*)weight, (const float*)nullptr, (const float*)save_mean, (const float*)save_invstd,           \
                       (float*)grad_x, (float*)grad_weight, (float*)grad_bias, g, (double*)scratch, (const float*)y, (float*)grad_res)
    const int S = bn_slices(B, C, HW, scratch);
    if (S > 1) {
        g.S = S;
        g.phase = 1;
        FFWM_BNR_BWD(kBnThreads, dim3((unsigned)C, (unsigned)S));
        g.phase = 2;
        FFWM_BNR_BWD(kBnThreads, dim3((unsigned)C, (unsigned)S));
    } else if (B * HW >= 2048) {
        FFWM_BNR_BWD(kBnThreads, dim3((unsigned)C));
    } else {
        FFWM_BNR_BWD(kWave, dim3((unsigned)((C + 15) / 16)));
    }
#undef FFWM_BNR_BWD
    return check_launch(fn);
}
