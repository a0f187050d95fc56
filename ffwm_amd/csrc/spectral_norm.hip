// spectral_norm.hip -- batched spectral normalisation of the conv weights of netG / netD for gfx950.
//
// The reference wraps every convolution of FFWM (netG) and MSDiscriminator (netD) in
// torch.nn.utils.spectral_norm (/root/reference/models/base_networks.py:5,218-264,381-413).  That
// hook runs, per layer and per forward call, one power iteration and a division:
//     v = normalize(W^T u);  u = normalize(W v);  sigma = u . (W v);  weight = W / sigma
// as ~12 tiny kernels (gemv, norm, clamp, div, dot, copies) forward and ~6 backward.  netG has 52
// such layers and netD is called three times per train step: ~1700 of the step's ~4600 launches.
//
// Here: three launches normalise up to FFWM_SN_MAX_LAYERS layers at once (W^T u | W v | W / sigma,
// each a grid over (layer, chunk) pairs of ALL layers) and two launches do the backward
//     dL/dW = G / sigma - (<G, W> / sigma^2) u v^T          (u, v constants, as in the hook)
// for all of them.  W is [rows, cols] row-major = weight.view(Cout, -1).
#include "common.hpp"

namespace ffwm {
namespace {

// Work decomposition: every phase is ONE launch over (layer, chunk) pairs of all layers; a block finds
// its pair from a prefix table that travels in the kernel arguments.
//   phase 1  v_raw = W^T u            chunk = 64 columns   (4 waves split the rows, LDS combine)
//   phase 2  wv    = W (v_raw/|v_raw|) chunk = 4 rows      (a wave per row, lanes stride the row)
//   phase 3  u = wv/|wv|, sigma = u.wv, weight_sn = W / sigma   chunk = 8192 elements
// |v_raw| and |wv| are recomputed by every block that needs them (<= 2304 / <= 1024 values): cheaper
// than a grid-wide synchronisation.  The first chunk of a layer writes the layer's u, v, sigma.
constexpr int kSnBlock = 256;
constexpr int kSnWaves = kSnBlock / kWave;
constexpr int kColChunk = 64, kRowChunk = 4, kElemChunk = 8192;

struct SnFwdArgs {
    ffwm_sn_layer l[FFWM_SN_MAX_LAYERS];
    int start[FFWM_SN_MAX_LAYERS + 1];      // prefix sum of chunk counts of the current phase
    int n;
};
struct SnBwdArgs {
    ffwm_sn_grad_layer l[FFWM_SN_MAX_LAYERS];
    int start[FFWM_SN_MAX_LAYERS + 1];
    int n;
};

template <typename A>
__device__ __forceinline__ int find_layer(const A& a, int& chunk) {
    const int b = blockIdx.x;
    int k = 0;
    while (k + 1 < a.n && a.start[k + 1] <= b) ++k;     // block-uniform scan, <= 32 steps
    chunk = b - a.start[k];
    return k;
}

template <typename T>
__device__ __forceinline__ T block_sum(T v, T* red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    __syncthreads();                 // red[] may still be read from a previous reduction
    if (lane == 0) red[wave] = v;
    __syncthreads();
    T s = 0;
#pragma unroll
    for (int w = 0; w < kSnWaves; ++w) s += red[w];
    return s;
}

// sum of squares of x[0..n) by the whole block
template <typename T>
__device__ __forceinline__ T block_sumsq(const T* __restrict__ x, int n, T* red) {
    T p = 0;
    for (int i = threadIdx.x; i < n; i += kSnBlock) p += x[i] * x[i];
    return block_sum(p, red);
}

template <typename T>
__global__ void __launch_bounds__(kSnBlock)
sn_phase1_kernel(SnFwdArgs a) {          // v_raw[j] = sum_i W[i, j] u[i]   (written to L.v)
    __shared__ T part[kSnWaves][kColChunk];
    int chunk;
    const ffwm_sn_layer L = a.l[find_layer(a, chunk)];
    const T* W = static_cast<const T*>(L.weight);
    const T* u = static_cast<const T*>(L.u);
    T* v = static_cast<T*>(L.v);
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    const int j = chunk * kColChunk + lane;
    T s = 0;
    if (j < L.cols) {
        const T* wc = W + j;
        int i = wave;
        for (; i + 3 * kSnWaves < L.rows; i += 4 * kSnWaves) {       // four independent loads in flight per lane
            const T w0 = wc[static_cast<size_t>(i) * L.cols], w1 = wc[static_cast<size_t>(i + kSnWaves) * L.cols];
            const T w2 = wc[static_cast<size_t>(i + 2 * kSnWaves) * L.cols], w3 = wc[static_cast<size_t>(i + 3 * kSnWaves) * L.cols];
            s += w0 * u[i] + w1 * u[i + kSnWaves] + w2 * u[i + 2 * kSnWaves] + w3 * u[i + 3 * kSnWaves];
        }
        for (; i < L.rows; i += kSnWaves) s += wc[static_cast<size_t>(i) * L.cols] * u[i];
    }
    part[wave][lane] = s;
    __syncthreads();
    if (wave == 0 && j < L.cols) {
        T t = 0;
#pragma unroll
        for (int w = 0; w < kSnWaves; ++w) t += part[w][lane];
        v[j] = t;
    }
}

template <typename T>
__global__ void __launch_bounds__(kSnBlock)
sn_phase2_kernel(SnFwdArgs a, int power_iterations, T eps) {     // wv[i] = sum_j W[i, j] v[j] / max(|v_raw|, eps)
    __shared__ T red[kSnWaves];
    int chunk;
    const ffwm_sn_layer L = a.l[find_layer(a, chunk)];
    const T* W = static_cast<const T*>(L.weight);
    const T* v = static_cast<const T*>(L.v);
    T* wv = static_cast<T*>(L.wv);
    T nrm = 1;
    if (power_iterations) {              // v still holds v_raw: normalise on the fly
        nrm = sqrt(block_sumsq(v, L.cols, red));
        nrm = nrm > eps ? nrm : eps;     // F.normalize: x / max(||x||, eps)
    }
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    for (int r = wave; r < kRowChunk; r += kSnWaves) {
        const int i = chunk * kRowChunk + r;
        if (i >= L.rows) break;
        const T* wr = W + static_cast<size_t>(i) * L.cols;
        T s = 0;
        int j = lane;
        for (; j + 3 * kWave < L.cols; j += 4 * kWave)               // four independent loads in flight per lane
            s += wr[j] * (v[j] / nrm) + wr[j + kWave] * (v[j + kWave] / nrm) + wr[j + 2 * kWave] * (v[j + 2 * kWave] / nrm) +
                 wr[j + 3 * kWave] * (v[j + 3 * kWave] / nrm);
        for (; j < L.cols; j += kWave) s += wr[j] * (v[j] / nrm);
        s = wave_sum(s);
        if (lane == 0) wv[i] = s;
    }
}

template <typename T>
__global__ void __launch_bounds__(kSnBlock)
sn_phase3_kernel(SnFwdArgs a, int power_iterations, T eps) {
    __shared__ T red[kSnWaves];
    int chunk;
    const ffwm_sn_layer L = a.l[find_layer(a, chunk)];
    const T* W = static_cast<const T*>(L.weight);
    const T* wv = static_cast<const T*>(L.wv);
    T* u = static_cast<T*>(L.u);
    T* v = static_cast<T*>(L.v);
    T* out = static_cast<T*>(L.weight_sn);
    T sigma, unrm = 1;
    if (power_iterations) {
        // u = wv / max(|wv|, eps);  sigma = u . wv
        T nrm = sqrt(block_sumsq(wv, L.rows, red));
        unrm = nrm > eps ? nrm : eps;
        T p = 0;
        for (int i = threadIdx.x; i < L.rows; i += kSnBlock) p += (wv[i] / unrm) * wv[i];
        sigma = block_sum(p, red);
    } else {
        T p = 0;
        for (int i = threadIdx.x; i < L.rows; i += kSnBlock) p += u[i] * wv[i];
        sigma = block_sum(p, red);
    }
    if (chunk == 0) {                    // the layer's first chunk publishes u, v, sigma (and the saved copies)
        T vinv_nrm = 1;
        if (power_iterations) {
            T nrm = sqrt(block_sumsq(v, L.cols, red));
            vinv_nrm = nrm > eps ? nrm : eps;
        }
        __syncthreads();                 // every thread has read v_raw before anyone overwrites it
        T* us = static_cast<T*>(L.u_saved);
        T* vs = static_cast<T*>(L.v_saved);
        for (int i = threadIdx.x; i < L.rows; i += kSnBlock) {
            const T ui = power_iterations ? wv[i] / unrm : u[i];
            if (power_iterations) u[i] = ui;
            if (us) us[i] = ui;
        }
        for (int j = threadIdx.x; j < L.cols; j += kSnBlock) {
            const T vj = power_iterations ? v[j] / vinv_nrm : v[j];
            if (power_iterations) v[j] = vj;
            if (vs) vs[j] = vj;
        }
        if (threadIdx.x == 0) *static_cast<T*>(L.sigma) = sigma;
    }
    const size_t n = static_cast<size_t>(L.rows) * L.cols;
    const size_t e0 = static_cast<size_t>(chunk) * kElemChunk;
    const size_t e1 = e0 + kElemChunk < n ? e0 + kElemChunk : n;
    for (size_t i = e0 + threadIdx.x; i < e1; i += kSnBlock) out[i] = W[i] / sigma;
}

// backward: phase A partial <G, W> per chunk -> scratch; phase B sums the layer's partials and writes
// grad_weight = G / sigma - (<G, W> / sigma^2) u v^T for its chunk.
template <typename T>
__global__ void __launch_bounds__(kSnBlock)
sn_bwd_dot_kernel(SnBwdArgs a) {
    __shared__ T red[kSnWaves];
    int chunk;
    const ffwm_sn_grad_layer L = a.l[find_layer(a, chunk)];
    const T* W = static_cast<const T*>(L.weight);
    const T* G = static_cast<const T*>(L.grad_weight_sn);
    const size_t n = static_cast<size_t>(L.rows) * L.cols;
    const size_t e0 = static_cast<size_t>(chunk) * kElemChunk;
    const size_t e1 = e0 + kElemChunk < n ? e0 + kElemChunk : n;
    T p = 0;
    for (size_t i = e0 + threadIdx.x; i < e1; i += kSnBlock) p += G[i] * W[i];
    p = block_sum(p, red);
    if (threadIdx.x == 0) static_cast<T*>(L.partials)[chunk] = p;
}

template <typename T>
__global__ void __launch_bounds__(kSnBlock)
sn_bwd_apply_kernel(SnBwdArgs a) {
    __shared__ T red[kSnWaves];
    int chunk;
    const ffwm_sn_grad_layer L = a.l[find_layer(a, chunk)];
    const T* W = static_cast<const T*>(L.weight);
    const T* G = static_cast<const T*>(L.grad_weight_sn);
    const T* u = static_cast<const T*>(L.u);
    const T* v = static_cast<const T*>(L.v);
    T* out = static_cast<T*>(L.grad_weight);
    const T sigma = *static_cast<const T*>(L.sigma);
    const size_t n = static_cast<size_t>(L.rows) * L.cols;
    const int nchunks = static_cast<int>((n + kElemChunk - 1) / kElemChunk);
    const T* partials = static_cast<const T*>(L.partials);
    T p = 0;
    for (int k = threadIdx.x; k < nchunks; k += kSnBlock) p += partials[k];
    const T coef = block_sum(p, red) / (sigma * sigma);
    const size_t e0 = static_cast<size_t>(chunk) * kElemChunk;
    const size_t e1 = e0 + kElemChunk < n ? e0 + kElemChunk : n;
    (void)W;
    for (size_t i = e0 + threadIdx.x; i < e1; i += kSnBlock) {
        const int r = static_cast<int>(i / L.cols), c = static_cast<int>(i - static_cast<size_t>(r) * L.cols);
        out[i] = G[i] / sigma - coef * (u[r] * v[c]);
    }
}

}  // namespace
}  // namespace ffwm

using namespace ffwm;

static int chunks_of(int64_t n, int per) { return static_cast<int>((n + per - 1) / per); }

extern "C" int ffwm_spectral_norm_forward(const ffwm_sn_layer* layers, int n_layers, int power_iterations,
                                          double eps, int dtype, void* stream) {
    const char* fn = "ffwm_spectral_norm_forward";
    FFWM_REQUIRE(dtype_ok(dtype), FFWM_ERR_DTYPE, "%s: dtype %d is not FFWM_F32/FFWM_F64", fn, dtype);
    FFWM_REQUIRE(layers && n_layers >= 0 && (power_iterations == 0 || power_iterations == 1), FFWM_ERR_ARG,
                 "%s: bad arguments (power_iterations must be 0 or 1)", fn);
    hipStream_t st = static_cast<hipStream_t>(stream);
    for (int i = 0; i < n_layers; ++i) {
        const ffwm_sn_layer& l = layers[i];
        FFWM_REQUIRE(l.weight && l.u && l.v && l.wv && l.weight_sn && l.sigma && l.rows > 0 && l.cols > 0, FFWM_ERR_ARG,
                     "%s: layer %d has a NULL pointer or a non-positive size", fn, i);
    }
    const double esz = dtype == FFWM_F32 ? 4 : 8;
    for (int base = 0; base < n_layers; base += FFWM_SN_MAX_LAYERS) {
        const int n = n_layers - base < FFWM_SN_MAX_LAYERS ? n_layers - base : FFWM_SN_MAX_LAYERS;
        SnFwdArgs a;
        a.n = n;
        double elems = 0;
        for (int i = 0; i < n; ++i) {
            a.l[i] = layers[base + i];
            elems += static_cast<double>(a.l[i].rows) * a.l[i].cols;
        }
        auto fill = [&](auto count) {
            a.start[0] = 0;
            for (int i = 0; i < n; ++i) a.start[i + 1] = a.start[i] + count(a.l[i]);
            return a.start[n];
        };
        if (power_iterations) {
            const int g1 = fill([](const ffwm_sn_layer& l) { return chunks_of(l.cols, kColChunk); });
            LaunchScope ls("spectral_norm_fwd_wtu", st, esz * elems);
            if (dtype == FFWM_F32) hipLaunchKernelGGL((sn_phase1_kernel<float>), dim3(g1), dim3(kSnBlock), 0, st, a);
            else hipLaunchKernelGGL((sn_phase1_kernel<double>), dim3(g1), dim3(kSnBlock), 0, st, a);
        }
        {
            const int g2 = fill([](const ffwm_sn_layer& l) { return chunks_of(l.rows, kRowChunk); });
            LaunchScope ls("spectral_norm_fwd_wv", st, esz * elems);
            if (dtype == FFWM_F32)
                hipLaunchKernelGGL((sn_phase2_kernel<float>), dim3(g2), dim3(kSnBlock), 0, st, a, power_iterations, static_cast<float>(eps));
            else
                hipLaunchKernelGGL((sn_phase2_kernel<double>), dim3(g2), dim3(kSnBlock), 0, st, a, power_iterations, eps);
        }
        {
            const int g3 = fill([](const ffwm_sn_layer& l) { return chunks_of(static_cast<int64_t>(l.rows) * l.cols, kElemChunk); });
            LaunchScope ls("spectral_norm_fwd_div", st, 2 * esz * elems);
            if (dtype == FFWM_F32)
                hipLaunchKernelGGL((sn_phase3_kernel<float>), dim3(g3), dim3(kSnBlock), 0, st, a, power_iterations, static_cast<float>(eps));
            else
                hipLaunchKernelGGL((sn_phase3_kernel<double>), dim3(g3), dim3(kSnBlock), 0, st, a, power_iterations, eps);
        }
    }
    return check_launch(fn);
}

extern "C" int ffwm_spectral_norm_backward(const ffwm_sn_grad_layer* layers, int n_layers, int dtype, void* stream) {
    const char* fn = "ffwm_spectral_norm_backward";
    FFWM_REQUIRE(dtype_ok(dtype), FFWM_ERR_DTYPE, "%s: dtype %d is not FFWM_F32/FFWM_F64", fn, dtype);
    FFWM_REQUIRE(layers && n_layers >= 0, FFWM_ERR_ARG, "%s: bad arguments", fn);
    hipStream_t st = static_cast<hipStream_t>(stream);
    for (int i = 0; i < n_layers; ++i) {
        const ffwm_sn_grad_layer& l = layers[i];
        FFWM_REQUIRE(l.weight && l.u && l.v && l.sigma && l.grad_weight_sn && l.grad_weight && l.partials && l.rows > 0 &&
                         l.cols > 0, FFWM_ERR_ARG, "%s: layer %d has a NULL pointer or a non-positive size", fn, i);
    }
    const double esz = dtype == FFWM_F32 ? 4 : 8;
    for (int base = 0; base < n_layers; base += FFWM_SN_MAX_LAYERS) {
        const int n = n_layers - base < FFWM_SN_MAX_LAYERS ? n_layers - base : FFWM_SN_MAX_LAYERS;
        SnBwdArgs a;
        a.n = n;
        a.start[0] = 0;
        double elems = 0;
        for (int i = 0; i < n; ++i) {
            a.l[i] = layers[base + i];
            const int64_t e = static_cast<int64_t>(a.l[i].rows) * a.l[i].cols;
            a.start[i + 1] = a.start[i] + chunks_of(e, kElemChunk);
            elems += static_cast<double>(e);
        }
        {
            LaunchScope ls("spectral_norm_bwd_dot", st, 2 * esz * elems);
            if (dtype == FFWM_F32) hipLaunchKernelGGL((sn_bwd_dot_kernel<float>), dim3(a.start[n]), dim3(kSnBlock), 0, st, a);
            else hipLaunchKernelGGL((sn_bwd_dot_kernel<double>), dim3(a.start[n]), dim3(kSnBlock), 0, st, a);
        }
        {
            LaunchScope ls("spectral_norm_bwd_apply", st, 2 * esz * elems);
            if (dtype == FFWM_F32) hipLaunchKernelGGL((sn_bwd_apply_kernel<float>), dim3(a.start[n]), dim3(kSnBlock), 0, st, a);
            else hipLaunchKernelGGL((sn_bwd_apply_kernel<double>), dim3(a.start[n]), dim3(kSnBlock), 0, st, a);
        }
    }
    return check_launch(fn);
}
