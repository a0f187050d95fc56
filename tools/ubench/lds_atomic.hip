// Micro-benchmark: LDS atomic-add throughput on gfx950 (ds_add_f32 vs ds_write_b32 vs ds_add_u32),
// conflict-free and with same-address collisions.  Build: hipcc --offload-arch=gfx950 -O3 lds_atomic.hip -o lds_atomic
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

constexpr int N = 4096;

template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, int iters, int spread) {
    __shared__ float A[N];
    for (int i = threadIdx.x; i < N; i += 256) A[i] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned h = threadIdx.x * 2654435761u + blockIdx.x;
    float v = 1.0f + lane;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            h = h * 1664525u + 1013904223u;
            const int off = spread ? (int)((h >> 16) % (unsigned)spread) : 0;
            const int idx = (wave * 1024 + (u & 3) * 64 + (u >> 2) + lane + off) & (N - 1);
            if (MODE == 0) __hip_atomic_fetch_add(&A[idx], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (MODE == 1) A[idx] = v;
            if (MODE == 2) __hip_atomic_fetch_add(reinterpret_cast<unsigned*>(&A[idx]), (unsigned)lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (MODE == 3) v += A[idx];
            if (MODE == 4) { float t = A[idx]; A[idx] = t + v; }
        }
    }
    __syncthreads();
    float s = v;
    for (int i = threadIdx.x; i < N; i += 256) s += A[i];
    if (s == 12345.678f) out[0] = s;
}

template <int MODE>
void run(const char* name, int spread) {
    float* d;
    hipMalloc(&d, 4);
    const int blocks = 256 * 4, iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 10, spread);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters, spread);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double ops = (double)blocks * 256 * iters * 16;           // lane-ops
    const double per_cu_clk = ops / 256.0 / (ms * 1e-3 * 2.4e9);    // lane-ops per clock per CU (at 2.4 GHz)
    printf("%-28s spread=%d  %.3f ms  %.1f G lane-ops/s  %.2f lane-ops/clk/CU  (%.1f clk per wave-instr)\n", name, spread, ms,
           ops / ms / 1e6, per_cu_clk, 64.0 / per_cu_clk);
    hipFree(d);
}

int main() {
    for (int spread : {0, 4, 1}) {
        if (spread == 1) spread = 0;
        run<0>("ds_add_f32", spread);
        run<1>("ds_write_b32", spread);
        run<2>("ds_add_u32", spread);
        run<3>("ds_read_b32", spread);
        run<4>("read+add+write (racy)", spread);
    }
    return 0;
}
