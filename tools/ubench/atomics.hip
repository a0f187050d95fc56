// Micro-benchmark: float atomic-add throughput on gfx950 -- LDS (ds_add_f32 / rtn / f64 / CAS) and
// global memory at workgroup / agent / system scope, block-private conflict-free addresses.
// Build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics atomics.hip -o atomics
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr int N = 4096;   // floats per block region (16 KiB)

template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, float* gbuf, int iters) {
    __shared__ float A[N];
    __shared__ double D[N / 2];
    for (int i = threadIdx.x; i < N; i += 256) A[i] = 0;
    for (int i = threadIdx.x; i < N / 2; i += 256) D[i] = 0;
    __syncthreads();
    float* G = gbuf + (size_t)blockIdx.x * N;
    float v = 1.0f + (threadIdx.x & 63);
    float acc = 0;
    if (MODE == 10) __builtin_amdgcn_s_setreg(1 | (4 << 6) | (1 << 11), 0);     // MODE.FP_DENORM[5:4] = 0: flush fp32 denormals (in and out)
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int idx = (threadIdx.x + u * 256) & (N - 1);      // conflict-free, consecutive per wave
            if (MODE == 0 || MODE == 10) __hip_atomic_fetch_add(&A[idx], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (MODE == 1) acc += __hip_atomic_fetch_add(&A[idx], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (MODE == 2) __hip_atomic_fetch_add(&D[idx & (N / 2 - 1)], (double)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (MODE == 3) {   // CAS loop
                unsigned* p = reinterpret_cast<unsigned*>(&A[idx]);
                unsigned old = *p, assumed;
                do {
                    assumed = old;
                    old = atomicCAS(p, assumed, __float_as_uint(__uint_as_float(assumed) + v));
                } while (old != assumed);
            }
            if (MODE == 4) __hip_atomic_fetch_add(&G[idx], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (MODE == 5) __hip_atomic_fetch_add(&G[idx], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (MODE == 6) __hip_atomic_fetch_add(&G[idx], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if (MODE == 7) __hip_atomic_fetch_add(reinterpret_cast<unsigned*>(&A[idx]), 3u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (MODE == 8) { float t = G[idx]; G[idx] = t + v; }    // plain global RMW (L1/L2 resident)
            if (MODE == 9) __hip_atomic_fetch_max(reinterpret_cast<int*>(&A[idx]), (int)u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    __syncthreads();
    float s = v + acc;
    for (int i = threadIdx.x; i < N; i += 256) s += A[i] + (float)D[i / 2];
    if (s == 12345.678f) out[0] = s;
}

template <int MODE>
void run(const char* name, int blocks, int iters) {
    float *d, *g;
    hipMalloc(&d, 4);
    hipMalloc(&g, (size_t)blocks * N * 4);
    hipMemset(g, 0, (size_t)blocks * N * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, g, 2);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, g, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double ops = (double)blocks * 256 * iters * 16;
    const double per_cu_clk = ops / 256.0 / (ms * 1e-3 * 2.4e9);
    printf("%-34s blocks=%d %.3f ms  %.1f G lane-ops/s  %.2f lane-ops/clk/CU  (%.1f clk per wave-instr)\n", name, blocks, ms,
           ops / ms / 1e6, per_cu_clk, 64.0 / per_cu_clk);
    (void)hipFree(d);
    (void)hipFree(g);
}

int main() {
    const int B = 1024;
    run<7>("lds ds_add_u32", B, 2000);
    run<9>("lds ds_max_i32", B, 2000);
    run<0>("lds ds_add_f32", B, 200);
    run<10>("lds ds_add_f32, MODE.FP_DENORM f32 = flush", B, 200);
    run<1>("lds ds_add_rtn_f32", B, 200);
    run<2>("lds ds_add_f64", B, 200);
    run<3>("lds CAS loop f32", B, 200);
    run<8>("global plain RMW (private 16K)", B, 200);
    run<4>("global atomic f32 scope=workgroup", B, 200);
    run<5>("global atomic f32 scope=agent", B, 200);
    run<6>("global atomic f32 scope=system", B, 200);
    return 0;
}
