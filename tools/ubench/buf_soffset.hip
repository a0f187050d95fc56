// Does the buffer range check of gfx950 see the scalar offset?  raw_buffer_load_b32(rsrc, voffset, soffset):
//   case A: voffset = 0xFFFFFFF0 (sentinel), soffset = 256      -> 0 expected if the check is on voffset alone or in > 32 bits
//   case B: voffset = 16, soffset = num_records (in-range voffset, sum past the end)
//   case C: voffset = 16, soffset = 256 (plain)
// hipcc --offload-arch=gfx950 tools/ubench/buf_soffset.hip -o tools/ubench/buf_soffset && tools/ubench/buf_soffset
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const float* p, unsigned bytes, float* out) {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, static_cast<int>(bytes), 0x00020000);
    out[0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, 0xFFFFFFF0u, 256, 0));
    out[1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, 16u, static_cast<int>(bytes), 0));
    out[2] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, 16u, 256, 0));
    out[3] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, 0xFFFFFFF0u, 0, 0));
    out[4] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, bytes - 4u, 256, 0));
    out[5] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, bytes - 260u, 256, 0));
    out[6] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, 0xFFFFFFFFu, 0, 0));
    out[7] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, 0xFFFFFFFFu, 64, 0));
}
int main() {
    const int n = 4096;
    std::vector<float> h(n);
    for (int i = 0; i < n; ++i) h[i] = 1000.f + i;
    float *d, *o;
    hipMalloc(&d, n * 4); hipMalloc(&o, 64);
    hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
    hipMemset(o, 0, 64);
    k<<<1, 1>>>(d, 2048 * 4, o);          // the resource covers the first half of the allocation
    float r[8];
    hipMemcpy(r, o, 32, hipMemcpyDeviceToHost);
    printf("A sentinel + soffset 256        : %g (0 = dropped; 1060 = wrapped to byte 240)\n", r[0]);
    printf("B voffset 16 + soffset = records: %g (0 = the check sees soffset; %g = it does not)\n", r[1], h[2048 + 4]);
    printf("C voffset 16 + soffset 256      : %g (expect %g)\n", r[2], h[(16 + 256) / 4]);
    printf("D sentinel, soffset 0           : %g (expect 0)\n", r[3]);
    printf("E voffset records-4 + soffset 256: %g (0 = the check sees soffset; %g = it does not)\n", r[4], h[2047 + 64]);
    printf("F voffset records-260 + soffset 256: %g (expect %g)\n", r[5], h[2047]);
    printf("G unaligned sentinel 0xFFFFFFFF, soffset 0 / 64: %g %g (expect 0 0)\n", r[6], r[7]);
    return 0;
}
