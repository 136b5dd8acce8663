// Micro-test (round 6): is the scalar offset (soffset) of a raw buffer load part of the range check on gfx950?
// buffer of 1024 bytes; loads at (voffset, soffset) pairs around the end; out-of-range loads return 0.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(const unsigned *src, unsigned *out)
{
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, 1024, 0x00020000);
    out[0] = __builtin_amdgcn_raw_buffer_load_b32(r, 512, 256, 0);      // in range: element 192
    out[1] = __builtin_amdgcn_raw_buffer_load_b32(r, 512, 768, 0);      // sum 1280: out of range only if soffset counts
    out[2] = __builtin_amdgcn_raw_buffer_load_b32(r, 1020, 0, 0);       // last element
    out[3] = __builtin_amdgcn_raw_buffer_load_b32(r, 1024, 0, 0);       // voffset out of range
    out[4] = __builtin_amdgcn_raw_buffer_load_b32(r, 0, 0x40000000, 0); // soffset 2^30
    out[5] = __builtin_amdgcn_raw_buffer_load_b32(r, 16, 0x7ffffff0, 0);
}
int main()
{
    unsigned h[1024], *src, *out, ho[8];
    for (int e = 0; e < 1024; ++e) h[e] = 0xA0000 + e;
    (void)hipMalloc(&src, sizeof(h)); (void)hipMalloc(&out, 64);
    (void)hipMemcpy(src, h, sizeof(h), hipMemcpyHostToDevice);
    k<<<1, 1>>>(src, out);
    (void)hipMemcpy(ho, out, 32, hipMemcpyDeviceToHost);
    printf("in range (v512+s256): %x (expect a00c0)\nv512+s768 (sum past the end): %x\nlast: %x\nv1024: %x\ns=2^30: %x\nv16+s0x7ffffff0: %x\n", ho[0], ho[1], ho[2], ho[3], ho[4], ho[5]);
    return 0;
}
