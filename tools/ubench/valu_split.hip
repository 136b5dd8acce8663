// Micro-benchmark (round 6): issue rate of the vector instructions of the three-way bf16 split (v_and_b32 with a 32-bit literal or a
// register mask, v_sub_f32, v_perm_b32) at 1, 2, 3, 4 waves per SIMD; ns per instruction and SIMD by the wall clock.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_split.hip -o /tmp/valu_split && /tmp/valu_split
#include <hip/hip_runtime.h>
#include <stdio.h>
// one "unit" = 12 instructions on 2 values: and, sub, and, sub per value + 3 perms + 1 extra sub
#define UNIT_LIT(a, b, p) \
    "v_and_b32 v20, 0xffff0000, " a "\n v_and_b32 v21, 0xffff0000, " b "\n v_sub_f32 v22, " a ", v20\n v_sub_f32 v23, " b ", v21\n" \
    "v_perm_b32 " p ", " b ", " a ", %1\n v_and_b32 v20, 0xffff0000, v22\n v_and_b32 v21, 0xffff0000, v23\n v_sub_f32 v24, v22, v20\n v_sub_f32 v25, v23, v21\n" \
    "v_perm_b32 v26, v23, v22, %1\n v_perm_b32 v27, v25, v24, %1\n v_xor_b32 " p ", v26, v27\n"
#define UNIT_REG(a, b, p) \
    "v_and_b32 v20, %2, " a "\n v_and_b32 v21, %2, " b "\n v_sub_f32 v22, " a ", v20\n v_sub_f32 v23, " b ", v21\n" \
    "v_perm_b32 " p ", " b ", " a ", %1\n v_and_b32 v20, %2, v22\n v_and_b32 v21, %2, v23\n v_sub_f32 v24, v22, v20\n v_sub_f32 v25, v23, v21\n" \
    "v_perm_b32 v26, v23, v22, %1\n v_perm_b32 v27, v25, v24, %1\n v_xor_b32 " p ", v26, v27\n"
#define UNIT_ADD(a, b, p) \
    "v_add_f32 v20, " a ", " b "\n v_add_f32 v21, " b ", " a "\n v_sub_f32 v22, " a ", v20\n v_sub_f32 v23, " b ", v21\n" \
    "v_add_f32 " p ", " b ", " a "\n v_add_f32 v20, v22, v22\n v_add_f32 v21, v23, v23\n v_sub_f32 v24, v22, v20\n v_sub_f32 v25, v23, v21\n" \
    "v_add_f32 v26, v23, v22\n v_add_f32 v27, v25, v24\n v_add_f32 " p ", v26, v27\n"
template <int MODE, int NT> __global__ __launch_bounds__(NT) void k(float *out, const float *in, int iters)
{
    float a = in[threadIdx.x], b = in[threadIdx.x + 64];
    unsigned sel = 0x07060302u, mask = 0xffff0000u;
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE == 0)
            asm volatile(UNIT_LIT("%0", "%3", "v28") UNIT_LIT("%3", "%0", "v29") UNIT_LIT("%0", "%3", "v30") UNIT_LIT("%3", "%0", "v31")
                         : "+v"(a) : "s"(sel), "s"(mask), "v"(b) : "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31");
        else if constexpr (MODE == 1)
            asm volatile(UNIT_REG("%0", "%3", "v28") UNIT_REG("%3", "%0", "v29") UNIT_REG("%0", "%3", "v30") UNIT_REG("%3", "%0", "v31")
                         : "+v"(a) : "s"(sel), "s"(mask), "v"(b) : "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31");
        else
            asm volatile(UNIT_ADD("%0", "%3", "v28") UNIT_ADD("%3", "%0", "v29") UNIT_ADD("%0", "%3", "v30") UNIT_ADD("%3", "%0", "v31")
                         : "+v"(a) : "s"(sel), "s"(mask), "v"(b) : "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31");
    }
    out[blockIdx.x * NT + threadIdx.x] = a;
}
template <int MODE, int NT> void run(float *out, float *in)
{
    const int iters = 4000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<MODE, NT><<<256, NT>>>(out, in, iters);
    (void)hipEventRecord(e0, 0);
    k<MODE, NT><<<256, NT>>>(out, in, iters);
    (void)hipEventRecord(e1, 0);
    (void)hipDeviceSynchronize();
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double per = ms * 1e6 / iters / 48.0 / (NT / 256.0);
    printf("%s, %d waves/SIMD: %.3f ns per instruction and SIMD (%.2f cycles at 2.4 GHz)\n", MODE == 0 ? "split, literal mask " : MODE == 1 ? "split, register mask" : "adds / subs only     ", NT / 256, per, per * 2.4);
}
int main()
{
    float *out, *in;
    (void)hipMalloc(&out, 256 * 1024 * 4); (void)hipMalloc(&in, 4096 * 4);
    float h[4096];
    for (int e = 0; e < 4096; ++e) h[e] = 0.001f * (e % 97) + 0.04f;
    (void)hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    run<0, 256>(out, in); run<0, 512>(out, in); run<0, 768>(out, in); run<0, 1024>(out, in);
    run<1, 256>(out, in); run<1, 512>(out, in); run<1, 768>(out, in); run<1, 1024>(out, in);
    run<2, 256>(out, in); run<2, 512>(out, in); run<2, 768>(out, in); run<2, 1024>(out, in);
    return 0;
}
