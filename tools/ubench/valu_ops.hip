// Micro-benchmark (round 6): issue rate of single vector instructions (8 independent destination registers per loop body, 48 per body) at
// 1, 2, 4 waves per SIMD: ns (and cycles at 2.4 GHz) per instruction and SIMD by the wall clock.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_ops.hip -o /tmp/valu_ops && /tmp/valu_ops
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define R8(OP) OP("v20") OP("v21") OP("v22") OP("v23") OP("v24") OP("v25") OP("v26") OP("v27")
#define R48(OP) R8(OP) R8(OP) R8(OP) R8(OP) R8(OP) R8(OP)
#define OP_ADD(d) "v_add_f32 " d ", %0, %3\n"
#define OP_SUB(d) "v_sub_f32 " d ", " d ", %3\n"
#define OP_ANDL(d) "v_and_b32 " d ", 0xffff0000, %0\n"
#define OP_ANDS(d) "v_and_b32 " d ", %2, %0\n"
#define OP_PERM(d) "v_perm_b32 " d ", %0, %3, %1\n"
#define OP_CVT(d) "v_cvt_pk_bf16_f32 " d ", %0, %3\n"
#define OP_PACK(d) "v_pack_b32_f16 " d ", %0, %3 op_sel:[1,1,0]\n"
#define OP_ANDOR(d) "v_and_or_b32 " d ", %0, %2, %3\n"
#define OP_BFI(d) "v_bfi_b32 " d ", %2, %0, %3\n"
#define OP_ALIGN(d) "v_alignbit_b32 " d ", %0, %3, 16\n"
#define OP_LSHR(d) "v_lshrrev_b32 " d ", 16, %0\n"
#define OP_XOR(d) "v_xor_b32 " d ", %0, %3\n"
#define OP_FMA(d) "v_fma_f32 " d ", %0, %3, " d "\n"
#define OP_MOV(d) "v_mov_b32 " d ", %0\n"
#define OP_PKADD(d) "v_pk_add_f32 v[30:31], %4, %5\n"
#define OP_PKADDN(d) "v_pk_add_f32 v[30:31], %4, %5 neg_lo:[0,1] neg_hi:[0,1]\n"
#define OP_PKMUL(d) "v_pk_mul_f32 v[30:31], %4, %5\n"
#define OP_MED3(d) "v_med3_f32 " d ", %0, %3, %3\n"
#define OP_MAX(d) "v_max_f32 " d ", %0, %3\n"
#define OP_ADD3(d) "v_add3_u32 " d ", %0, %3, %3\n"
#define OP_ADDU(d) "v_add_u32 " d ", %0, %3\n"
#define OP_ADDUS(d) "v_add_u32 " d ", %2, %3\n"
#define OP_SDWA(d) "v_or_b32_sdwa " d ", %0, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n"
#define OP_MAXI(d) "v_max_i32 " d ", 0, %0\n"
#define OP_OR(d) "v_or_b32 " d ", %0, %3\n"
#define OP_SNOP(d) "s_nop 0\n"
#define OP_SADD(d) "s_add_i32 s40, s41, s42\n"
#define CLOB : "+v"(a) : "s"(sel), "s"(mask), "v"(b), "v"(pa), "v"(pb) : "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v30", "v31", "s40", "s41", "s42"
template <int MODE, int NT> __global__ __launch_bounds__(NT) void k(float *out, const float *in, int iters)
{
    float a = in[threadIdx.x], b = in[threadIdx.x + 64];
    unsigned sel = 0x07060302u, mask = 0xffff0000u;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 pa = {a, b}, pb = {b, a};
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE == 0) asm volatile(R48(OP_ADD) CLOB);
        if constexpr (MODE == 1) asm volatile(R48(OP_SUB) CLOB);
        if constexpr (MODE == 2) asm volatile(R48(OP_ANDL) CLOB);
        if constexpr (MODE == 3) asm volatile(R48(OP_ANDS) CLOB);
        if constexpr (MODE == 4) asm volatile(R48(OP_PERM) CLOB);
        if constexpr (MODE == 5) asm volatile(R48(OP_CVT) CLOB);
        if constexpr (MODE == 6) asm volatile(R48(OP_PACK) CLOB);
        if constexpr (MODE == 7) asm volatile(R48(OP_ANDOR) CLOB);
        if constexpr (MODE == 8) asm volatile(R48(OP_BFI) CLOB);
        if constexpr (MODE == 9) asm volatile(R48(OP_ALIGN) CLOB);
        if constexpr (MODE == 10) asm volatile(R48(OP_LSHR) CLOB);
        if constexpr (MODE == 11) asm volatile(R48(OP_XOR) CLOB);
        if constexpr (MODE == 12) asm volatile(R48(OP_FMA) CLOB);
        if constexpr (MODE == 13) asm volatile(R48(OP_MOV) CLOB);
        if constexpr (MODE == 14) asm volatile(R48(OP_PKADD) CLOB);
        if constexpr (MODE == 15) asm volatile(R48(OP_PKADDN) CLOB);
        if constexpr (MODE == 16) asm volatile(R48(OP_PKMUL) CLOB);
        if constexpr (MODE == 17) asm volatile(R48(OP_MED3) CLOB);
        if constexpr (MODE == 18) asm volatile(R48(OP_MAX) CLOB);
        if constexpr (MODE == 19) asm volatile(R48(OP_ADD3) CLOB);
        if constexpr (MODE == 20) asm volatile(R48(OP_ADDU) CLOB);
        if constexpr (MODE == 21) asm volatile(R48(OP_ADDUS) CLOB);
        if constexpr (MODE == 22) asm volatile(R48(OP_SNOP) CLOB);
        if constexpr (MODE == 23) asm volatile(R48(OP_SNOP) CLOB);
        if constexpr (MODE == 24) asm volatile(R48(OP_SDWA) CLOB);
        if constexpr (MODE == 25) asm volatile(R48(OP_MAXI) CLOB);
        if constexpr (MODE == 26) asm volatile(R48(OP_OR) CLOB);
    }
    out[blockIdx.x * NT + threadIdx.x] = a;
}
static const char *names[] = {"v_add_f32", "v_sub_f32 (dependent on itself, 8 chains)", "v_and_b32 literal", "v_and_b32 sgpr", "v_perm_b32", "v_cvt_pk_bf16_f32", "v_pack_b32_f16 op_sel hi,hi",
                              "v_and_or_b32", "v_bfi_b32", "v_alignbit_b32", "v_lshrrev_b32", "v_xor_b32", "v_fma_f32 (8 chains)", "v_mov_b32", "v_pk_add_f32", "v_pk_add_f32 neg", "v_pk_mul_f32", "v_med3_f32", "v_max_f32", "v_add3_u32", "v_add_u32", "v_add_u32 sgpr", "s_nop 0", "s_nop 0 (again)", "v_or_b32_sdwa src1 WORD_1", "v_max_i32", "v_or_b32"};
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
    printf("%-44s %d waves/SIMD: %.3f ns = %.2f cycles at 2.4 GHz\n", names[MODE], NT / 256, per, per * 2.4);
}
template <int MODE> void run3(float *out, float *in) { run<MODE, 256>(out, in); run<MODE, 512>(out, in); run<MODE, 1024>(out, in); }
template <int M> void maybe(int want, float *out, float *in) { if (want == M || want < 0) run3<M>(out, in); }
int main(int argc, char **argv)
{
    const int want = argc > 1 ? atoi(argv[1]) : -1;
    float *out, *in;
    (void)hipMalloc(&out, 256 * 1024 * 4); (void)hipMalloc(&in, 4096 * 4);
    float h[4096];
    for (int e = 0; e < 4096; ++e) h[e] = 0.001f * (e % 97) + 0.04f;
    (void)hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    maybe<0>(want, out, in);
    maybe<1>(want, out, in);
    maybe<2>(want, out, in);
    maybe<3>(want, out, in);
    maybe<4>(want, out, in);
    maybe<5>(want, out, in);
    maybe<6>(want, out, in);
    maybe<7>(want, out, in);
    maybe<8>(want, out, in);
    maybe<9>(want, out, in);
    maybe<10>(want, out, in);
    maybe<11>(want, out, in);
    maybe<12>(want, out, in);
    maybe<13>(want, out, in);
    maybe<14>(want, out, in);
    maybe<15>(want, out, in);
    maybe<16>(want, out, in);
    maybe<17>(want, out, in);
    maybe<18>(want, out, in);
    maybe<19>(want, out, in);
    maybe<20>(want, out, in);
    maybe<21>(want, out, in);
    maybe<22>(want, out, in);
    maybe<23>(want, out, in);
    maybe<24>(want, out, in);
    maybe<25>(want, out, in);
    maybe<26>(want, out, in);
    return 0;
}
