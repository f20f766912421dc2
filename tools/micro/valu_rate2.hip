// Issue-rate microbenchmark 2 for gfx950 (development tool): which encodings / instruction classes run at the 2-cycle
// rate, what scalar instructions cost beside vector ones.  Wall-clock ns per wave-instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define R8(X) X X X X X X X X
#define OPS(S) asm volatile(S : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : "v"(b), "v"(c), "s"(sb) : "vcc", "scc");
#define V8(INS) INS("%0") INS("%1") INS("%2") INS("%3") INS("%4") INS("%5") INS("%6") INS("%7")

template <int OP>
__global__ __launch_bounds__(256) void k_rate(unsigned long long *out, int iters, uint32_t seed)
{
    uint32_t a0 = threadIdx.x * 3 + seed, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    uint32_t b = seed | 1, c = seed + 77;
    uint32_t s0 = seed, s1 = seed + 1, s2 = seed + 2, s3 = seed + 3, sb = seed + 9;
    for (int it = 0; it < iters; ++it) {
        if constexpr (OP == 0) {
#define A0(r) "v_add_u32 " r ", " r ", %12\n"
            OPS(R8(V8(A0)))
        } else if constexpr (OP == 1) {
#define A1(r) "v_add_u32_e64 " r ", " r ", %12\n"
            OPS(R8(V8(A1)))
        } else if constexpr (OP == 2) {
#define A2(r) "v_and_b32 " r ", " r ", %12\n"
            OPS(R8(V8(A2)))
        } else if constexpr (OP == 3) {
#define A3(r) "v_max_u32 " r ", " r ", %12\n"
            OPS(R8(V8(A3)))
        } else if constexpr (OP == 4) {
#define A4(r) "v_min_i32 " r ", " r ", %12\n"
            OPS(R8(V8(A4)))
        } else if constexpr (OP == 5) {
#define A5(r) "v_lshlrev_b32 " r ", 1, " r "\n"
            OPS(R8(V8(A5)))
        } else if constexpr (OP == 6) {
#define A6(r) "v_sub_u32 " r ", " r ", %12\n"
            OPS(R8(V8(A6)))
        } else if constexpr (OP == 7) {
#define A7(r) "v_cndmask_b32 " r ", " r ", %12, vcc\n"
            OPS(R8(V8(A7)))
        } else if constexpr (OP == 8) {
#define A8(r) "v_mov_b32 " r ", %12\n"
            OPS(R8(V8(A8)))
        } else if constexpr (OP == 9) {
#define A9(r) "v_max_u16 " r ", " r ", %12\n"
            OPS(R8(V8(A9)))
        } else if constexpr (OP == 10) {
#define A10(r) "v_mul_u32_u24 " r ", " r ", %12\n"
            OPS(R8(V8(A10)))
        } else if constexpr (OP == 11) {
#define A11(r) "v_min3_i32 " r ", " r ", %12, %13\n"
            OPS(R8(V8(A11)))
        } else if constexpr (OP == 12) {
#define A12(r) "v_add3_u32 " r ", " r ", %12, %13\n"
            OPS(R8(V8(A12)))
        } else if constexpr (OP == 13) {
#define A13(r) "v_and_or_b32 " r ", " r ", %12, %13\n"
            OPS(R8(V8(A13)))
        } else if constexpr (OP == 14) {
#define A14(r) "v_cmp_gt_u32 vcc, " r ", %12\n"
            OPS(R8(V8(A14)))
        } else if constexpr (OP == 15) {
#define A15(r) "v_cmp_gt_u32_e64 s[20:21], " r ", %12\n"
            asm volatile(R8(V8(A15)) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : "v"(b), "v"(c), "s"(sb) : "vcc", "scc", "s20", "s21");
        } else if constexpr (OP == 16) {   // SALU only
            OPS(R8("s_and_b32 %8, %8, %14\n s_add_u32 %9, %9, %14\n s_and_b32 %10, %10, %14\n s_add_u32 %11, %11, %14\n"
                   "s_and_b32 %8, %8, %14\n s_add_u32 %9, %9, %14\n s_and_b32 %10, %10, %14\n s_add_u32 %11, %11, %14\n"))
        } else if constexpr (OP == 17) {   // VOP3 : SALU = 1 : 1
#define A17(r) "v_min3_i32 " r ", " r ", %12, %13\n s_add_u32 %8, %8, %14\n"
            OPS(R8(V8(A17)))
        } else if constexpr (OP == 18) {   // VOP3 : SALU = 2 : 1 (64 VALU + 32 SALU per block)
#define A18(r) "v_min3_i32 " r ", " r ", %12, %13\n"
            OPS(R8(A18("%0") A18("%1") "s_add_u32 %8, %8, %14\n" A18("%2") A18("%3") "s_and_b32 %9, %9, %14\n" A18("%4") A18("%5") "s_add_u32 %10, %10, %14\n" A18("%6") A18("%7") "s_and_b32 %11, %11, %14\n"))
        } else if constexpr (OP == 19) {   // VOP2 : SALU = 1 : 1
#define A19(r) "v_add_u32 " r ", " r ", %12\n s_add_u32 %8, %8, %14\n"
            OPS(R8(V8(A19)))
        } else if constexpr (OP == 20) {
#define A20(r) "v_pk_sub_u16 " r ", " r ", %12 clamp\n"
            OPS(R8(V8(A20)))
        } else if constexpr (OP == 21) {
#define A21(r) "v_min_u16 " r ", " r ", %12\n"
            OPS(R8(V8(A21)))
        } else if constexpr (OP == 22) {
#define A22(r) "v_add_f32 " r ", " r ", %12\n"
            OPS(R8(V8(A22)))
        } else if constexpr (OP == 23) {
#define A23(r) "v_fma_f32 " r ", " r ", %12, %13\n"
            OPS(R8(V8(A23)))
        } else if constexpr (OP == 24) {
#define A24(r) "v_mul_f32 " r ", " r ", %12\n"
            OPS(R8(V8(A24)))
        } else if constexpr (OP == 25) {
#define A25(r) "v_cvt_f32_i32 " r ", " r "\n"
            OPS(R8(V8(A25)))
        } else if constexpr (OP == 26) {
#define A26(r) "v_xor_b32 " r ", " r ", %14\n"
            OPS(R8(V8(A26)))
        } else if constexpr (OP == 27) {
#define A27(r) "v_add_u32_dpp " r ", " r ", " r " row_shr:1 row_mask:0xf bank_mask:0xf\n"
            OPS(R8(V8(A27)))
        } else if constexpr (OP == 28) {
#define A28(r) "v_add_u32_sdwa " r ", " r ", %12 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n"
            OPS(R8(V8(A28)))
        } else if constexpr (OP == 29) {
#define A29(r) "v_max_i16 " r ", " r ", %12\n"
            OPS(R8(V8(A29)))
        }
    }
    uint32_t r = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ s0 ^ s1 ^ s2 ^ s3;
    if (r == 0x12345u) out[0] = r;
}

template <int OP>
void run(const char *name, unsigned long long *d_out, int per_iter)
{
    const int iters = 1024;
    printf("%-34s", name);
    for (int wps = 2; wps <= 8; wps *= 2) {
        const int blocks = 256 * wps;
        hipLaunchKernelGGL(k_rate<OP>, dim3(blocks), dim3(256), 0, 0, d_out, 8, 12345u);
        (void)hipDeviceSynchronize();
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k_rate<OP>, dim3(blocks), dim3(256), 0, 0, d_out, iters, 12345u);
        (void)hipEventRecord(e1);
        (void)hipDeviceSynchronize();
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        const double n = (double)iters * per_iter;
        printf("  w%d: %6.3f ns/ins/simd", wps, ms * 1e6 / (n * wps));
    }
    printf("\n"); fflush(stdout);
}

int main()
{
    unsigned long long *d_out;
    (void)hipMalloc(&d_out, 64);
    run<0>("v_add_u32 (VOP2)", d_out, 64);
    run<1>("v_add_u32_e64 (VOP3)", d_out, 64);
    run<2>("v_and_b32", d_out, 64);
    run<26>("v_xor_b32 v, v, sgpr", d_out, 64);
    run<3>("v_max_u32", d_out, 64);
    run<4>("v_min_i32", d_out, 64);
    run<5>("v_lshlrev_b32", d_out, 64);
    run<6>("v_sub_u32", d_out, 64);
    run<7>("v_cndmask_b32", d_out, 64);
    run<8>("v_mov_b32", d_out, 64);
    run<9>("v_max_u16 (VOP2)", d_out, 64);
    run<21>("v_min_u16 (VOP2)", d_out, 64);
    run<29>("v_max_i16 (VOP2)", d_out, 64);
    run<10>("v_mul_u32_u24 (VOP2)", d_out, 64);
    run<11>("v_min3_i32", d_out, 64);
    run<12>("v_add3_u32", d_out, 64);
    run<13>("v_and_or_b32", d_out, 64);
    run<20>("v_pk_sub_u16 clamp", d_out, 64);
    run<14>("v_cmp_gt_u32 vcc (VOPC)", d_out, 64);
    run<15>("v_cmp_gt_u32_e64 sgpr", d_out, 64);
    run<22>("v_add_f32", d_out, 64);
    run<24>("v_mul_f32", d_out, 64);
    run<23>("v_fma_f32", d_out, 64);
    run<25>("v_cvt_f32_i32", d_out, 64);
    run<27>("v_add_u32_dpp row_shr", d_out, 64);
    run<28>("v_add_u32_sdwa", d_out, 64);
    run<16>("SALU only (s_and/s_add)", d_out, 64);
    run<17>("v_min3 + s_add 1:1 (per pair)", d_out, 64);
    run<18>("v_min3 x2 + salu (per 2v+1s)", d_out, 32);
    run<19>("v_add_u32 + s_add 1:1 (per pair)", d_out, 64);
    return 0;
}
