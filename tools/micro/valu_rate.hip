// Issue-rate microbenchmark for gfx950 (development tool, not part of the library):
// cycles per wave64 instruction per SIMD for the integer / packed / LDS instructions the ORB kernels lean on,
// at 1, 2, 4 and 8 resident waves per SIMD.  Build: hipcc -O3 --offload-arch=gfx950 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>

#define REP8(X) X X X X X X X X
template <int OP>
__global__ __launch_bounds__(256) void k_rate(unsigned long long *out, int iters, uint32_t seed)
{
    __shared__ uint32_t sh[4096];
    uint32_t a0 = threadIdx.x * 3 + seed, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    uint32_t b = seed | 1, c = seed + 77;
    for (int i = threadIdx.x; i < 4096; i += 256) sh[i] = i * seed;
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#define OPS(INS) \
        asm volatile(INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7) \
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc", "scc");
        if constexpr (OP == 0) {
#define I0(n) "v_add_u32 %" #n ", %" #n ", %8\n"
            REP8(OPS(I0))
        } else if constexpr (OP == 1) {
#define I1(n) "v_min3_i32 %" #n ", %" #n ", %8, %9\n"
            REP8(OPS(I1))
        } else if constexpr (OP == 2) {
#define I2(n) "v_pk_add_u16 %" #n ", %" #n ", %8 clamp\n"
            REP8(OPS(I2))
        } else if constexpr (OP == 3) {
#define I3(n) "v_perm_b32 %" #n ", %" #n ", %8, %9\n"
            REP8(OPS(I3))
        } else if constexpr (OP == 4) {
#define I4(n) "v_dot4_u32_u8 %" #n ", %" #n ", %8, %9\n"
            REP8(OPS(I4))
        } else if constexpr (OP == 5) {
#define I5(n) "v_mad_i32_i24 %" #n ", %" #n ", %8, %9\n"
            REP8(OPS(I5))
        } else if constexpr (OP == 6) {
#define I6(n) "v_mul_lo_u32 %" #n ", %" #n ", %8\n"
            REP8(OPS(I6))
        } else if constexpr (OP == 7) {
#define I7(n) "v_pk_max_u16 %" #n ", %" #n ", %8\n"
            REP8(OPS(I7))
        } else if constexpr (OP == 8) {
#define I8(n) "v_cmp_gt_u32 vcc, %" #n ", %8\n"
            REP8(OPS(I8))
        } else if constexpr (OP == 9) {
#define I9(n) "v_dot2_u32_u16 %" #n ", %" #n ", %8, %9\n"
            REP8(OPS(I9))
        } else if constexpr (OP == 10) {
#define I10(n) "v_alignbyte_b32 %" #n ", %" #n ", %8, 3\n"
            REP8(OPS(I10))
        } else if constexpr (OP == 11) {
#define I11(n) "v_mbcnt_lo_u32_b32 %" #n ", %8, %" #n "\n"
            REP8(OPS(I11))
        } else if constexpr (OP == 12) {   // 4 VALU + 1 SALU interleaved (scalar unit beside vector)
#define I12(n) "v_add_u32 %" #n ", %" #n ", %8\n s_and_b64 vcc, vcc, exec\n"
            REP8(OPS(I12))
        } else if constexpr (OP == 13) {   // ds_read_u8 stream (byte gathers of the score phase)
            uint32_t ad = (a0 * 37) & 16380;
#define I13(n) "ds_read_u8 %" #n ", %8 offset:" #n "\n"
            asm volatile(REP8(I13(0) I13(1) I13(2) I13(3) I13(4) I13(5) I13(6) I13(7)) "s_waitcnt lgkmcnt(0)\n"
                         : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3), "=v"(a4), "=v"(a5), "=v"(a6), "=v"(a7) : "v"(ad) : "memory");
        } else if constexpr (OP == 14) {   // ds_read_b32 stream
            uint32_t ad = (threadIdx.x * 4) & 16380;
#define I14(n) "ds_read_b32 %" #n ", %8 offset:" #n "*4\n"
            asm volatile(REP8(I14(0) I14(1) I14(2) I14(3) I14(4) I14(5) I14(6) I14(7)) "s_waitcnt lgkmcnt(0)\n"
                         : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3), "=v"(a4), "=v"(a5), "=v"(a6), "=v"(a7) : "v"(ad) : "memory");
        } else if constexpr (OP == 15) {   // v_max3 + v_min3 mix as in the score network on i16? plain i32
#define I15(n) "v_max3_i32 %" #n ", %" #n ", %8, %9\n"
            REP8(OPS(I15))
        } else if constexpr (OP == 16) {
#define I16(n) "v_sad_u8 %" #n ", %" #n ", %8, %9\n"
            REP8(OPS(I16))
        } else if constexpr (OP == 17) {
#define I17(n) "v_pk_min_i16 %" #n ", %" #n ", %8\n"
            REP8(OPS(I17))
        } else if constexpr (OP == 18) {
#define I18(n) "v_bfe_u32 %" #n ", %" #n ", 8, 8\n"
            REP8(OPS(I18))
        } else if constexpr (OP == 19) {
#define I19(n) "v_fma_f64 %[d" #n "], %[d" #n "], %[e], %[e]\n"
            double d0 = a0, d1 = a1, d2 = a2, d3 = a3, e = 1.0000001;
            asm volatile(REP8("v_fma_f64 %0, %0, %4, %4\n v_fma_f64 %1, %1, %4, %4\n v_fma_f64 %2, %2, %4, %4\n v_fma_f64 %3, %3, %4, %4\n"
                              "v_fma_f64 %0, %0, %4, %4\n v_fma_f64 %1, %1, %4, %4\n v_fma_f64 %2, %2, %4, %4\n v_fma_f64 %3, %3, %4, %4\n")
                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(e));
            a0 = (uint32_t)d0; a1 = (uint32_t)d1; a2 = (uint32_t)d2; a3 = (uint32_t)d3;
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    uint32_t r = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
    if (r == 0x12345u) out[1 << 20] = r;
    if ((threadIdx.x & 63) == 0) out[(blockIdx.x * 4 + (threadIdx.x >> 6))] = t1 - t0;
}

template <int OP>
void run(const char *name, unsigned long long *d_out, int per_iter)
{
    const int iters = 256;
    printf("%-28s", name);
    for (int wps = 1; wps <= 8; wps *= 2) {
        // wps waves per SIMD: blocks of 256 threads = 1 wave per SIMD; wps blocks per CU, 256 CUs
        const int blocks = 256 * wps;
        hipLaunchKernelGGL(k_rate<OP>, dim3(blocks), dim3(256), 0, 0, d_out, iters, 12345u);
        hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_rate<OP>, dim3(blocks), dim3(256), 0, 0, d_out, iters, 12345u);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(blocks * 4);
        hipMemcpy(h.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        const double cyc = (double)h[h.size() / 2];
        const double n = (double)iters * per_iter;
        // s_memtime ticks at a fixed 100 MHz? report both: ticks per instr per wave, and wall ns per instr per SIMD
        printf("  w%d: %6.2f tick/ins/wave %6.3f ns/ins/simd", wps, cyc / n, ms * 1e6 / (n * wps));
    }
    printf("\n"); fflush(stdout);
}

int main()
{
    unsigned long long *d_out;
    hipMalloc(&d_out, ((1 << 20) + 8) * 8);
    run<0>("v_add_u32", d_out, 64);
    run<1>("v_min3_i32", d_out, 64);
    run<15>("v_max3_i32", d_out, 64);
    run<2>("v_pk_add_u16 clamp", d_out, 64);
    run<7>("v_pk_max_u16", d_out, 64);
    run<17>("v_pk_min_i16", d_out, 64);
    run<3>("v_perm_b32", d_out, 64);
    run<4>("v_dot4_u32_u8", d_out, 64);
    run<9>("v_dot2_u32_u16", d_out, 64);
    run<5>("v_mad_i32_i24", d_out, 64);
    run<6>("v_mul_lo_u32", d_out, 64);
    run<8>("v_cmp_gt_u32 -> vcc", d_out, 64);
    run<10>("v_alignbyte_b32", d_out, 64);
    run<11>("v_mbcnt_lo", d_out, 64);
    run<16>("v_sad_u8", d_out, 64);
    run<18>("v_bfe_u32", d_out, 64);
    run<12>("v_add_u32 + s_and_b64 (pairs)", d_out, 64);
    run<13>("ds_read_u8", d_out, 64);
    run<14>("ds_read_b32", d_out, 64);
    run<19>("v_fma_f64", d_out, 64);
    return 0;
}
