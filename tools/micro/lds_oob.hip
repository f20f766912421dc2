// Development check for gfx950: a DS write past the workgroup's LDS allocation must be dropped, not land in the LDS of
// another workgroup on the same CU (k_fast_cells' capped survivor list relies on it).  Every one-wavefront workgroup fills
// its own 4 KB with a pattern, writes junk to the 8 KB BEHIND its allocation, idles while its neighbours do the same, and
// checks that its own bytes are untouched; out-of-range reads are counted when they return anything but zero.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ __launch_bounds__(64) void k_oob(unsigned long long *bad, int rounds)
{
    extern __shared__ uint32_t lds[];
    const int lane = threadIdx.x, words = 1024;            // 4 KB allocated
    const uint32_t tag = 0x51000000u ^ (blockIdx.x * 2654435761u);
    for (int r = 0; r < rounds; ++r) {
        for (int i = lane; i < words; i += 64) lds[i] = tag + i + r;
        __syncthreads();
        for (int i = lane; i < 2048; i += 64) lds[words + i] = 0xdeadbeefu;            // out of range
        __syncthreads();
        for (volatile int spin = 0; spin < 200; ++spin) { }
        unsigned miss = 0, nonzero = 0;
        for (int i = lane; i < words; i += 64) miss += lds[i] != tag + i + r;
        for (int i = lane; i < 2048; i += 64) nonzero += lds[words + i] != 0u;
        if (miss) atomicAdd(&bad[0], (unsigned long long)miss);
        if (nonzero) atomicAdd(&bad[1], (unsigned long long)nonzero);
        __syncthreads();
    }
}

int main()
{
    unsigned long long *d_bad, h[2] = {0, 0};
    (void)hipMalloc(&d_bad, 16);
    (void)hipMemset(d_bad, 0, 16);
    hipLaunchKernelGGL(k_oob, dim3(256 * 32 * 4), dim3(64), 4096, 0, d_bad, 50);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h, d_bad, 16, hipMemcpyDeviceToHost);
    printf("workgroups %d x 50 rounds: own words corrupted %llu, out-of-range reads non-zero %llu, last error: %s\n", 256 * 32 * 4, h[0], h[1],
           hipGetErrorString(hipGetLastError()));
    return h[0] != 0;
}
