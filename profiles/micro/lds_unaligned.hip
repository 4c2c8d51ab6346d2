// lds_unaligned.hip -- do the wide LDS reads (ds_read_b64 / b96 / b128) work at DWORD alignment on gfx950, and what do they cost?
// Each lane reads K bytes at byte address 12 * lane + 4 * (lane & 1) + row * 384 (dword aligned, mostly not 8 / 16 byte aligned), checks the
// values and the kernel is timed over many rounds.  hipcc --offload-arch=gfx950 -O3 lds_unaligned.hip -o lds_unaligned
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template<int MODE> __global__ __launch_bounds__(256) void k(uint32_t* out, int iters, int shift)
{
    __shared__ uint32_t lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = i * 2654435761u;
    __syncthreads();
    const uint32_t addr0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)lds + (threadIdx.x & 63) * 8 + shift;   // 8-byte lane stride + shift
    uint32_t acc = 0, bad = 0;
    for (int it = 0; it < iters; it++)
    {
#pragma unroll
        for (int r = 0; r < 16; r++)
        {
            const uint32_t a = addr0 + r * 528;
            uint32_t x = 0, y = 0, z = 0, w = 0;
            if (MODE == 0) { asm volatile("ds_read2_b32 %0, %2 offset1:1\n ds_read_b32 %1, %2 offset:8\n s_waitcnt lgkmcnt(0)" : "=v"(*(uint64_t*)&x), "=v"(z) : "v"(a)); }
            if (MODE == 1) { uint64_t q; asm volatile("ds_read_b64 %0, %2\n ds_read_b32 %1, %2 offset:8\n s_waitcnt lgkmcnt(0)" : "=v"(q), "=v"(z) : "v"(a)); x = (uint32_t)q; y = (uint32_t)(q >> 32); }
            if (MODE == 2) { typedef uint32_t u3 __attribute__((ext_vector_type(3))); u3 q; asm volatile("ds_read_b96 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(q) : "v"(a)); x = q.x; y = q.y; z = q.z; }
            if (MODE == 3) { typedef uint32_t u4 __attribute__((ext_vector_type(4))); u4 q; asm volatile("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(q) : "v"(a)); x = q.x; y = q.y; z = q.z; w = q.w; }
            if (MODE == 0) { y = (uint32_t)((*(uint64_t*)&x) >> 32); }
            const uint32_t i0 = (a - (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)lds) >> 2;
            bad |= (x != i0 * 2654435761u) | (y != (i0 + 1) * 2654435761u) | (z != (i0 + 2) * 2654435761u) | (MODE == 3 && w != (i0 + 3) * 2654435761u);
            acc += x ^ y ^ z ^ w;
        }
    }
    if (bad) out[0] = 0xBAD;
    if (acc == 0x12345) out[1] = acc;
}

template<int MODE> void run(const char* name, uint32_t* d, int shift)
{
    hipMemset(d, 0, 16);
    const int blocks = 256 * 2, iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 10, shift);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters, shift);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    uint32_t h[2]; hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
    const double reads = (double)blocks * 4 * iters * 16;            // wave-level 12(16)-byte reads
    printf("%-28s shift %d: %s  %.2f clocks per wave-read per CU (2.4 GHz, 8 waves per CU)\n", name, shift, h[0] == 0xBAD ? "WRONG VALUES" : "values ok",
           ms * 1e-3 * 2.4e9 * 256 / reads);
}

int main()
{
    uint32_t* d; hipMalloc(&d, 64);
    for (int shift : { 0, 4 })
    {
        run<0>("ds_read2_b32 + ds_read_b32", d, shift); run<1>("ds_read_b64 + ds_read_b32", d, shift);
        run<2>("ds_read_b96", d, shift); run<3>("ds_read_b128", d, shift);
    }
    return 0;
}
