// valu_rate.hip -- issue rate of the search kernels' own vector instructions on gfx950: v_sad_u16, v_sad_u8, v_alignbyte, v_add_u32,
// as wave64 instructions per second over the whole chip, for 1, 2, 4 and 8 waves per SIMD.  This is the denominator of bench.py's
// roofline_valu (N_SIMD x clock / VALU_CYCLES): build with  hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template<int OP> __global__ __launch_bounds__(256) void rate_kernel(uint32_t* out, int iters, uint32_t seed)
{
    uint32_t a[8], s = seed + threadIdx.x;
#pragma unroll
    for (int i = 0; i < 8; i++) a[i] = s * (i + 1);
    for (int it = 0; it < iters; it++)
    {
#pragma unroll
        for (int u = 0; u < 8; u++)          // 8 x 8 = 64 independent-enough instructions per iteration
#pragma unroll
            for (int i = 0; i < 8; i++)
            {
                if (OP == 0) a[i] = __builtin_amdgcn_sad_u16(a[i], s, a[(i + 1) & 7]);
                if (OP == 1) a[i] = __builtin_amdgcn_sad_u8(a[i], s, a[(i + 1) & 7]);
                if (OP == 2) a[i] = __builtin_amdgcn_alignbyte(a[i], a[(i + 1) & 7], s & 3);
                if (OP == 3) a[i] = a[i] + a[(i + 1) & 7];
            }
    }
    uint32_t r = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) r ^= a[i];
    if (r == 0x12345) out[threadIdx.x] = r;
}

template<int OP> void run(const char* name, uint32_t* d)
{
    for (int wps : { 1, 2, 4, 8 })
    {
        const int blocks = 256 * wps, iters = 20000;      // 256 threads = 4 waves per block -> one wave per SIMD per resident block
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(rate_kernel<OP>, dim3(blocks), dim3(256), 0, 0, d, 100, 1u);
        hipEventRecord(e0);
        hipLaunchKernelGGL(rate_kernel<OP>, dim3(blocks), dim3(256), 0, 0, d, iters, 1u);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double insts = (double)blocks * 4 * iters * 64;
        printf("%-12s waves/SIMD %d: %.1f G wave-instr/s  (%.2f clocks per instruction per SIMD at 2.4 GHz)\n", name, wps, insts / (ms * 1e-3) / 1e9,
               1024 * 2.4e9 / (insts / (ms * 1e-3)));
    }
}

int main()
{
    uint32_t* d; hipMalloc(&d, 4096);
    run<0>("v_sad_u16", d); run<1>("v_sad_u8", d); run<2>("v_alignbyte", d); run<3>("v_add_u32", d);
    return 0;
}
