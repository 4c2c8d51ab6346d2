// L1/TA micro-benchmark: cost of one wave-wide load instruction for the address patterns the ME kernels use.
// build: hipcc --offload-arch=gfx950 -O3 -o l1bench l1bench.hip ; usage: l1bench  (prints cycles per wave-instruction per CU at full occupancy)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
template<int BYTES> __device__ __forceinline__ uint32_t ld(const char* p)
{
    if (BYTES == 4) { uint32_t a; __builtin_memcpy(&a, p, 4); return a; }
    if (BYTES == 8) { uint2 a; __builtin_memcpy(&a, p, 8); return a.x ^ a.y; }
    if (BYTES == 12) { struct { uint32_t x, y, z; } a; __builtin_memcpy(&a, __builtin_assume_aligned(p, 4), 12); return a.x ^ a.y ^ a.z; }
    uint4 a; __builtin_memcpy(&a, p, 16); return a.x ^ a.y ^ a.z ^ a.w;
}
template<int BYTES>
__global__ __launch_bounds__(256) void k(const char* base, int lanesPerRow, int rowStride, int misalign, int iters, int rowsSpan, uint32_t* out)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = lane / lanesPerRow, col = lane % lanesPerRow;
    const char* p = base + (size_t)(blockIdx.x % 8) * 4096 + wave * 64 + (size_t)row * rowStride + col * BYTES + misalign;
    uint32_t acc = 0;
    for (int i = 0; i < iters; i++)
    {
#pragma unroll
        for (int u = 0; u < 8; u++) acc ^= ld<BYTES>(p + (size_t)((i * 8 + u) % rowsSpan) * rowStride);
    }
    if (acc == 0x12345678) out[0] = acc;
}
int main()
{
    const size_t N = 64u << 20;
    char* d; uint32_t* o; hipMalloc(&d, N); hipMalloc(&o, 4); hipMemset(d, 1, N);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000, blocks = 256 * 8;
    struct P { int bytes, lpr, stride, mis; };
    std::vector<P> ps;
    for (int bytes : {8, 12, 16}) for (int lpr : {16, 4, 2, 1}) for (int mis : {0, 2, 4, 8, 12}) ps.push_back({bytes, lpr, 2112, mis});
    for (auto& q : ps)
    {
        float ms = 0;
        for (int rep = 0; rep < 2; rep++)
        {
            hipEventRecord(e0);
            if (q.bytes == 4) hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(256), 0, 0, d, q.lpr, q.stride, q.mis, iters, 4, o);
            else if (q.bytes == 8) hipLaunchKernelGGL(k<8>, dim3(blocks), dim3(256), 0, 0, d, q.lpr, q.stride, q.mis, iters, 4, o);
            else if (q.bytes == 12) hipLaunchKernelGGL(k<12>, dim3(blocks), dim3(256), 0, 0, d, q.lpr, q.stride, q.mis, iters, 4, o);
            else hipLaunchKernelGGL(k<16>, dim3(blocks), dim3(256), 0, 0, d, q.lpr, q.stride, q.mis, iters, 4, o);
            hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        }
        // wave-instructions per CU: blocks*4 waves * iters*8 / 256 CUs
        const double instrPerCU = (double)blocks * 4 * iters * 8 / 256.0;
        printf("bytes/lane %2d lanes/row %2d misalign %d : %.1f ns/instr/CU  (~%.1f clk @2.4GHz)  %.0f GB/s/CU-agg\n", q.bytes, q.lpr, q.mis,
               ms * 1e6 / instrPerCU, ms * 1e6 / instrPerCU * 2.4, (double)blocks * 4 * iters * 8 * 64 * q.bytes / (ms * 1e-3) / 1e9);
    }
    return 0;
}
