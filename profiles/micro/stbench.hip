// Store-pattern micro-benchmark: a 2-D tiled writer (like kern_planes) with W bytes per row segment per workgroup.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template<int BYTES> __global__ __launch_bounds__(256) void k(char* out, int tileW, int tileH, int stride, int rows, size_t planeBytes, int nplanes)
{
    // tile (blockIdx.x, blockIdx.y) of tileW bytes x tileH rows; every thread writes BYTES bytes per (row, plane) it owns
    const int lanesPerRow = tileW / BYTES, rowsPerPass = 256 / lanesPerRow;
    const int lx = threadIdx.x % lanesPerRow, ly = threadIdx.x / lanesPerRow;
    const size_t x = (size_t)blockIdx.x * tileW + (size_t)lx * BYTES;
    for (int p = 0; p < nplanes; p++)
        for (int y = ly; y < tileH; y += rowsPerPass)
        {
            char* a = out + (size_t)p * planeBytes + ((size_t)blockIdx.y * tileH + y) * stride + x;
            if (BYTES == 4) *(uint32_t*)a = threadIdx.x + p;
            else if (BYTES == 8) *(uint2*)a = make_uint2(threadIdx.x, p);
            else *(uint4*)a = make_uint4(threadIdx.x, p, y, 0);
        }
}
int main()
{
    const int stride = 2112 * 2, rows = 10240 / 2;           // 21.6 MB planes
    const size_t planeBytes = (size_t)stride * rows; const int nplanes = 16;
    char* d; hipMalloc(&d, planeBytes * nplanes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    struct P { int bytes, tw, th; } ps[] = { {4, 64, 16}, {4, 128, 8}, {4, 256, 4}, {8, 128, 16}, {8, 256, 8}, {16, 256, 16}, {16, 512, 8}, {4, 64, 64}, {16, 1024, 4}, {8, 64, 16}, {16, 64, 16}, {8, 64, 32}, {16, 64, 64} };
    for (auto& q : ps)
    {
        float ms = 0;
        dim3 grid(stride / q.tw, rows / q.th);
        for (int rep = 0; rep < 3; rep++)
        {
            hipEventRecord(e0);
            if (q.bytes == 4) hipLaunchKernelGGL(k<4>, grid, dim3(256), 0, 0, d, q.tw, q.th, stride, rows, planeBytes, nplanes);
            else if (q.bytes == 8) hipLaunchKernelGGL(k<8>, grid, dim3(256), 0, 0, d, q.tw, q.th, stride, rows, planeBytes, nplanes);
            else hipLaunchKernelGGL(k<16>, grid, dim3(256), 0, 0, d, q.tw, q.th, stride, rows, planeBytes, nplanes);
            hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        }
        printf("bytes/lane %2d tile %4d B x %2d rows : %.3f ms  %.0f GB/s\n", q.bytes, q.tw, q.th, ms, planeBytes * nplanes / (ms * 1e-3) / 1e9);
    }
    return 0;
}
