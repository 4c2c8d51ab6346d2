#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* o, const int* in)
{
    int a = in[0], b = in[1], c = in[2], d, e; int accs = 1000;
    asm volatile("v_dot4_i32_i8 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(accs));
    asm volatile("v_dot2_i32_i16 %0, %1, %2, %3" : "=v"(e) : "v"(a), "v"(b), "v"(c));
    o[0] = d; o[1] = e;
}
int main()
{
    int h[3] = { (int)0x04FD02FFu, (int)0x0A0B0C0Du, 7 }; int* o; int* d; hipMalloc(&o, 256); hipMalloc(&d, 12);
    hipMemcpy(d, h, 12, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o, d); int r[2]; hipMemcpy(r, o, 8, hipMemcpyDeviceToHost);
    // dot4: bytes a = {-1, 2, -3, 4}, b = {13, 12, 11, 10}: -13 + 24 - 33 + 40 = 18 (+1000)
    // dot2: a = {0x02FF=767, 0x04FD=1277}, b = {0x0C0D=3085, 0x0A0B=2571}: 767*3085 + 1277*2571 = 2366195 + 3283167 = 5649362 (+7)
    printf("dot4 %d (expect 1018)  dot2 %d (expect 5649369)\n", r[0], r[1]);
}
