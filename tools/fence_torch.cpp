// fence_torch.cpp -- the fence allocator (csrc/xh_fence.cpp) behind torch's pluggable device allocator, so that the buffers the TESTS hand to the C ABI (torch tensors) end at
// an unmapped page as well.  Test tooling: tests/conftest.py installs it when X265HIP_FENCE_TORCH names this library (tools/fence_run.sh); the product never loads it.
#define X265HIP_FENCE 1
#include "../x265-mod-by-patman_amd/csrc/xh_fence.cpp"
#include <sys/types.h>

extern "C" __attribute__((visibility("default"))) void* xh_fence_torch_malloc(ssize_t size, int device, hipStream_t)
{
    void* p = nullptr;
    (void)hipSetDevice(device);
    if (xh::dev_alloc(&p, size > 0 ? (size_t)size : 1, "torch") != hipSuccess) return nullptr;
    return p;
}
extern "C" __attribute__((visibility("default"))) void xh_fence_torch_free(void* p, ssize_t, int, hipStream_t)
{
    (void)xh::dev_free(p);
}
