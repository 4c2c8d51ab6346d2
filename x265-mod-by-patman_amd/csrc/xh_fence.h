// xh_fence.h -- every device allocation of the library goes through xh::dev_alloc / xh::dev_free.
//
// Release build: hipMalloc / hipFree, nothing else.
//
// Fence build (`make FENCE=1 OUT=<dir>/ OBJ=<dir>/obj`, -DX265HIP_FENCE): an electric fence for device memory.  Each allocation gets its own virtual-address
// reservation (hipMemAddressReserve) of which only the middle is backed by memory (hipMemCreate / hipMemMap); the granules before and after stay unmapped for the life of
// the process.  The block handed out ENDS at the last mapped byte (X265HIP_FENCE=end, the default: any read or write past the end -- by one element -- is a page fault at
// once, "Memory access fault by GPU ... Page not present", the address names the block) or STARTS at the first one (X265HIP_FENCE=start: under-runs).  Every block and
// every kernel launch (XH_LAUNCH_CHECK, synchronous in this build) is written to stderr or to X265HIP_FENCE_LOG, so the last launch line of a dead run is the faulting
// kernel and the fault address falls into one block's guard.  tools/fence_report.py maps one onto the other.  This is what AddressSanitizer would do for device heap
// blocks; the image has the device ASan bitcode but no ASan build of the HIP runtime (/opt/rocm/lib/asan), so the blocks' shadow would never be poisoned.
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>

namespace xh {

#ifndef X265HIP_FENCE
inline hipError_t dev_alloc(void** p, size_t bytes, const char* /*tag*/) { return hipMalloc(p, bytes); }
inline hipError_t dev_free(void* p) { return hipFree(p); }
inline void launch_note(const char*, int) {}
inline const char* alloc_tag(const char*, int) { return ""; }
inline hipError_t dev_alloc_pooled(void** p, size_t bytes, const char*) { return hipMalloc(p, bytes); }
inline void dev_free_pooled(void* p) { (void)hipFree(p); }
constexpr bool kFence = false;
#else
hipError_t dev_alloc(void** p, size_t bytes, const char* tag);
hipError_t dev_free(void* p);
void launch_note(const char* file, int line);     // log the launch, then wait for it (a fault is then this launch's)
hipError_t dev_alloc_pooled(void** p, size_t bytes, const char* tag);      // blocks that come and go by the hundred thousand (the slot path's staging blocks): kept mapped per size
void dev_free_pooled(void* p);
const char* alloc_tag(const char* file, int line);      // "file:line" of an allocation's CALLER (the per-object alloc() helpers pass __builtin_FILE / __builtin_LINE)
constexpr bool kFence = true;
#endif

} // namespace xh
#define XH_ALLOC_TAG_2(f, l) f ":" #l
#define XH_ALLOC_TAG_1(f, l) XH_ALLOC_TAG_2(f, l)
#define XH_ALLOC_TAG XH_ALLOC_TAG_1(__FILE__, __LINE__)
