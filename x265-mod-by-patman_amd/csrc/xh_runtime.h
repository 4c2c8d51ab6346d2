// xh_runtime.h -- host-side plumbing for the per-slot (drop-in table) functions.
// Every calling thread gets its own stream + device arena (the reference calls primitives from
// every pool worker concurrently and the slots must be reentrant: SURVEY 8b / threadpool.cpp).
#pragma once
#include "xh_common.h"
#include <vector>

namespace xh {

struct ThreadCtx
{
    hipStream_t stream = nullptr;
    char* arena = nullptr;       // device scratch (bump allocated per slot call; grows on demand)
    size_t arenaSize = 0, arenaUsed = 0;
    std::vector<char*> retired;  // smaller chunks replaced during the current call, freed by the next reset()
    char* zeros = nullptr;       // 256 zero bytes (offsets of single-item batches)
    char* pinned = nullptr;      // small pinned host block for scalar results / offsets
    static ThreadCtx& get();     // creates on first use; fatal() if there is no usable GPU
    void reset();
    void* dalloc(size_t bytes);  // bump allocation, 256-B aligned
    void sync();
};

// A host 2-D block staged into dense device memory (pitch = width elements).
// If the host stride is smaller than the width (the reference's IPFilterHarness uses such
// strides for sources, ipfilterharness.cpp:62-95) the span is copied 1:1 and the host stride kept.
struct DevBlock { void* ptr; intptr_t stride; };
DevBlock stage_in(ThreadCtx& c, const void* host, intptr_t strideElems, int w, int h, int elemSize);
// dense device output block (pitch = w), copied back into a strided host block after the launch
void* stage_out_alloc(ThreadCtx& c, int w, int h, int elemSize);
void stage_out_copy(ThreadCtx& c, void* host, intptr_t strideElems, const void* dev, int w, int h, int elemSize);
const int32_t* dev_zero_offsets(ThreadCtx& c);   // device int32[8] of zeros (offsets of single-item batches)

} // namespace xh
