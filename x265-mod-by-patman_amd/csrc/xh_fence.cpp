// xh_fence.cpp -- the fence build's allocator (see xh_fence.h); empty in a release build.
#include "xh_fence.h"
#ifdef X265HIP_FENCE
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdint>
#include <atomic>
#include <csignal>
#include <mutex>
#include <vector>
#include <unordered_map>
#include <utility>
#include <unistd.h>

namespace xh {
namespace {

struct Block { char* base; size_t reserved, mapped, bytes; char* mapAt; hipMemGenericAllocationHandle_t handle; };
std::mutex g_mu;
std::unordered_map<void*, Block> g_blocks;
std::unordered_map<size_t, std::vector<void*>> g_pool;     // dev_free_pooled: fenced blocks kept mapped for the next request of the same size
FILE* g_log = nullptr;
bool g_startMode = false;
size_t g_align = 16;
unsigned long g_serial = 0;
bool g_sync = true;
// the last launches of the process: written by the SIGABRT handler (the HSA runtime reports a memory fault and calls abort())
struct Note { const char* file; int line; };
constexpr int kRing = 32;
Note g_ring[kRing];
std::atomic<unsigned> g_ringAt{0};

void (*g_prevAbort)(int) = nullptr;
void on_abort(int)
{
    char buf[256];
    const unsigned at = g_ringAt.load();
    for (unsigned i = at > kRing ? at - kRing : 0; i < at; i++)
    {
        const Note& n = g_ring[i % kRing];
        const int len = snprintf(buf, sizeof(buf), "[fence] launch #%u %s:%d%s\n", i, n.file ? n.file : "?", n.line, i + 1 == at ? "   <-- last" : "");
        if (len > 0) { (void)!write(2, buf, (size_t)len); if (g_log && g_log != stderr) (void)!write(fileno(g_log), buf, (size_t)len); }
    }
    // (several copies of this file may live in one process -- the 8-bit and the 16-bit library, the tests' torch allocator shim: each prints its own launches, then hands on)
    signal(SIGABRT, (g_prevAbort && g_prevAbort != SIG_IGN) ? g_prevAbort : SIG_DFL);
    raise(SIGABRT);
}

FILE* log_file()
{
    if (g_log) return g_log;
    const char* path = getenv("X265HIP_FENCE_LOG");
    g_log = path ? fopen(path, "a") : nullptr;
    if (!g_log) g_log = stderr;
    setvbuf(g_log, nullptr, _IOLBF, 0);
    const char* m = getenv("X265HIP_FENCE");
    g_startMode = m && !strcmp(m, "start");
    if (const char* a = getenv("X265HIP_FENCE_ALIGN")) { const long v = atol(a); if (v >= 1 && v <= 4096 && !(v & (v - 1))) g_align = (size_t)v; }
    if (const char* y = getenv("X265HIP_FENCE_SYNC")) g_sync = atoi(y) != 0;
    g_prevAbort = signal(SIGABRT, on_abort);
    fprintf(g_log, "[fence] pid %d mode %s align %zu sync %d\n", (int)getpid(), g_startMode ? "start" : "end", g_align, (int)g_sync);
    return g_log;
}

} // namespace

hipError_t dev_alloc(void** p, size_t bytes, const char* tag)
{
    std::lock_guard<std::mutex> lock(g_mu);
    FILE* log = log_file();
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    hipMemAllocationProp prop{};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = dev;
    size_t gran = 0;
    if ((e = hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum)) != hipSuccess) return e;
    if (!gran) gran = 1 << 21;
    if (!bytes) bytes = 1;
    Block b{};
    b.mapped = (bytes + gran - 1) / gran * gran;
    b.reserved = b.mapped + 2 * gran;                                   // an unmapped granule either side
    void* base = nullptr;
    if ((e = hipMemAddressReserve(&base, b.reserved, gran, nullptr, 0)) != hipSuccess) return e;
    b.base = (char*)base; b.mapAt = b.base + gran;
    if ((e = hipMemCreate(&b.handle, b.mapped, &prop, 0)) != hipSuccess) { (void)hipMemAddressFree(base, b.reserved); return e; }
    if ((e = hipMemMap(b.mapAt, b.mapped, 0, b.handle, 0)) != hipSuccess) { (void)hipMemRelease(b.handle); (void)hipMemAddressFree(base, b.reserved); return e; }
    hipMemAccessDesc acc{};
    acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    if ((e = hipMemSetAccess(b.mapAt, b.mapped, &acc, 1)) != hipSuccess) { (void)hipMemUnmap(b.mapAt, b.mapped); (void)hipMemRelease(b.handle); (void)hipMemAddressFree(base, b.reserved); return e; }
    const size_t padded = (bytes + g_align - 1) / g_align * g_align;
    char* user = g_startMode ? b.mapAt : b.mapAt + b.mapped - padded;
    // the rest of the mapped range holds a pattern no search result, pixel or coefficient looks like; a kernel that reads it produces loud garbage instead of plausible zeros
    (void)hipMemset(b.mapAt, 0xA5, b.mapped);
    (void)hipDeviceSynchronize();           // the fill runs on the null stream: it must not land behind a copy the caller queues on its own (non-blocking) stream
    b.bytes = bytes;
    g_blocks[user] = b;
    fprintf(log, "[fence] alloc #%lu %p..%p (%zu B) mapped %p..%p tag %s\n", g_serial++, (void*)user, (void*)(user + bytes), bytes, (void*)b.mapAt, (void*)(b.mapAt + b.mapped), tag);
    *p = user;
    return hipSuccess;
}

hipError_t dev_free(void* p)
{
    if (!p) return hipSuccess;
    std::lock_guard<std::mutex> lock(g_mu);
    auto it = g_blocks.find(p);
    if (it == g_blocks.end()) return hipErrorInvalidValue;
    const Block b = it->second;
    g_blocks.erase(it);
    (void)hipDeviceSynchronize();
    (void)hipMemUnmap(b.mapAt, b.mapped);
    (void)hipMemRelease(b.handle);
    // the reservation is NOT used again: a later block must never land where a stale pointer still points (use after free = page fault as well).  (Handing reservations back
    // with hipMemAddressFree, or mapping new memory into old ones, made six slot-path tests fault that pass without either: this runtime does not take re-used ranges well.
    // The blocks that come by the hundred thousand -- the slot path's staging blocks -- are pooled instead: dev_alloc_pooled.)
    fprintf(log_file(), "[fence] free %p\n", p);
    return hipSuccess;
}

const char* alloc_tag(const char* file, int line)
{
    static thread_local char buf[160];
    const char* s = strrchr(file, '/');
    snprintf(buf, sizeof(buf), "%s:%d", s ? s + 1 : file, line);
    return buf;
}

// The staging blocks of the slot path (one per host block of every slot call: hundreds of thousands per encode) are not unmapped when the call is over but kept for the
// next request of the SAME size: the block still ends at an unmapped page (the pointer's place depends on the size), no address space is burnt, and the fenced
// suite runs in minutes.  What these blocks lose is the use-after-free check.
hipError_t dev_alloc_pooled(void** p, size_t bytes, const char* tag)
{
    {
        std::lock_guard<std::mutex> lock(g_mu);
        auto it = g_pool.find(bytes ? bytes : 1);
        if (it != g_pool.end() && !it->second.empty()) { *p = it->second.back(); it->second.pop_back(); return hipSuccess; }
    }
    return dev_alloc(p, bytes, tag);
}
void dev_free_pooled(void* p)
{
    if (!p) return;
    std::lock_guard<std::mutex> lock(g_mu);
    auto it = g_blocks.find(p);
    if (it == g_blocks.end()) return;
    g_pool[it->second.bytes].push_back(p);
}

void launch_note(const char* file, int line)
{
    { std::lock_guard<std::mutex> lock(g_mu); (void)log_file(); }
    const char* s = strrchr(file, '/');
    const unsigned at = g_ringAt.fetch_add(1);
    g_ring[at % kRing] = Note{ s ? s + 1 : file, line };
    if (g_sync) (void)hipDeviceSynchronize();
}

} // namespace xh
#endif
