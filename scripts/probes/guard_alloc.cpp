// Guard-page device allocator for out-of-bounds hunts: plugs into PyTorch through torch.cuda.memory.CUDAPluggableAllocator
// (scripts/guard_run.py).  Every allocation gets its OWN virtual-address reservation: whole granules of physical memory mapped at the front, one
// granule of UNMAPPED address space behind them, and the tensor is placed so that it ENDS where the mapping ends (FVK_GUARD_MODE=front: so that it
// STARTS where the mapping starts, with the previous reservation's unmapped granule in front of it).  A read or write past the end (or in front of
// the start) of ANY tensor — by 16 bytes or by gigabytes — is then a GPU page fault ("Memory access fault by GPU ... address 0x...") instead of a
// silent read of a mapped neighbour, which is what the caching allocator turns every small overrun into.  A freed tensor's ADDRESS RANGE is never
// handed out again (the first version unmapped and re-reserved at once: the second test of every file then read stale data through re-used
// virtual addresses, profiles/r06d_guard_page_runs.md); its physical memory goes to a quarantine that is unmapped oldest-first beyond
// FVK_GUARD_QUARANTINE_GB (default 48), so a late use after free faults as well.
//   hipcc -O2 -shared -fPIC scripts/probes/guard_alloc.cpp -o scripts/probes/libguard_alloc.so
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <deque>
#include <mutex>
#include <unordered_map>

namespace {
struct Rec {
    void* base;
    size_t reserved, mapped;
    hipMemGenericAllocationHandle_t h;
};
std::unordered_map<void*, Rec> g_live;
std::deque<Rec> g_quarantine;
size_t g_quarantine_bytes = 0, g_quarantine_cap = 48UL << 30;
std::mutex g_mu;
size_t g_gran = 0, g_align = 16;
bool g_front = false;
long g_allocs = 0, g_bytes = 0;

#define GCHECK(x)                                                                                       \
    do {                                                                                                \
        hipError_t e_ = (x);                                                                            \
        if (e_ != hipSuccess) {                                                                         \
            fprintf(stderr, "[guard_alloc] %s failed: %s\n", #x, hipGetErrorString(e_));                \
            abort();                                                                                    \
        }                                                                                               \
    } while (0)
}  // namespace

extern "C" void* guard_malloc(ssize_t size, int device, hipStream_t) {
    if (size <= 0) size = 1;
    hipMemAllocationProp prop{};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = device;
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_gran) {
        GCHECK(hipMemGetAllocationGranularity(&g_gran, &prop, hipMemAllocationGranularityMinimum));
        if (const char* a = getenv("FVK_GUARD_ALIGN")) g_align = (size_t)atol(a);
        if (const char* m = getenv("FVK_GUARD_MODE")) g_front = m[0] == 'f';
        if (const char* q = getenv("FVK_GUARD_QUARANTINE_GB")) g_quarantine_cap = (size_t)atol(q) << 30;
        fprintf(stderr, "[guard_alloc] granule %zu B, tensor alignment %zu B, tensors placed at the %s of their mapping\n", g_gran, g_align,
                g_front ? "START" : "END");
    }
    const size_t need = ((size_t)size + g_align - 1) / g_align * g_align;
    const size_t mapped = (need + g_gran - 1) / g_gran * g_gran;
    const size_t reserved = mapped + g_gran;  // one unmapped granule behind the mapping
    void* base = nullptr;
    GCHECK(hipMemAddressReserve(&base, reserved, 0, nullptr, 0));
    Rec r{base, reserved, mapped, {}};
    GCHECK(hipMemCreate(&r.h, mapped, &prop, 0));
    GCHECK(hipMemMap(base, mapped, 0, r.h, 0));
    hipMemAccessDesc acc{};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    GCHECK(hipMemSetAccess(base, mapped, &acc, 1));
    void* p = g_front ? base : (void*)((char*)base + (mapped - need));
    g_live[p] = r;
    ++g_allocs;
    g_bytes += size;
    return p;
}

extern "C" void guard_free(void* p, ssize_t, int, hipStream_t stream) {
    if (!p) return;
    (void)stream;
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_live.find(p);
    if (it == g_live.end()) {
        fprintf(stderr, "[guard_alloc] free of unknown pointer %p\n", p);
        return;
    }
    g_quarantine.push_back(it->second);
    g_quarantine_bytes += it->second.mapped;
    g_live.erase(it);
    if (g_quarantine_bytes > g_quarantine_cap) {
        GCHECK(hipDeviceSynchronize());  // nothing in flight may still touch what is unmapped now
        while (g_quarantine_bytes > g_quarantine_cap / 2 && !g_quarantine.empty()) {
            const Rec r = g_quarantine.front();
            g_quarantine.pop_front();
            g_quarantine_bytes -= r.mapped;
            GCHECK(hipMemUnmap(r.base, r.mapped));
            GCHECK(hipMemRelease(r.h));
            // (the reservation itself is kept for the life of the process: its addresses are never re-used)
        }
    }
}

extern "C" void guard_stats(long* allocs, long* bytes, long* live) {
    std::lock_guard<std::mutex> lk(g_mu);
    *allocs = g_allocs;
    *bytes = g_bytes;
    *live = (long)g_live.size();
}
