// wr_dma.cpp -- see wr_dma.h.
#include "wr_dma.h"

#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>

#include <mutex>

namespace wrdma {

namespace {

std::once_flag g_once;
bool g_ok = false;
double g_ticks_per_ms = 0;
std::once_flag g_timing_once;
bool g_timing = false;

void init()
{
    // HIP has initialised ROCr already; hsa_init only takes another reference
    if (hsa_init() != HSA_STATUS_SUCCESS) return;
    uint64_t freq = 0;
    if (hsa_system_get_info(HSA_SYSTEM_INFO_TIMESTAMP_FREQUENCY, &freq) == HSA_STATUS_SUCCESS && freq)
        g_ticks_per_ms = (double)freq / 1e3;
    g_ok = true;
}

// owner agent of an allocation ROCr made (device memory, hipHostMalloc); false for anything else.  Pageable host
// memory is unknown to ROCr.  Memory the caller pinned with hipHostRegister (HSA_EXT_POINTER_TYPE_LOCKED) is known, but
// the address the GPU sees it at is agentBaseAddress + offset, not the host address: hsa_amd_memory_async_copy does not
// translate, HIP's own memcpy does -- so such buffers take the hipMemcpyAsync path like pageable ones (they still move
// by DMA there, without the runtime's staging copy).
bool owner_of(const void* p, hsa_agent_t* agent)
{
    hsa_amd_pointer_info_t info;
    info.size = sizeof info;
    if (hsa_amd_pointer_info(const_cast<void*>(p), &info, nullptr, nullptr, nullptr) != HSA_STATUS_SUCCESS) return false;
    if (info.type != HSA_EXT_POINTER_TYPE_HSA) return false;
    *agent = info.agentOwner;
    return true;
}

}  // namespace

bool available()
{
    std::call_once(g_once, init);
    return g_ok;
}

Signal signal_create()
{
    if (!available()) return 0;
    hsa_signal_t s;
    if (hsa_signal_create(0, 0, nullptr, &s) != HSA_STATUS_SUCCESS) return 0;
    return s.handle;
}

void signal_destroy(Signal s)
{
    if (s) { hsa_signal_t h; h.handle = s; (void)hsa_signal_destroy(h); }
}

void signal_arm(Signal s, int count)
{
    hsa_signal_t h; h.handle = s;
    hsa_signal_store_screlease(h, count);
}

bool can_copy(const void* dst, const void* src)
{
    if (!available()) return false;
    hsa_agent_t a, b;
    return owner_of(dst, &a) && owner_of(src, &b);
}

void signal_cancel(Signal s, int count)
{
    hsa_signal_t h; h.handle = s;
    hsa_signal_subtract_screlease(h, count);
}

int copy_async(void* dst, const void* src, size_t bytes, Signal s)
{
    if (!available() || !s) return 1;
    hsa_agent_t da, sa;
    if (!owner_of(dst, &da) || !owner_of(src, &sa)) return 1;
    hsa_signal_t h; h.handle = s;
    const hsa_status_t st = hsa_amd_memory_async_copy(dst, da, src, sa, bytes, 0, nullptr, h);
    return st == HSA_STATUS_SUCCESS ? 0 : -1;
}

int wait(Signal s)
{
    hsa_signal_t h; h.handle = s;
    for (;;) {
        const hsa_signal_value_t v = hsa_signal_wait_scacquire(h, HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_BLOCKED);
        if (v == 0) return 0;
        if (v < 0) return -1;
    }
}

void enable_timing()
{
    if (!available()) return;
    // engine timestamps on the completion signals: a process-wide ROCr switch, so only thrown when a caller asks for
    // timings (wr_timings), and never back (other copies may be in flight)
    std::call_once(g_timing_once, [] { g_timing = hsa_amd_profiling_async_copy_enable(true) == HSA_STATUS_SUCCESS; });
}

double last_copy_ms(Signal s)
{
    if (!g_ticks_per_ms || !g_timing) return -1;
    hsa_signal_t h; h.handle = s;
    hsa_amd_profiling_async_copy_time_t t;
    if (hsa_amd_profiling_get_async_copy_time(h, &t) != HSA_STATUS_SUCCESS) return -1;
    return (double)(t.end - t.start) / g_ticks_per_ms;
}

}  // namespace wrdma
