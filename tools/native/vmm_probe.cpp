// vmm_probe.cpp -- can a quantized plane live in a reserved virtual address range backed by physical chunks that come and go
// (hipMemAddressReserve / hipMemCreate / hipMemMap / hipMemUnmap)?  Checks, on the GPU box: granularity; a kernel writing and
// reading a range mapped from several chunks; what ROCr says about such a pointer (hsa_amd_pointer_info) and whether
// hsa_amd_memory_async_copy moves it to pinned host memory on an SDMA engine (with the device's agent named explicitly);
// unmapping the first chunks while the rest stays usable; mapping the freed chunks into ANOTHER range; cost of map / unmap.
// Build: hipcc -O2 --offload-arch=gfx950 vmm_probe.cpp -o vmm_probe -lhsa-runtime64
#include <hip/hip_runtime.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

__global__ void k_fill(uint8_t* p, size_t n, unsigned salt)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (uint8_t)((i * 2654435761u + salt) >> 13);
}
__global__ void k_sum(const uint8_t* p, size_t n, unsigned long long* out)
{
    unsigned long long s = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += p[i];
    atomicAdd(out, s);
}

int main()
{
    int dev = 0;
    CK(hipSetDevice(dev));
    hipMemAllocationProp prop;
    memset(&prop, 0, sizeof prop);
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = dev;
    size_t gmin = 0, grec = 0;
    CK(hipMemGetAllocationGranularity(&gmin, &prop, hipMemAllocationGranularityMinimum));
    CK(hipMemGetAllocationGranularity(&grec, &prop, hipMemAllocationGranularityRecommended));
    printf("granularity: minimum %zu, recommended %zu bytes\n", gmin, grec);
    const size_t C = (size_t)64 << 20, NCH = 16, total = C * NCH;
    std::vector<hipMemGenericAllocationHandle_t> hs(NCH);
    double t = now();
    for (size_t k = 0; k < NCH; k++) CK(hipMemCreate(&hs[k], C, &prop, 0));
    printf("hipMemCreate of %zu x %zu MiB: %.2f ms each\n", NCH, C >> 20, (now() - t) * 1e3 / NCH);
    void* va = nullptr;
    CK(hipMemAddressReserve(&va, total, 0, nullptr, 0));
    hipMemAccessDesc acc;
    memset(&acc, 0, sizeof acc);
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    t = now();
    for (size_t k = 0; k < NCH; k++) {
        CK(hipMemMap((char*)va + k * C, C, 0, hs[k], 0));
        CK(hipMemSetAccess((char*)va + k * C, C, &acc, 1));
    }
    printf("hipMemMap + hipMemSetAccess: %.3f ms per chunk\n", (now() - t) * 1e3 / NCH);
    uint8_t* p = (uint8_t*)va;
    unsigned long long* d_sum; CK(hipMalloc(&d_sum, 8)); CK(hipMemset(d_sum, 0, 8));
    hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, 0, p, total, 7u);
    hipLaunchKernelGGL(k_sum, dim3(2048), dim3(256), 0, 0, p, total, d_sum);
    unsigned long long s0 = 0; CK(hipMemcpy(&s0, d_sum, 8, hipMemcpyDeviceToHost));
    printf("kernel wrote and summed %zu MiB across %zu chunks: sum %llu\n", total >> 20, NCH, s0);

    // ROCr's view and an SDMA copy to pinned host memory
    hsa_init();
    hsa_amd_pointer_info_t info; info.size = sizeof info;
    hsa_status_t st = hsa_amd_pointer_info(p + C + 4096, &info, nullptr, nullptr, nullptr);
    printf("hsa_amd_pointer_info(mapped): status %d type %d agentOwner.handle %llx agentBase %p size %zu\n", (int)st, (int)info.type,
           (unsigned long long)info.agentOwner.handle, info.agentBaseAddress, (size_t)info.sizeInBytes);
    void* plain; CK(hipMalloc(&plain, 1 << 20));
    hsa_amd_pointer_info_t pi; pi.size = sizeof pi;
    hsa_amd_pointer_info(plain, &pi, nullptr, nullptr, nullptr);
    printf("hsa_amd_pointer_info(hipMalloc): type %d agentOwner.handle %llx\n", (int)pi.type, (unsigned long long)pi.agentOwner.handle);
    void* host; CK(hipHostMalloc(&host, total, hipHostMallocDefault));
    hsa_amd_pointer_info_t hi; hi.size = sizeof hi;
    hsa_amd_pointer_info(host, &hi, nullptr, nullptr, nullptr);
    hsa_signal_t sig; hsa_signal_create(1, 0, nullptr, &sig);
    // a copy that spans a chunk boundary, device agent taken from an ordinary allocation
    const size_t off = C - (8 << 20), len = (size_t)16 << 20;
    t = now();
    st = hsa_amd_memory_async_copy((char*)host, hi.agentOwner, p + off, pi.agentOwner, len, 0, nullptr, sig);
    if (st != HSA_STATUS_SUCCESS) printf("hsa_amd_memory_async_copy: status %d\n", (int)st);
    else {
        while (hsa_signal_wait_scacquire(sig, HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_BLOCKED) > 0) {}
        const double dt = now() - t;
        size_t bad = 0;
        for (size_t i = 0; i < len; i++) if (((uint8_t*)host)[i] != (uint8_t)(((off + i) * 2654435761u + 7u) >> 13)) bad++;
        printf("SDMA copy of 16 MiB across a chunk boundary: %.2f ms, %zu wrong bytes\n", dt * 1e3, bad);
    }
    // and the same through HIP
    CK(hipMemcpy(host, p + off, len, hipMemcpyDeviceToHost));
    // free the first half of the chunks; the rest must stay usable
    t = now();
    for (size_t k = 0; k < NCH / 2; k++) CK(hipMemUnmap((char*)va + k * C, C));
    printf("hipMemUnmap: %.3f ms per chunk\n", (now() - t) * 1e3 / (NCH / 2));
    CK(hipMemset(d_sum, 0, 8));
    hipLaunchKernelGGL(k_sum, dim3(2048), dim3(256), 0, 0, p + total / 2, total / 2, d_sum);
    unsigned long long s1 = 0; CK(hipMemcpy(&s1, d_sum, 8, hipMemcpyDeviceToHost));
    printf("second half still readable after unmapping the first: sum %llu\n", s1);
    // the freed chunks back another range
    void* vb = nullptr;
    CK(hipMemAddressReserve(&vb, total / 2, 0, nullptr, 0));
    t = now();
    for (size_t k = 0; k < NCH / 2; k++) CK(hipMemMap((char*)vb + k * C, C, 0, hs[k], 0));
    const double t_map = now() - t;
    t = now();
    for (size_t k = 0; k < NCH / 2; k++) CK(hipMemSetAccess((char*)vb + k * C, C, &acc, 1));
    printf("re-mapping chunks that were mapped before: hipMemMap %.3f ms, hipMemSetAccess %.3f ms per chunk\n", t_map * 1e3 / (NCH / 2), (now() - t) * 1e3 / (NCH / 2));
    t = now();
    CK(hipMemSetAccess((char*)vb, C * (NCH / 2), &acc, 1));
    printf("hipMemSetAccess over the whole range of %zu chunks at once: %.3f ms\n", NCH / 2, (now() - t) * 1e3);
    for (int rep = 0; rep < 3; rep++) {
        t = now();
        for (size_t k = 0; k < NCH / 2; k++) CK(hipMemUnmap((char*)vb + k * C, C));
        const double tu = now() - t;
        t = now();
        for (size_t k = 0; k < NCH / 2; k++) CK(hipMemMap((char*)vb + k * C, C, 0, hs[k], 0));
        CK(hipMemSetAccess((char*)vb, C * (NCH / 2), &acc, 1));
        printf("round %d: unmap %.3f ms, map + one set-access %.3f ms per chunk\n", rep, tu * 1e3 / (NCH / 2), (now() - t) * 1e3 / (NCH / 2));
    }
    CK(hipMemset(d_sum, 0, 8));
    hipLaunchKernelGGL(k_sum, dim3(2048), dim3(256), 0, 0, (uint8_t*)vb, total / 2, d_sum);
    unsigned long long s2 = 0; CK(hipMemcpy(&s2, d_sum, 8, hipMemcpyDeviceToHost));
    printf("freed chunks mapped into another range keep their bytes: sum %llu (first half was %llu)\n", s2, s0 - s1);
    // the decoder's pattern: chunks a kernel wrote through one range are mapped into another, an SDMA engine fills them from
    // pinned host memory, a kernel reads them: does it see the new bytes?
    {
        uint8_t* hb = (uint8_t*)host;
        const size_t half = total / 2;
        unsigned long long want = 0;
        for (size_t i = 0; i < half; i++) { hb[i] = (uint8_t)((i * 7 + 3) & 0x3f); want += hb[i]; }
        hsa_signal_store_screlease(sig, 1);
        st = hsa_amd_memory_async_copy(vb, pi.agentOwner, host, hi.agentOwner, half, 0, nullptr, sig);
        if (st != HSA_STATUS_SUCCESS) printf("H2D hsa_amd_memory_async_copy: status %d\n", (int)st);
        while (hsa_signal_wait_scacquire(sig, HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_BLOCKED) > 0) {}
        CK(hipMemset(d_sum, 0, 8));
        hipLaunchKernelGGL(k_sum, dim3(2048), dim3(256), 0, 0, (uint8_t*)vb, half, d_sum);
        unsigned long long s3 = 0; CK(hipMemcpy(&s3, d_sum, 8, hipMemcpyDeviceToHost));
        printf("SDMA host->device into re-mapped chunks, then a kernel reads them: sum %llu, expected %llu -> %s\n", s3, want, s3 == want ? "ok" : "STALE / WRONG");
        // once more right after a kernel has WRITTEN them (lines of that write may still sit in the L2s)
        hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, 0, (uint8_t*)vb, half, 99u);
        CK(hipDeviceSynchronize());
        for (size_t i = 0; i < half; i++) hb[i] = (uint8_t)((i * 13 + 1) & 0x1f);
        want = 0; for (size_t i = 0; i < half; i++) want += hb[i];
        hsa_signal_store_screlease(sig, 1);
        st = hsa_amd_memory_async_copy(vb, pi.agentOwner, host, hi.agentOwner, half, 0, nullptr, sig);
        while (hsa_signal_wait_scacquire(sig, HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_BLOCKED) > 0) {}
        CK(hipMemset(d_sum, 0, 8));
        hipLaunchKernelGGL(k_sum, dim3(2048), dim3(256), 0, 0, (uint8_t*)vb, half, d_sum);
        CK(hipMemcpy(&s3, d_sum, 8, hipMemcpyDeviceToHost));
        printf("kernel write, SDMA host->device over it, kernel read (mapped range): sum %llu, expected %llu -> %s\n", s3, want, s3 == want ? "ok" : "STALE / WRONG");
        // the same on a plain hipMalloc buffer for comparison
        uint8_t* pm; CK(hipMalloc((void**)&pm, half));
        hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, 0, pm, half, 99u);
        CK(hipDeviceSynchronize());
        hsa_amd_pointer_info_t pmi; pmi.size = sizeof pmi; hsa_amd_pointer_info(pm, &pmi, nullptr, nullptr, nullptr);
        hsa_signal_store_screlease(sig, 1);
        st = hsa_amd_memory_async_copy(pm, pmi.agentOwner, host, hi.agentOwner, half, 0, nullptr, sig);
        while (hsa_signal_wait_scacquire(sig, HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_BLOCKED) > 0) {}
        CK(hipMemset(d_sum, 0, 8));
        hipLaunchKernelGGL(k_sum, dim3(2048), dim3(256), 0, 0, pm, half, d_sum);
        CK(hipMemcpy(&s3, d_sum, 8, hipMemcpyDeviceToHost));
        printf("kernel write, SDMA host->device over it, kernel read (hipMalloc buffer): sum %llu, expected %llu -> %s\n", s3, want, s3 == want ? "ok" : "STALE / WRONG");
    }
    // The case that bit: ONE address range backed first by chunks A (a kernel writes them, an SDMA engine reads them), then by
    // chunks B (an SDMA engine fills them, a kernel reads them).  Stale translations anywhere show as wrong sums.
    {
        const size_t half = total / 2, nh = NCH / 2;
        uint8_t* hb = (uint8_t*)host;
        void* vc = nullptr;
        CK(hipMemAddressReserve(&vc, half, 0, nullptr, 0));
        for (int variant = 0; variant < 3; variant++) {
            // chunks A = hs[0 .. nh), chunks B = hs[nh .. NCH); unmap whatever backs vb / va first
            for (size_t k = 0; k < nh; k++) (void)hipMemUnmap((char*)vb + k * C, C);
            for (size_t k = nh; k < NCH; k++) (void)hipMemUnmap((char*)va + k * C, C);
            (void)hipGetLastError();
            for (size_t k = 0; k < nh; k++) CK(hipMemMap((char*)vc + k * C, C, 0, hs[k], 0));
            CK(hipMemSetAccess(vc, half, &acc, 1));
            hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, 0, (uint8_t*)vc, half, 5u);
            CK(hipDeviceSynchronize());
            hsa_signal_store_screlease(sig, 1);
            st = hsa_amd_memory_async_copy(host, hi.agentOwner, vc, pi.agentOwner, half, 0, nullptr, sig);   // engine reads range (A)
            while (hsa_signal_wait_scacquire(sig, HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_BLOCKED) > 0) {}
            for (size_t k = 0; k < nh; k++) CK(hipMemUnmap((char*)vc + k * C, C));
            if (variant == 1) CK(hipDeviceSynchronize());
            if (variant == 2) { CK(hipMemAddressFree(vc, half)); CK(hipMemAddressReserve(&vc, half, 0, nullptr, 0)); }
            for (size_t k = 0; k < nh; k++) CK(hipMemMap((char*)vc + k * C, C, 0, hs[nh + (nh - 1 - k)], 0));          // chunks B, other order
            CK(hipMemSetAccess(vc, half, &acc, 1));
            unsigned long long want = 0;
            for (size_t i = 0; i < half; i++) { hb[i] = (uint8_t)((i * 11 + variant) & 0x0f); want += hb[i]; }
            hsa_signal_store_screlease(sig, 1);
            st = hsa_amd_memory_async_copy(vc, pi.agentOwner, host, hi.agentOwner, half, 0, nullptr, sig);   // engine fills range (B)
            while (hsa_signal_wait_scacquire(sig, HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_BLOCKED) > 0) {}
            CK(hipMemset(d_sum, 0, 8));
            hipLaunchKernelGGL(k_sum, dim3(2048), dim3(256), 0, 0, (uint8_t*)vc, half, d_sum);
            unsigned long long s4 = 0; CK(hipMemcpy(&s4, d_sum, 8, hipMemcpyDeviceToHost));
            // and what the engine reads back
            memset(hb, 0xEE, half);
            hsa_signal_store_screlease(sig, 1);
            st = hsa_amd_memory_async_copy(host, hi.agentOwner, vc, pi.agentOwner, half, 0, nullptr, sig);
            while (hsa_signal_wait_scacquire(sig, HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_BLOCKED) > 0) {}
            unsigned long long s5 = 0; for (size_t i = 0; i < half; i++) s5 += hb[i];
            printf("same range, other chunks (%s): kernel reads sum %llu, engine reads back %llu, expected %llu -> %s\n",
                   variant == 0 ? "plain" : variant == 1 ? "device synchronised after the unmap" : "address range freed and reserved again",
                   s4, s5, want, (s4 == want && s5 == want) ? "ok" : "WRONG");
            for (size_t k = 0; k < nh; k++) CK(hipMemUnmap((char*)vc + k * C, C));
            // put things back the way the tail of this program expects them
            for (size_t k = 0; k < nh; k++) { CK(hipMemMap((char*)vb + k * C, C, 0, hs[k], 0)); }
            CK(hipMemSetAccess(vb, half, &acc, 1));
            for (size_t k = nh; k < NCH; k++) CK(hipMemMap((char*)va + k * C, C, 0, hs[k], 0));
            CK(hipMemSetAccess((char*)va + half, half, &acc, 1));
        }
        CK(hipMemAddressFree(vc, half));
    }
    size_t fr = 0, tot = 0; CK(hipMemGetInfo(&fr, &tot));
    printf("hipMemGetInfo: free %.1f GiB of %.1f\n", fr / 1073741824.0, tot / 1073741824.0);
    for (size_t k = 0; k < NCH / 2; k++) CK(hipMemUnmap((char*)vb + k * C, C));
    for (size_t k = NCH / 2; k < NCH; k++) CK(hipMemUnmap((char*)va + k * C, C));
    for (size_t k = 0; k < NCH; k++) CK(hipMemRelease(hs[k]));
    CK(hipMemAddressFree(va, total)); CK(hipMemAddressFree(vb, total / 2));
    printf("vmm probe done\n");
    return 0;
}
