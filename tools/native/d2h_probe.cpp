// d2h_probe.cpp -- how should 1-8 GB leave the device?  Measures device -> pinned host copies done
//   A  by hipMemcpyAsync on a stream that never ran a kernel (what the runtime picks: SDMA or a blit kernel
//      shows in `rocprofv3 --kernel-trace --memory-copy-trace`),
//   B  by a hand-written copy kernel of G workgroups storing straight into the pinned buffer,
//   C  by hsa_amd_memory_async_copy (ROCr's own engine choice),
// and the same for host -> device.  Build: hipcc -O2 --offload-arch=gfx950 d2h_probe.cpp -o d2h_probe -lhsa-runtime64
#include <hip/hip_runtime.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void k_copy(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    // 4 loads in flight per lane
    for (; i + 3 * stride < n16; i += 4 * stride) {
        uint4 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
        dst[i] = a; dst[i + stride] = b; dst[i + 2 * stride] = c; dst[i + 3 * stride] = d;
    }
    for (; i < n16; i += stride) dst[i] = src[i];
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv)
{
    const size_t bytes = (argc > 1 ? atof(argv[1]) : 2.0) * (1 << 30);
    void *d, *h;
    CK(hipMalloc(&d, bytes));
    CK(hipHostMalloc(&h, bytes, hipHostMallocDefault));
    CK(hipMemset(d, 1, bytes));
    memset(h, 0, bytes);
    hipStream_t s, s2;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    for (int dir = 0; dir < 2; dir++) {
        const char* name = dir ? "H2D" : "D2H";
        void* dst = dir ? d : h; void* src = dir ? h : d;
        for (int r = 0; r < 3; r++) {
            double t = now();
            CK(hipMemcpyAsync(dst, src, bytes, dir ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost, s));
            CK(hipStreamSynchronize(s));
            printf("A %s hipMemcpyAsync lone stream: %.1f GB/s\n", name, bytes / (now() - t) / 1e9);
        }
        for (int g : {2, 4, 8, 16, 32, 64, 256, 1024}) {
            double best = 0;
            for (int r = 0; r < 2; r++) {
                double t = now();
                hipLaunchKernelGGL(k_copy, dim3(g), dim3(256), 0, s2, (const uint4*)src, (uint4*)dst, bytes / 16);
                CK(hipStreamSynchronize(s2));
                double v = bytes / (now() - t) / 1e9;
                if (v > best) best = v;
            }
            printf("B %s copy kernel, %4d workgroups: %.1f GB/s\n", name, g, best);
        }
    }
    // C: ROCr copy
    hsa_status_t st = hsa_init();
    hsa_amd_pointer_info_t pi; pi.size = sizeof pi;
    hsa_amd_pointer_info_t ph; ph.size = sizeof ph;
    st = hsa_amd_pointer_info(d, &pi, nullptr, nullptr, nullptr);
    hsa_status_t st2 = hsa_amd_pointer_info(h, &ph, nullptr, nullptr, nullptr);
    printf("pointer_info: dev st=%d type=%d, host st=%d type=%d\n", (int)st, (int)pi.type, (int)st2, (int)ph.type);
    hsa_signal_t sig;
    hsa_signal_create(1, 0, nullptr, &sig);
    for (int dir = 0; dir < 2; dir++) {
        for (int r = 0; r < 3; r++) {
            hsa_signal_store_relaxed(sig, 1);
            double t = now();
            if (dir == 0) st = hsa_amd_memory_async_copy(h, ph.agentOwner, d, pi.agentOwner, bytes, 0, nullptr, sig);
            else st = hsa_amd_memory_async_copy(d, pi.agentOwner, h, ph.agentOwner, bytes, 0, nullptr, sig);
            if (st != HSA_STATUS_SUCCESS) { printf("hsa copy failed %d\n", (int)st); break; }
            while (hsa_signal_wait_scacquire(sig, HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_BLOCKED) >= 1) {}
            printf("C %s hsa_amd_memory_async_copy: %.1f GB/s\n", dir ? "H2D" : "D2H", bytes / (now() - t) / 1e9);
        }
    }
    // D: hipMemcpyAsync D2H while a kernel runs on another stream / after a wait on a kernel event
    hipEvent_t ev; CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    for (int r = 0; r < 2; r++) {
        hipLaunchKernelGGL(k_copy, dim3(1024), dim3(256), 0, s2, (const uint4*)d, (uint4*)d, (size_t)1 << 20);
        CK(hipEventRecord(ev, s2));
        CK(hipStreamWaitEvent(s, ev, 0));
        double t = now();
        CK(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, s));
        CK(hipStreamSynchronize(s));
        printf("D D2H hipMemcpyAsync after hipStreamWaitEvent on a kernel's event: %.1f GB/s\n", bytes / (now() - t) / 1e9);
    }
    return 0;
}
