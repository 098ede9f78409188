// wr_dma.h -- bulk host <-> device copies on the GPU's DMA (SDMA) engines through ROCr.
//
// Why not hipMemcpyAsync: with a couple of dozen streams in a process the HIP runtime runs most
// device -> host copies as blit KERNELS (__amd_rocclr_copyBuffer) that sit on the compute units for the
// 20-170 ms a gigabyte-sized PCIe transfer takes and stretch every kernel running beside them 2-4 x
// (profiles/r02/b_hipmemcpy_kernel_stats.csv).  hsa_amd_memory_async_copy always lands on an SDMA engine: no
// compute unit is involved, copies of one direction queue up in call order.
// There is no stream here: the caller orders copies and kernels from the host (wait for the kernel's HIP
// event, then start the copy; wait for the copy's signal, then launch the kernel).
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace wrdma {

// Opaque completion signal (an hsa_signal_t handle); 0 = none.
typedef uint64_t Signal;

bool available();                  // ROCr initialised and usable
Signal signal_create();            // 0 on failure
void signal_destroy(Signal s);
// Arms `s` for `count` copies that will name it as their completion signal.
void signal_arm(Signal s, int count);
// dst / src are a (device memory, hipHostMalloc'd host memory) pair, i.e. copy_async will take them.  Pageable memory
// and memory pinned by the caller with hipHostRegister are refused: the caller sends them through hipMemcpyAsync.
bool can_copy(const void* dst, const void* src);
// Takes `count` copies that were armed but never started off the signal again (error paths).
void signal_cancel(Signal s, int count);
// Starts an asynchronous copy of `bytes` from src to dst; exactly one of them is device memory, the other
// host memory allocated by ROCr (hipHostMalloc / wr_host_alloc).  Returns 0 when queued on a DMA engine,
// 1 when the host pointer is pageable or hipHostRegister'd memory (the caller then falls back to hipMemcpyAsync),
// -1 on error.  `s` is decremented when the copy has finished.
int copy_async(void* dst, const void* src, size_t bytes, Signal s);
// Blocks (without spinning) until every copy armed on `s` has finished.  Returns 0, or -1 if a copy failed.
int wait(Signal s);
// Engine timestamps for last_copy_ms: off until someone asks (a process-wide ROCr profiling switch).
void enable_timing();
// Duration of the last copy that completed on `s`, in milliseconds (engine timestamps); < 0 if unknown
// (enable_timing not called before the copy started).
double last_copy_ms(Signal s);

}  // namespace wrdma
