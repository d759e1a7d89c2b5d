// Per-launch kernel timestamps for the measurement hooks (engine.hip: omx_qwen3_time_step_kernels).
// A caller arms an event pair for the NEXT kernel launch of this host thread; the launch sites of the decode step's kernels go through
// OMX_LAUNCH_TIMED, which hands an armed pair to hipExtLaunchKernelGGL: the events then carry the dispatch's own begin / end timestamps
// (what rocprofv3's kernel trace reads), not those of separate marker packets around it -- a hipEventRecord pair around a ~10 us kernel
// measured 2.5-2.8 us too long in the step, and an empty pair 4.7 us.  Unarmed launches are plain <<<>>> launches.
#pragma once
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>

namespace omx {

struct LaunchEvents { hipEvent_t start = nullptr, stop = nullptr; };
inline thread_local LaunchEvents g_launch_events;

inline void arm_launch_events(hipEvent_t start, hipEvent_t stop) { g_launch_events = {start, stop}; }
inline LaunchEvents take_launch_events() { LaunchEvents e = g_launch_events; g_launch_events = {}; return e; }

}  // namespace omx

#define OMX_LAUNCH_TIMED(kernel, grid, block, shmem, stream, ...)                                                                   \
    do {                                                                                                                            \
        const ::omx::LaunchEvents ev_ = ::omx::take_launch_events();                                                                \
        if (ev_.start) hipExtLaunchKernelGGL((kernel), (grid), (block), (uint32_t)(shmem), (stream), ev_.start, ev_.stop, 0, __VA_ARGS__); \
        else (kernel)<<<(grid), (block), (shmem), (stream)>>>(__VA_ARGS__);                                                         \
    } while (0)
