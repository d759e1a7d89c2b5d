// Per-launch kernel timestamps for the measurement hooks (engine.hip: omx_qwen3_time_step_kernels).
// A caller arms an event pair for the NEXT kernel launch of this host thread; the launch sites of the decode step's kernels go through
// OMX_LAUNCH_TIMED, which hands an armed pair to hipExtLaunchKernelGGL: the events then carry the dispatch's own begin / end timestamps
// (what rocprofv3's kernel trace reads), not those of separate marker packets around it -- a hipEventRecord pair around a ~10 us kernel
// measured 2.5-2.8 us too long in the step, and an empty pair 4.7 us.  Unarmed launches are plain <<<>>> launches.
//
// The same launch sites can be RECORDED instead of launched (aql_step.hpp): while a recorder is installed on the host thread, every
// OMX_LAUNCH_TIMED / OMX_LAUNCH stores {kernel handle, grid, block, LDS bytes, the argument bytes laid out like the kernarg segment} and
// launches nothing; the decode engine replays the recorded step as raw AQL packets on its own HSA queue.
#pragma once
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>

#include <cstring>
#include <vector>

namespace omx {

struct LaunchEvents { hipEvent_t start = nullptr, stop = nullptr; };
inline thread_local LaunchEvents g_launch_events;

inline void arm_launch_events(hipEvent_t start, hipEvent_t stop) { g_launch_events = {start, stop}; }
inline LaunchEvents take_launch_events() { LaunchEvents e = g_launch_events; g_launch_events = {}; return e; }

// ---- launch recorder ----
struct RecordedLaunch {
    const void* fn;                    // the kernel's host-side handle (what hipLaunchKernel takes)
    dim3 grid, block;
    uint32_t lds;                      // dynamic LDS bytes
    int tag;                           // caller's label of the launch (kernel class for the timing hook), -1 = none
    std::vector<unsigned char> args;   // explicit arguments at their kernarg offsets
};
struct LaunchRecorder {
    std::vector<RecordedLaunch> launches;
    int next_tag = -1;
};
inline thread_local LaunchRecorder* g_launch_recorder = nullptr;

template <class P>
inline void record_arg(std::vector<unsigned char>& buf, const P& v) {
    const size_t off = (buf.size() + alignof(P) - 1) / alignof(P) * alignof(P);
    buf.resize(off + sizeof(P), 0);
    std::memcpy(buf.data() + off, &v, sizeof(P));
}
// the parameter types come from the kernel's signature, so an `int` literal passed for a `size_t` parameter is widened as a launch would
template <class... P, class... A>
inline void record_launch(void (*kernel)(P...), dim3 grid, dim3 block, size_t lds, A&&... a) {
    static_assert(sizeof...(P) == sizeof...(A), "record_launch: argument count differs from the kernel's parameters");
    RecordedLaunch r;
    r.fn = reinterpret_cast<const void*>(kernel);
    r.grid = grid; r.block = block; r.lds = (uint32_t)lds;
    r.tag = g_launch_recorder->next_tag;
    g_launch_recorder->next_tag = -1;
    (record_arg<P>(r.args, static_cast<P>(a)), ...);
    g_launch_recorder->launches.push_back(std::move(r));
}

}  // namespace omx

#define OMX_LAUNCH_TIMED(kernel, grid, block, shmem, stream, ...)                                                                   \
    do {                                                                                                                            \
        if (::omx::g_launch_recorder) { ::omx::record_launch((kernel), dim3(grid), dim3(block), (shmem), __VA_ARGS__); break; }     \
        const ::omx::LaunchEvents ev_ = ::omx::take_launch_events();                                                                \
        if (ev_.start) hipExtLaunchKernelGGL((kernel), (grid), (block), (uint32_t)(shmem), (stream), ev_.start, ev_.stop, 0, __VA_ARGS__); \
        else (kernel)<<<(grid), (block), (shmem), (stream)>>>(__VA_ARGS__);                                                         \
    } while (0)
// a launch site that is never timed but may be recorded
#define OMX_LAUNCH(kernel, grid, block, shmem, stream, ...)                                                                         \
    do {                                                                                                                            \
        if (::omx::g_launch_recorder) { ::omx::record_launch((kernel), dim3(grid), dim3(block), (shmem), __VA_ARGS__); break; }     \
        (kernel)<<<(grid), (block), (shmem), (stream)>>>(__VA_ARGS__);                                                              \
    } while (0)
