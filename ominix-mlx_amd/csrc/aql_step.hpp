// Replay of a recorded decode step as raw AQL packets on an HSA queue the engine owns (aql_step.hip).
//
// Why: a decode step is ~147 dependent launches of 8-35 us each, and every launch boundary of a HIP stream / hipGraph replay costs
// what the packet processor does between two dispatches: the barrier, the packet's acquire fence (invalidate) and release fence
// (L2 write-back) and HIP's own bookkeeping packets.  tools/aql_probe measured 0.6-1.0 us per launch for own packets with the barrier bit
// kept (ordering stays the hardware's job) and the per-packet fences dropped.  The kernels of the step are the ones HIP loaded: their
// kernel descriptors are looked up by name in the loaded executables (hsa_ven_amd_loader), the kernarg segment is the recorded explicit
// arguments + the code-object-v5 hidden arguments, all packets of a step are written once per replay and the doorbell rung once per step.
#pragma once
#include "launch_timing.hpp"

namespace omx {

struct AqlProgram;   // one recorded step: packets + kernargs in device memory

// fence mode of the packets between the first and the last of a replay (the first always acquires, the last always releases, both at
// system scope, and carries the completion signal):
//   0 = agent-scope acquire + release on every packet (what a HIP stream does; needs nothing from the kernels)
//   1 = no fences (kernels must move every cross-kernel value with write-through stores and coherent loads)
//   2 = acquire only, 3 = release only (measurement)
enum { AQL_FENCE_AGENT = 0, AQL_FENCE_NONE = 1, AQL_FENCE_ACQUIRE = 2, AQL_FENCE_RELEASE = 3 };

// null (with the reason in the error slot) when the HSA side is unavailable, a kernel cannot be resolved or needs something the replay
// does not provide -- the caller stays on hipGraph
AqlProgram* aql_build(const LaunchRecorder& rec, int fence_mode);
void aql_destroy(AqlProgram* p);
int aql_launches(const AqlProgram* p);
// `times` back-to-back replays; returns after the last packet completed.  wall_ms: doorbell -> completion.
// per_launch_us (optional, [launches]): mean device duration of each launch over the replays (queue profiling on for this call)
int aql_replay(AqlProgram* p, int times, double* wall_ms, float* per_launch_us = nullptr);

}  // namespace omx
