// In-process stand-in for an RCCL communicator: `world` engine instances driven by `world` host threads
// of ONE process on ONE GPU exchange through it.  It exists so that the tensor-parallel paths (engine.hip,
// dit.hip) can be run with REAL shards (world > 1) on a single-GPU test box; the production path hands the
// engines ncclAllReduce (ominix-mlx_amd/comm.py).  Same call signature as ncclAllReduce, same dtype/op codes.
//
// all-reduce = copy my contribution to my slot, stream-sync, host barrier, reduce all slots on my stream in
// rank order (every rank computes the identical sum), stream-sync, host barrier (slots may be reused).
#include <condition_variable>
#include <mutex>
#include <vector>

#include "common.hpp"

namespace {

constexpr int kNcclUint64 = 5, kNcclFloat32 = 7, kNcclBfloat16 = 9, kNcclSum = 0, kNcclMax = 2;

struct Group {
    int world = 0;
    size_t slot_bytes = 0;
    std::vector<void*> slots;
    std::mutex mu;
    std::condition_variable cv;
    int arrived = 0;
    long generation = 0;
    bool broken = false;
    void** slots_dev = nullptr;
};
struct RankComm {
    Group* g;
    int rank;
};

// returns false when the group was torn down while waiting
bool barrier(Group* g) {
    std::unique_lock<std::mutex> lk(g->mu);
    const long gen = g->generation;
    if (++g->arrived == g->world) {
        g->arrived = 0;
        ++g->generation;
        g->cv.notify_all();
        return !g->broken;
    }
    g->cv.wait(lk, [&] { return g->generation != gen || g->broken; });
    return !g->broken;
}

template <class T, bool MAX>
__global__ void reduce_slots_kernel(T* out, T* const* slots, int world, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        T acc = slots[0][i];
        for (int r = 1; r < world; ++r) {
            const T v = slots[r][i];
            acc = MAX ? (v > acc ? v : acc) : acc + v;
        }
        out[i] = acc;
    }
}
__global__ void reduce_slots_bf16_kernel(omx::bf16_t* out, omx::bf16_t* const* slots, int world, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float acc = omx::bf16_to_f32(slots[0][i]);
        for (int r = 1; r < world; ++r) acc += omx::bf16_to_f32(slots[r][i]);
        out[i] = omx::f32_to_bf16(acc);
    }
}

}  // namespace

struct omx_loopback_ {
    Group g;
    std::vector<RankComm> ranks;
};

extern "C" int omx_loopback_create(omx_loopback* out, int world, size_t max_bytes) {
    OMX_REQUIRE(out && world >= 1 && world <= 16 && max_bytes > 0, "omx_loopback_create: bad arguments");
    omx_loopback L = new omx_loopback_();
    L->g.world = world;
    L->g.slot_bytes = max_bytes;
    L->g.slots.resize(world, nullptr);
    for (int r = 0; r < world; ++r) OMX_HIP_CHECK(hipMalloc(&L->g.slots[r], max_bytes));
    OMX_HIP_CHECK(hipMalloc(&L->g.slots_dev, sizeof(void*) * world));
    OMX_HIP_CHECK(hipMemcpy(L->g.slots_dev, L->g.slots.data(), sizeof(void*) * world, hipMemcpyHostToDevice));
    for (int r = 0; r < world; ++r) L->ranks.push_back(RankComm{&L->g, r});
    *out = L;
    return 0;
}

extern "C" int omx_loopback_destroy(omx_loopback L) {
    if (!L) return 0;
    {
        std::lock_guard<std::mutex> lk(L->g.mu);
        L->g.broken = true;
    }
    L->g.cv.notify_all();
    for (void* p : L->g.slots) (void)hipFree(p);
    (void)hipFree(L->g.slots_dev);
    delete L;
    return 0;
}

/* the `comm` argument rank `rank` passes to omx_loopback_allreduce (== what omx_*_set_comm receives) */
extern "C" void* omx_loopback_rank_comm(omx_loopback L, int rank) {
    return (L && rank >= 0 && rank < L->g.world) ? (void*)&L->ranks[rank] : nullptr;
}

/* a rank that failed elsewhere releases its peers instead of leaving them in the barrier */
extern "C" int omx_loopback_abort(omx_loopback L) {
    if (!L) return 0;
    {
        std::lock_guard<std::mutex> lk(L->g.mu);
        L->g.broken = true;
    }
    L->g.cv.notify_all();
    return 0;
}

extern "C" int omx_loopback_allreduce(const void* send, void* recv, size_t count, int dtype, int op, void* comm,
                                      hipStream_t stream) {
    RankComm* rc = (RankComm*)comm;
    if (!rc || !rc->g) return 4;   // ncclInvalidArgument
    Group* g = rc->g;
    const size_t esz1 = dtype == kNcclUint64 ? 8 : dtype == kNcclFloat32 ? 4 : dtype == kNcclBfloat16 ? 2 : 0;
    if (g->world == 1 && esz1 != 0) {
        // a one-rank group: the reduction is the identity, issued as ONE small kernel on the caller's stream and therefore capturable --
        // the launch a real one-hop reduction would cost, without its link latency (tools/tp_shard_step.py times a tensor-parallel
        // rank's step at its real shard shapes this way)
        const unsigned blocks1 = (unsigned)((count + 255) / 256 < 64 ? (count + 255) / 256 : 64);
        if (dtype == kNcclFloat32) reduce_slots_kernel<float, false><<<blocks1, 256, 0, stream>>>((float*)recv, (float* const*)g->slots_dev, 0, 0);
        else if (dtype == kNcclUint64) reduce_slots_kernel<unsigned long long, true><<<blocks1, 256, 0, stream>>>((unsigned long long*)recv, (unsigned long long* const*)g->slots_dev, 0, 0);
        else reduce_slots_bf16_kernel<<<blocks1, 256, 0, stream>>>((omx::bf16_t*)recv, (omx::bf16_t* const*)g->slots_dev, 0, 0);
        if (send != recv && hipMemcpyAsync(recv, send, count * esz1, hipMemcpyDeviceToDevice, stream) != hipSuccess) return 1;
        return hipGetLastError() == hipSuccess ? 0 : 1;
    }
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) return 5;   // ncclInvalidUsage: not capturable
    const size_t esz = dtype == kNcclUint64 ? 8 : dtype == kNcclFloat32 ? 4 : dtype == kNcclBfloat16 ? 2 : 0;
    if (esz == 0 || count * esz > g->slot_bytes) return 4;
    if (hipMemcpyAsync(g->slots[rc->rank], send, count * esz, hipMemcpyDeviceToDevice, stream) != hipSuccess) return 1;
    if (hipStreamSynchronize(stream) != hipSuccess) return 1;
    if (!barrier(g)) return 3;
    const unsigned blocks = (unsigned)((count + 255) / 256 < 1024 ? (count + 255) / 256 : 1024);
    if (dtype == kNcclFloat32 && op == kNcclSum)
        reduce_slots_kernel<float, false><<<blocks, 256, 0, stream>>>((float*)recv, (float* const*)g->slots_dev, g->world, count);
    else if (dtype == kNcclUint64 && op == kNcclMax)
        reduce_slots_kernel<unsigned long long, true><<<blocks, 256, 0, stream>>>((unsigned long long*)recv, (unsigned long long* const*)g->slots_dev, g->world, count);
    else if (dtype == kNcclBfloat16 && op == kNcclSum)
        reduce_slots_bf16_kernel<<<blocks, 256, 0, stream>>>((omx::bf16_t*)recv, (omx::bf16_t* const*)g->slots_dev, g->world, count);
    else
        return 4;
    if (hipStreamSynchronize(stream) != hipSuccess) return 1;
    if (!barrier(g)) return 3;
    return 0;
}
