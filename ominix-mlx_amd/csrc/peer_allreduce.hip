// One-shot all-reduce of the tensor-parallel decode step over xGMI peer stores (two all-reduces of the hidden
// vector per layer, one of the argmax key per token).  A decode step at TP = N hands RCCL 72 reductions of 16 KB: at that size
// a ring is pure latency (2 (N - 1) dependent hops, a launch each).  xGMI is a point-to-point fabric and every GPU can store straight
// into every other GPU's memory, so a vector this small is reduced in ONE hop:
//   * every rank owns an INBOX [2 parities][world][8192] of 8-byte granules {32-bit payload word, 32-bit tag} in fine-grained device
//     memory, exported with hipIpcGetMemHandle and mapped by every peer (one process per GPU);
//   * a call = one kernel: thread i stores word i of its rank's contribution, tagged with the call's sequence number, into slot
//     [parity][rank][i] of EVERY inbox (system-scope 8-byte stores: payload and tag travel together, so there is no flag, no fence
//     and no second round -- the LL idea of the collective libraries), then polls the `world` slots [parity][*][i] of its OWN inbox
//     until all carry the tag and reduces them in rank order: every rank computes the identical f32 sum;
//   * two parities: a rank can be at most one call ahead of a peer (it cannot finish call n + 1 before the peer has started it, i.e.
//     has finished reading call n), so call n + 1 never overwrites data of call n that is still being read;
//   * the sequence number lives in device memory and is advanced by the kernel's last block: a captured step graph replays it.
// Same signature and codes as ncclAllReduce (the engines take either, omx_qwen3_set_comm); calls it does not cover (more than 8192
// words: the batched prefill's [T, hidden] reductions, the DiT's 28 MB ones; other dtypes) go to the RCCL communicator given at
// creation.  Waits are bounded: a peer that never shows up raises the abort word (omx_peer_comm_status) instead of hanging the GPU.
#include <cstring>

#include "common.hpp"
#include "peer.hpp"

namespace omx {
namespace {

typedef int (*nccl_allreduce_fn)(const void*, void*, size_t, int, int, void*, hipStream_t);
constexpr int kNcclUint64 = 5, kNcclFloat32 = 7, kNcclSum = 0, kNcclMax = 2;

struct PeerComm {
    int rank = 0, world = 1;
    uint64_t* inbox = nullptr;                 // [2][world][kPeerMaxWords], fine-grained, exported
    uint64_t* peers[kPeerMaxWorld] = {};       // every rank's inbox as mapped here (own: inbox)
    bool mapped[kPeerMaxWorld] = {};           // opened with hipIpcOpenMemHandle (to be closed)
    uint32_t* state = nullptr;                 // device: [0] sequence number, [1] blocks done, [2] abort
    PeerDev* dev = nullptr;                    // device copy of the table the kernels read (written once every inbox is mapped)
    void* rccl = nullptr;
    nccl_allreduce_fn rccl_fn = nullptr;
    hipIpcMemHandle_t handle;
    bool connected = false;
};

struct PeerArgs {
    const PeerDev* dev;
    const uint32_t* send;
    uint32_t* recv;
    int words, max64;
};

__global__ __launch_bounds__(256) void peer_allreduce_kernel(const PeerArgs a) {
    const PeerDev* p = a.dev;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const unsigned tag = peer_tag(p);
    if (i < a.words) {
        peer_store_word(p, tag, i, a.send[i]);
        if (!a.max64) {
            a.recv[i] = __float_as_uint(peer_poll_sum_f32(p, tag, i));
        } else {
            // 64-bit keys travel as two words: lanes 2k (low) and 2k + 1 (high) of one key sit next to each other in the wave
            unsigned w[kPeerMaxWorld];
            const bool ok = peer_poll_words(p, tag, i, w);
            unsigned best_lo = 0, best_hi = 0;
#pragma unroll
            for (int r = 0; r < kPeerMaxWorld; ++r) {
                if (r >= p->world) continue;
                const unsigned other = __shfl_xor(w[r], 1, 64);
                const unsigned lo = (i & 1) ? other : w[r], hi = (i & 1) ? w[r] : other;
                if (hi > best_hi || (hi == best_hi && lo > best_lo)) { best_hi = hi; best_lo = lo; }
            }
            a.recv[i] = ok ? ((i & 1) ? best_hi : best_lo) : 0u;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) peer_block_done(p, tag, gridDim.x);
}

int upload_table(PeerComm* c) {
    PeerDev t = {};
    for (int r = 0; r < c->world; ++r) t.peers[r] = c->peers[r];
    t.inbox = c->inbox; t.state = c->state; t.rank = c->rank; t.world = c->world;
    if (!c->dev) OMX_HIP_CHECK(hipMalloc((void**)&c->dev, sizeof(PeerDev)));
    OMX_HIP_CHECK(hipMemcpy(c->dev, &t, sizeof(PeerDev), hipMemcpyHostToDevice));
    c->connected = true;
    return 0;
}

}  // namespace
}  // namespace omx

extern "C" {

int omx_peer_comm_create(void** out, int rank, int world, void* rccl_comm, void* rccl_allreduce_fn) {
    using namespace omx;
    OMX_REQUIRE(out, "omx_peer_comm_create: null output");
    OMX_REQUIRE(world >= 1 && world <= kPeerMaxWorld && rank >= 0 && rank < world, "omx_peer_comm_create: rank %d of %d (at most %d ranks)", rank,
                world, kPeerMaxWorld);
    PeerComm* c = new PeerComm();
    c->rank = rank; c->world = world; c->rccl = rccl_comm; c->rccl_fn = (nccl_allreduce_fn)rccl_allreduce_fn;
    const size_t bytes = (size_t)2 * world * kPeerMaxWords * sizeof(uint64_t);
    // fine-grained: stores of a peer become visible to a polling kernel without a cache invalidate at a kernel boundary
    if (hipExtMallocWithFlags((void**)&c->inbox, bytes, hipDeviceMallocFinegrained) != hipSuccess || hipMemset(c->inbox, 0, bytes) != hipSuccess ||
        hipMalloc((void**)&c->state, 16) != hipSuccess || hipMemset(c->state, 0, 16) != hipSuccess) {
        (void)hipGetLastError();
        if (c->inbox) (void)hipFree(c->inbox);
        if (c->state) (void)hipFree(c->state);
        delete c;
        return set_error("omx_peer_comm_create: device allocation of the inbox failed");
    }
    std::memset(&c->handle, 0, sizeof(c->handle));
    if (world > 1 && hipIpcGetMemHandle(&c->handle, c->inbox) != hipSuccess) {
        (void)hipGetLastError();
        (void)hipFree(c->inbox); (void)hipFree(c->state);
        delete c;
        return set_error("omx_peer_comm_create: hipIpcGetMemHandle failed (HSA_ENABLE_IPC_MODE_LEGACY=0 is required on this driver)");
    }
    c->peers[rank] = c->inbox;
    if (world == 1 && upload_table(c)) return 1;
    OMX_HIP_CHECK(hipDeviceSynchronize());
    *out = c;
    return 0;
}

int omx_peer_comm_handle(void* comm, void* out64) {
    using namespace omx;
    OMX_REQUIRE(comm && out64, "omx_peer_comm_handle: null argument");
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "the host side exchanges 64-byte handles");
    std::memcpy(out64, &static_cast<PeerComm*>(comm)->handle, 64);
    return 0;
}

// handles: world x 64 bytes, rank-major, as gathered by the host side
int omx_peer_comm_connect(void* comm, const void* handles) {
    using namespace omx;
    OMX_REQUIRE(comm && handles, "omx_peer_comm_connect: null argument");
    PeerComm* c = static_cast<PeerComm*>(comm);
    for (int r = 0; r < c->world; ++r) {
        if (r == c->rank || c->peers[r]) continue;
        hipIpcMemHandle_t h;
        std::memcpy(&h, static_cast<const char*>(handles) + (size_t)r * 64, 64);
        void* p = nullptr;
        if (hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess) != hipSuccess) {
            (void)hipGetLastError();
            return set_error("omx_peer_comm_connect: hipIpcOpenMemHandle of rank %d's inbox failed", r);
        }
        c->peers[r] = static_cast<uint64_t*>(p);
        c->mapped[r] = true;
    }
    return upload_table(c);
}

// the table the kernels read (device memory; NULL until every inbox is mapped): the engines hand it to the GEMV whose epilogue
// reduces its own output rows over the peers (gemv.hip)
const void* omx_peer_comm_device(void* comm) {
    omx::PeerComm* c = static_cast<omx::PeerComm*>(comm);
    return c && c->connected ? c->dev : nullptr;
}

// ncclAllReduce's signature; comm = the handle of omx_peer_comm_create
int omx_peer_allreduce(const void* send, void* recv, size_t count, int dtype, int op, void* comm, omx_stream stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    using namespace omx;
    if (!comm || !send || !recv) return 1;
    PeerComm* c = static_cast<PeerComm*>(comm);
    if (!c->connected) return 1;
    const bool f32sum = dtype == kNcclFloat32 && op == kNcclSum, u64max = dtype == kNcclUint64 && op == kNcclMax;
    const size_t words = f32sum ? count : 2 * count;
    if ((!f32sum && !u64max) || words > (size_t)kPeerMaxWords || count == 0) {
        if (!c->rccl_fn) return 1;   // nothing to hand the call to
        return c->rccl_fn(send, recv, count, dtype, op, c->rccl, stream);
    }
    PeerArgs a = {};
    a.dev = c->dev;
    a.send = static_cast<const uint32_t*>(send); a.recv = static_cast<uint32_t*>(recv);
    a.words = (int)words; a.max64 = u64max ? 1 : 0;
    peer_allreduce_kernel<<<(unsigned)((words + 255) / 256), 256, 0, stream>>>(a);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

void* omx_peer_allreduce_fn(void) { return (void*)&omx_peer_allreduce; }

int omx_peer_comm_status(void* comm, unsigned* aborted) {
    using namespace omx;
    OMX_REQUIRE(comm && aborted, "omx_peer_comm_status: null argument");
    uint32_t st[4] = {};
    OMX_HIP_CHECK(hipMemcpy(st, static_cast<PeerComm*>(comm)->state, 16, hipMemcpyDeviceToHost));
    *aborted = st[2];
    return 0;
}

int omx_peer_comm_destroy(void* comm) {
    using namespace omx;
    if (!comm) return 0;
    PeerComm* c = static_cast<PeerComm*>(comm);
    (void)hipDeviceSynchronize();
    for (int r = 0; r < c->world; ++r)
        if (c->mapped[r]) (void)hipIpcCloseMemHandle(c->peers[r]);
    (void)hipFree(c->inbox);
    (void)hipFree(c->state);
    if (c->dev) (void)hipFree(c->dev);
    delete c;
    return 0;
}

}  // extern "C"
