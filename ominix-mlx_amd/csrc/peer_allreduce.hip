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
//
// The LARGE path (round 4; f32 / bf16 sums of any size, used when no RCCL communicator was given or OMX_PEER_LARGE=1): a two-shot
// all-reduce in ONE kernel over the same fully connected fabric -- every byte crosses exactly one link, all links at once:
//   1. rank r PUSHES slice s of its input into slot r of rank s's stage1 (16-byte peer stores), fences, and -- its last block to
//      arrive -- stores the call's tag into flag [0][r] of every rank;
//   2. once all `world` flags [0][*] carry the tag, rank s sums the `world` slots of its stage1 in rank order (f32 accumulation, one
//      rounding for bf16) and PUSHES the reduced slice into slot s of every rank's stage2; fence, flag [1][s] everywhere;
//   3. once all flags [1][*] carry the tag, every rank copies its stage2 to the destination; the last block advances the sequence.
// One slice is reduced by one rank and broadcast, so all ranks hold identical bits.  stage1 / stage2 need no second parity: a peer can
// only start pushing call n + 1 after it has seen this rank's phase-2 flag of call n (all reads of stage1 done), and only reaches
// phase 2 of call n + 1 after this rank's phase-1 flag of call n + 1 (its call-n kernel, copy included, has ended).  Messages above the
// stage size (OMX_PEER_STAGE_MB, default 64) go in chunks.  It also lets several ranks share ONE GPU (tests; bench.py OMX_BENCH_ONE_GPU=1).
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
    unsigned long long n_small = 0, n_large = 0, n_moe = 0, n_rccl = 0;   // launches by path (host-side, omx_peer_comm_counts)
    size_t stage_bytes = 0;                    // large path: bytes per stage (0: off)
    size_t off_flags = 0, off_stage1 = 0, off_stage2 = 0;   // byte offsets inside the exported allocation
    int sys_scope = 1;                         // large path: system-scope release / acquire (the default; omx_peer_comm_set_scope)
};

constexpr int kNcclBfloat16 = 9;
constexpr int kLargeBlocks = 128;              // all of them wait on each other: far below one workgroup per CU
// OMX_PEER_LARGE_BLOCKS=n (8..128): fewer blocks per large-path launch.  Only for several ranks on ONE GPU (bench.py OMX_BENCH_ONE_GPU=1 sets
// it): there the waiting blocks of N - 1 ranks sit on the CUs the N-th rank's matrix-core kernels need whole (a four- or eight-wave GEMM
// workgroup takes a CU's entire register file), and at N = 8 the bounded waits expired.  On a node every rank has its own GPU.
static int large_blocks() {
    static const int v = [] {
        const char* e = getenv("OMX_PEER_LARGE_BLOCKS");
        const int n = e ? atoi(e) : kLargeBlocks;
        return n < 8 ? 8 : n > kLargeBlocks ? kLargeBlocks : n;
    }();
    return v;
}

struct PeerLargeArgs {
    const PeerDev* dev;
    const u32x4* send;
    u32x4* recv;
    size_t nvec, slice_vec;                    // 16-byte vectors in the message / per slice
};

__device__ __forceinline__ void large_publish(const PeerDev* p, int phase, unsigned tag, uint32_t* counter) {
    // every thread's pushes have left this GPU before its block reports in; the last block to arrive tells every rank.
    // Two forms (PeerDev::sys_scope, wave-uniform):
    //  * system scope (the default whenever ranks sit on different GPUs): release fence at system scope by every thread, the flags stored
    //    with release at system scope -- what the memory model asks for when the payload is written and read by OTHER devices;
    //  * agent scope (all ranks on ONE GPU, or OMX_PEER_SCOPE=agent): the stages are fine-grained (uncached) memory, a store is on its
    //    way when it retires, so the release is "wait for my stores" (vmcnt) -- a system-scope fence also writes the whole L2 back,
    //    140 us per call measured with three of them in this kernel (MI355X_MICROARCH.md handoff-flag: payload -> vmcnt(0) -> flag).
    //    Between processes on one GPU the "peer" stage is local HBM, so that form cannot fail there; across xGMI it is unproven
    //    (ADVICE r4): it is never chosen automatically for ranks on different devices.
    if (p->sys_scope) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    else __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned done = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (done == gridDim.x - 1) {
            __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (int r = 0; r < p->world; ++r) {
                peer_gu64* f = (peer_gu64*)(p->flags[r] + (size_t)phase * p->world + p->rank);
                if (p->sys_scope) __hip_atomic_store(f, (unsigned long long)tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                else __hip_atomic_store(f, (unsigned long long)tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // (every block's stores had retired before it reported in)
            }
        }
    }
}

// thread 0 polls this rank's `world` flags of the phase; false when a peer never arrived (abort word raised)
__device__ __forceinline__ bool large_wait(const PeerDev* p, int phase, unsigned tag) {
    __shared__ int s_ok;
    if (threadIdx.x == 0) {
        int ok = 1;
        for (int r = 0; r < p->world && ok; ++r) {
            const peer_gu64* f = (const peer_gu64*)(p->flags[p->rank] + (size_t)phase * p->world + r);
            unsigned spins = 0;
            while ((unsigned)__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != tag) {   // (an acquire per poll would drop the L2 each time)
                if (++spins >= kPeerSpinLimit) { __hip_atomic_store(p->state + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); ok = 0; break; }
                __builtin_amdgcn_s_sleep(2);
            }
        }
        s_ok = ok;
    }
    __syncthreads();
    // ONE acquire after the polls (MI355X_MICROARCH.md: relaxed poll -> one acquire -> plain loads)
    if (p->sys_scope) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    else __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // (drops this CU's L1 lines of the stage; the L2 does not cache fine-grained memory)
    return s_ok != 0;
}

template <bool BF16>
__global__ __launch_bounds__(256) void peer_allreduce_large_kernel(const PeerLargeArgs a) {
    const PeerDev* p = a.dev;
    const unsigned tag = peer_tag(p);
    const int W = p->world, me = p->rank;
    const size_t gtid = (size_t)blockIdx.x * 256 + threadIdx.x, gsize = (size_t)gridDim.x * 256;
    // 1. push: vector v belongs to slice v / slice_vec
    for (size_t v = gtid; v < a.nvec; v += gsize) {
        const size_t sl = v / a.slice_vec, off = v - sl * a.slice_vec;
        *(reinterpret_cast<u32x4*>(p->stage1[sl]) + (size_t)me * a.slice_vec + off) = a.send[v];
    }
    large_publish(p, 0, tag, p->state + 3);
    bool ok = large_wait(p, 0, tag);
    // 2. reduce the own slice in rank order, push the result to every rank
    const size_t lo = (size_t)me * a.slice_vec, cnt = lo < a.nvec ? min(a.slice_vec, a.nvec - lo) : 0;
    for (size_t o = gtid; o < cnt && ok; o += gsize) {
        u32x4 out;
        if (BF16) {
            float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int r = 0; r < W; ++r) {
                const u32x4 x = *(reinterpret_cast<const u32x4*>(p->stage1[me]) + (size_t)r * a.slice_vec + o);
#pragma unroll
                for (int q = 0; q < 4; ++q) { acc[2 * q] += bf16lo(x[q]); acc[2 * q + 1] += bf16hi(x[q]); }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) out[q] = pack_bf16(acc[2 * q], acc[2 * q + 1]);
        } else {
            float acc[4] = {0, 0, 0, 0};
            for (int r = 0; r < W; ++r) {
                const u32x4 x = *(reinterpret_cast<const u32x4*>(p->stage1[me]) + (size_t)r * a.slice_vec + o);
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] += __uint_as_float(x[q]);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) out[q] = __float_as_uint(acc[q]);
        }
        for (int d = 0; d < W; ++d) *(reinterpret_cast<u32x4*>(p->stage2[d]) + lo + o) = out;
    }
    large_publish(p, 1, tag, p->state + 4);
    ok = large_wait(p, 1, tag) && ok;
    // 3. the whole message is in this rank's stage2
    for (size_t v = gtid; v < a.nvec; v += gsize)
        a.recv[v] = ok ? *(reinterpret_cast<const u32x4*>(p->stage2[me]) + v) : u32x4{0u, 0u, 0u, 0u};
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned done = __hip_atomic_fetch_add(p->state + 5, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (done == gridDim.x - 1) {
            __hip_atomic_store(p->state + 5, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(p->state, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

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

// Expert-parallel prompt: the weighted sum of a sparse-MoE block as an ALL-TO-ALL COMBINE + all-gather instead of an all-reduce of
// [T, hidden] f32 partials (SURVEY.md 8e row 2 / BASELINE config 3: "expert-parallel all-to-all over xGMI").  Rank o OWNS tokens
// [o * chunk, (o + 1) * chunk).  One kernel, the flags and counters of the two-shot all-reduce:
//   1. every rank pushes bf16(y_j * score_j) of each slot j routed to ITS experts into slot (j - o * chunk * k) of the stage1 of the
//      token's owner o -- every slot has exactly one writer, a rank sends ~T k / N rows instead of all T;
//   2. the owner sums its tokens' k slot rows in slot order (f32), forms bf16(resid + bf16(sum)) -- the roundings of
//      moe_combine_partial_kernel + ep_fold_kernel -- and pushes the finished residual rows into every rank's stage2 (attention is
//      replicated: every rank needs every row back);
//   3. every rank copies stage2 to its residual stream.
// Per rank and layer at T = 2 048, k = 2, 8 ranks: 4.2 MB + 14.7 MB pushed, against 58 MB for the f32 all-reduce.
struct PeerMoeArgs {
    const PeerDev* dev;
    const bf16_t* y;
    const uint32_t* pos_of_slot;
    const uint32_t* inds;
    const bf16_t* scores;
    const bf16_t* resid;
    bf16_t* out;
    int T, hidden, k, e_lo, e_n, chunk;
};

__global__ __launch_bounds__(256) void peer_moe_combine_kernel(const PeerMoeArgs a) {
    const PeerDev* p = a.dev;
    const unsigned tag = peer_tag(p);
    const int W = p->world, me = p->rank;
    const int vec = a.hidden / 8;                         // 16-byte vectors per row
    const int slots = a.T * a.k;
    // 1. push this rank's slots to their tokens' owners
    for (int j = blockIdx.x; j < slots; j += gridDim.x) {
        const int e = (int)a.inds[j];
        if (e < a.e_lo || e >= a.e_lo + a.e_n) continue;                // (block-uniform)
        const int t = j / a.k, o = min(t / a.chunk, W - 1);
        const float sc = bf16_to_f32(a.scores[j]);
        const u32x4* src = reinterpret_cast<const u32x4*>(a.y + (size_t)a.pos_of_slot[j] * a.hidden);
        u32x4* dst = reinterpret_cast<u32x4*>(p->stage1[o]) + (size_t)(j - o * a.chunk * a.k) * vec;
        for (int v = threadIdx.x; v < vec; v += 256) {
            const u32x4 x = src[v];
            u32x4 r;
#pragma unroll
            for (int q = 0; q < 4; ++q) r[q] = pack_bf16(bf16lo(x[q]) * sc, bf16hi(x[q]) * sc);
            dst[v] = r;
        }
    }
    large_publish(p, 0, tag, p->state + 3);
    bool ok = large_wait(p, 0, tag);
    // 2. the owner's tokens: sum the k slot rows in slot order, add the residual, push the finished row to every rank
    const int t0 = me * a.chunk, n_own = max(0, min(a.chunk, a.T - t0));
    for (int tt = blockIdx.x; tt < n_own && ok; tt += gridDim.x) {
        const int t = t0 + tt;
        const u32x4* rows = reinterpret_cast<const u32x4*>(p->stage1[me]) + (size_t)tt * a.k * vec;
        const u32x4* rs = reinterpret_cast<const u32x4*>(a.resid + (size_t)t * a.hidden);
        for (int v = threadIdx.x; v < vec; v += 256) {
            float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int s2 = 0; s2 < a.k; ++s2) {
                const u32x4 x = rows[(size_t)s2 * vec + v];
#pragma unroll
                for (int q = 0; q < 4; ++q) { acc[2 * q] += bf16lo(x[q]); acc[2 * q + 1] += bf16hi(x[q]); }
            }
            const u32x4 r = rs[v];
            u32x4 o;
#pragma unroll
            for (int q = 0; q < 4; ++q) o[q] = pack_bf16(bf16lo(r[q]) + round_bf16(acc[2 * q]), bf16hi(r[q]) + round_bf16(acc[2 * q + 1]));
            for (int d = 0; d < W; ++d) *(reinterpret_cast<u32x4*>(p->stage2[d]) + (size_t)t * vec + v) = o;
        }
    }
    large_publish(p, 1, tag, p->state + 4);
    ok = large_wait(p, 1, tag) && ok;
    // 3. all T rows are in this rank's stage2
    const size_t nvec = (size_t)a.T * vec;
    for (size_t v = (size_t)blockIdx.x * 256 + threadIdx.x; v < nvec; v += (size_t)gridDim.x * 256)
        reinterpret_cast<u32x4*>(a.out)[v] = ok ? *(reinterpret_cast<const u32x4*>(p->stage2[me]) + v) : u32x4{0u, 0u, 0u, 0u};
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned done = __hip_atomic_fetch_add(p->state + 5, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (done == gridDim.x - 1) {
            __hip_atomic_store(p->state + 5, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(p->state, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

int upload_table(PeerComm* c) {
    PeerDev t = {};
    for (int r = 0; r < c->world; ++r) {
        t.peers[r] = c->peers[r];
        unsigned char* base = reinterpret_cast<unsigned char*>(c->peers[r]);
        t.flags[r] = reinterpret_cast<uint64_t*>(base + c->off_flags);
        t.stage1[r] = base + c->off_stage1;
        t.stage2[r] = base + c->off_stage2;
    }
    t.stage_bytes = c->stage_bytes;
    t.sys_scope = c->sys_scope;
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
    const size_t inbox_bytes = (size_t)2 * world * kPeerMaxWords * sizeof(uint64_t);
    // the large path: on when there is no RCCL communicator to hand big calls to, or when asked for (every rank must decide alike)
    const char* le = getenv("OMX_PEER_LARGE");
    const bool large = le ? le[0] == '1' : rccl_allreduce_fn == nullptr;
    if (large) {
        const char* se = getenv("OMX_PEER_STAGE_MB");
        const long mb = se ? atol(se) : 64;
        c->stage_bytes = (size_t)(mb > 0 ? mb : 64) << 20;
    }
    // hand-off scope of the large path: system unless asked otherwise (OMX_PEER_SCOPE=agent|system; comm.py picks agent by itself only
    // when every rank reports the same device)
    if (const char* sc = getenv("OMX_PEER_SCOPE")) c->sys_scope = (sc[0] == 'a' || sc[0] == 'A') ? 0 : 1;
    c->off_flags = inbox_bytes;
    c->off_stage1 = inbox_bytes + 4096;
    c->off_stage2 = c->off_stage1 + (c->stage_bytes ? c->stage_bytes + 4096 : 0);
    const size_t bytes = c->off_stage2 + (c->stage_bytes ? c->stage_bytes + 4096 : 0);
    // fine-grained: stores of a peer become visible to a polling kernel without a cache invalidate at a kernel boundary
    // (only the granules and flags are cleared: the stages are written before they are read)
    if (hipExtMallocWithFlags((void**)&c->inbox, bytes, hipDeviceMallocFinegrained) != hipSuccess || hipMemset(c->inbox, 0, inbox_bytes + 4096) != hipSuccess ||
        hipMalloc((void**)&c->state, 64) != hipSuccess || hipMemset(c->state, 0, 64) != hipSuccess) {
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
        const bool bf16sum = dtype == kNcclBfloat16 && op == kNcclSum;
        const size_t esz = f32sum ? 4 : 2, nbytes = count * esz;
        if (c->stage_bytes && (f32sum || bf16sum) && count > 0 && nbytes % 16 == 0 &&
            ((reinterpret_cast<uintptr_t>(send) | reinterpret_cast<uintptr_t>(recv)) & 15u) == 0) {
            // two-shot over the peers' stages, a chunk of at most one stage per launch (slices of whole 16-byte vectors)
            const size_t chunk_vec = c->stage_bytes / 16;
            for (size_t v0 = 0, nvec = nbytes / 16; v0 < nvec; v0 += chunk_vec) {
                PeerLargeArgs a = {};
                a.dev = c->dev;
                a.send = static_cast<const u32x4*>(send) + v0; a.recv = static_cast<u32x4*>(recv) + v0;
                a.nvec = std::min(chunk_vec, nvec - v0);
                a.slice_vec = (a.nvec + c->world - 1) / c->world;
                if (bf16sum) peer_allreduce_large_kernel<true><<<large_blocks(), 256, 0, stream>>>(a);
                else peer_allreduce_large_kernel<false><<<large_blocks(), 256, 0, stream>>>(a);
                if (hipGetLastError() != hipSuccess) return 1;
                ++c->n_large;
            }
            return 0;
        }
        if (!c->rccl_fn) return 1;   // nothing to hand the call to
        ++c->n_rccl;
        return c->rccl_fn(send, recv, count, dtype, op, c->rccl, stream);
    }
    PeerArgs a = {};
    a.dev = c->dev;
    a.send = static_cast<const uint32_t*>(send); a.recv = static_cast<uint32_t*>(recv);
    a.words = (int)words; a.max64 = u64max ? 1 : 0;
    peer_allreduce_kernel<<<(unsigned)((words + 255) / 256), 256, 0, stream>>>(a);
    ++c->n_small;
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

void* omx_peer_allreduce_fn(void) { return (void*)&omx_peer_allreduce; }

// scope of the large path's stage hand-offs: 1 = system-scope release / acquire, 0 = agent-scope fences + relaxed flags.  Every rank
// must choose alike (the forms interoperate, but a system-scope consumer behind an agent-scope producer gains nothing).  Not during a
// reduction or a stream capture: the device table is rewritten.
int omx_peer_comm_set_scope(void* comm, int system_scope) {
    using namespace omx;
    OMX_REQUIRE(comm, "omx_peer_comm_set_scope: null communicator");
    PeerComm* c = static_cast<PeerComm*>(comm);
    const int was = c->sys_scope;
    c->sys_scope = system_scope ? 1 : 0;
    if (c->connected) {
        // the host field follows the device table, never the other way round: a failed upload leaves both on the old scope
        hipError_t e = hipDeviceSynchronize();
        if (e == hipSuccess && upload_table(c)) e = hipErrorUnknown;
        if (e == hipSuccess) e = hipDeviceSynchronize();
        if (e != hipSuccess) {
            c->sys_scope = was;
            return set_error("omx_peer_comm_set_scope: the device table could not be rewritten (%s); the scope stays %s", hipGetErrorString(e),
                             was ? "system" : "agent");
        }
    }
    return 0;
}
int omx_peer_comm_scope(void* comm) {
    omx::PeerComm* c = static_cast<omx::PeerComm*>(comm);
    return c ? c->sys_scope : -1;
}
// PCI bus id of the calling thread's current device ("0000:c1:00.0"): ranks compare them to learn whether they share one GPU
int omx_peer_device_id(char* out, int len) {
    using namespace omx;
    OMX_REQUIRE(out && len >= 16, "omx_peer_device_id: buffer of at least 16 bytes");
    int dev = 0;
    OMX_HIP_CHECK(hipGetDevice(&dev));
    OMX_HIP_CHECK(hipDeviceGetPCIBusId(out, len, dev));
    return 0;
}

// bytes of one stage of the two-shot / exchange path (0: that path is off -- an RCCL communicator takes the large calls)
size_t omx_peer_comm_stage_bytes(void* comm) {
    omx::PeerComm* c = static_cast<omx::PeerComm*>(comm);
    return c && c->connected ? c->stage_bytes : 0;
}

// out [T, hidden] = resid + the MoE block's weighted expert outputs, exchanged as described at peer_moe_combine_kernel.
// 0: done; 2: not available on this communicator / for this size (the caller keeps its all-reduce); 1: error
int omx_peer_moe_combine(void* out, const void* resid, const omx_moe_ep_slots* sl, int T, int hidden, int top_k, int e_lo, int e_n, void* comm,
                         omx_stream stream_) {
    using namespace omx;
    if (!comm || !out || !resid || !sl || !sl->y) return 1;
    PeerComm* c = static_cast<PeerComm*>(comm);
    if (!c->connected || !c->stage_bytes || T <= 0 || hidden % 8 != 0 || top_k < 1) return 2;
    const int chunk = (T + c->world - 1) / c->world;
    if ((size_t)chunk * top_k * hidden * 2 > c->stage_bytes || (size_t)T * hidden * 2 > c->stage_bytes) return 2;
    PeerMoeArgs a = {};
    a.dev = c->dev;
    a.y = static_cast<const bf16_t*>(sl->y); a.pos_of_slot = sl->pos_of_slot; a.inds = sl->inds; a.scores = static_cast<const bf16_t*>(sl->scores);
    a.resid = static_cast<const bf16_t*>(resid); a.out = static_cast<bf16_t*>(out);
    a.T = T; a.hidden = hidden; a.k = top_k; a.e_lo = e_lo; a.e_n = e_n; a.chunk = chunk;
    peer_moe_combine_kernel<<<large_blocks(), 256, 0, (hipStream_t)stream_>>>(a);
    ++c->n_moe;
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

// launches issued through this communicator by path: [0] one-shot, [1] two-shot chunks, [2] MoE combine, [3] handed to RCCL (calls made while
// a graph was being captured count once, not per replay)
int omx_peer_comm_counts(void* comm, unsigned long long* out4) {
    using namespace omx;
    OMX_REQUIRE(comm && out4, "omx_peer_comm_counts: null argument");
    const PeerComm* c = static_cast<const PeerComm*>(comm);
    out4[0] = c->n_small; out4[1] = c->n_large; out4[2] = c->n_moe; out4[3] = c->n_rccl;
    return 0;
}

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
