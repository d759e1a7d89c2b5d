// See aql_step.hpp.
#include "aql_step.hpp"

#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <hsa/hsa_ven_amd_loader.h>

#include <stdlib.h>

#include <chrono>
#include <map>
#include <mutex>
#include <string>

#include "common.hpp"

namespace omx {

namespace {

struct KernelInfo {
    uint64_t object = 0;
    uint32_t kernarg_size = 0, group_static = 0, priv = 0;
};

struct Runtime {
    bool ok = false;
    std::string why;
    hsa_agent_t gpu = {};
    hsa_ven_amd_loader_1_03_pfn_t loader = {};
    hsa_queue_t* q = nullptr;
    hsa_signal_t done = {};
    uint64_t ticks_per_s = 0;
    bool queue_error = false;
    std::map<std::string, KernelInfo> kernels;
    std::vector<hsa_signal_t> launch_signals;   // per-launch timing
    std::mutex mu;
};

constexpr uint32_t kQueuePackets = 16384;

struct AgentPick { uint32_t bus, dev; bool have_id; hsa_agent_t first, match; bool have_first, have_match; };
hsa_status_t pick_agent(hsa_agent_t a, void* data) {
    AgentPick* p = (AgentPick*)data;
    hsa_device_type_t t;
    if (hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t) != HSA_STATUS_SUCCESS || t != HSA_DEVICE_TYPE_GPU) return HSA_STATUS_SUCCESS;
    if (!p->have_first) { p->first = a; p->have_first = true; }
    uint32_t bdf = 0;
    if (p->have_id && hsa_agent_get_info(a, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_BDFID, &bdf) == HSA_STATUS_SUCCESS &&
        ((bdf >> 8) & 0xFF) == p->bus && ((bdf >> 3) & 0x1F) == p->dev && !p->have_match) {
        p->match = a; p->have_match = true;
    }
    return HSA_STATUS_SUCCESS;
}

void queue_error_cb(hsa_status_t, hsa_queue_t*, void* data) { ((Runtime*)data)->queue_error = true; }

Runtime& runtime() {
    static Runtime rt;
    static std::once_flag once;
    std::call_once(once, [] {
        Runtime& r = rt;
        auto fail = [&](const char* what) { r.why = what; };
        if (hsa_init() != HSA_STATUS_SUCCESS) return fail("hsa_init failed");
        AgentPick pick = {};
        int dev = 0, bus = 0, pdev = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&bus, hipDeviceAttributePciBusId, dev) == hipSuccess &&
            hipDeviceGetAttribute(&pdev, hipDeviceAttributePciDeviceId, dev) == hipSuccess) {
            pick.bus = (uint32_t)bus; pick.dev = (uint32_t)pdev; pick.have_id = true;
        }
        (void)hipGetLastError();
        if (hsa_iterate_agents(pick_agent, &pick) != HSA_STATUS_SUCCESS || !pick.have_first) return fail("no HSA GPU agent");
        r.gpu = pick.have_match ? pick.match : pick.first;
        if (hsa_system_get_major_extension_table(HSA_EXTENSION_AMD_LOADER, 1, sizeof(r.loader), &r.loader) != HSA_STATUS_SUCCESS ||
            !r.loader.hsa_ven_amd_loader_iterate_executables)
            return fail("the HSA loader extension 1.03 (iterate_executables) is not available");
        if (hsa_queue_create(r.gpu, kQueuePackets, HSA_QUEUE_TYPE_SINGLE, queue_error_cb, &r, UINT32_MAX, UINT32_MAX, &r.q) != HSA_STATUS_SUCCESS)
            return fail("hsa_queue_create failed");
        if (hsa_signal_create(1, 0, nullptr, &r.done) != HSA_STATUS_SUCCESS) return fail("hsa_signal_create failed");
        hsa_system_get_info(HSA_SYSTEM_INFO_TIMESTAMP_FREQUENCY, &r.ticks_per_s);
        r.ok = true;
    });
    return rt;
}

struct FindSym { Runtime* r; const char* name; KernelInfo out; bool found; };
hsa_status_t find_in_executable(hsa_executable_t exe, void* data) {
    FindSym* f = (FindSym*)data;
    hsa_executable_symbol_t sym;
    if (hsa_executable_get_symbol_by_name(exe, f->name, &f->r->gpu, &sym) != HSA_STATUS_SUCCESS) return HSA_STATUS_SUCCESS;
    hsa_symbol_kind_t kind;
    if (hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_TYPE, &kind) != HSA_STATUS_SUCCESS || kind != HSA_SYMBOL_KIND_KERNEL)
        return HSA_STATUS_SUCCESS;
    KernelInfo k;
    hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_OBJECT, &k.object);
    hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_KERNARG_SEGMENT_SIZE, &k.kernarg_size);
    hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_GROUP_SEGMENT_SIZE, &k.group_static);
    hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_PRIVATE_SEGMENT_SIZE, &k.priv);
    if (!k.object) return HSA_STATUS_SUCCESS;
    f->out = k;
    f->found = true;
    return HSA_STATUS_INFO_BREAK;
}

// the kernel descriptor HIP itself dispatches for this host handle
int resolve_kernel(Runtime& r, const void* fn, KernelInfo* out) {
    hipFuncAttributes attr;
    OMX_HIP_CHECK(hipFuncGetAttributes(&attr, fn));            // (also makes HIP load the module's code object)
    const char* name = hipKernelNameRefByPtr(fn, nullptr);
    OMX_REQUIRE(name && name[0], "aql: no kernel name for a recorded launch");
    const std::string key = std::string(name) + ".kd";
    auto it = r.kernels.find(key);
    if (it == r.kernels.end()) {
        FindSym f = {&r, key.c_str(), {}, false};
        const hsa_status_t st = r.loader.hsa_ven_amd_loader_iterate_executables(find_in_executable, &f);
        OMX_REQUIRE((st == HSA_STATUS_SUCCESS || st == HSA_STATUS_INFO_BREAK) && f.found, "aql: kernel descriptor %s not found in the loaded executables", key.c_str());
        it = r.kernels.emplace(key, f.out).first;
    }
    *out = it->second;
    return 0;
}

// code object v5 hidden arguments, relative to the 8-aligned end of the explicit ones (LLVM AMDGPUUsage "Code Object V5 Metadata")
constexpr size_t kHiddenBytes = 256;
void fill_hidden(unsigned char* h, const RecordedLaunch& L) {
    const uint32_t bc[3] = {L.grid.x, L.grid.y, L.grid.z};
    const uint16_t gs[3] = {(uint16_t)L.block.x, (uint16_t)L.block.y, (uint16_t)L.block.z};
    std::memcpy(h + 0, bc, 12);          // hidden_block_count_{x,y,z}
    std::memcpy(h + 12, gs, 6);          // hidden_group_size_{x,y,z}
    // remainders (18..23) and global offsets (40..63) stay zero: grids are whole workgroups
    const uint16_t dims = L.grid.z > 1 ? 3 : L.grid.y > 1 ? 2 : 1;
    std::memcpy(h + 64, &dims, 2);       // hidden_grid_dims
    const uint32_t dyn = L.lds;
    std::memcpy(h + 120, &dyn, 4);       // hidden_dynamic_lds_size
}

}  // namespace

struct AqlProgram {
    std::vector<hsa_kernel_dispatch_packet_t> pkts;   // complete but for header / completion signal
    std::vector<int> tags;
    void* kernargs = nullptr;                         // device memory
    int fence_mode = AQL_FENCE_AGENT;
};

int aql_launches(const AqlProgram* p) { return p ? (int)p->pkts.size() : 0; }

void aql_destroy(AqlProgram* p) {
    if (!p) return;
    if (p->kernargs) (void)hipFree(p->kernargs);
    delete p;
}

static int build_into(Runtime& r, const LaunchRecorder& rec, AqlProgram* p) {
    const size_t n = rec.launches.size();
    OMX_REQUIRE(n > 0 && n < kQueuePackets / 2, "aql: %zu recorded launches do not fit the queue", n);
    std::vector<KernelInfo> info(n);
    std::vector<size_t> off(n);
    size_t total = 0;
    for (size_t i = 0; i < n; ++i) {
        const RecordedLaunch& L = rec.launches[i];
        if (resolve_kernel(r, L.fn, &info[i])) return 1;
        const size_t base = (L.args.size() + 7) / 8 * 8;
        // the compiler's segment = explicit arguments + the hidden block: a different size means the recorded bytes are not laid out the
        // way the kernel reads them
        OMX_REQUIRE(info[i].kernarg_size == base + kHiddenBytes || info[i].kernarg_size == base,
                    "aql: kernarg segment of launch %zu is %u bytes, recorded %zu explicit", i, info[i].kernarg_size, L.args.size());
        OMX_REQUIRE(L.block.x * L.block.y * L.block.z <= 1024 && L.grid.x && L.grid.y && L.grid.z, "aql: bad launch geometry");
        off[i] = total;
        total += (base + kHiddenBytes + 63) / 64 * 64;
    }
    std::vector<unsigned char> stage(total, 0);
    for (size_t i = 0; i < n; ++i) {
        const RecordedLaunch& L = rec.launches[i];
        std::memcpy(stage.data() + off[i], L.args.data(), L.args.size());
        fill_hidden(stage.data() + off[i] + (L.args.size() + 7) / 8 * 8, L);
    }
    OMX_HIP_CHECK(hipMalloc(&p->kernargs, total));
    OMX_HIP_CHECK(hipMemcpy(p->kernargs, stage.data(), total, hipMemcpyHostToDevice));
    OMX_HIP_CHECK(hipDeviceSynchronize());
    p->pkts.resize(n);
    p->tags.resize(n);
    for (size_t i = 0; i < n; ++i) {
        const RecordedLaunch& L = rec.launches[i];
        hsa_kernel_dispatch_packet_t k = {};
        k.setup = (uint16_t)(L.grid.z > 1 || L.block.z > 1 ? 3 : L.grid.y > 1 || L.block.y > 1 ? 2 : 1);
        k.workgroup_size_x = (uint16_t)L.block.x; k.workgroup_size_y = (uint16_t)L.block.y; k.workgroup_size_z = (uint16_t)L.block.z;
        k.grid_size_x = L.grid.x * L.block.x; k.grid_size_y = L.grid.y * L.block.y; k.grid_size_z = L.grid.z * L.block.z;
        k.private_segment_size = info[i].priv;
        k.group_segment_size = info[i].group_static + L.lds;
        k.kernel_object = info[i].object;
        k.kernarg_address = (char*)p->kernargs + off[i];
        p->pkts[i] = k;
        p->tags[i] = L.tag;
    }
    return 0;
}

AqlProgram* aql_build(const LaunchRecorder& rec, int fence_mode) {
    Runtime& r = runtime();
    if (!r.ok) { set_error("aql: %s", r.why.c_str()); return nullptr; }
    std::lock_guard<std::mutex> lock(r.mu);
    AqlProgram* p = new AqlProgram;
    p->fence_mode = fence_mode;
    if (build_into(r, rec, p)) { aql_destroy(p); return nullptr; }
    return p;
}

static uint16_t packet_header(int acquire, int release, bool barrier = true) {
    return (uint16_t)((HSA_PACKET_TYPE_KERNEL_DISPATCH << HSA_PACKET_HEADER_TYPE) | ((barrier ? 1u : 0u) << HSA_PACKET_HEADER_BARRIER) |
                      ((unsigned)acquire << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) | ((unsigned)release << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE));
}

int aql_replay(AqlProgram* p, int times, double* wall_ms, float* per_launch_us) {
    OMX_REQUIRE(p && times > 0, "aql_replay: bad arguments");
    Runtime& r = runtime();
    OMX_REQUIRE(r.ok && !r.queue_error, "aql: the queue is unusable (%s)", r.queue_error ? "a packet was rejected" : r.why.c_str());
    std::lock_guard<std::mutex> lock(r.mu);
    hsa_queue_t* q = r.q;
    const size_t n = p->pkts.size();
    const int mid_acq = (p->fence_mode == AQL_FENCE_AGENT || p->fence_mode == AQL_FENCE_ACQUIRE) ? HSA_FENCE_SCOPE_AGENT : HSA_FENCE_SCOPE_NONE;
    const int mid_rel = (p->fence_mode == AQL_FENCE_AGENT || p->fence_mode == AQL_FENCE_RELEASE) ? HSA_FENCE_SCOPE_AGENT : HSA_FENCE_SCOPE_NONE;
    const bool timed = per_launch_us != nullptr;
    const unsigned nobarrier_mask = getenv("OMX_AQL_NOBARRIER") ? (unsigned)strtoul(getenv("OMX_AQL_NOBARRIER"), nullptr, 0) : 0u;
    if (timed) {
        while (r.launch_signals.size() < n) {
            hsa_signal_t s;
            OMX_REQUIRE(hsa_signal_create(1, 0, nullptr, &s) == HSA_STATUS_SUCCESS, "aql: hsa_signal_create failed");
            r.launch_signals.push_back(s);
        }
        OMX_REQUIRE(hsa_amd_profiling_set_profiler_enabled(q, 1) == HSA_STATUS_SUCCESS, "aql: queue profiling unavailable");
        for (size_t i = 0; i < n; ++i) per_launch_us[i] = 0.f;
    }
    const uint64_t wait_ticks = r.ticks_per_s ? r.ticks_per_s : 1000000000ull;   // one second per look
    const auto t0 = std::chrono::steady_clock::now();
    for (int t = 0; t < times; ++t) {
        const bool last_replay = t == times - 1 || timed;
        // room for a whole step
        for (unsigned spins = 0; hsa_queue_load_write_index_relaxed(q) + n - hsa_queue_load_read_index_scacquire(q) > q->size; ++spins)
            OMX_REQUIRE(spins < (1u << 30) && !r.queue_error, "aql: the queue stopped draining");
        if (last_replay) hsa_signal_store_relaxed(r.done, 1);
        const uint64_t base = hsa_queue_add_write_index_relaxed(q, n);
        for (size_t i = 0; i < n; ++i) {
            hsa_kernel_dispatch_packet_t* slot = (hsa_kernel_dispatch_packet_t*)q->base_address + ((base + i) & (q->size - 1));
            hsa_kernel_dispatch_packet_t k = p->pkts[i];
            const bool first = t == 0 && i == 0, last = last_replay && i == n - 1;
            if (last) k.completion_signal = r.done;
            else if (timed) { hsa_signal_store_relaxed(r.launch_signals[i], 1); k.completion_signal = r.launch_signals[i]; }
            // measurement only (OMX_AQL_NOBARRIER = bit mask of launch tags): those packets do not wait for their predecessors -- results are
            // void unless the kernels order themselves
            const bool barrier = first || last || p->tags[i] < 0 || !((nobarrier_mask >> p->tags[i]) & 1u);
            const uint16_t header = packet_header(first || (timed && i == 0) ? HSA_FENCE_SCOPE_SYSTEM : mid_acq, last ? HSA_FENCE_SCOPE_SYSTEM : mid_rel, barrier);
            std::memcpy((char*)slot + 4, (char*)&k + 4, sizeof(k) - 4);
            __atomic_store_n((uint32_t*)slot, (uint32_t)header | ((uint32_t)k.setup << 16), __ATOMIC_RELEASE);
        }
        hsa_signal_store_screlease(q->doorbell_signal, (hsa_signal_value_t)(base + n - 1));
        if (last_replay) {
            int looks = 0;
            while (hsa_signal_wait_scacquire(r.done, HSA_SIGNAL_CONDITION_LT, 1, wait_ticks, HSA_WAIT_STATE_BLOCKED) != 0) {
                OMX_REQUIRE(++looks < 60 && !r.queue_error, "aql: replay did not complete (%s)", r.queue_error ? "queue error" : "timeout");
            }
            if (timed) {
                for (size_t i = 0; i + 1 < n; ++i) {
                    hsa_amd_profiling_dispatch_time_t dt = {};
                    if (hsa_amd_profiling_get_dispatch_time(r.gpu, r.launch_signals[i], &dt) == HSA_STATUS_SUCCESS && r.ticks_per_s)
                        per_launch_us[i] += (float)((double)(dt.end - dt.start) * 1e6 / (double)r.ticks_per_s);
                }
                hsa_amd_profiling_dispatch_time_t dt = {};
                if (hsa_amd_profiling_get_dispatch_time(r.gpu, r.done, &dt) == HSA_STATUS_SUCCESS && r.ticks_per_s)
                    per_launch_us[n - 1] += (float)((double)(dt.end - dt.start) * 1e6 / (double)r.ticks_per_s);
            }
        }
    }
    const auto t1 = std::chrono::steady_clock::now();
    if (timed) {
        (void)hsa_amd_profiling_set_profiler_enabled(q, 0);
        for (size_t i = 0; i < n; ++i) per_launch_us[i] /= (float)times;
    }
    if (wall_ms) *wall_ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
    return 0;
}

}  // namespace omx
