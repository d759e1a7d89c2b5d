// Can a dependent-free chain of GEMV-shaped launches overlap its kernel boundaries when the AQL packets carry no barrier bit?
// (hipExtAnyOrderLaunch is unsupported on gfx9xx; this goes to the HSA queue directly.)  Submits `chain` dispatches of the probe kernel
// over distinct weight buffers to an own HSA queue, with and without the barrier bit, and reports the period per launch.
//   build: see tools/aql_probe/build.sh          run on the GPU box: ./tools/aql_probe/probe tools/aql_probe/kernel.hsaco
#include <hip/hip_runtime.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <vector>

#define HSA(x) do { hsa_status_t s_ = (x); if (s_ != HSA_STATUS_SUCCESS) { const char* m = nullptr; hsa_status_string(s_, &m); \
    fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, m ? m : "?"); exit(1); } } while (0)
#define HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Args { const uint16_t* w; const uint16_t* x; float* out; int N; int K; int rows_per_wave; int pad; };

static hsa_agent_t g_gpu; static bool g_have = false;
static hsa_status_t find_gpu(hsa_agent_t a, void*) {
    hsa_device_type_t t; hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t);
    if (t == HSA_DEVICE_TYPE_GPU && !g_have) { g_gpu = a; g_have = true; }
    return HSA_STATUS_SUCCESS;
}
static hsa_amd_memory_pool_t g_kernarg_pool; static bool g_have_pool = false;
static hsa_status_t find_pool(hsa_amd_memory_pool_t p, void*) {
    hsa_amd_segment_t seg; hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_SEGMENT, &seg);
    uint32_t flags = 0; hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_GLOBAL_FLAGS, &flags);
    if (seg == HSA_AMD_SEGMENT_GLOBAL && (flags & HSA_AMD_MEMORY_POOL_GLOBAL_FLAG_KERNARG_INIT) && !g_have_pool) { g_kernarg_pool = p; g_have_pool = true; }
    return HSA_STATUS_SUCCESS;
}
static hsa_agent_t g_cpu; static bool g_have_cpu = false;
static hsa_status_t find_cpu(hsa_agent_t a, void*) {
    hsa_device_type_t t; hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t);
    if (t == HSA_DEVICE_TYPE_CPU && !g_have_cpu) { g_cpu = a; g_have_cpu = true; }
    return HSA_STATUS_SUCCESS;
}

int main(int argc, char** argv) {
    const char* path = argc > 1 ? argv[1] : "tools/aql_probe/kernel.hsaco";
    const int N = argc > 2 ? atoi(argv[2]) : 6144, K = 4096, rpw = argc > 3 ? atoi(argv[3]) : 4, chain = 64, copies = 24;
    HIP(hipSetDevice(0));
    HIP(hipFree(0));
    HSA(hsa_init());                                     // reference-counted: HIP already initialised the same runtime
    HSA(hsa_iterate_agents(find_gpu, nullptr));
    HSA(hsa_iterate_agents(find_cpu, nullptr));
    HSA(hsa_amd_agent_iterate_memory_pools(g_cpu, find_pool, nullptr));
    if (!g_have || !g_have_pool) { fprintf(stderr, "no gpu agent / kernarg pool\n"); return 1; }
    // ---- load the code object ----
    std::ifstream f(path, std::ios::binary);
    std::vector<char> blob((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    if (blob.empty()) { fprintf(stderr, "cannot read %s\n", path); return 1; }
    hsa_code_object_reader_t reader; hsa_executable_t exe;
    HSA(hsa_code_object_reader_create_from_memory(blob.data(), blob.size(), &reader));
    HSA(hsa_executable_create_alt(HSA_PROFILE_FULL, HSA_DEFAULT_FLOAT_ROUNDING_MODE_DEFAULT, nullptr, &exe));
    HSA(hsa_executable_load_agent_code_object(exe, g_gpu, reader, nullptr, nullptr));
    HSA(hsa_executable_freeze(exe, nullptr));
    hsa_executable_symbol_t sym;
    HSA(hsa_executable_get_symbol_by_name(exe, "stream_gemv.kd", &g_gpu, &sym));
    uint64_t kobj = 0; uint32_t kernarg_size = 0, group = 0, priv = 0;
    HSA(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_OBJECT, &kobj));
    HSA(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_KERNARG_SEGMENT_SIZE, &kernarg_size));
    HSA(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_GROUP_SEGMENT_SIZE, &group));
    HSA(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_PRIVATE_SEGMENT_SIZE, &priv));
    printf("kernel object %#llx kernarg %u B group %u B private %u B\n", (unsigned long long)kobj, kernarg_size, group, priv);
    // ---- buffers (HIP allocations are visible to HSA dispatches of the same process) ----
    std::vector<uint16_t*> w(copies);
    for (auto& p : w) { HIP(hipMalloc((void**)&p, (size_t)N * K * 2)); HIP(hipMemset(p, 0x3c, (size_t)N * K * 2)); }
    uint16_t* x; float* out;
    HIP(hipMalloc((void**)&x, K * 2)); HIP(hipMemset(x, 0x3c, K * 2));
    HIP(hipMalloc((void**)&out, (size_t)N * 4));
    HIP(hipDeviceSynchronize());
    char* kernargs = nullptr;
    const size_t ka_stride = (kernarg_size + 63) / 64 * 64;
    // kernargs in DEVICE memory (argv[4] = "host": the CPU agent's kernarg pool instead -- every wave's first s_load is then a PCIe
    // round trip, which is what made the first version of this probe read 17.5 us per barrier launch)
    const bool host_ka = argc > 4 && !strcmp(argv[4], "host");
    std::vector<char> stage(ka_stride * chain, 0);
    for (int i = 0; i < chain; ++i) {
        Args a = {w[i % copies], x, out, N, K, rpw, 0};
        memcpy(stage.data() + i * ka_stride, &a, sizeof(a));
    }
    if (host_ka) {
        HSA(hsa_amd_memory_pool_allocate(g_kernarg_pool, ka_stride * chain, 0, (void**)&kernargs));
        HSA(hsa_amd_agents_allow_access(1, &g_gpu, nullptr, kernargs));
        memcpy(kernargs, stage.data(), stage.size());
    } else {
        HIP(hipMalloc((void**)&kernargs, stage.size()));
        HIP(hipMemcpy(kernargs, stage.data(), stage.size(), hipMemcpyHostToDevice));
        HIP(hipDeviceSynchronize());
    }
    hsa_queue_t* q = nullptr;
    HSA(hsa_queue_create(g_gpu, 4096, HSA_QUEUE_TYPE_SINGLE, nullptr, nullptr, UINT32_MAX, UINT32_MAX, &q));
    hsa_signal_t done;
    HSA(hsa_signal_create(1, 0, nullptr, &done));
    const int groups = (N + rpw - 1) / rpw, blocks = (groups + 3) / 4;
    auto run = [&](bool barrier, int fence) -> double {
        hsa_signal_store_relaxed(done, 1);
        const uint64_t base = hsa_queue_add_write_index_relaxed(q, chain);
        for (int i = 0; i < chain; ++i) {
            hsa_kernel_dispatch_packet_t* pk = (hsa_kernel_dispatch_packet_t*)q->base_address + ((base + i) & (q->size - 1));
            hsa_kernel_dispatch_packet_t p = {};
            p.setup = 1;                                                   // 1 dimension
            p.workgroup_size_x = 256; p.workgroup_size_y = 1; p.workgroup_size_z = 1;
            p.grid_size_x = (uint32_t)blocks * 256; p.grid_size_y = 1; p.grid_size_z = 1;
            p.private_segment_size = priv; p.group_segment_size = group;
            p.kernel_object = kobj; p.kernarg_address = kernargs + i * ka_stride;
            p.completion_signal = (i == chain - 1) ? done : hsa_signal_t{0};
            memcpy((char*)pk + 4, (char*)&p + 4, sizeof(p) - 4);
            uint16_t header = (HSA_PACKET_TYPE_KERNEL_DISPATCH << HSA_PACKET_HEADER_TYPE) | (fence << HSA_PACKET_HEADER_ACQUIRE_FENCE_SCOPE) |
                              (fence << HSA_PACKET_HEADER_RELEASE_FENCE_SCOPE) | ((barrier ? 1 : 0) << HSA_PACKET_HEADER_BARRIER);
            if (i == chain - 1) header |= 1 << HSA_PACKET_HEADER_BARRIER;  // the last one waits for all: its signal ends the chain
            __atomic_store_n((uint32_t*)pk, (uint32_t)header | ((uint32_t)p.setup << 16), __ATOMIC_RELEASE);
        }
        const auto t0 = std::chrono::steady_clock::now();
        hsa_signal_store_screlease(q->doorbell_signal, base + chain - 1);
        while (hsa_signal_wait_scacquire(done, HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_ACTIVE) != 0) {}
        const auto t1 = std::chrono::steady_clock::now();
        return std::chrono::duration<double, std::micro>(t1 - t0).count() / chain;
    };
    // the same code object through HIP's own stream (barrier bit on every packet, HIP's fences)
    hipModule_t mod; hipFunction_t fn;
    HIP(hipModuleLoadData(&mod, blob.data()));
    HIP(hipModuleGetFunction(&fn, mod, "stream_gemv"));
    hipStream_t hs; HIP(hipStreamCreate(&hs));
    auto run_hip = [&]() -> double {
        HIP(hipStreamSynchronize(hs));
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < chain; ++i) {
            Args a = {w[i % copies], x, out, N, K, rpw, 0};
            size_t sz = sizeof(a);
            void* cfg[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &a, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
            HIP(hipModuleLaunchKernel(fn, blocks, 1, 1, 256, 1, 1, 0, hs, nullptr, cfg));
        }
        HIP(hipStreamSynchronize(hs));
        const auto t1 = std::chrono::steady_clock::now();
        return std::chrono::duration<double, std::micro>(t1 - t0).count() / chain;
    };
    run_hip();
    const double mb = (double)N * K * 2 / 1e6;
    for (int rep = 0; rep < 3; ++rep) {
        const double a = run(true, HSA_FENCE_SCOPE_AGENT), b = run(false, HSA_FENCE_SCOPE_AGENT), c = run(false, HSA_FENCE_SCOPE_NONE),
                     d = run(true, HSA_FENCE_SCOPE_NONE);
        const double h = run_hip();
        printf("HIP stream %.2f us | ", h);
        printf("N=%d rpw=%d blocks=%d %.1f MB | barrier+agent fences %.2f us (%.2f TB/s) | no barrier, agent fences %.2f us | no barrier, no fences %.2f us (%.2f TB/s) | barrier, no fences %.2f us\n",
               N, rpw, blocks, mb, a, mb / a / 1e3, b, c, mb / c / 1e3, d);
    }
    return 0;
}
