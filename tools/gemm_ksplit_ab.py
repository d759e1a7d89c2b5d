"""A/B of the 256^2 GEMM kernel's two-way K split on the prefill shapes whose 8 x 16 tile grid covers half of the chip
(OMX_GEMM_KSPLIT=0/1, read per launch).  usage: python tools/gemm_ksplit_ab.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import omx_import
omx = omx_import.load_package()
for M, N, K in ((2048, 4096, 4096), (2048, 4096, 12288), (2048, 2048, 8192), (1024, 4096, 12288), (4608, 3072, 3072), (512, 9216, 3072), (512, 3072, 3072), (512, 3072, 9216)):
    x = omx.ops.fill_uniform((M, K), 1, 1.0); w = omx.ops.fill_uniform((N, K), 2, 0.05)
    out = omx.ops.Tensor((M, N), "bf16")
    run = lambda: omx.check(omx.lib.omx_linear(out.ptr, x.ptr, w.ptr, None, M, N, K, x.dtype, None))   # (no allocation in the timed loop)
    for mode in ("0", "1", "tile128", "tile256"):
        os.environ.pop("OMX_GEMM_TILE", None)
        os.environ["OMX_GEMM_KSPLIT"] = "0" if mode != "1" else "1"
        if mode.startswith("tile"):
            os.environ["OMX_GEMM_TILE"] = mode[4:]
        for _ in range(3):
            run()
        omx.ops.synchronize()
        t = time.perf_counter()
        for _ in range(20):
            run()
        omx.ops.synchronize()
        dt = (time.perf_counter() - t) / 20
        print(f"M={M} N={N} K={K} mode {mode:8s} {dt * 1e6:8.1f} us  {2.0 * M * N * K / dt / 1e12:7.1f} TF/s", flush=True)
