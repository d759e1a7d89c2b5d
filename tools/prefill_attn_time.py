"""Causal flash attention at prefill shapes (Qwen3-8B heads): TFLOP/s by prompt length."""
import sys, os, time
sys.path.insert(0, os.getcwd())
import omx_import
omx = omx_import.load_package()
T = omx.ops.Tensor
H, Hkv, D = 32, 8, 128
for S in (2048, 8192, 16384):
    q = omx.ops.fill_uniform((1, H, S, D), 1, 1.0); k = omx.ops.fill_uniform((1, Hkv, S, D), 2, 1.0); v = omx.ops.fill_uniform((1, Hkv, S, D), 3, 1.0)
    out = T((1, H, S, D), "bf16")
    def run(n):
        for _ in range(n):
            omx.check(omx.lib.omx_sdpa(out.ptr, q.ptr, k.ptr, v.ptr, 1, H, Hkv, S, S, D, Hkv * S * D, S * D, 0.088, 1, None, 12, None))
        omx.ops.synchronize()
    run(2)
    t = time.perf_counter(); run(5); dt = (time.perf_counter() - t) / 5
    print("causal S", S, "ms", round(dt * 1e3, 3), "TF", round(2.0 * S * S * D * H / dt / 1e12, 1), flush=True)
