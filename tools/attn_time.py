import sys, os, ctypes, time
sys.path.insert(0, os.getcwd())
import numpy as np, omx_import
omx = omx_import.load_package()
T = omx.ops.Tensor
S, H, D = 4608, 24, 128
q = omx.ops.fill_uniform((1, H, S, D), 1, 1.0); k = omx.ops.fill_uniform((1, H, S, D), 2, 1.0); v = omx.ops.fill_uniform((1, H, S, D), 3, 1.0)
out = T((1, H, S, D), "bf16")
def run(n):
    for _ in range(n):
        omx.check(omx.lib.omx_sdpa(out.ptr, q.ptr, k.ptr, v.ptr, 1, H, H, S, S, D, H*S*D, S*D, 0.088, 0, None, 12, None))
    omx.ops.synchronize()
run(2)
t = time.perf_counter(); run(5); dt = (time.perf_counter() - t) / 5
print("dbg", os.environ.get("OMX_ATTN_DEBUG", "0"), "ms", round(dt * 1e3, 3), "TF", round(4.0 * S * S * D * H / dt / 1e12, 1), flush=True)
