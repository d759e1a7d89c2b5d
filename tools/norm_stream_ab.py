"""Row-norm launches of the DiT / prefill against a plain streaming kernel of the same bytes, in isolation (device events over
back-to-back launches).  usage: python tools/norm_stream_ab.py [rows] [dim]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import omx_import  # noqa: E402
omx = omx_import.load_package()
from ominix_mlx_amd import ops  # noqa: E402
import torch  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 4608
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 3072
rng = np.random.default_rng(0)
x = ops.Tensor.from_numpy(rng.standard_normal((1, rows, dim)).astype(np.float32), "bf16")
sh = ops.Tensor.from_numpy(rng.standard_normal((1, dim)).astype(np.float32), "bf16")
sc = ops.Tensor.from_numpy(rng.standard_normal((1, dim)).astype(np.float32), "bf16")
w = ops.Tensor.from_numpy(rng.standard_normal((dim,)).astype(np.float32), "bf16")
out = ops.empty_like(x)
lib = omx.lib


def timed(name, fn, n=200):
    for _ in range(10):
        fn()
    omx.check(lib.omx_synchronize(None))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    import time
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    omx.check(lib.omx_synchronize(None))
    dt = (time.perf_counter() - t0) / n
    gb = rows * dim * 2 * 2 / 1e9
    print(f"{name:28s} {dt * 1e6:7.2f} us   {gb / dt / 1e3:5.2f} TB/s (read + write of [{rows}, {dim}] bf16)", flush=True)


for keep in ("1", "0"):
    os.environ["OMX_NORM_KEEP"] = keep
    timed(f"fused_modulate keep={keep}", lambda: omx.check(lib.omx_fused_modulate(out.ptr, x.ptr, sh.ptr, sc.ptr, 1, rows, dim, 1e-6, x.dtype, None)))
    timed(f"rms_norm keep={keep}", lambda: omx.check(lib.omx_rms_norm(out.ptr, x.ptr, w.ptr, rows, dim, 1e-6, x.dtype, None)))
timed("add (3 streams: x1.5 bytes)", lambda: omx.check(lib.omx_add(out.ptr, x.ptr, x.ptr, rows * dim, x.dtype, None)))
