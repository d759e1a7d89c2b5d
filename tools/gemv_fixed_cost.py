"""What a decode-step launch costs beyond its bytes: time of the streaming GEMV (RMSNorm prologue, bf16 store; csrc/gemv.hip through
omx_bench_gemv: back-to-back launches over distinct weight buffers, HIP events) for N x 4096 matrices from 4 MB to 1.2 GB, and the
least-squares line  t = t0 + bytes / rate.  t0 is what every launch of the step pays whatever it streams (boundary + first-byte
latency + tail); rate is the chip's streaming ceiling for this access pattern.  The packed 4-bit GEMV (csrc/quant.hip) likewise.
usage: python tools/gemv_fixed_cost.py"""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import omx_import  # noqa: E402

omx = omx_import.load_package()
lib = omx.lib
for f in (lib.omx_bench_gemv, lib.omx_bench_qgemv):
    f.restype = ctypes.c_int
lib.omx_bench_gemv.argtypes = [ctypes.c_int] * 7 + [ctypes.POINTER(ctypes.c_float)]
lib.omx_bench_qgemv.argtypes = [ctypes.c_int] * 7 + [ctypes.POINTER(ctypes.c_float)]
K = 4096
ROWS = [512, 1024, 2048, 4096, 6144, 8192, 12288, 24576, 49152, 98304, 151936]


def fit(pts, label):
    b = np.array([p[0] for p in pts], float)
    t = np.array([p[1] for p in pts], float)
    A = np.stack([np.ones_like(b), b], 1)
    (t0, inv), *_ = np.linalg.lstsq(A, t, rcond=None)
    print(f"{label}: t = {t0 * 1e6:.2f} us + bytes / {1 / inv / 1e12:.2f} TB/s   (max residual {np.abs(A @ [t0, inv] - t).max() * 1e6:.2f} us)")


pts = []
for N in ROWS:
    nbytes = N * K * 2
    copies = max(3, int(1.5e9 // nbytes))
    copies = min(copies, 256)
    ms = ctypes.c_float()
    omx.check(lib.omx_bench_gemv(N, K, 1, 0, 0, copies, copies * 3, ctypes.byref(ms)))      # RMSNorm prologue, bf16 store
    pts.append((nbytes, ms.value * 1e-3))
    print(f"bf16  N={N:6d}: {nbytes / 1e6:8.1f} MB  {ms.value * 1e3:8.2f} us  {nbytes / ms.value / 1e9:7.2f} TB/s", flush=True)
fit(pts, "bf16 GEMV ")
pts = []
for N in ROWS:
    nbytes = N * K * (0.5 + 4 / 64)
    copies = min(256, max(3, int(6e8 // nbytes)))
    ms = ctypes.c_float()
    omx.check(lib.omx_bench_qgemv(N, K, 4, 1, 0, copies, copies * 3, ctypes.byref(ms)))
    pts.append((nbytes, ms.value * 1e-3))
    print(f"4-bit N={N:6d}: {nbytes / 1e6:8.1f} MB  {ms.value * 1e3:8.2f} us  {nbytes / ms.value / 1e9:7.2f} TB/s", flush=True)
fit(pts, "4-bit GEMV")
