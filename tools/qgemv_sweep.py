"""Packed-weight GEMV microbenchmark (run on the GPU box): achieved GB/s of packed bytes per shape."""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import omx_import
omx = omx_import.load_package()
lib = omx.lib
lib.omx_bench_qgemv.restype = ctypes.c_int
lib.omx_bench_qgemv.argtypes = [ctypes.c_int] * 7 + [ctypes.POINTER(ctypes.c_float)]
PRO = {"none": 0, "rms": 1}
EPI = {"store": 0, "resid": 1, "swiglu": 2, "argmax": 3}
shapes = [("qkv", 6144, 4096, "rms", "store"), ("o_proj", 4096, 4096, "none", "resid"), ("gate_up", 12288, 4096, "rms", "swiglu"),
          ("down", 4096, 12288, "none", "resid"), ("lm_head", 151936, 4096, "rms", "argmax")]
bits = int(sys.argv[1]) if len(sys.argv) > 1 else 4
for name, N, K, pro, epi in shapes:
    mats = 2 if epi == "swiglu" else 1
    nbytes = mats * N * K * (bits / 8 + 4 / 64)
    copies = max(2, int(6e8 // nbytes))
    ms = ctypes.c_float()
    omx.check(lib.omx_bench_qgemv(N, K, bits, PRO[pro], EPI[epi], copies, copies * 3, ctypes.byref(ms)))
    print(f"{name:8s} N={N:6d} K={K:5d}: {ms.value*1e3:8.1f} us  {nbytes/ms.value/1e6:7.1f} GB/s  ({nbytes/1e6:.1f} MB)", flush=True)
