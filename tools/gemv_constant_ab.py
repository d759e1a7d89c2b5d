"""Launch constants of the bf16 and the packed 4-bit GEMV: back-to-back launches on tiny matrices (the bytes are negligible, the time is the
kernel's own latency chain + the launch boundary), per prologue / epilogue.  usage: python tools/gemv_constant_ab.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import omx_import
omx = omx_import.load_package()
lib = omx.lib
lib.omx_bench_qgemv.restype = ctypes.c_int
lib.omx_bench_qgemv.argtypes = [ctypes.c_int] * 7 + [ctypes.POINTER(ctypes.c_float)]
lib.omx_bench_gemv.restype = ctypes.c_int
lib.omx_bench_gemv.argtypes = [ctypes.c_int] * 7 + [ctypes.POINTER(ctypes.c_float)]
PRO = {"none": 0, "rms": 1}
EPI = {"store": 0, "resid": 1, "swiglu": 2}
for K in (4096, 12288):
    for N in (512, 2048, 4096):
        for pro, epi in (("none", "store"), ("rms", "store"), ("none", "resid"), ("rms", "swiglu")):
            if K != 4096 and pro == "rms":
                continue
            ms = ctypes.c_float(); ms2 = ctypes.c_float()
            omx.check(lib.omx_bench_gemv(N, K, PRO[pro], EPI[epi], 0, 8, 400, ctypes.byref(ms)))
            omx.check(lib.omx_bench_qgemv(N, K, 4, PRO[pro], EPI[epi], 8, 400, ctypes.byref(ms2)))
            mats = 2 if epi == "swiglu" else 1
            print(f"N={N:5d} K={K:5d} {pro:4s}/{epi:6s}: bf16 {ms.value*1e3:6.2f} us ({mats*N*K*2/1e6:6.1f} MB)   4-bit {ms2.value*1e3:6.2f} us ({mats*N*K*0.5625/1e6:6.1f} MB)", flush=True)
