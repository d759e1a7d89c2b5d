"""GEMV microbenchmark sweep (run on the GPU box): achieved algorithmic GB/s per shape/config."""
import ctypes, sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import omx_import
omx = omx_import.load_package()
lib = omx.lib
lib.omx_bench_gemv.restype = ctypes.c_int
lib.omx_bench_gemv.argtypes = [ctypes.c_int] * 7 + [ctypes.POINTER(ctypes.c_float)]
PRO = {"none": 0, "rms": 1}
EPI = {"store": 0, "resid": 1, "swiglu": 2, "argmax": 3, "f32": 4}
shapes = [  # (name, N, K, pro, epi)
    ("qkv", 6144, 4096, "rms", "store"),
    ("o_proj", 4096, 4096, "none", "resid"),
    ("gate_up", 12288, 4096, "rms", "swiglu"),
    ("down", 4096, 12288, "none", "resid"),
    ("lm_head", 151936, 4096, "rms", "argmax"),
]
rpws = [int(a) for a in sys.argv[1:]] or [0, 2, 4, 8, 16]
for name, N, K, pro, epi in shapes:
    mats = 2 if epi == "swiglu" else 1
    nbytes = N * K * 2 * mats
    copies = max(2, int(600e6 // nbytes) + 1)
    for rpw in rpws:
        ms = ctypes.c_float()
        st = lib.omx_bench_gemv(N, K, PRO[pro], EPI[epi], rpw, copies, 50, ctypes.byref(ms))
        if st:
            print(name, rpw, "ERR", lib.omx_last_error().decode()); continue
        print(json.dumps({"kernel": name, "N": N, "K": K, "rpw": rpw, "us": round(ms.value * 1e3, 2),
                          "GBps": round(nbytes / ms.value / 1e6, 1)}), flush=True)
