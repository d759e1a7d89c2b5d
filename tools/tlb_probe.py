"""Where does the ~2.5 us that a COLD weight matrix costs a GEMV come from: address translation or data?"""
import ctypes, sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import omx_import
omx = omx_import.load_package()
lib = omx.lib
lib.omx_bench_gemv_warm.restype = ctypes.c_int
lib.omx_bench_gemv_warm.argtypes = [ctypes.c_int] * 7 + [ctypes.c_long, ctypes.c_int, ctypes.POINTER(ctypes.c_float)]
which = sys.argv[1:] or ["o", "qkv"]
shapes = {"o": ("o_proj 33MB", 4096, 4096, 0, 1), "qkv": ("qkv 50MB", 6144, 4096, 1, 0), "down": ("down 100MB", 4096, 12288, 0, 1)}
for key in which:
    name, N, K, pro, epi = shapes[key]
    nbytes = N * K * 2
    copies = int(1.6e9 // nbytes)
    for label, mode, stride, one in [("cold, separate allocations", 0, 0, 0), ("cold, one allocation", 0, 0, 1),
                                     ("touch 2MB stride", 1, 2 << 20, 1), ("touch 64KB stride", 1, 64 << 10, 1),
                                     ("touch 4KB stride", 1, 4 << 10, 1), ("stream whole matrix first", 2, 0, 1),
                                     ("same matrix every time", 0, 0, -1)]:
        ms = ctypes.c_float()
        c = 1 if one < 0 else copies
        omx.check(lib.omx_bench_gemv_warm(N, K, pro, epi, c, 60, mode, stride, max(one, 0), ctypes.byref(ms)))
        print(json.dumps({"kernel": name, "case": label, "us_per_iter": round(ms.value * 1e3, 2)}), flush=True)
