"""Per-tile overhead O and per-step time X of the 256^2 GEMM kernels from a two-point fit over K (run on the GPU box):
8 192 x 8 192 outputs = 1 024 tiles = exactly four rounds of 256, so t(K) = 4 (O + X K / 64).  OMX_GEMM_W4=0 / 1 picks the kernel."""
import ctypes, sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["OMX_GEMM_TILE"] = "256"
import omx_import
omx = omx_import.load_package()
lib = omx.lib
lib.omx_bench_gemm.restype = ctypes.c_int
lib.omx_bench_gemm.argtypes = [ctypes.c_int] * 5 + [ctypes.POINTER(ctypes.c_float)]
def t_us(M, N, K):
    ms = ctypes.c_float()
    omx.check(lib.omx_bench_gemm(M, N, K, 3, 20, ctypes.byref(ms)))
    return ms.value * 1e3
variants = [("0", "0"), ("1", "0")] + ([("1", "8")] if os.environ.get("OMX_LIB_VARIANT") else [])      # (diag build: + the K loop without its epilogue)
for w4, var in variants:
    os.environ["OMX_GEMM_W4"] = w4
    os.environ["OMX_GEMM_W4_VAR"] = var
    for rnd in range(2):
        t1, t2, t3 = t_us(8192, 8192, 2048), t_us(8192, 8192, 4096), t_us(8192, 8192, 8192)
        x = (t3 - t1) / 4 / ((8192 - 2048) / 64)
        o = t1 / 4 - x * 2048 / 64
        print(json.dumps({"kernel": ("four-wave" if w4 == "1" else "eight-wave") + (", no epilogue" if var == "8" else ""), "us_K2048": round(t1, 1), "us_K4096": round(t2, 1), "us_K8192": round(t3, 1),
                          "X_us_per_64k_step": round(x, 3), "O_us_per_tile": round(o, 2), "check_K4096": round(4 * (o + x * 64), 1)}), flush=True)
