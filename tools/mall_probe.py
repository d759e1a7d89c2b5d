"""Does a weight matrix that was just read stay in the Infinity Cache / L2?  Same GEMV, same
buffer re-read every iteration (copies=1) vs rotating buffers (cold)."""
import ctypes, sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import omx_import
omx = omx_import.load_package()
lib = omx.lib
lib.omx_bench_gemv.restype = ctypes.c_int
lib.omx_bench_gemv.argtypes = [ctypes.c_int] * 7 + [ctypes.POINTER(ctypes.c_float)]
for name, N, K, pro, epi in [("o_proj 33MB", 4096, 4096, 0, 1), ("qkv 50MB", 6144, 4096, 1, 0), ("down 100MB", 4096, 12288, 0, 1),
                             ("gate_up 201MB", 12288, 4096, 1, 2), ("8k x 4k 67MB", 8192, 4096, 0, 0), ("16k x 4k 134MB", 16384, 4096, 0, 0),
                             ("28k x 4k 235MB", 28672, 4096, 0, 0), ("40k x 4k 335MB", 40960, 4096, 0, 0)]:
    mats = 2 if epi == 2 else 1
    nbytes = N * K * 2 * mats
    for copies in (1, max(2, int(800e6 // nbytes) + 1)):
        ms = ctypes.c_float()
        omx.check(lib.omx_bench_gemv(N, K, pro, epi, 0, copies, 40, ctypes.byref(ms)))
        print(json.dumps({"kernel": name, "copies": copies, "us": round(ms.value * 1e3, 2), "GBps": round(nbytes / ms.value / 1e6, 1)}), flush=True)
