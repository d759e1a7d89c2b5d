"""What this chip streams in ONE dependency-free launch: the plain bf16 GEMV kernel (csrc/gemv.hip, no prologue, no epilogue) over a
single [N, 4096] matrix of a layer's bytes (386 MB), the head's (1.24 GB) and the whole decode step's (15.44 GB for Qwen3-8B at context
2 048) -- the ceiling the step's 0.64 of 8 TB/s is to be read against (VERDICT r5 "Next" 2).  HIP events on the launch's stream,
rotating buffers where the matrix is smaller than the 256 MiB Infinity Cache would need.
usage: python tools/stream_ceiling.py"""
import ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import omx_import  # noqa: E402
omx = omx_import.load_package()
omx.lib.omx_bench_gemv.restype = ctypes.c_int
omx.lib.omx_bench_gemv.argtypes = [ctypes.c_int] * 7 + [ctypes.POINTER(ctypes.c_float)]


def ceiling(nbytes, K=4096, iters=5):
    N = int(nbytes // (2 * K)) // 512 * 512
    copies = 1 if N * K * 2 > (2 << 30) else 3
    ms = ctypes.c_float(0)
    omx.check(omx.lib.omx_bench_gemv(N, K, 0, 0, 0, copies, iters, ctypes.byref(ms)))
    return {"bytes": N * K * 2, "us": round(ms.value * 1e3, 2), "GBps": round(N * K * 2 / (ms.value * 1e-3) / 1e9, 1)}


if __name__ == "__main__":
    for label, b in (("layer", 386e6), ("lm_head", 1.2447e9), ("step", 15.44e9)):
        print(json.dumps({"what": label, **ceiling(b)}), flush=True)
