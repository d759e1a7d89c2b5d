"""bf16 MFMA GEMM microbenchmark (run on the GPU box): TFLOP/s per shape."""
import ctypes, sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import omx_import
omx = omx_import.load_package()
lib = omx.lib
lib.omx_bench_gemm.restype = ctypes.c_int
lib.omx_bench_gemm.argtypes = [ctypes.c_int] * 5 + [ctypes.POINTER(ctypes.c_float)]
shapes = [("prefill q/o", 2048, 4096, 4096), ("prefill gate", 2048, 12288, 4096), ("prefill down", 2048, 4096, 12288),
          ("klein qkv_mlp", 4608, 27648, 3072), ("klein to_out", 4608, 3072, 12288), ("klein mlp_in", 4096, 18432, 3072),
          ("square 4k", 4096, 4096, 4096), ("square 8k", 8192, 8192, 8192),
          # grids the 128^2 kernel cannot fill (64^2 ring kernel; OMX_GEMM_TILE=128 for the A/B)
          ("paraformer qkv", 501, 1536, 512), ("paraformer out", 501, 512, 512), ("paraformer ffn up", 501, 2048, 512),
          ("paraformer ffn down", 501, 512, 2048), ("paraformer decoder ffn", 216, 512, 2048),
          ("encoder-4b qkv", 512, 6144, 2560), ("encoder-4b down", 512, 2560, 9728), ("klein txt to_q", 512, 3072, 3072),
          ("klein txt mlp_out", 512, 3072, 9216), ("prompt-128 q", 128, 4096, 4096)]
if len(sys.argv) > 1 and sys.argv[1] == "skinny":
    shapes = [sh for sh in shapes if sh[1] <= 512]
for name, M, N, K in shapes:
    ms = ctypes.c_float()
    omx.check(lib.omx_bench_gemm(M, N, K, 3, 20, ctypes.byref(ms)))
    print(json.dumps({"gemm": name, "M": M, "N": N, "K": K, "us": round(ms.value * 1e3, 1), "TFLOPs": round(2.0 * M * N * K / ms.value / 1e9, 1)}), flush=True)
