"""A/B of one GEMM environment switch on one box: TF/s per shape for each value, interleaved twice.
    python tools/gemm_env_ab.py OMX_GEMM_W4_PERSIST 0 1      (further KEY=VALUE arguments are set for the whole run)"""
import ctypes, sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
args = [a for a in sys.argv[1:] if "=" not in a]
for a in sys.argv[1:]:
    if "=" in a:
        k, v = a.split("=", 1)
        os.environ[k] = v
import omx_import
omx = omx_import.load_package()
lib = omx.lib
lib.omx_bench_gemm.restype = ctypes.c_int
lib.omx_bench_gemm.argtypes = [ctypes.c_int] * 5 + [ctypes.POINTER(ctypes.c_float)]
shapes = [("prefill q/o", 2048, 4096, 4096), ("prefill gate/up", 2048, 24576, 4096), ("prefill down", 2048, 4096, 12288), ("klein qkv_mlp", 4608, 27648, 3072),
          ("klein to_out", 4608, 3072, 12288), ("klein txt+img mlp_in", 4608, 18432, 3072), ("square 8k", 8192, 8192, 8192)]
key, values = args[0], args[1:]
for name, M, N, K in shapes:
    row = {"gemm": name}
    for rnd in range(2):
        for v in values:
            os.environ[key] = v
            ms = ctypes.c_float()
            omx.check(lib.omx_bench_gemm(M, N, K, 3, 20, ctypes.byref(ms)))
            row.setdefault(f"{key}={v}", []).append(round(2.0 * M * N * K / ms.value / 1e9))
    print(json.dumps(row), flush=True)
