import ctypes, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import omx_import
omx = omx_import.load_package()
lib = omx.lib
lib.omx_bench_gemm.restype = ctypes.c_int
lib.omx_bench_gemm.argtypes = [ctypes.c_int] * 5 + [ctypes.POINTER(ctypes.c_float)]
for M, N, K in ((8192, 8192, 8192), (4608, 27648, 3072), (2048, 24576, 4096)):
    ms = ctypes.c_float()
    omx.check(lib.omx_bench_gemm(M, N, K, 1, 2, ctypes.byref(ms)))
    x = (torch.randn(M, K, device="cuda") * 0.1).to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") * 0.1).to(torch.bfloat16)
    for _ in range(3):
        y = torch.nn.functional.linear(x, w)
    torch.cuda.synchronize()
