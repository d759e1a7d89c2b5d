"""What the vendor library reaches on THIS box for the shapes the hot path runs (run on the GPU box): torch.matmul (hipBLASLt / rocBLAS
behind it) next to omx's own GEMM, interleaved per shape so DVFS and box-to-box variance hit both alike.  A measurement of the
ceiling a bf16 NT GEMM is known to reach on gfx950 -- not a code path of the product (nothing in the package calls torch.matmul)."""
import ctypes, sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import omx_import
omx = omx_import.load_package()
lib = omx.lib
lib.omx_bench_gemm.restype = ctypes.c_int
lib.omx_bench_gemm.argtypes = [ctypes.c_int] * 5 + [ctypes.POINTER(ctypes.c_float)]
import os
if len(sys.argv) > 1 and sys.argv[1] == "few":      # a few CUs only: no chip-wide power limit, what is left is the kernel's own stalls
    shapes = [("16 tiles", 1024, 1024, 16384), ("64 tiles", 2048, 2048, 16384), ("256 tiles", 4096, 4096, 16384)]
    os.environ["OMX_GEMM_TILE"] = "256"
else:
  shapes = [("prefill q/o", 2048, 4096, 4096), ("prefill gate/up", 2048, 24576, 4096), ("prefill down", 2048, 4096, 12288),
          ("klein qkv_mlp", 4608, 27648, 3072), ("klein to_out", 4608, 3072, 12288), ("klein img mlp_in", 4096, 18432, 3072),
          ("klein txt to_q", 512, 3072, 3072), ("square 4k", 4096, 4096, 4096), ("square 8k", 8192, 8192, 8192)]


def torch_us(M, N, K, reps=20):
    x = (torch.randn(M, K, device="cuda") * 0.1).to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") * 0.1).to(torch.bfloat16)
    for _ in range(3):
        y = torch.nn.functional.linear(x, w)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        y = torch.nn.functional.linear(x, w)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


for name, M, N, K in shapes:
    row = {"gemm": name, "M": M, "N": N, "K": K}
    for rnd in range(2):
        ms = ctypes.c_float()
        omx.check(lib.omx_bench_gemm(M, N, K, 3, 20, ctypes.byref(ms)))
        row[f"omx_TF_{rnd}"] = round(2.0 * M * N * K / ms.value / 1e9, 1)
        row[f"blaslt_TF_{rnd}"] = round(2.0 * M * N * K / torch_us(M, N, K) / 1e6, 1)
    tiles = ((M + 255) // 256) * ((N + 255) // 256)
    per_cu = K // 64 * ((tiles + 255) // 256)
    for k in [k for k in row if "_TF_" in k]:
        row[k.replace("_TF_", "_cyc24_")] = round(2.0 * M * N * K / row[k] / 1e12 / per_cu * 2.4e9)     # 2.4 GHz cycles per 64-k step of a tile
    print(json.dumps(row), flush=True)
