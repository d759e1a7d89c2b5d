"""Where the four-wave GEMM tile's time goes (run on the GPU box, library built with `python tools/gen_gemm5_asm.py --diag && make VARIANT=g5d
VARIANT_FLAGS=-DOMX_G5_DIAG=1`, loaded with OMX_LIB_VARIANT=g5d): the K loop with its DMA, its fragment reads, its barriers / landing waits, or all
of them removed (results are garbage: timing only), interleaved three times."""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["OMX_GEMM_W4"] = "1"
import omx_import
omx = omx_import.load_package()
lib = omx.lib
lib.omx_bench_gemm.restype = ctypes.c_int
lib.omx_bench_gemm.argtypes = [ctypes.c_int] * 5 + [ctypes.POINTER(ctypes.c_float)]
names = {0: "full", 1: "no DMA", 2: "no fragment reads", 3: "no barriers / landing waits", 4: "MFMA only", 5: "fragment reads only", 6: "DMA only", 7: "barriers / landing waits only"}
for M, N, K in ((8192, 8192, 8192), (2048, 2048, 16384), (1024, 1024, 16384)):
    for rnd in range(2):
        for var in range(8):
            os.environ["OMX_GEMM_W4_VAR"] = str(var)
            ms = ctypes.c_float()
            omx.check(lib.omx_bench_gemm(M, N, K, 2, 10, ctypes.byref(ms)))
            tiles = ((M + 255) // 256) * ((N + 255) // 256)
            steps = K // 64 * max(1, (tiles + 255) // 256)       # K steps per CU
            print(f"{M}x{N}x{K}  {names[var]:28s} {2.0 * M * N * K / ms.value / 1e9:8.1f} TF   {ms.value * 1e6 / steps * 2.4 / 1e3 * 1e3:8.0f} cycles @2.4GHz per K step", flush=True)
