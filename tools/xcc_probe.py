"""Which XCD does block b of a plain 1-D launch run on?  (XCC_ID hardware register per block.)  The XCD-aware tile orders
in gemm.hip / attn_prefill.hip and any XCD-local synchronisation assume b % 8."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import omx_import
omx = omx_import.load_package()
lib = omx.lib
lib.omx_bench_xcc_ids.restype = ctypes.c_int
lib.omx_bench_xcc_ids.argtypes = [ctypes.c_int, ctypes.c_void_p]
for nb in (64, 256, 512, 2048):
    out = np.zeros(nb, np.uint32)
    omx.check(lib.omx_bench_xcc_ids(nb, out.ctypes.data))
    match = float((out == (np.arange(nb) % 8)).mean())
    print(f"blocks {nb:5d}: first 24 XCC ids {out[:24].tolist()}  fraction with xcc == b % 8: {match:.3f}  histogram {np.bincount(out, minlength=8).tolist()}", flush=True)
