"""One f32 linear of a given shape, a few launches: python tools/f32_gemm_one.py M N K [launches]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import omx_import
omx = omx_import.load_package()
T = omx.ops.Tensor
M, N, K = (int(a) for a in sys.argv[1:4])
n = int(sys.argv[4]) if len(sys.argv) > 4 else 3
g = np.random.default_rng(0)
x = T.from_numpy(g.standard_normal((M, K)).astype(np.float32), "f32")
w = T.from_numpy(g.standard_normal((N, K)).astype(np.float32), "f32")
for _ in range(n):
    y = omx.ops.linear(x, w, None)
omx.ops.synchronize()
