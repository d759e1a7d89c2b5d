"""The four f32 GEMM shapes of a Paraformer encoder layer (30 s of audio: 501 rows), 20 launches each -- run under rocprofv3 --kernel-trace
and read tools/ktrace_shapes.py for per-shape durations."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import omx_import
omx = omx_import.load_package()
T = omx.ops.Tensor
g = np.random.default_rng(0)
M = int(sys.argv[1]) if len(sys.argv) > 1 else 501
for N, K in [(1536, 512), (512, 512), (2048, 512), (512, 2048)]:
    x = T.from_numpy(g.standard_normal((M, K)).astype(np.float32), "f32")
    w = T.from_numpy(g.standard_normal((N, K)).astype(np.float32), "f32")
    for _ in range(20):
        y = omx.ops.linear(x, w, None)
    omx.ops.synchronize()
