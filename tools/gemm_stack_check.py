import sys, numpy as np
sys.path.insert(0, "/root/repo")
import omx_import
omx = omx_import.load_package()
from ominix_mlx_amd import mlx_c as mx
from oracle import ref_core as rc
g = np.random.default_rng(0)
M, K = int(sys.argv[1]), int(sys.argv[2])
Ns = [int(v) for v in sys.argv[3:]]
x = rc.bf16_round(g.standard_normal((1, M, K)).astype(np.float32))
ws = [rc.bf16_round((g.standard_normal((n, K)) * 0.05).astype(np.float32)) for n in Ns]
X = mx.Array.from_numpy(x)
W = [mx.Array.from_numpy(w) for w in ws]
outs = {}
for fuse in (False, True):
    mx.lazy_mode(True, fuse)
    ys = [mx.matmul(X, mx.transpose(w)) for w in W]
    mx.eval(*ys)
    outs[fuse] = [y.numpy() for y in ys]
for i, n in enumerate(Ns):
    d = np.abs(outs[True][i] - outs[False][i])
    bad = np.argwhere(d > 0)
    print("N", n, "max diff", float(d.max()), "n_bad", len(bad), "first bad", bad[:3].tolist(), "stats", mx.lazy_stats()["fused_launches"])
