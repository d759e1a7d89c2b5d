import sys, numpy as np
sys.path.insert(0, "/root/repo")
import omx_import
omx = omx_import.load_package()
from ominix_mlx_amd import mlx_c as mx
a = np.arange(64, dtype=np.float32).reshape(2, 32) / 8
A = mx.Array.from_numpy(a, mx.FLOAT32)
for name, fn in (("sig", mx.sigmoid), ("neg", mx.negative), ("exp", mx.exp), ("neg2", mx.negative)):
    print(name, fn(A).numpy().ravel()[:6])
for lazy in (False, True):
    mx.lazy_mode(lazy, True)
    print("lazy", lazy, "neg", mx.negative(A).numpy().ravel()[:6])
