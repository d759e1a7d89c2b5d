import sys, numpy as np
sys.path.insert(0, "/root/repo")
import omx_import
omx = omx_import.load_package()
from ominix_mlx_amd import mlx_c as mx
from oracle import ref_core as rc
g = np.random.default_rng(0)
for dt, name in ((mx.BFLOAT16, "bf16"), (mx.FLOAT32, "f32")):
    for shape in ((1, 5, 8), (3, 7), (2, 64), (4, 1000)):
        a = g.standard_normal(shape).astype(np.float32); b = g.standard_normal(shape).astype(np.float32)
        if name == "bf16": a, b = rc.bf16_round(a), rc.bf16_round(b)
        A, B = mx.Array.from_numpy(a, dt), mx.Array.from_numpy(b, dt)
        rnd = (lambda v: rc.bf16_round(v)) if name == "bf16" else (lambda v: v.astype(np.float32))
        for opn, got, ref in (("neg", mx.negative(A).numpy(), -a), ("add", mx.add(A, B).numpy(), rnd(a + b)), ("mul", mx.multiply(A, B).numpy(), rnd(a * b)),
                              ("sub", mx.subtract(A, B).numpy(), rnd(a - b)), ("sig", mx.sigmoid(A).numpy(), rnd(1 / (1 + np.exp(-a.astype(np.float64))))),
                              ("exp", mx.exp(A).numpy(), rnd(np.exp(a.astype(np.float64))))):
            err = np.abs(got.astype(np.float64) - ref.astype(np.float64)).max()
            print(name, shape, opn, "max err", float(err), "OK" if err <= 2e-2 * max(1.0, np.abs(ref).max()) else "BAD")
