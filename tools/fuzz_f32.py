"""Random-shape soak of the float32 kernels behind Paraformer (run on the GPU box): `python tools/fuzz_f32.py [cases] [seed]`.
omx.ops.linear (f32) against float64 over random M, N, K (aligned and ragged: every kernel and tail form of gemm_f32.hip), and
omx_paraformer_attention_f32 against numpy's explicit form over random Tq, Tk, heads (every width, split and the fallback).  Prints the worst
relative deviation of each family and the first failing case, exit code 1 on a failure."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import omx_import
omx = omx_import.load_package()
from ominix_mlx_amd import paraformer  # noqa: F401  (registers the attention entry point's signature)
T = omx.ops.Tensor
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
g = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
worst_lin, worst_att, bad = 0.0, 0.0, 0
for c in range(cases):
    M = int(g.choice([1, 2, 7, 16, 33, 64, 65, 127, 215, 256, 501, 777]))
    N = int(g.choice([1, 3, 4, 60, 64, 66, 128, 130, 512, 516, 1536, 2048, 8404]))
    K = int(g.choice([4, 8, 45, 60, 64, 68, 128, 200, 512, 560, 1026, 2048, 5632]))
    if M * N * K > 3e9:
        continue
    x = g.standard_normal((M, K)).astype(np.float32)
    w = (g.standard_normal((N, K)) * 0.1).astype(np.float32)
    b = g.standard_normal(N).astype(np.float32) if g.random() < 0.5 else None
    got = omx.ops.linear(T.from_numpy(x, "f32"), T.from_numpy(w, "f32"), T.from_numpy(b, "f32") if b is not None else None).numpy()
    want = x.astype(np.float64) @ w.astype(np.float64).T + (b.astype(np.float64) if b is not None else 0.0)
    mag = np.abs(x).astype(np.float64) @ np.abs(w).astype(np.float64).T + 1.0
    dev = float((np.abs(got - want) / mag).max())
    worst_lin = max(worst_lin, dev)
    if not np.isfinite(got).all() or dev > 1e-6:
        bad += 1
        print(f"linear FAIL M={M} N={N} K={K} bias={b is not None}: {dev:.3e}", flush=True)
for c in range(cases):
    heads = int(g.choice([1, 2, 4, 8]))
    Tq = int(g.choice([1, 5, 16, 17, 100, 215, 256, 501, 640]))
    Tk = int(g.choice([1, 2, 15, 64, 97, 128, 129, 200, 256, 257, 400, 501, 512, 513, 700]))
    D = heads * 128
    q = g.standard_normal((Tq, D)).astype(np.float32)
    kv = g.standard_normal((Tk, 2 * D)).astype(np.float32)
    q_d, kv_d, out = T.from_numpy(q, "f32"), T.from_numpy(kv, "f32"), T((Tq, D), "f32")
    omx.check(omx.lib.omx_paraformer_attention_f32(out.ptr, q_d.ptr, kv_d.ptr, kv_d.ptr + 4 * D, D, 2 * D, D, Tq, Tk, heads, None))
    got = out.numpy()
    ref = np.empty((Tq, D))
    for h in range(heads):
        sl = slice(128 * h, 128 * h + 128)
        sc = (q[:, sl].astype(np.float64) @ kv[:, :D][:, sl].astype(np.float64).T) / np.sqrt(128.0)
        p = np.exp(sc - sc.max(axis=1, keepdims=True))
        ref[:, sl] = (p / p.sum(axis=1, keepdims=True)) @ kv[:, D:][:, sl].astype(np.float64)
    dev = float(np.abs(got - ref).max() / (max(1.0, np.abs(ref).max()) * np.sqrt(Tk)))
    worst_att = max(worst_att, dev)
    if not np.isfinite(got).all() or dev > 2e-6:
        bad += 1
        print(f"attention FAIL Tq={Tq} Tk={Tk} heads={heads}: {dev:.3e}", flush=True)
print(f"{cases} cases each: worst linear deviation {worst_lin:.3e} of sum|a b| (bound 1e-6), worst attention {worst_att:.3e} of max|out| sqrt(Tk) (bound 2e-6), failures {bad}")
sys.exit(1 if bad else 0)
