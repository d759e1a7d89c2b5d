"""Differential soak of the deferred-execution layer behind the mlx-c ABI (csrc/mlxc_lazy.hpp), on the GPU box:
`python tools/fuzz_lazy.py [programs] [steps] [seed]`.  Every program is a random sequence over the ops the decode / prompt idioms are
built from (rms_norm, x W^T, add, multiply, sigmoid, slice_update into a cache, slice of it, argmax + item, eval of a subset, dropping
held references) and is run three times on the same inputs: ops launched as called (lazy off), recorded and launched as recorded (fuse
off), recorded and rewritten by the peephole pass (default).  All surviving arrays must agree BIT FOR BIT (the fused kernels keep the
per-op rounding points).  Prints the first diverging program (seed, step list) and exits 1."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import omx_import
omx = omx_import.load_package()
from ominix_mlx_amd import mlx_c as mx
from oracle import ref_core as rc  # bf16 rounding of the inputs only

programs = int(sys.argv[1]) if len(sys.argv) > 1 else 100
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
seed0 = int(sys.argv[3]) if len(sys.argv) > 3 else 0
H, I, C = 256, 512, 96


def make_program(g):
    """-> (inputs: dict name -> np array, ops: list of tuples).  Names: h* / i* activations [1, M, H|I], c the cache [1, C, H]."""
    M = int(g.choice([1, 1, 1, 16, 64]))
    inputs = {"h0": rc.bf16_round(g.standard_normal((1, M, H)).astype(np.float32)),
              "c": rc.bf16_round(g.standard_normal((1, C, H)).astype(np.float32) * 0.1),
              "nw": rc.bf16_round((1 + 0.1 * g.standard_normal(H)).astype(np.float32))}
    for k in range(3):
        inputs[f"whh{k}"] = rc.bf16_round((g.standard_normal((H, H)) * 0.05).astype(np.float32))
        inputs[f"whi{k}"] = rc.bf16_round((g.standard_normal((I, H)) * 0.05).astype(np.float32))
        inputs[f"wih{k}"] = rc.bf16_round((g.standard_normal((H, I)) * 0.05).astype(np.float32))
    hs, is_, ops, n = ["h0"], [], [], 0
    for _ in range(steps):
        kind = g.choice(["norm", "hh", "hi", "ih", "add", "mul", "sig", "swiglu", "cache_w", "cache_r", "argmax", "eval", "drop"],
                        p=[.12, .1, .12, .1, .1, .08, .08, .08, .06, .04, .04, .04, .04])
        n += 1
        if kind == "norm":
            ops.append(("norm", f"h{n}", g.choice(hs))); hs.append(f"h{n}")
        elif kind == "hh":
            ops.append(("mm", f"h{n}", g.choice(hs), f"whh{g.integers(3)}")); hs.append(f"h{n}")
        elif kind == "hi":
            ops.append(("mm", f"i{n}", g.choice(hs), f"whi{g.integers(3)}")); is_.append(f"i{n}")
        elif kind == "ih" and is_:
            ops.append(("mm", f"h{n}", g.choice(is_), f"wih{g.integers(3)}")); hs.append(f"h{n}")
        elif kind in ("add", "mul"):
            pool = hs if (not is_ or g.random() < 0.6) else is_
            out = ("h" if pool is hs else "i") + str(n)
            ops.append((kind, out, g.choice(pool), g.choice(pool))); pool.append(out)
        elif kind == "sig":
            pool = hs if (not is_ or g.random() < 0.5) else is_
            out = ("h" if pool is hs else "i") + str(n)
            ops.append(("sig", out, g.choice(pool))); pool.append(out)
        elif kind == "swiglu" and len(hs) >= 1:   # (g * sigmoid(g)) * u with g, u products of one activation: the idiom itself
            x = g.choice(hs)
            ops.append(("mm", f"i{n}a", x, f"whi{g.integers(3)}")); ops.append(("sig", f"i{n}b", f"i{n}a"))
            ops.append(("mul", f"i{n}c", f"i{n}a", f"i{n}b")); ops.append(("mm", f"i{n}d", x, f"whi{g.integers(3)}"))
            ops.append(("mul", f"i{n}", f"i{n}c", f"i{n}d")); is_.append(f"i{n}")
            if g.random() < 0.5:
                ops.append(("drop", f"i{n}a")); ops.append(("drop", f"i{n}b")); ops.append(("drop", f"i{n}c")); ops.append(("drop", f"i{n}d"))
        elif kind == "cache_w":
            ops.append(("cache_w", g.choice(hs), int(g.integers(0, C - M + 1))))
        elif kind == "cache_r":
            lo = int(g.integers(0, C - M + 1))
            ops.append(("cache_r", f"h{n}", lo)); hs.append(f"h{n}")
        elif kind == "argmax":
            ops.append(("argmax", g.choice(hs)))
        elif kind == "eval":
            ops.append(("eval", [str(x) for x in g.choice(hs, size=min(2, len(hs)), replace=False)]))
        elif kind == "drop" and len(hs) > 2:
            victim = hs.pop(int(g.integers(1, len(hs))))
            ops.append(("drop", victim))
    return M, inputs, ops


def run(M, inputs, ops):
    live = {k: mx.Array.from_numpy(v) for k, v in inputs.items()}
    items = []
    for op in ops:
        k = op[0]
        if k == "norm":
            if op[2] in live: live[op[1]] = mx.rms_norm(live[op[2]], live["nw"], 1e-6)
        elif k == "mm":
            if op[2] in live: live[op[1]] = mx.matmul(live[op[2]], mx.transpose(live[op[3]]))
        elif k in ("add", "mul"):
            if op[2] in live and op[3] in live: live[op[1]] = (mx.add if k == "add" else mx.multiply)(live[op[2]], live[op[3]])
        elif k == "sig":
            if op[2] in live: live[op[1]] = mx.sigmoid(live[op[2]])
        elif k == "cache_w":
            if op[1] in live: live["c"] = mx.slice_update(live["c"], live[op[1]], [0, op[2], 0], [1, op[2] + M, H])
        elif k == "cache_r":
            live[op[1]] = mx.slice(live["c"], [0, op[2], 0], [1, op[2] + M, H])
        elif k == "argmax":
            if op[1] in live:
                a = mx.argmax_axis(mx.reshape(live[op[1]], [M * H]), 0)
                items.append(int(a.item()))
        elif k == "eval":
            mx.eval(*[live[x] for x in op[1] if x in live])
        elif k == "drop":
            live.pop(op[1], None)
    names = sorted(n for n in live if n[0] in "hic" and n not in inputs or n == "c")
    return {n: live[n].numpy() for n in names}, items


bad = 0
stats0 = mx.lazy_stats()
for p in range(programs):
    g = np.random.default_rng(seed0 * 100003 + p)
    M, inputs, ops = make_program(g)
    outs = []
    for lazy, fuse in ((False, False), (True, False), (True, True)):
        mx.lazy_mode(lazy, fuse)
        outs.append(run(M, inputs, ops))
    mx.lazy_mode(True, True)
    for mode, (vals, items) in zip(("recorded", "fused"), outs[1:]):
        diff = [n for n in outs[0][0] if not np.array_equal(outs[0][0][n], vals[n], equal_nan=True)]
        if diff or items != outs[0][1]:
            bad += 1
            print(f"program {p} (seed {seed0}, M={M}) diverges in mode {mode}: arrays {diff[:6]}, items equal {items == outs[0][1]}")
            for o in ops:
                print("   ", o)
            break
    if bad:
        break
stats1 = mx.lazy_stats()
print(f"{programs} programs x {steps} steps, three modes each: {'OK, all arrays bit-identical' if not bad else 'DIVERGED'}; "
      + ", ".join(f"{k} +{stats1[k] - stats0[k]}" for k in stats1 if isinstance(stats1[k], int)))
sys.exit(1 if bad else 0)
