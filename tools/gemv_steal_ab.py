"""The decode step with and without the GEMV launches' dynamic tail (gemv.hpp steal_ctr), interleaved in one process; tokens must agree.
usage: python tools/gemv_steal_ab.py [steps] [rounds] [pct,pct,...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench  # noqa: E402
import omx_import  # noqa: E402
omx = omx_import.load_package()
from ominix_mlx_amd import engine  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 128
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
pcts = [int(p) for p in sys.argv[3].split(",")] if len(sys.argv) > 3 else [10]
ids = bench.prompt_ids(2048, bench.QWEN3_8B["vocab_size"])
variants = [("off", "0", 0)] + [(f"steal {p} %", "1", p) for p in pcts]
models = {}
for name, on, pct in variants:
    os.environ["OMX_GEMV_STEAL"] = on
    os.environ["OMX_GEMV_STEAL_PCT"] = str(pct or 10)
    m = engine.Model(max_context=2048 + (rounds + 1) * steps + 64, **bench.QWEN3_8B)
    m.synth_weights()
    m.prefill(ids)
    toks = m.decode(16)          # builds the graph under this environment
    models[name] = (m, toks)
ref = models["off"][1]
for name, (m, toks) in models.items():
    assert np.array_equal(toks, ref), f"{name}: tokens differ from the static launch"
best = {name: 1e9 for name in models}
seqs = {name: [] for name in models}
for r in range(rounds):
    for name, (m, _) in models.items():
        t0 = time.perf_counter()
        seqs[name].append(m.decode(steps))
        best[name] = min(best[name], (time.perf_counter() - t0) / steps)
for name in models:
    assert all(np.array_equal(a, b) for a, b in zip(seqs[name], seqs["off"])), f"{name}: tokens differ from the static launch"
    print(f"{name:12s} {best[name] * 1e3:.4f} ms / step  {1 / best[name]:.1f} tok/s  ({models[name][0].decode_path()})", flush=True)
