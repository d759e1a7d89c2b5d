"""Cost of the speculative-decoding verify pass (Model.verify: n tokens batched on top of the cache, [n, V] lm_head GEMM, per-row argmax)
against n ordinary decode steps, Qwen3-8B shapes, 2048 tokens of context.  With synthetic weights a draft model cannot agree with the
target (flat logits), so this reports the MECHANISM's ceiling: tokens per second if every draft token were accepted, draft cost excluded.
usage: python tools/speculative_bench.py [n ...]      (default n = 2 3 5 9 17)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import omx_import  # noqa: E402

omx = omx_import.load_package()
from ominix_mlx_amd import engine  # noqa: E402

cfg = dict(bench.QWEN3_8B)
m = engine.Model(max_context=2048 + 512, **cfg)
m.synth_weights()
prompt = bench.prompt_ids(2048, cfg["vocab_size"])
m.prefill(prompt)
t0 = time.perf_counter(); m.decode(32); step_ms = (time.perf_counter() - t0) * 1e3 / 32
print(f"decode step {step_ms:.3f} ms")
for n in ([int(a) for a in sys.argv[1:]] or (2, 3, 5, 9, 17)):
    toks = [int(t) for t in prompt[:n]]
    m.verify(toks); m.trim(n, toks[0])              # warm (buffers, kernels)
    t0 = time.perf_counter()
    reps = 8
    for _ in range(reps):
        m.verify(toks)
        m.trim(n, toks[0])
    ms = (time.perf_counter() - t0) * 1e3 / reps
    print(f"verify {n:2d} tokens: {ms:7.3f} ms  = {ms / step_ms:4.2f} decode steps  -> {n / ms * 1e3:7.1f} tok/s if all {n - 1} drafts are accepted "
          f"(plain decode {1e3 / step_ms:.1f})")
m.close()
