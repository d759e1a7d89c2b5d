"""Profiler workload (run on the GPU box under rocprofv3): the bench's decode path and nothing else -- Qwen3-8B shapes, 2048-token
batched prefill, then N greedy decode steps through the engine's step graph.  `python3 tools/decode_steps.py [steps] [prompt] [bits]`
(bits 4 / 8: the same shapes as an MLX-quantized checkpoint)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import omx_import  # noqa: E402

omx = omx_import.load_package()
from ominix_mlx_amd import engine  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 16
prompt = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
cfg = dict(bench.QWEN3_8B)
bits = int(sys.argv[3]) if len(sys.argv) > 3 else 0
if bits:
    cfg["quantization"] = {"bits": bits, "group_size": 64}
m = engine.Model(max_context=prompt + steps + 8, **cfg)
m.synth_weights()
first = m.prefill(bench.prompt_ids(prompt, cfg["vocab_size"]))
toks = m.decode(steps)
print("first tokens", int(first), [int(t) for t in toks[:4]], "path", m.decode_path())
m.close()
