"""Profiler workload for the MFMA-busy PMC pass: one 2048-token batched prefill at Qwen3-8B shapes (gemm_bf16_nt_* + attn_prefill_kernel)
and two FLUX.2-klein 1024x1024 DiT steps.  `python3 tools/prefill_flux_steps.py`."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import omx_import  # noqa: E402

omx = omx_import.load_package()
from ominix_mlx_amd import engine  # noqa: E402

cfg = dict(bench.QWEN3_8B)
m = engine.Model(max_context=2048 + 16, **cfg)
m.synth_weights()
for _ in range(2):
    m.reset()
    m.prefill(bench.prompt_ids(2048, cfg["vocab_size"]))
print("prefill device ms", m.last_prefill_ms())
m.close()
print(bench.flux_secondary(omx, steps=2))
