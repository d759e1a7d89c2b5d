"""Decode loop of the Qwen3-8B-shaped model as an MLX 4-bit checkpoint (for rocprofv3 / quick timing)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import omx_import
omx = omx_import.load_package()
from ominix_mlx_amd import engine
bits = int(sys.argv[1]) if len(sys.argv) > 1 else 4
ctx = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
m = engine.Model(hidden_size=4096, num_hidden_layers=36, intermediate_size=12288, num_attention_heads=32, num_key_value_heads=8,
                 head_dim=128, vocab_size=151936, max_context=ctx + 200, quantization={"bits": bits, "group_size": 64} if bits else None)
m.synth_weights()
m.prefill(((np.arange(ctx, dtype=np.uint32) * 7919) + 13) % 151936)
m.decode(8)
t0 = time.perf_counter(); m.decode(64); dt = time.perf_counter() - t0
print(f"bits {bits} ctx {ctx}: {64/dt:.1f} tok/s  {dt/64*1e3:.3f} ms/step  device {m.last_decode_ms()/64:.3f} ms", flush=True)
m.close()
