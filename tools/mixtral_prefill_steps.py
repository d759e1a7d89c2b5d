"""Three 2 048-token prompts through the single-GPU Mixtral-8x7B engine (for rocprofv3 --kernel-trace --stats)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import omx_import
omx = omx_import.load_package()
from ominix_mlx_amd import engine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
m = engine.Model(max_context=n + 64, **bench.MIXTRAL_8X7B)
m.synth_weights()
ids = bench.prompt_ids(n, bench.MIXTRAL_8X7B["vocab_size"])
for i in range(3):
    m.reset()
    m.prefill(ids)
    print(f"prompt of {n} tokens: {m.last_prefill_ms():.2f} ms", flush=True)
m.close()
