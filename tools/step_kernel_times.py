"""Per-kernel in-step durations (HIP events around every launch of real eager steps) of the decode step on Qwen3-8B shapes, for the
launch-per-op step and the engine modes.  usage: python tools/step_kernel_times.py [layers] [prompt]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import omx_import  # noqa: E402
omx = omx_import.load_package()
from ominix_mlx_amd import engine  # noqa: E402
L = int(sys.argv[1]) if len(sys.argv) > 1 else 8
prompt = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
cfg = dict(bench.QWEN3_8B)
cfg["num_hidden_layers"] = L
ids = bench.prompt_ids(prompt, cfg["vocab_size"])
for mode in ("0", "2", "1"):
    os.environ["OMX_STEP_ENGINE"] = mode
    m = engine.Model(max_context=prompt + 200, **cfg)
    m.synth_weights()
    m.prefill(ids)
    m.decode(16)
    ms = min(m.last_decode_ms() / 16 for _ in range(3) if m.decode(16) is not None)
    us = m.time_step_kernels(4)
    per_layer = sum(v for k, v in us.items() if k not in ("lm_head", "step_engine"))
    print(json.dumps({"mode": mode, "graph_ms_per_step": round(ms, 4), "kernels_us": {k: round(v, 2) for k, v in us.items()},
                      "sum_layer_kernels_us": round(per_layer, 2),
                      "sum_step_us": round(per_layer * L + us["lm_head"] + us["step_engine"], 1)}), flush=True)
    m.close()
