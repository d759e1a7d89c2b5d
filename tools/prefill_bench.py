"""Prefill timing of the dense decoder on one MI355X: Qwen3-8B shapes, synthetic weights, `n` prompt tokens through the batched
prefill (MFMA GEMMs + flash attention); prints the device time of each of `reps` prefills from an empty cache."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import omx_import
omx = omx_import.load_package()
from ominix_mlx_amd import engine
import bench


def run(n=2048, reps=3, model="qwen3-8b"):
    cfg = bench.MODELS[model]
    m = engine.Model(max_context=n + 64, **cfg)
    m.synth_weights()
    prompt = bench.prompt_ids(n, cfg["vocab_size"])
    ms = []
    for _ in range(reps):
        m.reset()
        m.prefill(prompt)
        ms.append(round(m.last_prefill_ms(), 3))
    m.close()
    return {"workload": f"{model} bf16 prefill of {n} tokens", "device_ms": ms, "tokens_per_sec": round(n / min(ms) * 1e3, 1)}


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    print(json.dumps(run(n)), flush=True)
