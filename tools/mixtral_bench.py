"""BASELINE config 3 on ONE MI355X: Mixtral-8x7B shapes (mixtral-mlx/src/model.rs:44-52) in bf16 -- 93 GB of weights fit the
288 GB of one GPU -- 2048-token prompt, greedy decode.  Reports tok/s and the algorithmic HBM fraction (router + top-2
experts + attention weights + lm_head per token).  Also Qwen3-30B-A3B shapes (128 experts, top-8)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import omx_import
omx = omx_import.load_package()
from ominix_mlx_amd import engine

MODELS = {
    "mixtral-8x7b": dict(hidden_size=4096, num_hidden_layers=32, intermediate_size=14336, num_attention_heads=32, num_key_value_heads=8,
                         head_dim=128, vocab_size=32000, rms_norm_eps=1e-5, rope_theta=1e6, num_experts=8, num_experts_per_tok=2,
                         moe_intermediate_size=14336, moe_mode="mixtral", qk_norm=False),
    "qwen3-30b-a3b": dict(hidden_size=2048, num_hidden_layers=48, intermediate_size=6144, num_attention_heads=32, num_key_value_heads=4,
                          head_dim=128, vocab_size=151936, rms_norm_eps=1e-6, rope_theta=1e6, num_experts=128, num_experts_per_tok=8,
                          moe_intermediate_size=768, moe_mode="qwen3_moe", norm_topk_prob=True),
}
bits = 0
args = [a for a in sys.argv[1:]]
if '--bits' in args:
    i = args.index('--bits'); bits = int(args[i + 1]); del args[i:i + 2]
which = args or list(MODELS)
n_prompt, warm, steps = 2048, 8, 64
for name in which:
    cfg = MODELS[name]
    if bits and cfg['moe_intermediate_size'] % 512:
        continue
    m = engine.Model(max_context=n_prompt + warm + steps + 8, quantization={'bits': bits, 'group_size': 64} if bits else None, **cfg)
    t0 = time.perf_counter(); m.synth_weights(); omx.ops.synchronize(); synth_s = time.perf_counter() - t0
    prompt = ((np.arange(n_prompt, dtype=np.int64) * 7919 + 13) % cfg["vocab_size"]).astype(np.uint32)
    first = m.prefill(prompt)
    m.decode(warm)
    t0 = time.perf_counter(); toks = m.decode(steps); dt = time.perf_counter() - t0
    step_bytes = m.step_bytes(n_prompt + warm + steps // 2)
    print(json.dumps({"model": name, "bits": bits or 16, "decode_tokens_per_sec": round(steps / dt, 1), "ms_per_token": round(dt / steps * 1e3, 3),
                      "device_ms_per_token": round(m.last_decode_ms() / steps, 3), "prefill_ms": round(m.last_prefill_ms(), 1),
                      "algorithmic_GB_per_token": round(step_bytes / 1e9, 2), "hbm_GBps": round(step_bytes / (dt / steps) / 1e9, 1),
                      "frac_of_8TBps": round(step_bytes / (dt / steps) / 8e12, 3), "weights_synth_s": round(synth_s, 2),
                      "path": m.decode_path(), "first_tokens": [int(first)] + [int(t) for t in toks[:3]]}), flush=True)
    m.close()
