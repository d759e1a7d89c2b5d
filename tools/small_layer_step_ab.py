"""The persistent decode step (csrc/step_engine.hip, `make EXPERIMENTS=1`) against one launch per op on SMALL layers -- the sizes where
MI355X_MICROARCH.md ("Persistent kernels": engine-vs-launches 0.87-0.89x on a 121.6 MB layer) measures it winning and where it had never
been timed here: single-rank models with the per-rank SHAPES of Qwen3-8B's tensor-parallel shards (heads and
intermediate columns divided by TP: 193 / 97 / 48 MB of weights per layer; the shard's reductions are not part of this comparison, both
forms would pay them alike) and Qwen3-0.6B (BASELINE config 1: 9.4 MB per layer).  Both step forms live in one process and alternate.
usage: OMX_LIB_VARIANT=exp python tools/small_layer_step_ab.py [steps] [prompt] > profiles/r06_small_layer_step.md"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import omx_import  # noqa: E402
omx = omx_import.load_package()
from ominix_mlx_amd import engine  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 64
prompt = int(sys.argv[2]) if len(sys.argv) > 2 else 2048


def shard_shape(tp):
    c = dict(bench.QWEN3_8B)
    c["num_attention_heads"] //= tp
    c["num_key_value_heads"] //= tp
    c["intermediate_size"] //= tp
    return c


SHAPES = [("Qwen3-8B (TP 1)", dict(bench.QWEN3_8B))] + [(f"Qwen3-8B TP {tp} shard shape", shard_shape(tp)) for tp in (2, 4, 8)] + \
         [("Qwen3-0.6B", dict(bench.QWEN3_0_6B))]
FORMS = [("launches", {"OMX_STEP_ENGINE": "0"}),
         ("engine (whole step)", {"OMX_STEP_ENGINE": "1", "OMX_SE_NSWEEP": "3", "OMX_SE_INFLIGHT": "2", "OMX_SE_THIN": "1"}),
         ("hybrid (attention its own launch)", {"OMX_STEP_ENGINE": "2", "OMX_SE_NSWEEP": "3", "OMX_SE_INFLIGHT": "2", "OMX_SE_THIN": "1"})]

print("| shape | layer weights (MB) | form | decode path | ms / step | tok/s | vs launches | bit-identical |")
print("|---|---|---|---|---|---|---|---|")
for name, cfg in SHAPES:
    h, H, Hkv, D, I = cfg["hidden_size"], cfg["num_attention_heads"], cfg["num_key_value_heads"], cfg["head_dim"], cfg["intermediate_size"]
    layer_mb = 2 * (h * (H + 2 * Hkv) * D + H * D * h + 3 * h * I) / 1e6
    ids = bench.prompt_ids(prompt, cfg["vocab_size"])
    ref, base = None, None
    for form, env in FORMS:
        os.environ.update(env)
        try:
            m = engine.Model(max_context=prompt + 4 * steps + 16, **cfg)
            m.synth_weights()
            m.prefill(ids)
            m.decode(8)
            best, toks = 1e9, []
            for _ in range(3):
                toks += [int(t) for t in m.decode(steps)]
                best = min(best, m.last_decode_ms() / steps)
            path = m.decode_path()
            logits = m.last_logits()
            m.close()
        except Exception as e:  # noqa: BLE001  (a shape the engine refuses is a row of the table, not the end of the run)
            print(f"| {name} | {layer_mb:.1f} | {form} | refused: {str(e)[:80]} | | | | |", flush=True)
            continue
        same = ""
        if ref is None:
            ref, base = (toks, logits), best
        else:
            same = str(bool(toks == ref[0] and np.array_equal(logits, ref[1])))
        print(f"| {name} | {layer_mb:.1f} | {form} | {path} | {best:.4f} | {1e3 / best:.1f} | {best / base:.3f}x | {same} |", flush=True)
