"""A/B of the decode step on Qwen3-8B shapes: one launch per op (hipGraph) vs the persistent step engine (csrc/step_engine.hip).
Both engines live in one process and alternate, so box-to-box variance cancels.
usage: python tools/step_engine_ab.py [steps] [prompt] [layers]"""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import omx_import  # noqa: E402
omx = omx_import.load_package()
from ominix_mlx_amd import engine  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 64
prompt = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
cfg = dict(bench.QWEN3_8B)
if len(sys.argv) > 3:
    cfg["num_hidden_layers"] = int(sys.argv[3])
ids = bench.prompt_ids(prompt, cfg["vocab_size"])
variants = [("launches", {"OMX_STEP_ENGINE": "0"})]
for spec in os.environ.get("OMX_AB_VARIANTS", "1:2:1,3:2:1,3:3:1,1:2:0,3:2:0,3:3:0").split(","):
    if not spec:
        continue
    ns, inf, thin = spec.split(":")
    variants.append((f"engine nsweep={ns} inflight={inf} thin={thin}",
                     {"OMX_STEP_ENGINE": "1", "OMX_SE_NSWEEP": ns, "OMX_SE_INFLIGHT": inf, "OMX_SE_THIN": thin}))
for spec in os.environ.get("OMX_AB_HYBRID", "3:2:1").split(","):
    if spec:
        ns, inf, thin = spec.split(":")
        variants.append((f"hybrid nsweep={ns} inflight={inf} thin={thin}",
                         {"OMX_STEP_ENGINE": "2", "OMX_SE_NSWEEP": ns, "OMX_SE_INFLIGHT": inf, "OMX_SE_THIN": thin}))
ref = None
for name, env in variants:
    os.environ.update(env)
    m = engine.Model(max_context=prompt + 3 * steps + 16, **cfg)
    m.synth_weights()
    first = m.prefill(ids)
    m.decode(8)
    best = 1e9
    toks = []
    for _ in range(3):
        toks += [int(t) for t in m.decode(steps)]
        best = min(best, m.last_decode_ms() / steps)
    logits = m.last_logits()
    same = None
    if ref is None:
        ref = (toks, logits)
    else:
        same = bool(toks == ref[0] and np.array_equal(logits, ref[1]))
    print(json.dumps({"variant": name, "ms_per_step": round(best, 4), "tok_s": round(1e3 / best, 1), "bit_identical_to_launches": same,
                      "tokens": toks[:4]}), flush=True)
    m.close()
