"""A/B of the decode step on Qwen3-8B shapes: hipGraph replay vs the same launches as AQL packets on the engine's own HSA queue
(csrc/aql_step.hip; OMX_STEP_AQL = 1 agent-scope fences on every packet, 2 no fences, 3 acquire only, 4 release only).
Engines live in one process and alternate, so box-to-box variance cancels.  Host wall clock around decode() for every variant.
usage: python tools/aql_step_ab.py [steps] [prompt] [layers] [modes, e.g. 0,1,2]"""
import json, os, sys, time
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
modes = [int(x) for x in (sys.argv[4] if len(sys.argv) > 4 else "0,1,2,3,4").split(",")]
quant = int(os.environ.get("OMX_AB_QUANT", "0"))
masks = os.environ.get("OMX_AB_MASKS", "0").split(",")
# further environment per variant, e.g. OMX_AB_EXTRA='[{}, {"OMX_GEMV_RPW_QKV": "2"}]'
extras = json.loads(os.environ.get("OMX_AB_EXTRA", "[{}]"))
ids = bench.prompt_ids(prompt, cfg["vocab_size"])
os.environ["OMX_STEP_AQL_VERBOSE"] = "1"
ref = None
for rnd in range(int(os.environ.get("OMX_AB_ROUNDS", "2"))):
    for mode, mask, extra in [(md, mk, ex) for ex in extras for md in modes for mk in (masks if md else ["0"])]:
        for k in list(os.environ):
            if k.startswith("OMX_GEMV_RPW_") or k.startswith("OMX_PF") or k.startswith("OMX_GEMV_PERM"):
                del os.environ[k]
        os.environ.update(extra)
        os.environ["OMX_STEP_AQL"] = str(mode)
        os.environ["OMX_AQL_NOBARRIER"] = mask   # measurement only: packets of these launch classes carry no barrier bit (results void)
        kw = dict(cfg)
        if quant:
            kw.update(quant_bits=quant, quant_group=64)
        m = engine.Model(max_context=prompt + 3 * steps + 16, **kw)
        m.synth_weights()
        m.prefill(ids)
        m.decode(8)
        best = 1e9
        toks = []
        for _ in range(3):
            t0 = time.perf_counter()
            toks += [int(t) for t in m.decode(steps)]
            best = min(best, (time.perf_counter() - t0) * 1e3 / steps)
        logits = m.last_logits()
        same = None
        if ref is None:
            ref = (toks, logits)
        else:
            same = bool(toks == ref[0] and np.array_equal(logits, ref[1]))
        print(json.dumps({"aql_mode": mode, "nobarrier_mask": mask, "env": extra, "path": m.decode_path(), "ms_per_step": round(best, 4), "tok_s": round(1e3 / best, 1),
                          "bit_identical_to_first": same, "tokens": toks[:4]}), flush=True)
        m.close()
