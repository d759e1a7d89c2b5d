"""In-step A/B of the rows-per-wave choice of one decode GEMV class (OMX_GEMV_RPW_QKV / _O / _GU / _DOWN, read when the step is built):
Qwen3-8B shapes, graph step time and per-kernel HIP-event times.  usage: python tools/gemv_rpw_ab.py QKV 4 3 6 12 [layers]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import omx_import  # noqa: E402
omx = omx_import.load_package()
from ominix_mlx_amd import engine  # noqa: E402
which = sys.argv[1]
vals = [int(v) for v in sys.argv[2:] if int(v) < 100]
L = 36
cfg = dict(bench.QWEN3_8B)
cfg["num_hidden_layers"] = L
ids = bench.prompt_ids(2048, cfg["vocab_size"])
for v in vals:
    os.environ[f"OMX_GEMV_RPW_{which}"] = str(v)
    m = engine.Model(max_context=2048 + 400, **cfg)
    m.synth_weights()
    first = m.prefill(ids)
    toks = [int(t) for t in m.decode(16)]
    ms = min(m.last_decode_ms() / 64 for _ in range(3) if m.decode(64) is not None)
    us = m.time_step_kernels(4)
    print(json.dumps({"class": which, "rows_per_wave": v, "tok_per_s": round(1e3 / ms, 1), "ms_per_step": round(ms, 4),
                      "kernels_us": {k: round(x, 2) for k, x in us.items() if x}, "tokens": toks[:4]}), flush=True)
    m.close()
