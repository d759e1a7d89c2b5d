"""The drop-in route (csrc/per_op_route.hip: qwen3-mlx's forward + Generate::next replayed through the mlx-c ABI) timed next to the engine on
Qwen3-8B shapes, with the deferred list's counters (mlxc_lazy.hpp).  OMX_MLX_LAZY=0 / OMX_MLX_FUSE=0 for the A/B.
usage: python tools/per_op_route_time.py [prompt] [tokens] [bits: 0 | 4 | 8]"""
import ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import omx_import  # noqa: E402
omx = omx_import.load_package()
from ominix_mlx_amd import engine  # noqa: E402

n_prompt = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
n_tok = int(sys.argv[2]) if len(sys.argv) > 2 else 64
cfg = dict(bench.QWEN3_8B)
bits = int(sys.argv[3]) if len(sys.argv) > 3 else 0
if bits:
    cfg["quantization"] = {"bits": bits, "group_size": 64}       # the same shapes as an MLX-quantized checkpoint (qwen3-mlx/README.md:102)
ids = bench.prompt_ids(n_prompt, cfg["vocab_size"])
m = engine.Model(max_context=n_prompt + 2 * n_tok + 64, **cfg)
m.synth_weights()
first = m.prefill(ids)
eng = [int(first)] + [int(t) for t in m.decode(n_tok)]
eng_ms = m.last_decode_ms() / n_tok
m.per_op_route(ids, 1)
stats0 = (ctypes.c_long * 6)()
omx.lib.omx_mlx_lazy_stats(stats0)
r = m.per_op_route(ids, n_tok)
stats1 = (ctypes.c_long * 6)()
omx.lib.omx_mlx_lazy_stats(stats1)
d = [stats1[i] - stats0[i] for i in range(6)]
toks = [int(t) for t in r["tokens"]]
lead = 0
while lead < min(len(toks), len(eng)) and toks[lead] == eng[lead]:
    lead += 1
print(json.dumps({"engine_ms_per_token": round(eng_ms, 4), "route_ms_per_token": round(r["ms_per_token"], 4), "route_tok_s": round(1e3 / r["ms_per_token"], 1),
                  "of_engine": round(eng_ms / r["ms_per_token"], 3), "prefill_ms": round(r["prefill_ms"], 2), "calls_per_token": r["calls_per_token"],
                  "recorded": d[0], "launched_as_recorded": d[1], "fused_launches": d[2], "flushes": d[3], "flush_host_ms_per_token": round(d[4] / 1e6 / (n_tok + 1), 3), "rewrite_ms_per_token": round(d[5] / 1e6 / (n_tok + 1), 3),
                  "leading_tokens_equal_to_engine": f"{lead} of {len(toks)}", "route_tokens": toks[:6], "engine_tokens": eng[:6]}))
m.close()
