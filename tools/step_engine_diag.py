"""Phase-by-phase comparison of the persistent decode step against the launch-per-op step on a ONE-layer model: after each decode step
the engine's edge buffers (granules) are compared with the launch path's activation buffers.
usage: python tools/step_engine_diag.py [config] [prompt] [steps]   config: small | big"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import omx_import  # noqa: E402
omx = omx_import.load_package()
from ominix_mlx_amd import engine  # noqa: E402
from oracle import synth  # noqa: E402
lib = omx.lib

which = sys.argv[1] if len(sys.argv) > 1 else "small"
n_prompt = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 12
if which == "d64":
    cfg = dict(hidden_size=512, num_hidden_layers=1, intermediate_size=1536, num_attention_heads=8, num_key_value_heads=4, head_dim=64,
               vocab_size=2048, rms_norm_eps=1e-6, rope_theta=1e6, tie_word_embeddings=False)
elif which == "small":
    cfg = dict(hidden_size=1024, num_hidden_layers=1, intermediate_size=3072, num_attention_heads=8, num_key_value_heads=2, head_dim=128,
               vocab_size=4096, rms_norm_eps=1e-6, rope_theta=1e6, tie_word_embeddings=True)
else:
    cfg = dict(hidden_size=4096, num_hidden_layers=1, intermediate_size=12288, num_attention_heads=32, num_key_value_heads=8, head_dim=128,
               vocab_size=32768, rms_norm_eps=1e-6, rope_theta=1e6, tie_word_embeddings=False)
H, Hkv, D, hd, I = cfg["num_attention_heads"], cfg["num_key_value_heads"], cfg["head_dim"], cfg["hidden_size"], cfg["intermediate_size"]


def read(m, name, n, dtype=np.uint16):
    buf = np.zeros(n, dtype)
    omx.check(lib.omx_qwen3_debug_read(m._h, name.encode(), buf.ctypes.data, buf.nbytes // 2))
    return buf


def gran(m, name, n):
    g = read(m, name, n, np.uint64)
    return (g & np.uint64(0xFFFFFFFF)).astype(np.uint32).view(np.uint16), (g >> np.uint64(32)).astype(np.uint32)


models = {}
prompt = synth.prompt_ids(n_prompt, cfg["vocab_size"])
for mode in ("0", "1"):
    os.environ["OMX_STEP_ENGINE"] = mode
    os.environ["OMX_ATTN_OPROJ"] = "0"     # the launch path keeps attn_out in a buffer
    m = engine.Model(max_context=n_prompt + steps + 300, **cfg)
    m.synth_weights()
    first = m.prefill(prompt)
    models[mode] = m
bad = 0
for s in range(steps):
    os.environ["OMX_STEP_ENGINE"] = "0"
    t0 = models["0"].decode(1)
    os.environ["OMX_STEP_ENGINE"] = "1"
    t1 = models["1"].decode(1)
    ref = {"qkv": read(models["0"], "qkv", (H + 2 * Hkv) * D), "attn": read(models["0"], "attn_out", H * D), "act": read(models["0"], "act", I)}
    got = {"qkv": gran(models["1"], "g_qkv", (H + 2 * Hkv) * D // 2), "attn": gran(models["1"], "g_attn", H * D // 2),
           "act": gran(models["1"], "g_act", I // 2), "x1": gran(models["1"], "g_x1", hd // 2)}
    hr = read(models["0"], "h", hd)      # launch path: final residual of a 1-layer model with separate O lands in h (two swaps)
    h2r = read(models["0"], "h2", hd)
    he = read(models["1"], "h2", hd)
    line = [f"step {s} pos {n_prompt + s} tok {int(t0[0])}/{int(t1[0])}"]
    for k in ("qkv", "attn", "act"):
        v, tags = got[k]
        nbad = int((v != ref[k]).sum())
        line.append(f"{k}: {nbad} bad of {v.size} (tags {np.unique(tags)[:3]})")
        bad += nbad
        if nbad:
            idx = np.nonzero(v != ref[k])[0]
            line.append(f"   first bad idx {idx[:8]} ...")
    line.append(f"x1 vs h2(ref): {int((got['x1'][0] != h2r).sum())} / vs h(ref): {int((got['x1'][0] != hr).sum())}")
    line.append(f"final h: vs h {int((he != hr).sum())} vs h2 {int((he != h2r).sum())}")
    print("  ".join(line), flush=True)
    if s == 0:
        f = lambda a: (a.astype(np.uint32) << 16).view(np.float32)
        print("   attn ref", f(ref["attn"][:8]), "\n   attn got", f(got["attn"][0][:8]))
        bad_heads = [(int(h), int((got["attn"][0][h * D:(h + 1) * D] != ref["attn"][h * D:(h + 1) * D]).sum())) for h in range(H)]
        print("   bad per head", bad_heads)
        cap = (n_prompt + steps + 300 + 255) // 256 * 256
        for nm in ("k0", "v0"):
            a0 = read(models["0"], nm, Hkv * cap * D).reshape(Hkv, cap, D)
            a1 = read(models["1"], nm, Hkv * cap * D).reshape(Hkv, cap, D)
            rows = np.nonzero((a0 != a1).any(axis=2))
            print("   cache", nm, "rows differing:", list(zip(rows[0][:6].tolist(), rows[1][:6].tolist())), "count", rows[0].size)
print("total bad", bad)
