"""Phase timeline of the persistent decode kernel (csrc/decode_mega.hip) on Qwen3-8B shapes.
Prints, per phase, the median over layers of: span (first block in -> last block out), the median block's
busy time, and the wait between this phase's last arrival and the next phase's first start."""
import ctypes, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import omx_import
omx = omx_import.load_package()
from ominix_mlx_amd import engine
lib = omx.lib
lib.omx_qwen3_debug_trace_step.restype = ctypes.c_int
lib.omx_qwen3_debug_trace_step.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_int)]
ctx = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
L = 36
m = engine.Model(hidden_size=4096, num_hidden_layers=L, intermediate_size=12288, num_attention_heads=32,
                 num_key_value_heads=8, head_dim=128, vocab_size=151936, max_context=ctx + 256)
m.synth_weights()
prompt = (np.arange(ctx, dtype=np.uint32) * 7919) % 151936
m.prefill(prompt)
m.decode(8)
EV = 16
buf = np.zeros(L * EV * 1024, np.uint64)
nb = ctypes.c_int()
omx.check(lib.omx_qwen3_debug_trace_step(m._h, buf.ctypes.data, buf.size, ctypes.byref(nb)))
nb = nb.value
t = buf[:L * EV * nb].reshape(L, EV, nb).astype(np.int64)
t0 = t[0, 0].min()
us = (t - t0) / 100.0
names = ["qkv", "attn", "o", "gate_up", "down"]
print(f"blocks {nb}; step span {us.max():.1f} us (layers only)")
for p, nm in enumerate(names):
    s, e = us[1:, 2 * p], us[1:, 2 * p + 1]           # skip layer 0 (cold)
    busy = e - s
    act = busy > 0.02 if nm in ("attn", "o") else np.ones_like(busy, bool)
    span = e.max(axis=1) - s.min(axis=1)
    first_in = s.min(axis=1)
    last_out = e.max(axis=1)
    nxt = us[1:, 2 * p + 2].min(axis=1) if p < 4 else np.r_[us[2:, 0].min(axis=1), np.nan]
    gap = nxt - last_out
    bm = np.array([np.median(busy[i][act[i]]) if act[i].any() else 0 for i in range(L - 1)])
    bx = np.array([busy[i][act[i]].max() if act[i].any() else 0 for i in range(L - 1)])
    print(f"{nm:8s} span {np.median(span):6.2f} us   busy median {np.median(bm):6.2f} max {np.median(bx):6.2f}   "
          f"barrier gap after {np.nanmedian(gap):5.2f} us   active blocks {int(np.median(act.sum(axis=1)))}")
per_layer = np.diff(us[:, 0].min(axis=1))
print(f"per layer: median {np.median(per_layer):.2f} us  ->  x{L} = {np.median(per_layer) * L / 1000:.3f} ms")
# attention sub-steps of the blocks that ran attention (events 10..14), relative to the phase start (event 2)
ab = t[1:, 10] > 0
for ev, nm in ((10, "q normed+roped"), (11, "K/V loop done"), (12, "partials stored+acked"), (13, "arrival counter back"),
               (14, "combine done (last block)"), (3, "phase end")):
    sel = ab & (t[1:, ev] > 0)
    d = (t[1:, ev] - t[1:, 2])[sel] / 100.0
    if d.size:
        print(f"  attn +{nm:28s} median {np.median(d):6.2f} us   p90 {np.percentile(d, 90):6.2f}   max {d.max():6.2f}   n/layer {sel.sum() // (L - 1)}")
if os.environ.get("TRACE_DUMP"):
    np.save(os.environ["TRACE_DUMP"], us)
m.close()
