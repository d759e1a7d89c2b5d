"""Timeline of the decode-step attention kernel (csrc/attn_step.hip) on Qwen3-8B shapes: one eager step with the
kernel's wall-clock stamps on (100 MHz), reported relative to the first block start of each layer's launch.
usage: python tools/attn_step_trace.py [ctx] [layers]"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import omx_import
omx = omx_import.load_package()
from ominix_mlx_amd import engine
lib = omx.lib
ctx = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
L = int(sys.argv[2]) if len(sys.argv) > 2 else 12
m = engine.Model(hidden_size=4096, num_hidden_layers=L, intermediate_size=12288, num_attention_heads=32,
                 num_key_value_heads=8, head_dim=128, vocab_size=151936, max_context=ctx + 256)
m.synth_weights()
m.prefill((np.arange(ctx, dtype=np.uint32) * 7919) % 151936)
m.decode(8)
buf = np.zeros(L * 64 * 8 * 8, np.uint64)
nb = ctypes.c_int()
omx.check(lib.omx_qwen3_debug_trace_step(m._h, buf.ctypes.data, buf.size, ctypes.byref(nb)))
nb = nb.value
t = buf[:L * nb * 8].reshape(L, nb, 8).astype(np.int64)
names = ["block start", "loads landed, q/k normed+roped", "own chunk done", "granules stored", "gather + merge done (consumers)",
         "attention vector swept into LDS (O projection)", "O rows stored", "all waves' partials parked in LDS (block barrier)"]
print(f"{nb} blocks per launch, {L} layers, ctx {ctx}")
for ev, nm in enumerate(names):
    rows = []
    for l in range(1, L):
        t0 = t[l, :, 0][t[l, :, 0] > 0].min()
        v = t[l, :, ev]
        v = v[v > 0]
        if v.size:
            rows.append(((v - t0) / 100.0))
    if rows:
        allv = np.concatenate(rows)
        print(f"  {nm:34s} median {np.median(allv):6.2f} us   p90 {np.percentile(allv, 90):6.2f}   max(median over layers) "
              f"{np.median([r.max() for r in rows]):6.2f}   n/layer {allv.size // (L - 1)}")
m.close()
