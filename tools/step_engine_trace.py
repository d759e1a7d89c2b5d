"""Timeline of the persistent decode step (csrc/step_engine.hip) on Qwen3-8B shapes: one eager step with per-CU wall-clock stamps
(100 MHz) of consumer wave 0 for the first four layers, reported per phase as time since the layer started on that CU.
usage: python tools/step_engine_trace.py [ctx] [layers]"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("OMX_STEP_ENGINE", "1")
import bench  # noqa: E402
import omx_import  # noqa: E402
omx = omx_import.load_package()
from ominix_mlx_amd import engine  # noqa: E402
lib = omx.lib
ctx = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
L = int(sys.argv[2]) if len(sys.argv) > 2 else 8
cfg = dict(bench.QWEN3_8B)
cfg["num_hidden_layers"] = L
m = engine.Model(max_context=ctx + 256, **cfg)
m.synth_weights()
m.prefill(bench.prompt_ids(ctx, cfg["vocab_size"]))
m.decode(8)
buf = np.zeros(1024 * 64, np.uint64)
n = ctypes.c_int()
omx.check(lib.omx_qwen3_debug_trace_engine(m._h, buf.ctypes.data, buf.size, ctypes.byref(n)))
cus = n.value
t = buf[:cus * 64].reshape(cus, 64).astype(np.int64)
names = {0: "layer start", 1: "x ready (gathered)", 17: "x normalised", 2: "q/k/v rows done", 3: "group q/k/v swept", 4: "normed + roped",
         5: "K/V slots ready", 6: "rounds done", 7: "parked", 8: "own chunk done (all waves)", 9: "partials stored",
         10: "attention phase done", 11: "attention vector ready", 12: "o rows done", 13: "x1 ready", 18: "x1 normalised",
         14: "gate/up rows done", 15: "act ready", 16: "down rows done"}
order = [0, 1, 17, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 18, 14, 15, 16]
for l in (1, 2):
    tt = t[:, (l - 1) * 32:(l - 1) * 32 + 32]
    t0 = np.median(tt[:, 0])
    print(f"layer {l} (times in us since the median layer start)")
    prev = None
    for k in order:
        v = tt[:, k]
        v = (v[v > 0] - t0) / 100.0
        if not v.size:
            continue
        d = "" if prev is None else f"   +{np.median(v) - prev:6.2f}"
        print(f"   {names[k]:28s} median {np.median(v):8.2f}  min {v.min():8.2f}  max {v.max():8.2f}  n {v.size:3d}{d}")
        prev = np.median(v)
    for k in (14, 16):
        v = (tt[:, k] - np.median(tt[:, 0])) / 100.0
        dur = (tt[:, k] - tt[:, {14: 18, 16: 15}[k]]) / 100.0
        print(f"   {names[k]}: per-XCD median of the phase duration", [round(float(np.median(dur[x::8])), 2) for x in range(8)])
        slow = np.argsort(-v)[:24]
        print("      slowest CUs (cu: finish, duration):", [(int(c), round(float(v[c]), 1), round(float(dur[c]), 1)) for c in slow])
        fast = np.argsort(v)[:8]
        print("      fastest CUs:", [(int(c), round(float(v[c]), 1), round(float(dur[c]), 1)) for c in fast])
m.close()
