"""Timeline of the persistent decode step (csrc/step_engine.hip) on Qwen3-8B shapes: one eager step with per-CU wall-clock stamps
(100 MHz) of consumer wave 0 for the first four layers, reported per phase as time since the layer started on that CU.
usage: python tools/step_engine_trace.py [ctx] [layers]"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["OMX_STEP_ENGINE"] = "1"
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
names = ["layer start", "x ready (gathered)", "q/k/v rows done", "attention partials stored", "attention phase done",
         "attention vector ready", "o rows done", "x1 ready", "gate/up rows done", "act ready", "down rows done"]
t0 = t[:, 0].min()
for l in range(min(4, L)):
    print(f"layer {l}: starts {np.median(t[:, l * 12] - t0) / 100:.2f} us after the step (median over CUs)")
    prev = None
    for k, nm in enumerate(names):
        v = (t[:, l * 12 + k] - t0) / 100.0
        v = v[t[:, l * 12 + k] > 0]
        if not v.size:
            continue
        d = "" if prev is None else f"   +{np.median(v) - prev:6.2f}"
        print(f"   {nm:28s} median {np.median(v):8.2f}  min {v.min():8.2f}  max {v.max():8.2f}{d}")
        prev = np.median(v)
        if k == 2:
            for kk, nn in enumerate(["  group q/k/v swept", "  normed + roped", "  own chunk done (before merge)"]):
                vv = (t[:, 48 + l * 4 + kk] - t0) / 100.0
                vv = vv[t[:, 48 + l * 4 + kk] > 0]
                if vv.size:
                    print(f"   {nn:28s} median {np.median(vv):8.2f}  min {vv.min():8.2f}  max {vv.max():8.2f}")
m.close()
