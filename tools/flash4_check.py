"""The 4-wave persistent flash-attention kernel (csrc/attn_flash4.hip, OMX_ATTN_W4=1) against the 8-wave two-phase kernel on the FLUX joint
attention shape (24 heads x 128, 4 608 tokens, no mask): largest difference, run-to-run determinism, time per call."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import omx_import
omx = omx_import.load_package()
T = omx.ops.Tensor
shapes = {"flux 24x4608": (24, 24, 4608), "8x2048": (8, 8, 2048), "32x2048 gqa8": (32, 8, 2048)}
for name, (H, Hkv, S) in shapes.items():
    D = 128
    q = omx.ops.fill_uniform((1, H, S, D), 1, 1.0); k = omx.ops.fill_uniform((1, Hkv, S, D), 2, 1.0); v = omx.ops.fill_uniform((1, Hkv, S, D), 3, 1.0)
    outs = {}
    for mode in ("0", "1", "1b", "thr0"):
        os.environ["OMX_ATTN_W4"] = "0" if mode == "0" else "1"
        os.environ["OMX_ATTN_W4_THR"] = "0" if mode == "thr0" else "8"
        out = T((1, H, S, D), "bf16")
        def run(n):
            for _ in range(n):
                omx.check(omx.lib.omx_sdpa(out.ptr, q.ptr, k.ptr, v.ptr, 1, H, Hkv, S, S, D, Hkv * S * D, S * D, D ** -0.5, 0, None, 12, None))
            omx.ops.synchronize()
        run(3)
        t = time.perf_counter(); run(20); dt = (time.perf_counter() - t) / 20
        flop = 4.0 * S * S * D * H
        outs[mode] = out.numpy().astype(np.float32)
        print(f"{name:16s} W4={mode:5s} {dt * 1e6:8.1f} us  {flop / dt / 1e12:7.1f} TF/s  ({flop / dt / 2.5e15:.3f} of 2.5 PF)", flush=True)
    d = np.abs(outs["1"] - outs["0"]); print("   max |new - old| =", d.max(), " max |old| =", np.abs(outs["0"]).max(), " mean diff", d.mean())
    print("   thr 0 vs old:", np.abs(outs["thr0"] - outs["0"]).max(), "  run-to-run identical:", bool(np.array_equal(outs["1"], outs["1b"])), " nan:", bool(np.isnan(outs["1"]).any()))
