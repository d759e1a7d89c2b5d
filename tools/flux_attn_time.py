"""Joint attention of a FLUX.2-klein block (24 heads x 128, 4608 tokens, no mask) and the causal prefill attention (Qwen3-8B heads,
2048 tokens) through omx_sdpa: TFLOP/s and run-to-run determinism.  (Round 3 used it to A/B a lazy-rescale variant -- see
EXPERIMENTS.md: a wave-uniform branch around the rescale cost the kernel 60 %.)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import omx_import
omx = omx_import.load_package()
T = omx.ops.Tensor
for name, (H, Hkv, S, mask) in {"flux 4608 none": (24, 24, 4608, 0), "prefill 2048 causal": (32, 8, 2048, 1)}.items():
    D = 128
    q = omx.ops.fill_uniform((1, H, S, D), 1, 1.0); k = omx.ops.fill_uniform((1, Hkv, S, D), 2, 1.0); v = omx.ops.fill_uniform((1, Hkv, S, D), 3, 1.0)
    outs = {}
    for lazy in ("a", "b"):
        out = T((1, H, S, D), "bf16")
        def run(n):
            for _ in range(n):
                omx.check(omx.lib.omx_sdpa(out.ptr, q.ptr, k.ptr, v.ptr, 1, H, Hkv, S, S, D, Hkv * S * D, S * D, D ** -0.5, mask, None, 12, None))
            omx.ops.synchronize()
        run(3)
        t = time.perf_counter(); run(10); dt = (time.perf_counter() - t) / 10
        flop = 4.0 * S * S * D * H * (0.5 if mask else 1.0)
        outs[lazy] = out.numpy()
        print(f"{name:22s} run {lazy}  {dt * 1e6:8.1f} us  {flop / dt / 1e12:7.1f} TF/s  ({flop / dt / 2.5e15:.3f} of 2.5 PF)", flush=True)
    print("   run-to-run identical:", bool(np.array_equal(outs["a"], outs["b"])))
