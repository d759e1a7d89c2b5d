"""128 x 256 against the shipped tile choice on the two prompt projections that fill half the chip, plus the whole 2 048-token prompt.
usage: python tools/gemm_rows128_ab.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench  # noqa: E402
import omx_import  # noqa: E402
omx = omx_import.load_package()
from ominix_mlx_amd import engine, ops  # noqa: E402

rng = np.random.default_rng(0)
T = ops.Tensor
for (M, N, K) in ((2048, 4096, 4096), (2048, 4096, 12288), (2048, 4096, 2048), (1024, 4096, 4096), (4096, 3072, 3072)):
    x = T.from_numpy(rng.standard_normal((M, K)).astype(np.float32))
    ws = [T.from_numpy((rng.standard_normal((N, K)) * 0.05).astype(np.float32)) for _ in range(6)]   # distinct weights: not one cached matrix
    out = ops.empty_like(x, (M, N))
    line = f"M={M} N={N} K={K}:"
    for mode in ("0", "1"):
        os.environ["OMX_GEMM_ROWS128"] = mode
        def run(n):
            for i in range(n):
                omx.check(omx.lib.omx_linear(out.ptr, x.ptr, ws[i % 6].ptr, None, M, N, K, x.dtype, None))
            omx.check(omx.lib.omx_synchronize(None))
        run(12)
        t0 = time.perf_counter(); run(120); dt = (time.perf_counter() - t0) / 120
        line += f"  rows128={mode} {dt * 1e6:7.1f} us ({2.0 * M * N * K / dt / 1e12:6.1f} TF/s)"
    print(line, flush=True)
del os.environ["OMX_GEMM_ROWS128"]
ids = bench.prompt_ids(2048, bench.QWEN3_8B["vocab_size"])
for mode in ("0", "-1", "0", "-1"):
    os.environ["OMX_GEMM_ROWS128"] = mode
    m = engine.Model(max_context=2048 + 64, **bench.QWEN3_8B)
    m.synth_weights()
    m.prefill(ids); m.reset(); m.prefill(ids)
    a = m.last_prefill_ms(); m.reset(); m.prefill(ids)
    print(f"2 048-token prompt, rows128 {'off' if mode == '0' else 'default'}: {a:.2f} / {m.last_prefill_ms():.2f} ms", flush=True)
    m.close()
