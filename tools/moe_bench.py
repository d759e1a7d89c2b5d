"""Sparse-MoE block timing at Mixtral-8x7B shapes (BASELINE config 3, one GPU's view): hidden 4096, 8 experts x
(gate/up 14336x4096, down 4096x14336) bf16, top-2.  Decode (1 token: expert-selected batched GEMV, HBM-bound:
2 experts x 3 matrices = 704.6 MB per layer) and prefill (2048 tokens: device sort + grouped MFMA GEMM)."""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import omx_import
omx = omx_import.load_package()
from ominix_mlx_amd import moe
ops = omx.ops
E, h, I, k = 8, 4096, 14336, 2
layers = 4          # distinct weight sets so that the 256 MB Infinity Cache cannot serve re-reads
blocks = []
for l in range(layers):
    gw = ops.fill_uniform((E, h), 10 + l, 0.5)
    blocks.append(moe.SparseMoeBlock(gw, ops.fill_uniform((E, I, h), 20 + l, 0.03), ops.fill_uniform((E, I, h), 30 + l, 0.03),
                                     ops.fill_uniform((E, h, I), 40 + l, 0.03), k, "mixtral"))
for n, reps in ((1, 40), (2048, 6)):
    x = ops.fill_uniform((n, h), 5, 1.0)
    for b in blocks:
        b.forward(x)
    ops.synchronize()
    t0 = time.perf_counter()
    for r in range(reps):
        for b in blocks:
            b.forward(x)
    ops.synchronize()
    dt = (time.perf_counter() - t0) / (reps * layers)
    if n == 1:
        byt = k * 3 * I * h * 2
        print(f"decode  n=1    : {dt*1e6:8.1f} us/layer  {byt/dt/1e9:7.1f} GB/s algorithmic ({byt/1e6:.1f} MB)  frac of 8 TB/s {byt/dt/8e12:.3f}")
    else:
        fl = 2.0 * n * k * 3 * I * h
        print(f"prefill n={n}: {dt*1e6:8.1f} us/layer  {fl/dt/1e12:7.1f} TFLOP/s  frac of 2.5 PF {fl/dt/2.5e15:.3f}")
